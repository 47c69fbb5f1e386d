// How do a 256 KB-per-CU weight stream (1 KB global_load_dwordx4 per wave instruction, all workgroups reading the SAME 256 KB from
// their XCD's L2) and the f32 MFMAs that consume it share a CU on gfx950? One 512-lane workgroup per CU (256), 8 waves = 2 per
// SIMD, every wave owns 32 KB of the buffer (32 loads) and 128 v_mfma_f32_16x16x4_f32 -- the rollout kernel's layer 2. Loads and
// MFMAs are inline asm (the first version of this probe, mfma_load_overlap_probe.hip, was rescheduled and CSE'd by the compiler:
// its "MFMAs consume in order" figure measured the compiler's load sinking, not the hardware). In-kernel s_memtime per wave.
//   mode 0  MFMAs only                      mode 1  loads only, one wait at the end
//   mode 2  waves 0-3 load, waves 4-7 MFMA (no data dependence): does load ISSUE on the sibling wave slow the MFMAs?
//   mode 3  all 32 loads issued, then the MFMAs consume them in k order (s_waitcnt vmcnt(31 - i) in front of chunk i)
//   mode 4  software pipeline: DEPTH loads ahead, then per chunk { 1 load, wait for the oldest, 4 MFMAs }
//   mode 6  ONE matrix wave per SIMD (waves 0-3): 64 loads + 256 MFMAs each, pipelined like mode 4 with 4 MFMA chains per load PAIR
//           (four 16-column tiles per wave: 8 accumulators); waves 4-7 idle (VALU 0) or running ~4,000 cycles of dependent VALU
//           work (VALU 1: the rollout kernel's noise draw)
//   mode 5  mode 3 on waves 4-7 from t = 0; waves 0-3 first wait ~LAG cycles (s_sleep), then the same (the kernel's "layer 1 first")
// Build: hipcc -O3 --offload-arch=gfx950 -o stream_mfma_probe tools/probes/stream_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

// "=&v": the destination never overlaps the address pair; every destination is read in full BEHIND an explicit s_waitcnt (the compiler
// does not know that an asm load's result lands later: a partly dead destination was reused as the next load's address register
// and overwritten by the returning data -- a fault at address 0 in the first run of this probe)
#define LOAD(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=&v"(dst) : "v"(ptr) : "memory")
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define WAITVM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// the wait in front of a destination's first use takes it as an operand: nothing the compiler generates can read it any earlier
#define WAITUSE(n, x) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(x)::"memory")
#define USE8(x) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"memory")

template <int I> struct WaitVm { static __device__ __forceinline__ void run(f32x4 &x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(I) : "memory"); } };

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(const float4 *__restrict__ w, float *out, unsigned long long *stamps, float a0, int lag)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 *src = w + (size_t)wave * 32 * 64 + lane;  // this wave's 32 KB: 32 chunks of 1 KB
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    float a = a0 + lane;
    f32x4 v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = f32x4{a0, a0, a0, a0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = t0;
    const bool loader = MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5 || (MODE == 2 && wave < 4);
    const bool mfma = MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5 || (MODE == 2 && wave >= 4);
    if (MODE == 5 && wave < 4) {
        for (int i = 0; i < lag; i += 64 * 16) __builtin_amdgcn_s_sleep(16);  // s_sleep n = 64 n cycles
    }
    if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) LOAD(v[i], src + i * 64, 0);
        t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i + DEPTH < 32) { LOAD(v[i + DEPTH], src + (i + DEPTH) * 64, 0); WaitVm<DEPTH>::run(v[i]); }
            else {
                switch (31 - i) {  // loads still behind chunk i
#define W(n) case n: WAITUSE(n, v[i]); break;
                W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15)
#undef W
                }
            }
            MFMA(c0, a, v[i].x); MFMA(c1, a, v[i].y); MFMA(c2, a, v[i].z); MFMA(c3, a, v[i].w);
        }
    } else {
        if (loader) {
#pragma unroll
            for (int i = 0; i < 32; ++i) LOAD(v[i], src + i * 64, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (mfma) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (MODE == 3 || MODE == 5) {
                    switch (31 - i) {
#define W(n) case n: WAITUSE(n, v[i]); break;
                    W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19) W(20)
                    W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31)
#undef W
                    }
                    MFMA(c0, a, v[i].x); MFMA(c1, a, v[i].y); MFMA(c2, a, v[i].z); MFMA(c3, a, v[i].w);
                } else {
                    MFMA(c0, a, a); MFMA(c1, a, a); MFMA(c2, a, a); MFMA(c3, a, a);
                }
            }
        }
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();  // everything ISSUED
    WAITVM(0);
    if (MODE == 1 || MODE == 2) { USE8((v + 0)); USE8((v + 8)); USE8((v + 16)); USE8((v + 24)); }
    f32x4 s = c0 + c1 + c2 + c3;  // waits for the MFMA results
    float t = s[0] + s[1] + s[2] + s[3];
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) t += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    if (t == 12345.678f) out[0] = t;
    if (lane == 0) {
        unsigned long long *st = stamps + ((size_t)blockIdx.x * 8 + wave) * 4;
        st[0] = t0; st[1] = t1; st[2] = t2; st[3] = t3;
    }
}

template <int DEPTH, int VALU>
__global__ __launch_bounds__(512) void probe6(const float4 *__restrict__ w, float *out, unsigned long long *stamps, float a0, int lag)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = t0, t2 = t0, t3 = t0;
    float t = 0.0f;
    if (wave < 4) {
        const float4 *src = w + (size_t)wave * 64 * 64 + lane;  // this wave's 64 KB: 64 chunks of 1 KB
        f32x4 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float a = a0 + lane;
        f32x4 v[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = f32x4{a0, a0, a0, a0};
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) LOAD(v[i], src + i * 64, 0);
        t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (i + DEPTH < 64) { LOAD(v[i + DEPTH], src + (i + DEPTH) * 64, 0); WaitVm<DEPTH>::run(v[i]); }
            else {
                switch (63 - i) {
#define W(n) case n: WAITUSE(n, v[i]); break;
                W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15)
#undef W
                }
            }
            const int g = 4 * (i & 1);
            MFMA(c[g + 0], a, v[i].x); MFMA(c[g + 1], a, v[i].y); MFMA(c[g + 2], a, v[i].z); MFMA(c[g + 3], a, v[i].w);
        }
        t2 = __builtin_amdgcn_s_memtime();
        WAITVM(0);
        f32x4 s = (c[0] + c[1]) + (c[2] + c[3]) + (c[4] + c[5]) + (c[6] + c[7]);
        t = s[0] + s[1] + s[2] + s[3];
        t3 = __builtin_amdgcn_s_memtime();
    } else if (VALU) {
        // ~4,000 cycles of dependent VALU work on the helper waves (one 4-cycle op per iteration)
        float x = a0 + lane;
        for (int i = 0; i < 1000; ++i) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
        t = x;
        t1 = t2 = t3 = __builtin_amdgcn_s_memtime();
    }
    if (t == 12345.678f) out[0] = t;
    if (lane == 0) {
        unsigned long long *st = stamps + ((size_t)blockIdx.x * 8 + wave) * 4;
        st[0] = t0; st[1] = t1; st[2] = t2; st[3] = t3;
    }
}

template <int MODE, int DEPTH, int VALU = 0>
static void run(const float4 *w, float *out, unsigned long long *stamps, const char *tag, int lag = 0)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() {
        if (MODE == 6) probe6<DEPTH, VALU><<<256, 512>>>(w, out, stamps, 1.0f, lag);
        else probe<MODE == 6 ? 0 : MODE, DEPTH><<<256, 512>>>(w, out, stamps, 1.0f, lag);
    };
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    const int reps = 200;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8 * 4);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    // per workgroup: relative to its earliest t0; medians over workgroups of (waves 0-3, waves 4-7) x (issued loads, issued all, done)
    std::vector<long long> col[2][3];
    for (int b = 0; b < 256; ++b) {
        unsigned long long base = ~0ull;
        for (int wv = 0; wv < 8; ++wv) base = std::min(base, h[(b * 8 + wv) * 4]);
        for (int g = 0; g < 2; ++g)
            for (int k = 0; k < 3; ++k) {
                long long m = 0;
                for (int wv = 4 * g; wv < 4 * g + 4; ++wv) m = std::max(m, (long long)(h[(b * 8 + wv) * 4 + 1 + k] - base));
                col[g][k].push_back(m);
            }
    }
    auto med = [](std::vector<long long> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-78s %6.2f us/launch | waves 0-3: loads issued %6lld all issued %6lld done %6lld | waves 4-7: %6lld %6lld %6lld cycles\n", tag,
           ms * 1e3 / reps, med(col[0][0]), med(col[0][1]), med(col[0][2]), med(col[1][0]), med(col[1][1]), med(col[1][2]));
}

int main()
{
    float4 *w;
    float *out;
    unsigned long long *stamps;
    hipMalloc(&w, 256 * 1024);
    hipMalloc(&out, 64);
    hipMalloc(&stamps, 256 * 8 * 4 * 8);
    hipMemset(w, 0, 256 * 1024);
    run<0, 0>(w, out, stamps, "0 MFMAs only (128 per wave, 2 waves per SIMD)");
    run<1, 0>(w, out, stamps, "1 loads only (32 x 1 KB per wave, 256 KB per CU, same bytes on every CU)");
    run<2, 0>(w, out, stamps, "2 waves 0-3 load (128 KB per CU), waves 4-7 MFMA, independent");
    run<3, 0>(w, out, stamps, "3 all loads issued, MFMAs consume in k order (vmcnt(31 - i))");
    run<4, 4>(w, out, stamps, "4 pipeline, 4 loads ahead, per chunk: 1 load / wait oldest / 4 MFMAs");
    run<4, 8>(w, out, stamps, "4 pipeline, 8 loads ahead");
    run<4, 12>(w, out, stamps, "4 pipeline, 12 loads ahead");
    run<6, 4>(w, out, stamps, "6 one matrix wave per SIMD (64 loads + 256 MFMAs), 4 loads ahead, helpers idle");
    run<6, 8>(w, out, stamps, "6 one matrix wave per SIMD, 8 loads ahead, helpers idle");
    run<6, 16>(w, out, stamps, "6 one matrix wave per SIMD, 16 loads ahead, helpers idle");
    run<6, 8, 1>(w, out, stamps, "6 one matrix wave per SIMD, 8 loads ahead, helpers run 1000 dependent FMAs");
    run<5, 0>(w, out, stamps, "5 mode 3, waves 0-3 start 3000 cycles late", 3000);
    run<5, 0>(w, out, stamps, "5 mode 3, waves 0-3 start 5000 cycles late", 5000);
    return 0;
}
