// What does ONE layer of a row-local MLP chain cost when the layer boundary is a barrier among the workgroups that share the
// rows (a "cluster") instead of a kernel launch?  (the learner's ~20 dependent launches at ~4.8 us each.)
//
// 256 workgroups x 256 lanes (4 waves, split-K like linear_act_fwd_kernel). A cluster = CS workgroups that own the SAME 16 rows
// and one 16-column tile each of a 256 x 256 Linear layer. Phase p: every workgroup reads the cluster's 16 x 256 activation panel
// (written in phase p-1 by the CS members, one 16 x 16 tile each), its own 16 x 256 weight panel, runs 64 MFMAs
// (v_mfma_f32_16x16x4_f32), combines the four split-K partials in LDS, stores its tile into the other panel and arrives at the
// cluster's counter; it proceeds when CS * (p + 1) arrivals are in. W is a column rotation (out[r][n] = in[r][(n + 1) % 256]), so
// after P phases the panel must be the input rotated by P columns: a stale or early read shows up as a wrong value.
//
// Modes
//   0  agent-scope release / acquire around the counter (what the memory model asks for in general: L2 write-back + invalidate)
//   1  cluster members on ONE XCD (workgroup b runs on XCD b % 8, MI355X guide): the XCD's L2 is the point of coherence, so the
//      tile stores only have to be acknowledged (s_waitcnt vmcnt(0)) before a relaxed arrival, and the panel is read with sc0
//      loads that miss the CU's vector L1
//   2  as 0 but ONE counter for the whole grid (a grid barrier per phase)
//   3/4  modes 0 / 1 without loads and MFMAs: the barrier alone
// Spins are bounded (SPIN_LIMIT polls, then an error word is set and the wave goes on): the grid always drains.
// Build: hipcc -O3 --offload-arch=gfx950 -o cluster_chain_probe tools/probes/cluster_chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 256, WAVES = 4, SPIN_LIMIT = 1 << 22;

struct Args {
    float *panel[2];        // [clusters][16][H] ping-pong
    const float *w;         // [H][H]
    unsigned int *counter;  // [clusters] (mode 2: [0] only), zeroed before every launch by the kernel's own epilogue
    unsigned int *error;
    int phases, cs, clusters;
};

template <int MODE>
__global__ __launch_bounds__(64 * WAVES) void chain_kernel(const Args a)
{
    __shared__ f32x4 part[WAVES - 1][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, h = lane >> 4;
    const int b = blockIdx.x;
    int cluster, member;
    if (MODE == 1 || MODE == 4) {  // XCD-local clusters: the workgroups b, b + 8, b + 16, ... share an XCD
        const int xcd = b & 7, j = b >> 3, per_xcd = a.clusters / 8;
        cluster = xcd * per_xcd + j / a.cs;
        member = j % a.cs;
    } else {
        cluster = b / a.cs;
        member = b % a.cs;
    }
    unsigned int *ctr = a.counter + (MODE == 2 ? 0 : cluster);
    const unsigned int arrivals_per_phase = MODE == 2 ? gridDim.x : a.cs;
    const bool compute = MODE <= 2;
    const int n0 = 16 * member;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, H * H * 4, 0x00020000);
    float4 bw[4];
    if (compute) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // the weight panel does not change: loaded once (a real chain loads another layer's per phase)
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, 4 * ((n0 + r) * H + 16 * (wave + WAVES * u) + 4 * h), 0, 0);
            bw[u] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    }
    for (int p = 0; p < a.phases; ++p) {
        if (compute) {
            const float *in = a.panel[p & 1] + (size_t)cluster * 16 * H;
            float *out = a.panel[(p + 1) & 1] + (size_t)cluster * 16 * H;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, 16 * H * 4, 0x00020000);
            float4 av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, 4 * (r * H + 16 * (wave + WAVES * u) + 4 * h), 0, MODE == 1 ? 1 : 0);
                av[u] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
            f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bw[u].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bw[u].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bw[u].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bw[u].w, acc1, 0, 0, 0);
            }
            f32x4 acc = acc0 + acc1;
            if (wave > 0) part[wave - 1][lane] = acc;
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int v = 0; v < WAVES - 1; ++v) acc += part[v][lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) out[(4 * h + e) * H + n0 + r] = acc[e];
            }
        }
        // ---- arrive ----
        if (wave == 0) {
            if (MODE == 1 || MODE == 4) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the wave's tile stores are acknowledged by the L2
                if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                // (every lane's stores precede lane 0's release in program order of the wave: one instruction stream)
            }
            // ---- wait ----
            if (lane == 0) {
                const unsigned int want = arrivals_per_phase * (unsigned int)(p + 1);
                int spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (++spins > SPIN_LIMIT) { atomicExch(a.error, 1u + (unsigned int)p); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (MODE == 1 || MODE == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

__global__ void reset_kernel(unsigned int *counter, int n) { if ((int)threadIdx.x < n) counter[threadIdx.x] = 0; }

template <int MODE>
static float run(hipStream_t s, Args a, int grid, int launches_per_graph, std::vector<float> &host_in, bool verify)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < launches_per_graph; ++i) {
        reset_kernel<<<1, 64, 0, s>>>(a.counter, a.clusters);
        chain_kernel<MODE><<<grid, 64 * WAVES, 0, s>>>(a);
    }
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t panel = (size_t)a.clusters * 16 * H;
    if (verify) {
        hipMemcpy(a.panel[0], host_in.data(), panel * 4, hipMemcpyHostToDevice);
        reset_kernel<<<1, 64, 0, s>>>(a.counter, a.clusters);
        chain_kernel<MODE><<<grid, 64 * WAVES, 0, s>>>(a);
        hipStreamSynchronize(s);
        std::vector<float> out(panel);
        hipMemcpy(out.data(), a.panel[a.phases & 1], panel * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int c = 0; c < a.clusters; ++c)
            for (int r = 0; r < 16; ++r)
                for (int n = 0; n < H; ++n)
                    bad += out[((size_t)c * 16 + r) * H + n] != host_in[((size_t)c * 16 + r) * H + (n + a.phases) % H];
        unsigned int err = 0;
        hipMemcpy(&err, a.error, 4, hipMemcpyDeviceToHost);
        printf("    verify: %zu wrong values of %zu, spin-limit error word %u\n", bad, panel, err);
    }
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (reps * launches_per_graph);
}

int main(int argc, char **argv)
{
    const int grid = 256, cs = argc > 1 ? atoi(argv[1]) : 16, clusters = grid / cs;
    Args a;
    const size_t panel = (size_t)clusters * 16 * H;
    hipMalloc(&a.panel[0], panel * 4); hipMalloc(&a.panel[1], panel * 4);
    float *w; hipMalloc(&w, H * H * 4);
    hipMalloc(&a.counter, 64 * 4); hipMalloc(&a.error, 4);
    hipMemset(a.counter, 0, 64 * 4); hipMemset(a.error, 0, 4);
    std::vector<float> hw((size_t)H * H, 0.0f), hin(panel);
    for (int n = 0; n < H; ++n) hw[(size_t)n * H + (n + 1) % H] = 1.0f;
    for (size_t i = 0; i < panel; ++i) hin[i] = (float)(rand() % 4096) / 16.0f;
    hipMemcpy(w, hw.data(), (size_t)H * H * 4, hipMemcpyHostToDevice);
    a.w = w; a.cs = cs; a.clusters = clusters;
    hipStream_t s; hipStreamCreate(&s);
    printf("cluster chain probe: %d workgroups x 256 lanes, clusters of %d (16 rows each), layer 256 x 256\n", grid, cs);
    const char *names[] = {"agent-scope release/acquire, cluster barrier", "XCD-local (sc0 loads, ack'd stores), cluster barrier",
                           "agent-scope, GRID barrier", "barrier only, agent scope, cluster", "barrier only, XCD-local, cluster"};
    for (int mode = 0; mode < 5; ++mode) {
        printf("mode %d: %s\n", mode, names[mode]);
        float t[3];
        const int phases[3] = {1, 9, 33};
        for (int k = 0; k < 3; ++k) {
            a.phases = phases[k];
            const bool v = mode <= 2 && k > 0;
            switch (mode) {
            case 0: t[k] = run<0>(s, a, grid, 20, hin, v); break;
            case 1: t[k] = run<1>(s, a, grid, 20, hin, v); break;
            case 2: t[k] = run<2>(s, a, grid, 20, hin, v); break;
            case 3: t[k] = run<3>(s, a, grid, 20, hin, false); break;
            default: t[k] = run<4>(s, a, grid, 20, hin, false); break;
            }
        }
        printf("    launch + reset with 1 / 9 / 33 phases: %.2f / %.2f / %.2f us -> %.3f us per phase (9 -> 33), %s\n", t[0], t[1], t[2],
               (t[2] - t[1]) / 24.0f, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
