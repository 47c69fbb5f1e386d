// Premise check for a loader / consumer split of the policy kernel's layer 2: do 4 loader waves streaming 256 KB per CU into LDS
// by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction, no VGPRs) overlap with 4 consumer waves that each issue 256
// v_mfma_f32_16x16x4_f32 with BOTH operands read from LDS (ds_read_b128)? One workgroup per CU (256), 8 waves.
// modes: 0 consumers only (operands from a static LDS image) | 1 loaders only | 2 both, no flow control (loaders overwrite a 64 KB
// region round robin; consumers read whatever is there: timing only)
// Build: hipcc -O3 --offload-arch=gfx950 -w -o ldsdma_mfma_probe tools/probes/ldsdma_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float4 *__restrict__ w, float *out)
{
    extern __shared__ float4 lds[];  // [64 KB B ring][16 KB A image]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *ring = lds, *aimg = lds + 4096;
    for (int i = threadIdx.x; i < 4096 + 1024; i += 512) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    if (wave >= 4) {  // loaders: 64 x 1 KB pieces each
        if (MODE == 0) return;
        const int ld = wave - 4;
        for (int i = 0; i < 64; ++i) {
            const float4 *src = w + (size_t)(ld * 64 + i) * 64 + lane;
            float4 *dst = ring + ((ld * 64 + i) & 63) * 64;  // wave-uniform LDS base, lane * 16 added by the instruction
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0)
        return;
    }
    if (MODE == 1) return;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    for (int t = 0; t < 4; ++t) {
#pragma unroll 4
        for (int c = 0; c < 16; ++c) {
            const float4 a = aimg[c * 64 + lane];
            const float4 b = ring[((wave * 4 + t) * 16 + c) % 64 * 64 + lane];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c3, 0, 0, 0);
        }
    }
    f32x4 s = c0 + c1 + c2 + c3;
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
}

template <int MODE>
static void run(const float4 *w, float *out, const char *tag)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = (4096 + 1024) * sizeof(float4);
    hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 20; ++i) probe<MODE><<<256, 512, lds>>>(w, out);
    hipDeviceSynchronize();
    const int reps = 200;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) probe<MODE><<<256, 512, lds>>>(w, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-72s %.2f us per launch   (%s)\n", tag, ms * 1e3 / reps, hipGetErrorString(hipGetLastError()));
}

int main()
{
    float4 *w;
    float *out;
    hipMalloc(&w, 256 * 1024);
    hipMemset(w, 0, 256 * 1024);
    hipMalloc(&out, 4);
    run<0>(w, out, "4 consumer waves: 256 MFMAs each, A and B from LDS");
    run<1>(w, out, "4 loader waves: 256 KB per CU by LDS-DMA");
    run<2>(w, out, "both at once (no flow control)");
    return 0;
}
