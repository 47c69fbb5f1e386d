// Do f32 MFMAs and a 256 KB-per-CU weight stream overlap on gfx950? One workgroup per CU (256), 8 waves; every wave issues
// 128 v_mfma_f32_16x16x4_f32 (operands in registers) and/or 32 x 1 KB global_load_dwordx4 from a 256 KB buffer shared by all
// workgroups (the policy kernel's layer-2 weight walk). Variants: mfma only | loads only | all loads then the MFMAs (independent
// of the loaded data) | 4 loads / 16 MFMAs interleaved | MFMAs CONSUMING the loaded data in order (the real dependency).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_load_overlap_probe tools/probes/mfma_load_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float4 *__restrict__ w, float *out, float a0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 *src = w + (size_t)wave * 32 * 64 + lane;  // this wave's 32 KB
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    float a = a0 + lane, b = a0 - lane;
    float4 v[32];
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
    } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = src[i * 64];
    } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = src[i * 64];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
    } else if (MODE == 3) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * g + i] = src[(4 * g + i) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            }
        }
    } else {  // MODE 4: all loads issued, MFMAs consume them in order (B operand = loaded data)
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = src[i * 64];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[i].x, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[i].y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[i].z, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[i].w, c3, 0, 0, 0);
        }
    }
    f32x4 s = c0 + c1 + c2 + c3;
    float t = s[0] + s[1] + s[2] + s[3];
    if (MODE != 0 && MODE != 4) {
#pragma unroll
        for (int i = 0; i < 32; ++i) t += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (t == 12345.678f) out[0] = t;
}

template <int MODE>
static void run(const float4 *w, float *out, const char *tag)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) probe<MODE><<<256, 512>>>(w, out, 1.0f);
    hipDeviceSynchronize();
    const int reps = 200;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) probe<MODE><<<256, 512>>>(w, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %.2f us per launch\n", tag, ms * 1e3 / reps);
}

int main()
{
    float4 *w;
    float *out;
    hipMalloc(&w, 256 * 1024);
    hipMemset(w, 0, 256 * 1024);
    hipMalloc(&out, 4);
    run<0>(w, out, "128 MFMAs per wave");
    run<1>(w, out, "32 x 1 KB loads per wave (256 KB per CU, shared)");
    run<2>(w, out, "all loads issued, then the MFMAs (independent of the data)");
    run<3>(w, out, "4 loads / 16 MFMAs interleaved (independent)");
    run<4>(w, out, "all loads issued, MFMAs consume them in order");
    return 0;
}
