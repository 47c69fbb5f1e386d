// f32 MFMA issue rate on gfx950 at the policy kernel's geometry: 256 workgroups (one per CU), W waves each, every wave issues N
// v_mfma_f32_16x16x4_f32 on ACC independent accumulators, operands in registers (no memory traffic at all).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_rate_probe tools/probes/mfma_rate_probe.hip ; run: ./mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int ACC>
__global__ __launch_bounds__(1024) void mfma_loop(float *out, int n, float a0, float b0)
{
    f32x4 c[ACC];
    for (int i = 0; i < ACC; ++i) c[i] = {0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int i = 0; i < n; i += ACC) {
#pragma unroll
        for (int j = 0; j < ACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
    }
    f32x4 s = c[0];
    for (int i = 1; i < ACC; ++i) s += c[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];  // keep the chain alive
}

template <int ACC>
static void run(int waves, int n, const char *tag)
{
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) mfma_loop<ACC><<<256, 64 * waves>>>(out, n, 1.0f, 2.0f);
    hipDeviceSynchronize();
    const int reps = 200;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) mfma_loop<ACC><<<256, 64 * waves>>>(out, n, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, per_simd = (double)n * waves / 4.0;
    printf("%s waves/CU %2d  acc %d  mfma/wave %5d  launch %.2f us  -> %.1f ns per MFMA per SIMD (32 cycles @2.4 GHz = 13.3 ns)  %.1f TFLOP/s\n",
           tag, waves, ACC, n, us, us * 1e3 / per_simd, 256.0 * waves * n * 2048.0 / us / 1e6);
}

int main()
{
    for (int n : {128, 256, 1024, 8192}) {
        run<4>(4, 2 * n, "1 wave/SIMD ");
        run<4>(8, n, "2 waves/SIMD");
        run<2>(8, n, "2 waves/SIMD");
        run<4>(16, n / 2, "4 waves/SIMD");
    }
    return 0;
}
