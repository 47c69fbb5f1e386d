// Micro-benchmark (MI355X): how long does one 512-lane workgroup need to pull a 256 x 256 f32 weight matrix (256 KB) from L2
// with (a) the MFMA-operand pattern of the policy kernel (lane (r, h) reads 16 B of row n0 + r at k = 16u + 4h: 16 half-used
// cache lines per instruction) and (b) full-line loads (16 consecutive lanes read 256 contiguous bytes of one row)?
// One workgroup per CU (grid = 256), every workgroup reads the SAME matrix, like the policy kernel's layer 2.
// build: hipcc -O3 --offload-arch=gfx950 load_pattern_probe.hip -o load_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int H = 256, WAVES = 8;

template <int MODE>
__global__ __launch_bounds__(64 * WAVES) void probe(const float *__restrict__ w, float *__restrict__ sink, unsigned long long *t)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, h = lane >> 4;
    const unsigned long long t0 = wall_clock64();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 0) {  // operand pattern: wave takes tiles wave and wave + 8
        for (int tile = wave; tile < 16; tile += WAVES) {
            const float *wr = w + (tile * 16 + r) * H;
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4 *>(wr + 16 * u + 4 * h);
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    } else {  // full lines: lane i of a wave reads bytes [16 i, 16 i + 16) of a 1 KB run = one row; the wave takes 32 rows
        for (int row = wave * 32; row < wave * 32 + 32; row += 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4 *>(w + (row + u) * H + 4 * lane);
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x * 512 + tid] = acc.x;
    if (tid == 0) t[blockIdx.x] = t1 - t0;
}

int main()
{
    float *w, *sink; unsigned long long *t;
    hipMalloc(&w, H * H * 4); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&t, 256 * 8);
    hipMemset(w, 0, H * H * 4);
    unsigned long long ht[256];
    for (int grid : {16, 256}) {
        for (int mode = 0; mode < 2; ++mode) {
            double best = 1e9, sum = 0;
            for (int rep = 0; rep < 20; ++rep) {
                if (mode == 0) probe<0><<<grid, 512>>>(w, sink, t); else probe<1><<<grid, 512>>>(w, sink, t);
                hipDeviceSynchronize();
                hipMemcpy(ht, t, grid * 8, hipMemcpyDeviceToHost);
                double m = 0; for (int i = 0; i < grid; ++i) m += ht[i] * 0.01; m /= grid;
                if (rep >= 5) { sum += m; if (m < best) best = m; }
            }
            printf("grid %3d  %s: mean %.2f us  best %.2f us per workgroup (256 KB)\n", grid, mode ? "full-line loads " : "operand pattern  ", sum / 15, best);
        }
    }
    return 0;
}
