// Does a dependent launch find the data its predecessor wrote faster when producer and consumer workgroup sit on the SAME XCD?
// 256 workgroups of one wave ping-pong a 256 KB buffer (16 x 16-float tiles, one per workgroup): workgroup b reads tile
// (b + shift) % 256 of `src` (written by workgroup (b + shift) % 256 of the previous launch) and writes tile b of `dst`.
// Workgroups are dispatched round robin over the 8 XCDs (workgroup b -> XCD b % 8): shift 0 = own tile, shift 8 = another
// workgroup of the same XCD, shift 1 = a workgroup of the next XCD. 200 launches per hipGraph, HIP events around replays.
// Build: hipcc -O3 --offload-arch=gfx950 -o xcd_locality_probe tools/probes/xcd_locality_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(64) void hop(const float4 *__restrict__ src, float4 *__restrict__ dst, const int shift, const int tiles)
{
    const int t = (blockIdx.x + shift) % tiles;
    float4 v = src[t * 64 + threadIdx.x];
    v.x += 1.0f;
    dst[blockIdx.x * 64 + threadIdx.x] = v;
}

__global__ __launch_bounds__(64) void empty_kernel(float4 *dst) { if (dst == nullptr) dst[0] = make_float4(0, 0, 0, 0); }

static float run(hipStream_t s, float4 *a, float4 *b, int shift, int tiles, int mode)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 200; ++i) {
        if (mode == 0) hop<<<tiles, 64, 0, s>>>(i & 1 ? b : a, i & 1 ? a : b, shift, tiles);
        else empty_kernel<<<tiles, 64, 0, s>>>(a);
    }
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / 2000.0f;
}

int main()
{
    const int tiles = 256;
    float4 *a, *b;
    hipMalloc(&a, tiles * 64 * sizeof(float4)); hipMalloc(&b, tiles * 64 * sizeof(float4));
    hipMemset(a, 0, tiles * 64 * sizeof(float4)); hipMemset(b, 0, tiles * 64 * sizeof(float4));
    hipStream_t s; hipStreamCreate(&s);
    printf("empty kernel, 256 workgroups                         %6.2f us per dependent launch\n", run(s, a, b, 0, tiles, 1));
    const int shifts[] = {0, 8, 64, 1, 3, 4, 9, 129};
    for (int k = 0; k < 8; ++k)
        printf("reads the tile workgroup b + %3d wrote (XCD %s)   %6.2f us per dependent launch   (%s)\n", shifts[k],
               shifts[k] % 8 == 0 ? "same " : "other", run(s, a, b, shifts[k], tiles, 0), hipGetErrorString(hipGetLastError()));
    for (int t : {8, 32, 64, 1024}) {
        printf("%4d workgroups: own tile %6.2f us, next workgroup's tile %6.2f us, + 8 %6.2f us\n", t, run(s, a, b, 0, t < 256 ? t : 256, 0),
               run(s, a, b, 1, t < 256 ? t : 256, 0), run(s, a, b, 8 % (t < 256 ? t : 256), t < 256 ? t : 256, 0));
    }
    return 0;
}
