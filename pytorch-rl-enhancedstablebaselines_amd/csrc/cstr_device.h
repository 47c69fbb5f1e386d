// cstr_device.h -- shared host/device helpers of libcstr_rl_hip (internal, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/cstr_rl_hip.h"

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

// Launch shape for one-lane-per-env streaming kernels. MI355X: 256 CUs in 8 XCDs. Small N (the 4096-env
// training case) is latency-bound: one wave per workgroup spreads the launch over as many CUs/XCDs as
// there are waves. Large N streams: 256-thread workgroups, capped at 8 per CU, grid-stride for the rest.
static inline void env_launch_shape(int64_t n, int &block, int &grid)
{
    // tuning knobs for the A/B microbenchmarks (tools/microbench_collect.py); unset in production
    static const int ov_block = getenv("CSTR_ENV_BLOCK") ? atoi(getenv("CSTR_ENV_BLOCK")) : 0;
    static const int ov_cap = getenv("CSTR_ENV_GRID_CAP") ? atoi(getenv("CSTR_ENV_GRID_CAP")) : 0;
    if (n <= 65536 && !ov_block) {
        block = 64;
        grid = (int)((n + 63) / 64);
    } else {
        // streaming regime (A/B on MI355X, N = 2^22, profiles/r01_notes.md): 512-thread workgroups, <= 4096 of them
        // (2 envs per lane) beat 256 x 2048 by ~5 %; > 4096 workgroups lose to the per-workgroup ticket atomics
        block = ov_block ? ov_block : 512;
        const int64_t cap = ov_cap ? ov_cap : 4096;
        int64_t g = (n + block - 1) / block;
        grid = (int)(g < cap ? g : cap);
    }
}

// Flat element-wise kernels over float4: same rule.
static inline void flat_launch_shape(int64_t n_vec, int &block, int &grid)
{
    // tuning knobs for A/B microbenchmarks (bench.py "kernels" section); unset in production
    static const int ov_block = getenv("CSTR_FLAT_BLOCK") ? atoi(getenv("CSTR_FLAT_BLOCK")) : 0;
    static const int ov_cap = getenv("CSTR_FLAT_GRID_CAP") ? atoi(getenv("CSTR_FLAT_GRID_CAP")) : 0;
    // A/B on MI355X (bench.py "kernels", r01_notes.md): 512-thread workgroups, <= 4096 of them: Adam at the learners' 136 k
    // parameters 4.5 -> 3.9 us, streaming fractions unchanged (0.63 Adam / 0.87 polyak of 8 TB/s for every shape tried)
    block = ov_block ? ov_block : 512;
    const int64_t cap = ov_cap ? ov_cap : 4096;
    int64_t g = (n_vec + block - 1) / block;
    if (g < 1) g = 1;
    grid = (int)(g < cap ? g : cap);
}

#ifdef __HIPCC__
// "Last workgroup out" ticket: every workgroup has finished READING the control words it needs before it
// takes a ticket (the barrier orders its waves' loads, whose values were already consumed for addressing),
// so the workgroup that draws gridDim.x-1 may advance them. No data is handed between workgroups inside the
// launch -- the next kernel on the stream sees the update through the kernel boundary.
__device__ __forceinline__ bool last_block_ticket(unsigned long long *ticket)
{
    __shared__ int is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = atomicAdd(ticket, 1ULL);
        is_last = (t == (unsigned long long)gridDim.x - 1ULL);
        if (is_last) *ticket = 0ULL;  // self-reset for the next launch
    }
    __syncthreads();
    return is_last != 0;
}

// ring_ctl = { pos, full, ticket, adds }: ReplayBuffer.add's epilogue (core/common/buffers.py:280-283)
// `rng_ctl` (may be NULL): a Philox stream control block whose offset this launch advances by `rng_advance` on behalf of the
// policy launch in front of it (cstr_policy_mlp_t.reserved bit 0): one ticket for both control blocks.
__device__ __forceinline__ void ring_advance_last_block(int64_t *ring_ctl, int64_t rows, uint64_t *rng_ctl = nullptr,
                                                        uint64_t rng_advance = 0)
{
    if (last_block_ticket(reinterpret_cast<unsigned long long *>(ring_ctl + 2)) && threadIdx.x == 0) {
        int64_t pos = ring_ctl[0] + 1;
        if (pos == rows) { ring_ctl[1] = 1; pos = 0; }
        ring_ctl[0] = pos;
        ring_ctl[3] += 1;
        if (rng_ctl) rng_ctl[1] += rng_advance;
    }
}
#endif
