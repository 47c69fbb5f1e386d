// cstr_mt_device.h -- NumPy-legacy MT19937 pieces shared by the replay sampler (cstr_replay.hip) and the rollout kernel
// (cstr_mlp.hip): the recurrence, tempering, and a ONE-WAVE index draw (no workgroup barriers) that reproduces
// `RandomState.randint(0, high, size)` (masked rejection, numpy/random/src/distributions/distributions.c) word for word.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int MT_N = 624, MT_M = 397;

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t far)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// mt19937_gen on ONE wave over the LDS image mt[624]: ten rounds of 64 consecutive words in ascending order. Word kk needs
// OLD mt[kk], OLD mt[kk + 1] (a later word of this round or of a later round; every lane reads before any lane writes) and
// mt[(kk + 397) % 624], which is OLD for kk < 227 and was produced >= 227 words (an earlier round) ago otherwise; kk = 623
// pairs with the NEW mt[0]. A wave's LDS operations complete in order, so no barrier is needed between rounds.
__device__ __forceinline__ void mt_twist_wave(uint32_t *mt, const int lane)
{
    for (int base = 0; base < MT_N; base += 64) {
        const int kk = base + lane;
        uint32_t v = 0;
        if (kk < MT_N) {
            const int k1 = (kk == MT_N - 1) ? 0 : kk + 1, kf = (kk < MT_N - MT_M) ? kk + MT_M : kk - (MT_N - MT_M);
            v = mt_mix(mt[kk], mt[k1], mt[kf]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // reads of this round before its writes (compiler ordering)
        if (kk < MT_N) mt[kk] = v;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// random_bounded_uint64_fill(off = 0, rng = high - 1, masked) for rng < 2^32 - 1 on ONE wave: `count` accepted values into
// out[] (global), consuming words from (mt, pos) in order; returns the new pos. Same contract as mt_randint_fill (cstr_replay.hip).
__device__ __forceinline__ int mt_randint_fill_wave(uint32_t *mt, int pos, const uint32_t rng, const int count, int32_t *__restrict__ out,
                                                    const int lane)
{
    if (rng == 0u) {  // high == 1: zeros, consumes nothing
        for (int i = lane; i < count; i += 64) out[i] = 0;
        return pos;
    }
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    int filled = 0;
    while (filled < count) {  // wave-uniform
        if (pos == MT_N) { mt_twist_wave(mt, lane); pos = 0; }
        const int chunk = min(MT_N - pos, 64), need = count - filled;
        uint32_t w = 0;
        bool acc = false;
        if (lane < chunk) {
            w = mt_temper(mt[pos + lane]) & mask;
            acc = (w <= rng);
        }
        const unsigned long long bal = __ballot(acc);
        const int rank = __popcll(bal & ((1ULL << lane) - 1ULL)), total = __popcll(bal);
        if (acc && rank < need) out[filled + rank] = (int32_t)w;
        const unsigned long long last = __ballot(acc && rank == need - 1);  // the draw stops right after the need-th accepted word
        pos += last ? (int)__builtin_ctzll(last) + 1 : chunk;
        filled += min(total, need);
    }
    return pos;
}

}  // namespace
