// cstr_rng_device.h -- the sampling heads' counter-based noise: Philox4x32-10 -> Box-Muller (internal, not part of the ABI).
// Shared by cstr_mlp.hip (gaussian head / policy / rollout kernels) and cstr_chain.hip (row-chain kernels): the same counters
// give the same draws whichever launch finalises a row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Box-Muller in two independent halves (the rollout kernel gives them to two waves): z0 = radius(a) * cos, z1 = radius(a) * sin
__device__ __forceinline__ float box_muller_radius(uint32_t a)
{
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1): 24 random bits, never 0
    return sqrtf(-2.0f * logf(u1));
}

// cos / sin(2 pi u2) through sincospif: one shared, exact argument reduction (the angle is given in half-turns) instead of two
// full-range reductions of 2 pi u2 -- the same distribution, a shorter dependent chain
__device__ __forceinline__ void box_muller_angle(uint32_t b, float &cs, float &sn)
{
    const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
    sincospif(2.0f * u2, &sn, &cs);
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &z0, float &z1)
{
    const float r = box_muller_radius(a);
    float sn, cs;
    box_muller_angle(b, cs, sn);
    z0 = r * cs;
    z1 = r * sn;
}

}  // namespace
