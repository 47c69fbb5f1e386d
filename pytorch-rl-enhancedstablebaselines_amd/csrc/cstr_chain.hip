// cstr_chain.hip -- row-chain kernels of the learners' gradient step for gfx950 (include/cstr_rl_hip.h, "row-chain kernels").
//
// Why: a SAC gradient step at batch 256 was ~20 dependent launches of ~4.8 us for ~0.03 GFLOP each. A launch
// boundary is nevertheless the CHEAPEST cross-workgroup hand-over this chip offers: an empty dependent launch costs 1.6 us, a barrier
// among the 16 workgroups that share 16 rows costs 4 us with hand-rolled XCD-local coherence (and reads stale data) and 14-17 us with
// the memory model's agent-scope release / acquire (tools/probes/cluster_chain_probe.hip, profiles/r03_notes.md). So these kernels
// remove boundaries without ANY hand-over inside a launch:
//   * a workgroup (4 waves) owns 16 batch rows x one column group (1, 2 or 4 MFMA tiles) of the chain's WIDE layer (H1 x H2) and
//     streams only that column group's weights (v_mfma_f32_16x16x4_f32, exact f32; A operand = a 16 x K panel in LDS);
//   * the cheap layer in FRONT of it is recomputed by every workgroup of the row group into that panel: the first Linear
//     (K = obs_dim (+ act_dim) <= 12 inputs), or an element-wise function of stored activations (dz2 = dq * w3 * relu'(h2));
//   * the narrow layer BEHIND it (Q head H2 -> 1, actor head H2 -> 2A, action gradient H1 -> A) leaves the launch as per-column-group
//     partial sums, which the next launch adds up in a fixed order in its prologue (deterministic; no float atomics, no tickets).
// SAC.train (core/sac/sac.py:215-287) = actor chain, Q chain (4 nets), Q backward chain, [dW/db sets], [Adam], Q chain (x_pi), Q backward
// chain (to the action), actor backward chain, [dW/db sets], [Adam + polyak]: 10 launches (6 of them here) instead of 20.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cstr_rl_hip.h"
#include "cstr_device.h"
#include "cstr_rng_device.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int BUF_OOB = 0x40000000;
constexpr int CH_THREADS = 256, CH_WAVES = 4;
constexpr float LOG_STD_MIN = -20.0f, LOG_STD_MAX = 2.0f;  // core/sac/policies.py:20-22
constexpr int MAX_A = CSTR_MAX_HEAD_ACT;                   // 4

// dynamic LDS layout (floats): part | red | xs | gs | panel
constexpr int SM_PART = 0;                          // f32x4 part[3][64]: split-K partial tiles
constexpr int SM_RED = SM_PART + 3 * 64 * 4;        // red[4 tiles][16 rows][8]: narrow-layer partials per tile
constexpr int SM_XS = SM_RED + 4 * 16 * 8;          // xs[16][16]: the row group's input rows
constexpr int SM_GS = SM_XS + 16 * 16;              // gs[16][8]: per-row gradients of the prologue
constexpr int SM_PANEL = SM_GS + 16 * 8;            // panel[16][K + 4]: A operand of the wide layer
static size_t chain_lds_bytes(int k) { return (size_t)(SM_PANEL + 16 * (k + 4)) * sizeof(float); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float *base, const int64_t floats)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)(floats * 4), 0x00020000);
}

__device__ __forceinline__ float4 ld128(const __amdgpu_buffer_rsrc_t rs, const int byte_off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ float ld32(const __amdgpu_buffer_rsrc_t rs, const int byte_off)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0));
}

// sum over the 16 lanes that share lane >> 4 (the 16 columns of an MFMA C/D tile row): xor butterfly, every lane gets the total
__device__ __forceinline__ float rowsum16(float v)
{
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// B operand of a FORWARD-layout pass: W [N][K] row-major, tile columns n0 .. n0 + 15; lane (r, h) takes W[n0 + r][16 c + 4 h .. + 3]
// of the wave's k chunks c = ks, ks + S, ... (rows / k beyond the matrix read as zeros by the descriptor's range check)
template <int NQ>
__device__ __forceinline__ void load_b_fwd(float4 (&bq)[NQ], const float *w, const int N, const int K, const int n0, const int ks, const int S)
{
    const int lane = threadIdx.x & 63, r = lane & 15, h = lane >> 4;
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(w, (int64_t)N * K);
    const int wo = 4 * (n0 + r) * K;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int k = 16 * (ks + S * u) + 4 * h;
        bq[u] = ld128(rw, (k < K && n0 + r < N) ? wo + 4 * k : BUF_OOB);
    }
}

// B operand of a BACKWARD-layout pass: out[m][k] = sum_n P[m][n] W[n][k], W [N][K] row-major, tile columns k0 .. k0 + 15; lane (r, h)
// takes W[16 c + 4 h + e][k0 + r], e = 0..3 (64-byte row segments across the 16 lanes)
template <int NQ>
__device__ __forceinline__ void load_b_bwd(float4 (&bq)[NQ], const float *w, const int N, const int K, const int k0, const int ks, const int S)
{
    const int lane = threadIdx.x & 63, r = lane & 15, h = lane >> 4;
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(w, (int64_t)N * K);
    const bool col_ok = k0 + r < K;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int n = 16 * (ks + S * u) + 4 * h;
        const int wo = 4 * (n * K + k0 + r);
        bq[u].x = ld32(rw, (col_ok && n < N) ? wo : BUF_OOB);
        bq[u].y = ld32(rw, (col_ok && n + 1 < N) ? wo + 4 * K : BUF_OOB);
        bq[u].z = ld32(rw, (col_ok && n + 2 < N) ? wo + 8 * K : BUF_OOB);
        bq[u].w = ld32(rw, (col_ok && n + 3 < N) ? wo + 12 * K : BUF_OOB);
    }
}

// 16 x 16 tile: acc = panel[16][kdim] (LDS, row stride ld) x B (registers), the wave's k chunks; two accumulators like the Linear kernels
template <int NQ>
__device__ __forceinline__ f32x4 tile_mma(const float *panel, const int ld, const int kdim, const float4 (&bq)[NQ], const int ks, const int S)
{
    const int lane = threadIdx.x & 63, r = lane & 15, h = lane >> 4;
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int k = 16 * (ks + S * u) + 4 * h;
        if (16 * (ks + S * u) < kdim) {  // wave-uniform
            float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k < kdim) a = *reinterpret_cast<const float4 *>(panel + r * ld + k);  // kdim % 4 == 0: a quad is inside or outside
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[u].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[u].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[u].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[u].w, acc1, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}

// split-K combine: wave = tile + T * ks; the ks > 0 waves park their partial tile in LDS, the ks == 0 wave of the tile adds them in
// ks order. Every wave of the workgroup calls this (one barrier).
__device__ __forceinline__ f32x4 combine_split_k(f32x4 acc, float *smem, const int T, const int S)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 *part = reinterpret_cast<f32x4 *>(smem + SM_PART);
    if (S > 1) {
        if (wave >= T) part[(wave - T) * 64 + lane] = acc;
        __syncthreads();
        if (wave < T) {
            for (int v = 1; v < S; ++v) acc += part[(wave + T * v - T) * 64 + lane];
        }
    }
    return acc;
}

// The SAC head's sampling arithmetic for ONE row (gaussian_head_gemm_fwd_kernel's expressions): p[0..A) = mean, p[A..2A) = raw
// log_std; eps given or Philox4x32-10 / Box-Muller at counter `ctr`.
struct HeadRow { float a[MAX_A], e[MAX_A], lp; };
__device__ __forceinline__ HeadRow sample_head_row(const float (&p)[2 * MAX_A], const int act_dim, const float *eps_row, const bool philox,
                                                   const uint64_t seed, const uint64_t ctr)
{
    const float half_log_2pi = 0.91893853320467274178f;
    HeadRow o;
    float lp = 0.0f, corr = 0.0f;
#pragma unroll
    for (int j0 = 0; j0 < MAX_A; j0 += 2) {
        if (j0 >= act_dim) break;
        float e[2] = {0.0f, 0.0f};
        if (philox) {
            uint32_t rr[4];
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)(j0 >> 1), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rr);
            box_muller(rr[0], rr[1], e[0], e[1]);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = j0 + jj;
            if (j >= act_dim) break;
            if (!philox) e[jj] = eps_row[j];
            float mu = 0.0f, raw = 0.0f;
#pragma unroll
            for (int q = 0; q < 2 * MAX_A; ++q) {  // selects with static register indices
                if (q == j) mu = p[q];
                if (q == act_dim + j) raw = p[q];
            }
            const float ls = fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX);
            const float sd = expf(ls);
            const float u = mu + sd * e[jj];
            const float a = tanhf(u);
            const float d = u - mu, var = sd * sd;
            lp += -(d * d) / (2.0f * var) - logf(sd) - half_log_2pi;
            corr += logf(1.0f - a * a + 1e-6f);
            o.a[j] = a;
            o.e[j] = e[jj];
        }
    }
    o.lp = lp - corr;
    return o;
}

// ---- SAC actor, forward chain -------------------------------------------------------------------------------------------------
struct ActorFwdArgs {
    cstr_sac_actor_t net;
    cstr_ring_t ring; int64_t *ring_ctl; int advance_ring; uint64_t *rng_ctl; uint64_t rng_advance;
    const int32_t *idx; int batch, tiles;
    float *x_data, *x_pi, *x_next, *out_done, *out_rew, *a_h1, *a_h2, *head_part;
};

template <int NQ>
__global__ __launch_bounds__(CH_THREADS) void sac_actor_chain_fwd_kernel(const ActorFwdArgs a)
{
    extern __shared__ __align__(16) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, h = lane >> 4;
    const int T = a.tiles, S = CH_WAVES / T, tile = wave % T, ks = wave / T;
    const int D = a.net.obs_dim, A = a.net.act_dim, W = D + A, H1 = a.net.h1, H2 = a.net.h2, B = a.batch, M = 2 * B;
    const int n0 = (blockIdx.x * T + tile) * 16, m0 = blockIdx.y * 16;
    float *xs = smem + SM_XS, *red = smem + SM_RED, *panel = smem + SM_PANEL;
    const int ld = H1 + 4;
    // (a) the wide layer's B operand does not depend on anything: requested first
    float4 bq[NQ];
    load_b_fwd<NQ>(bq, a.net.w2, H2, H1, n0, ks, S);
    const int col = n0 + r;
    const float bv = a.net.b2[min(col, H2 - 1)];
    // (b) the row group's 16 input rows: ReplayBuffer.sample's gather (buffers.py:316-323) or the already packed observation columns
    const int qpr = D / 4;  // 16-byte quads per row
    if (t < 16 * qpr) {
        const int row = t / qpr, qd = t % qpr, m = min(m0 + row, M - 1);
        const bool next = m >= B;
        const int b = next ? m - B : m;
        float4 v;
        if (a.idx) {
            const int64_t o = (int64_t)a.idx[b] * a.ring.n_envs + a.idx[B + b];
            v = *reinterpret_cast<const float4 *>((next ? a.ring.next_obs : a.ring.obs) + o * D + 4 * qd);
            if (blockIdx.x == 0 && m0 + row < M) {  // materialise the packed batch for the launches behind this one
                float *xo = (next ? a.x_next : a.x_pi) + (int64_t)b * W + 4 * qd;
                reinterpret_cast<float2 *>(xo)[0] = make_float2(v.x, v.y); reinterpret_cast<float2 *>(xo)[1] = make_float2(v.z, v.w);
                if (!next) {
                    float *xd = a.x_data + (int64_t)b * W;
                    reinterpret_cast<float2 *>(xd + 4 * qd)[0] = make_float2(v.x, v.y); reinterpret_cast<float2 *>(xd + 4 * qd)[1] = make_float2(v.z, v.w);
                    if (qd == 0) {
                        for (int j = 0; j < A; ++j) xd[D + j] = a.ring.act[o * A + j];
                        a.out_done[b] = a.ring.done[o] * (1.0f - a.ring.timeout[o]);  // buffers.py:322
                        a.out_rew[b] = a.ring.rew[o];
                    }
                }
            }
        } else {
            const float2 *src = reinterpret_cast<const float2 *>((next ? a.x_next : a.x_pi) + (int64_t)b * W + 4 * qd);
            const float2 lo = src[0], hi = src[1];
            v = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        *reinterpret_cast<float4 *>(xs + row * 16 + 4 * qd) = v;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0 && a.idx) {  // nobody reads these words in this launch
        if (a.advance_ring) {  // ReplayBuffer.add's epilogue (buffers.py:280-283), left over by the rollout launch
            int64_t pos = a.ring_ctl[0] + 1;
            if (pos == a.ring.rows) { a.ring_ctl[1] = 1; pos = 0; }
            a.ring_ctl[0] = pos;
            a.ring_ctl[3] += 1;
        }
        if (a.rng_ctl) a.rng_ctl[1] += a.rng_advance;
    }
    __syncthreads();
    // (c) layer 1, recomputed by every workgroup of the row group: h1 = relu(x W1^T + b1) -> panel (k ascending fma chain, then + bias)
    for (int c = t; c < H1; c += CH_THREADS) {
        float w[8];
        const float4 w0 = *reinterpret_cast<const float4 *>(a.net.w1 + (int64_t)c * D);
        w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w;
        if (D == 8) {
            const float4 w1v = *reinterpret_cast<const float4 *>(a.net.w1 + (int64_t)c * D + 4);
            w[4] = w1v.x; w[5] = w1v.y; w[6] = w1v.z; w[7] = w1v.w;
        }
        const float bb = a.net.b1[c];
        const bool keep = blockIdx.x == 0 && a.a_h1 != nullptr;
#pragma unroll 4
        for (int row = 0; row < 16; ++row) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < D) acc = __fmaf_rn(xs[row * 16 + k], w[k], acc);
            const float v = fmaxf(acc + bb, 0.0f);
            panel[row * ld + c] = v;
            if (keep && m0 + row < B) a.a_h1[(int64_t)(m0 + row) * H1 + c] = v;
        }
    }
    __syncthreads();
    // (d) layer 2: this wave's 16 x 16 tile (its share of K), (e) split-K combine
    f32x4 acc = tile_mma<NQ>(panel, ld, H1, bq, ks, S);
    acc = combine_split_k(acc, smem, T, S);
    // (f) epilogue of the tile's first wave: bias + ReLU, keep the pi(obs) rows for the backward, head partials over these 16 columns
    if (wave < T) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = col < H2 ? fmaxf(acc[e] + bv, 0.0f) : 0.0f;
            const int m = m0 + 4 * h + e;
            if (col < H2 && m < B && a.a_h2) a.a_h2[(int64_t)m * H2 + col] = v[e];
        }
        for (int j = 0; j < 2 * A; ++j) {
            const float wv = col < H2 ? a.net.hw[(int64_t)j * H2 + col] : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = rowsum16(v[e] * wv);
                if (r == 0) red[(tile * 16 + 4 * h + e) * 8 + j] = s;
            }
        }
    }
    __syncthreads();
    if (t < 16 * 2 * A) {
        const int row = t / (2 * A), j = t % (2 * A);
        float s = red[row * 8 + j];
        for (int tl = 1; tl < T; ++tl) s += red[(tl * 16 + row) * 8 + j];
        if (m0 + row < M) a.head_part[((int64_t)blockIdx.x * M + m0 + row) * (2 * A) + j] = s;
    }
}

// ---- Q networks, forward chain ------------------------------------------------------------------------------------------------
struct QFwdArgs {
    cstr_chain_net_t nets[CSTR_CHAIN_MAX_NETS];
    cstr_sac_head_fin_t fin;
    int n_nets, w_in, h1, h2, batch, tiles, has_fin;
};

template <int NQ>
__global__ __launch_bounds__(CH_THREADS) void q_chain_fwd_kernel(const QFwdArgs a)
{
    extern __shared__ __align__(16) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, h = lane >> 4;
    const int T = a.tiles, S = CH_WAVES / T, tile = wave % T, ks = wave / T;
    const cstr_chain_net_t &net = a.nets[blockIdx.z];
    const int W = a.w_in, H1 = a.h1, H2 = a.h2, B = a.batch;
    const int n0 = (blockIdx.x * T + tile) * 16, m0 = blockIdx.y * 16;
    float *xs = smem + SM_XS, *red = smem + SM_RED, *panel = smem + SM_PANEL;
    const int ld = H1 + 4;
    float4 bq[NQ];
    load_b_fwd<NQ>(bq, net.w2, H2, H1, n0, ks, S);
    const int col = n0 + r;
    const float bv = net.b2[min(col, H2 - 1)];
    const float w3v = col < H2 ? net.w3[col] : 0.0f;
    const int role = a.has_fin ? net.role : CSTR_CHAIN_ROLE_PLAIN;
    const int D = a.has_fin ? a.fin.obs_dim : W;
    // (b) input rows; with a pending actor head the action columns of the pi(next_obs) rows are finalised HERE (every workgroup of
    //     the row group for itself), and the column-group-0 workgroups store what later launches need
    if (t < 16 * W) {
        const int row = t / W, k = t % W;
        if (!(role >= CSTR_CHAIN_ROLE_NEXT && k >= D)) xs[row * 16 + k] = net.x[(int64_t)min(m0 + row, B - 1) * W + k];
    }
    const bool fin_rows = role >= CSTR_CHAIN_ROLE_NEXT || (role == CSTR_CHAIN_ROLE_STORE_PI && blockIdx.x == 0);
    if (fin_rows && t >= 64 && t < 80) {  // 16 lanes of wave 1, a row each (wave 0's low lanes load the inputs)
        const int row = t - 64, b = min(m0 + row, B - 1), A = a.fin.act_dim;
        const bool nxt = role >= CSTR_CHAIN_ROLE_NEXT;
        const int64_t i = (nxt ? B : 0) + b;  // row of the 2B-row actor pass
        float p[2 * MAX_A];
#pragma unroll
        for (int j = 0; j < 2 * MAX_A; ++j) {
            p[j] = 0.0f;
            if (j < 2 * A) {
                float s = a.fin.head_part[i * (2 * A) + j];
                for (int part = 1; part < a.fin.n_parts; ++part) s += a.fin.head_part[((int64_t)part * 2 * B + i) * (2 * A) + j];
                p[j] = s + a.fin.hb[j];
            }
        }
        const bool philox = a.fin.eps_in == nullptr;
        const uint64_t seed = philox ? a.fin.rng_ctl[0] : 0ull, base = philox ? a.fin.rng_ctl[1] : 0ull;
        const HeadRow o = sample_head_row(p, A, philox ? nullptr : a.fin.eps_in + i * A, philox, seed, base + (uint64_t)i);
        const bool store = blockIdx.x == 0 && m0 + row < B && role != CSTR_CHAIN_ROLE_NEXT;
#pragma unroll
        for (int j = 0; j < MAX_A; ++j) {
            if (j >= A) break;
            if (nxt) xs[row * 16 + D + j] = o.a[j];
            if (store) {
                (nxt ? a.fin.x_next : a.fin.x_pi)[(int64_t)b * W + D + j] = o.a[j];
                if (!nxt) {
                    float mu = 0.0f, raw = 0.0f;
#pragma unroll
                    for (int q = 0; q < 2 * MAX_A; ++q) {
                        if (q == j) mu = p[q];
                        if (q == A + j) raw = p[q];
                    }
                    a.fin.params[(int64_t)b * 2 * A + j] = mu;
                    a.fin.params[(int64_t)b * 2 * A + A + j] = raw;
                    a.fin.eps_out[(int64_t)b * A + j] = o.e[j];
                }
            }
        }
        if (store) (nxt ? a.fin.logp_next : a.fin.logp_pi)[b] = o.lp;
    }
    __syncthreads();
    // (c) layer 1 recomputed: K = W <= 12 inputs
    for (int c = t; c < H1; c += CH_THREADS) {
        float w[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) w[k] = k < W ? net.w1[(int64_t)c * W + k] : 0.0f;
        const float bb = net.b1[c];
        const bool keep = blockIdx.x == 0 && net.h1 != nullptr;
#pragma unroll 4
        for (int row = 0; row < 16; ++row) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (k < W) acc = __fmaf_rn(xs[row * 16 + k], w[k], acc);
            const float v = fmaxf(acc + bb, 0.0f);
            panel[row * ld + c] = v;
            if (keep && m0 + row < B) net.h1[(int64_t)(m0 + row) * H1 + c] = v;
        }
    }
    __syncthreads();
    f32x4 acc = tile_mma<NQ>(panel, ld, H1, bq, ks, S);
    acc = combine_split_k(acc, smem, T, S);
    if (wave < T) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = col < H2 ? fmaxf(acc[e] + bv, 0.0f) : 0.0f;
            const int m = m0 + 4 * h + e;
            if (col < H2 && m < B && net.h2) net.h2[(int64_t)m * H2 + col] = v;
            const float s = rowsum16(v * w3v);
            if (r == 0) red[(tile * 16 + 4 * h + e) * 8] = s;
        }
    }
    __syncthreads();
    if (t < 16 && m0 + t < B) {
        float s = red[t * 8];
        for (int tl = 1; tl < T; ++tl) s += red[(tl * 16 + t) * 8];
        net.q_part[(int64_t)blockIdx.x * B + m0 + t] = s;
    }
}

// ---- Q networks, loss root + backward chain -------------------------------------------------------------------------------------
struct QBwdArgs {
    cstr_chain_net_t nets[2];
    cstr_chain_root_t root;
    int n_nets, w_in, obs_dim, h1, h2, tiles;
    float *dz2, *dz1, *gact_part;
};

__device__ __forceinline__ float q_from_parts(const float *part, const float *b3, const int n_parts, const int batch, const int row)
{
    float s = part[row];
    for (int p = 1; p < n_parts; ++p) s += part[(int64_t)p * batch + row];
    return s + b3[0];
}

__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ float block_sum_256(float v, float *sm)
{
    v = wave_sum64(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    const float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}

// the batch reductions of the loss (td_twin_q_loss_kernel / sac_actor_loss_kernel / neg_mean_loss_kernel's expressions and tree)
__device__ void chain_loss_workgroup(const cstr_chain_root_t &rt, float *sm)
{
    const int tid = threadIdx.x, B = rt.batch, P = rt.n_parts;
    const bool with_alpha = rt.mode == 1 && rt.alpha.log_alpha != nullptr;
    const float la = with_alpha ? rt.alpha.log_alpha[0] : 0.0f;
    const float ec = with_alpha ? expf(la) : (rt.ent_coef ? rt.ent_coef[0] : 0.0f);
    const float inv = 1.0f / (float)B;
    if (rt.mode == 1) {
        float a1 = 0.0f, a2 = 0.0f, aa = 0.0f;
        for (int b = tid; b < B; b += 256) {
            float q = fminf(q_from_parts(rt.q_part[2], rt.b3[2], P, B, b), q_from_parts(rt.q_part[3], rt.b3[3], P, B, b));
            if (rt.next_logp) q = q - ec * rt.next_logp[b];
            const float tq = rt.rew[b] + (1.0f - rt.done[b]) * rt.gamma * q;
            const float d1 = q_from_parts(rt.q_part[0], rt.b3[0], P, B, b) - tq, d2 = q_from_parts(rt.q_part[1], rt.b3[1], P, B, b) - tq;
            a1 += d1 * d1;
            a2 += d2 * d2;
            if (with_alpha) aa += rt.alpha.logp_pi[b] + rt.alpha.target_entropy;
        }
        const float s1 = block_sum_256(a1, sm), s2 = block_sum_256(a2, sm);
        const float mean = with_alpha ? block_sum_256(aa, sm) / (float)B : 0.0f;
        if (tid == 0) {
            const float loss = rt.scale * (s1 / (float)B + s2 / (float)B);
            if (rt.loss_out) rt.loss_out[0] = loss;
            if (rt.loss_sum) rt.loss_sum[0] += loss;
            if (with_alpha) {
                rt.alpha.grad_out[0] = -mean;
                rt.alpha.ent_coef_out[0] = ec;
                if (rt.alpha.loss_out) rt.alpha.loss_out[0] = -(la * mean);
                if (rt.alpha.loss_sum) rt.alpha.loss_sum[0] += -(la * mean);
                if (rt.alpha.ent_coef_sum) rt.alpha.ent_coef_sum[0] += ec;
            }
        }
    } else if (rt.mode == 3) {
        float acc = 0.0f;
        for (int b = tid; b < B; b += 256) acc += q_from_parts(rt.q_part[0], rt.b3[0], P, B, b);
        const float sum = block_sum_256(acc, sm);
        if (tid == 0) {
            const float loss = -(sum * inv);
            if (rt.loss_out) rt.loss_out[0] = loss;
            if (rt.loss_sum) rt.loss_sum[0] += loss;
        }
    } else {
        float acc = 0.0f;
        for (int b = tid; b < B; b += 256) {
            const float x = q_from_parts(rt.q_part[0], rt.b3[0], P, B, b), c = q_from_parts(rt.q_part[1], rt.b3[1], P, B, b);
            acc += ec * rt.logp[b] - (x <= c ? x : c);
        }
        const float sum = block_sum_256(acc, sm);
        if (tid == 0) {
            const float loss = sum * inv;
            if (rt.loss_out) rt.loss_out[0] = loss;
            if (rt.loss_sum) rt.loss_sum[0] += loss;
        }
    }
    if (tid == 0 && rt.rng_ctl) rt.rng_ctl[1] += rt.rng_advance;  // nobody reads the offset in this launch
}

template <int NQ>
__global__ __launch_bounds__(CH_THREADS) void q_chain_bwd_kernel(const QBwdArgs a)
{
    extern __shared__ __align__(16) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, h = lane >> 4;
    const cstr_chain_root_t &rt = a.root;
    const int B = rt.batch, H1 = a.h1, H2 = a.h2, W = a.w_in;
    const int g = blockIdx.z;
    if ((int)blockIdx.y == B / 16) {  // the loss workgroup
        if (g == 0 && blockIdx.x == 0) chain_loss_workgroup(rt, smem + SM_RED);
        return;
    }
    const int T = a.tiles, S = CH_WAVES / T, tile = wave % T, ks = wave / T;
    const cstr_chain_net_t &net = a.nets[g];
    const int k0 = (blockIdx.x * T + tile) * 16, m0 = blockIdx.y * 16, col = k0 + r;
    float *red = smem + SM_RED, *gs = smem + SM_GS, *panel = smem + SM_PANEL;
    const int ld = H2 + 4;
    // (a) operands that depend on nothing: the tile's B operand (W2 read along n) and the epilogue's relu'(h1) mask values
    float4 bq[NQ];
    load_b_bwd<NQ>(bq, net.w2, H2, H1, k0, ks, S);
    float ty[4];
    {
        const __amdgpu_buffer_rsrc_t rh = rsrc_of(net.h1, (int64_t)B * H1);
#pragma unroll
        for (int e = 0; e < 4; ++e) ty[e] = ld32(rh, col < H1 ? 4 * ((m0 + 4 * h + e) * H1 + col) : BUF_OOB);
    }
    // (b) d(loss)/dq of this network for the row group's 16 rows (cstr_head_root_t's expressions)
    if (t < 16) {
        const int row = m0 + t, P = rt.n_parts;
        const bool with_alpha = rt.mode == 1 && rt.alpha.log_alpha != nullptr;
        const float ec = with_alpha ? expf(rt.alpha.log_alpha[0]) : (rt.ent_coef ? rt.ent_coef[0] : 0.0f);
        const float kq = rt.scale * 2.0f / (float)B, inv = 1.0f / (float)B;
        float gqv, qown = 0.0f;
        if (rt.mode == 1) {
            float q = fminf(q_from_parts(rt.q_part[2], rt.b3[2], P, B, row), q_from_parts(rt.q_part[3], rt.b3[3], P, B, row));
            if (rt.next_logp) q = q - ec * rt.next_logp[row];
            const float tq = rt.rew[row] + (1.0f - rt.done[row]) * rt.gamma * q;
            qown = q_from_parts(rt.q_part[g], rt.b3[g], P, B, row);
            gqv = kq * (qown - tq);
            if (g == 0 && blockIdx.x == 0 && rt.target_out) rt.target_out[row] = tq;
        } else if (rt.mode == 3) {
            qown = q_from_parts(rt.q_part[0], rt.b3[0], P, B, row);
            gqv = -inv;
        } else {
            const float q1 = q_from_parts(rt.q_part[0], rt.b3[0], P, B, row), q2 = q_from_parts(rt.q_part[1], rt.b3[1], P, B, row);
            const bool first = q1 <= q2;
            qown = g == 0 ? q1 : q2;
            gqv = (first == (g == 0)) ? -inv : 0.0f;
        }
        gs[t] = gqv;
        if (blockIdx.x == 0) {
            if (rt.q_out) rt.q_out[(int64_t)g * B + row] = qown;
            if (rt.gq_out) rt.gq_out[(int64_t)g * B + row] = gqv;
        }
    }
    __syncthreads();
    // (c) dz2 = dq * w3 * relu'(h2), recomputed by every workgroup of the row group into the panel; each stores its share of columns
    {
        const int cpg = (H2 + gridDim.x - 1) / gridDim.x;
        for (int c = t; c < H2; c += CH_THREADS) {
            const float wv = net.w3[c];
            const bool mine = a.dz2 != nullptr && c / cpg == (int)blockIdx.x;
            float y[16];
#pragma unroll
            for (int row = 0; row < 16; ++row) y[row] = net.h2[(int64_t)(m0 + row) * H2 + c];
#pragma unroll
            for (int row = 0; row < 16; ++row) {
                const float d = y[row] > 0.0f ? gs[row] * wv : 0.0f;
                panel[row * ld + c] = d;
                if (mine) a.dz2[((int64_t)g * B + m0 + row) * H2 + c] = d;
            }
        }
    }
    __syncthreads();
    // (d) dz1 tile = (dz2 W2)[rows][k0 .. k0 + 15], (e) split-K combine, (f) * relu'(h1); partial action gradient
    f32x4 acc = tile_mma<NQ>(panel, ld, H2, bq, ks, S);
    acc = combine_split_k(acc, smem, T, S);
    const int A = W - a.obs_dim;
    if (wave < T) {
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d[e] = (col < H1 && ty[e] > 0.0f) ? acc[e] : 0.0f;
            if (a.dz1 && col < H1) a.dz1[((int64_t)g * B + m0 + 4 * h + e) * H1 + col] = d[e];
        }
        if (a.gact_part) {
            for (int j = 0; j < A; ++j) {
                const float wv = col < H1 ? net.w1[(int64_t)col * W + a.obs_dim + j] : 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = rowsum16(d[e] * wv);
                    if (r == 0) red[(tile * 16 + 4 * h + e) * 8 + j] = s;
                }
            }
        }
    }
    if (a.gact_part) {
        __syncthreads();
        if (t < 16 * A) {
            const int row = t / A, j = t % A;
            float s = red[row * 8 + j];
            for (int tl = 1; tl < T; ++tl) s += red[(tl * 16 + row) * 8 + j];
            a.gact_part[(((int64_t)g * gridDim.x + blockIdx.x) * B + m0 + row) * A + j] = s;
        }
    }
}

// ---- SAC actor, backward chain -------------------------------------------------------------------------------------------------
struct ActorBwdArgs {
    cstr_sac_actor_t net;
    const float *gact_part; int n_nets, n_parts; const float *ent_coef;
    const float *x_pi, *params, *eps, *a_h1, *a_h2;
    float *g_params, *dz2, *dz1;
    int batch, tiles;
};

template <int NQ>
__global__ __launch_bounds__(CH_THREADS) void sac_actor_chain_bwd_kernel(const ActorBwdArgs a)
{
    extern __shared__ __align__(16) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, h = lane >> 4;
    const int T = a.tiles, S = CH_WAVES / T, tile = wave % T, ks = wave / T;
    const int D = a.net.obs_dim, A = a.net.act_dim, W = D + A, H1 = a.net.h1, H2 = a.net.h2, B = a.batch;
    const int k0 = (blockIdx.x * T + tile) * 16, m0 = blockIdx.y * 16, col = k0 + r;
    float *gs = smem + SM_GS, *panel = smem + SM_PANEL;
    const int ld = H2 + 4;
    float4 bq[NQ];
    load_b_bwd<NQ>(bq, a.net.w2, H2, H1, k0, ks, S);
    float ty[4];
    {
        const __amdgpu_buffer_rsrc_t rh = rsrc_of(a.a_h1, (int64_t)B * H1);
#pragma unroll
        for (int e = 0; e < 4; ++e) ty[e] = ld32(rh, col < H1 ? 4 * ((m0 + 4 * h + e) * H1 + col) : BUF_OOB);
    }
    // (b) d(loss)/d(action) from the critic's partial sums, then the squashed-Gaussian head's backward (gaussian_head_bwd_kernel's
    //     expressions): g_params[row] = (d/d mean | d/d log_std)
    if (t < 16 * A) {
        const int row = t / A, j = t % A, b = m0 + row;
        float ga = 0.0f;
        for (int g = 0; g < a.n_nets; ++g)
            for (int p = 0; p < a.n_parts; ++p) ga += a.gact_part[(((int64_t)g * a.n_parts + p) * B + b) * A + j];
        const float gl = a.ent_coef[0] * (1.0f / (float)B);  // d(loss)/d(logp) (sac_actor_loss_kernel)
        const float av = a.x_pi[(int64_t)b * W + D + j], raw = a.params[(int64_t)b * 2 * A + A + j];
        const float s = expf(fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX));
        const float one_m = 1.0f - av * av;
        const float gu = ga * one_m + gl * (2.0f * av * one_m / (one_m + 1e-6f));
        const float gls = (raw >= LOG_STD_MIN && raw <= LOG_STD_MAX) ? gu * a.eps[(int64_t)b * A + j] * s - gl : 0.0f;
        gs[row * 8 + j] = gu;
        gs[row * 8 + A + j] = gls;
        if (blockIdx.x == 0) {
            a.g_params[(int64_t)b * 2 * A + j] = gu;
            a.g_params[(int64_t)b * 2 * A + A + j] = gls;
        }
    }
    __syncthreads();
    // (c) dz2 = (g_params hw) * relu'(a_h2): 2A terms per element, j ascending (gaussian_head_bwd_input_kernel's chain)
    {
        const int cpg = (H2 + gridDim.x - 1) / gridDim.x;
        for (int c = t; c < H2; c += CH_THREADS) {
            float wq[2 * MAX_A];
#pragma unroll
            for (int j = 0; j < 2 * MAX_A; ++j) wq[j] = j < 2 * A ? a.net.hw[(int64_t)j * H2 + c] : 0.0f;
            const bool mine = c / cpg == (int)blockIdx.x;
            float y[16];
#pragma unroll
            for (int row = 0; row < 16; ++row) y[row] = a.a_h2[(int64_t)(m0 + row) * H2 + c];
#pragma unroll
            for (int row = 0; row < 16; ++row) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 2 * MAX_A; ++j)
                    if (j < 2 * A) acc = __fmaf_rn(gs[row * 8 + j], wq[j], acc);
                const float d = y[row] > 0.0f ? acc : 0.0f;
                panel[row * ld + c] = d;
                if (mine) a.dz2[(int64_t)(m0 + row) * H2 + c] = d;
            }
        }
    }
    __syncthreads();
    f32x4 acc = tile_mma<NQ>(panel, ld, H2, bq, ks, S);
    acc = combine_split_k(acc, smem, T, S);
    if (wave < T && col < H1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a.dz1[(int64_t)(m0 + 4 * h + e) * H1 + col] = ty[e] > 0.0f ? acc[e] : 0.0f;
    }
}

// chunks of 16 along the reduction a wave owns, rounded up to an instantiated size
static int nq_for(int kdim, int tiles)
{
    const int s = CH_WAVES / tiles, nch = (kdim + 15) / 16, per = (nch + s - 1) / s;
    return per <= 4 ? 4 : per <= 8 ? 8 : per <= 16 ? 16 : 32;
}

static bool chain_dims_ok(int h1, int h2, int64_t batch, int tiles)
{
    return h1 >= 16 && h2 >= 16 && h1 <= CSTR_CHAIN_MAX_WIDTH && h2 <= CSTR_CHAIN_MAX_WIDTH && h1 % 4 == 0 && h2 % 4 == 0 && batch >= 16 &&
           batch <= 1024 && batch % 16 == 0 && (tiles == 1 || tiles == 2 || tiles == 4);
}

#define CHAIN_DISPATCH(KERNEL, NQV, GRID, LDS, STREAM, ARGS)                                   \
    do {                                                                                       \
        if ((NQV) == 4) KERNEL<4><<<GRID, CH_THREADS, LDS, STREAM>>>(ARGS);                    \
        else if ((NQV) == 8) KERNEL<8><<<GRID, CH_THREADS, LDS, STREAM>>>(ARGS);               \
        else if ((NQV) == 16) KERNEL<16><<<GRID, CH_THREADS, LDS, STREAM>>>(ARGS);             \
        else KERNEL<32><<<GRID, CH_THREADS, LDS, STREAM>>>(ARGS);                              \
    } while (0)

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------------------------------------

static int check_actor(const cstr_sac_actor_t *n)
{
    if (!n || !n->w1 || !n->b1 || !n->w2 || !n->b2 || !n->hw || !n->hb) return CSTR_E_BADARG;
    if (!((n->obs_dim == 4 && n->act_dim == 2) || (n->obs_dim == 8 && (n->act_dim == 2 || n->act_dim == 4)))) return CSTR_E_UNSUPPORTED;
    if (!aligned16(n->w1) || !aligned16(n->w2)) return CSTR_E_BADARG;
    return CSTR_OK;
}

extern "C" int cstr_sac_actor_chain_fwd_f32(const cstr_sac_actor_t *actor, const cstr_ring_t *ring, int64_t *ring_ctl, int advance_ring,
                                            uint64_t *rollout_rng_ctl, uint64_t rollout_rng_advance, const int32_t *sample_idx, int64_t batch,
                                            float *x_data, float *x_pi, float *x_next, float *out_done, float *out_rew, float *a_h1,
                                            float *a_h2, float *head_part, int tiles, cstr_stream_t stream)
{
    const int rc = check_actor(actor);
    if (rc) return rc;
    if (!x_pi || !x_next || !head_part) return CSTR_E_BADARG;
    if (!chain_dims_ok(actor->h1, actor->h2, batch, tiles)) return CSTR_E_UNSUPPORTED;
    if (!aligned8(x_pi) || !aligned8(x_next)) return CSTR_E_BADARG;
    ActorFwdArgs a = {};
    a.net = *actor;
    if (sample_idx) {
        if (!ring || !ring->obs || !ring->next_obs || !ring->act || !ring->rew || !ring->done || !ring->timeout || !x_data || !out_done || !out_rew)
            return CSTR_E_BADARG;
        if (ring->obs_dim != actor->obs_dim || ring->act_dim != actor->act_dim) return CSTR_E_BADARG;
        if (advance_ring && !ring_ctl) return CSTR_E_BADARG;
        if (!aligned16(ring->obs) || !aligned16(ring->next_obs) || !aligned8(x_data)) return CSTR_E_BADARG;
        if (ring->rows >= 0xFFFFFFFFLL || ring->n_envs >= 0xFFFFFFFFLL) return CSTR_E_UNSUPPORTED;
        a.ring = *ring;
    }
    a.ring_ctl = ring_ctl; a.advance_ring = advance_ring; a.rng_ctl = rollout_rng_ctl; a.rng_advance = rollout_rng_advance;
    a.idx = sample_idx; a.batch = (int)batch; a.tiles = tiles;
    a.x_data = x_data; a.x_pi = x_pi; a.x_next = x_next; a.out_done = out_done; a.out_rew = out_rew;
    a.a_h1 = a_h1; a.a_h2 = a_h2; a.head_part = head_part;
    const dim3 grid((unsigned)((actor->h2 + 16 * tiles - 1) / (16 * tiles)), (unsigned)(2 * batch / 16));
    const size_t lds = chain_lds_bytes(actor->h1);
    const int nq = nq_for(actor->h1, tiles);
    hipStream_t s = (hipStream_t)stream;
    CHAIN_DISPATCH(sac_actor_chain_fwd_kernel, nq, grid, lds, s, a);
    return (int)hipGetLastError();
}

extern "C" int cstr_q_chain_fwd_f32(const cstr_chain_net_t *nets, int n_nets, int w_in, int h1, int h2, int64_t batch,
                                    const cstr_sac_head_fin_t *fin, int tiles, cstr_stream_t stream)
{
    if (!nets || n_nets < 1 || n_nets > CSTR_CHAIN_MAX_NETS || w_in < 1 || w_in > 12) return CSTR_E_BADARG;
    if (!chain_dims_ok(h1, h2, batch, tiles)) return CSTR_E_UNSUPPORTED;
    QFwdArgs a = {};
    for (int g = 0; g < n_nets; ++g) {
        const cstr_chain_net_t &n = nets[g];
        if (!n.w1 || !n.b1 || !n.w2 || !n.b2 || !n.w3 || !n.b3 || !n.x || !n.q_part || !aligned16(n.w2)) return CSTR_E_BADARG;
        if (n.role < 0 || n.role > 3 || (n.role != 0 && !fin)) return CSTR_E_BADARG;
        a.nets[g] = n;
    }
    if (fin) {
        if (!fin->head_part || !fin->hb || fin->n_parts < 1 || fin->act_dim < 1 || fin->act_dim > MAX_A || fin->obs_dim + fin->act_dim != w_in)
            return CSTR_E_BADARG;
        if (!fin->eps_in && !fin->rng_ctl) return CSTR_E_BADARG;
        if (!fin->x_pi || !fin->x_next || !fin->params || !fin->eps_out || !fin->logp_pi || !fin->logp_next) return CSTR_E_BADARG;
        a.fin = *fin;
        a.has_fin = 1;
    }
    a.n_nets = n_nets; a.w_in = w_in; a.h1 = h1; a.h2 = h2; a.batch = (int)batch; a.tiles = tiles;
    const dim3 grid((unsigned)((h2 + 16 * tiles - 1) / (16 * tiles)), (unsigned)(batch / 16), (unsigned)n_nets);
    const size_t lds = chain_lds_bytes(h1);
    const int nq = nq_for(h1, tiles);
    hipStream_t s = (hipStream_t)stream;
    CHAIN_DISPATCH(q_chain_fwd_kernel, nq, grid, lds, s, a);
    return (int)hipGetLastError();
}

extern "C" int cstr_q_chain_bwd_f32(const cstr_chain_net_t *nets, int n_nets, const cstr_chain_root_t *root, int w_in, int obs_dim, int h1,
                                    int h2, float *dz2, float *dz1, float *gact_part, int tiles, cstr_stream_t stream)
{
    if (!nets || !root || n_nets < 1 || n_nets > 2 || w_in < 1 || w_in > 12 || obs_dim < 0 || obs_dim > w_in) return CSTR_E_BADARG;
    if (root->mode < 1 || root->mode > 3 || root->n_parts < 1) return CSTR_E_BADARG;
    if ((root->mode == 3) != (n_nets == 1)) return CSTR_E_BADARG;
    if (!chain_dims_ok(h1, h2, root->batch, tiles)) return CSTR_E_UNSUPPORTED;
    if (w_in - obs_dim > MAX_A && gact_part) return CSTR_E_UNSUPPORTED;
    QBwdArgs a = {};
    for (int g = 0; g < n_nets; ++g) {
        const cstr_chain_net_t &n = nets[g];
        if (!n.w1 || !n.w2 || !n.w3 || !n.h1 || !n.h2) return CSTR_E_BADARG;
        a.nets[g] = n;
    }
    const int need = root->mode == 1 ? 4 : n_nets;
    for (int g = 0; g < need; ++g)
        if (!root->q_part[g] || !root->b3[g]) return CSTR_E_BADARG;
    if (root->mode == 1 && (!root->rew || !root->done)) return CSTR_E_BADARG;
    if (root->mode == 2 && (!root->logp || !root->ent_coef)) return CSTR_E_BADARG;
    if (root->mode == 1 && root->alpha.log_alpha && (!root->alpha.logp_pi || !root->alpha.grad_out || !root->alpha.ent_coef_out)) return CSTR_E_BADARG;
    a.root = *root;
    a.n_nets = n_nets; a.w_in = w_in; a.obs_dim = obs_dim; a.h1 = h1; a.h2 = h2; a.tiles = tiles;
    a.dz2 = dz2; a.dz1 = dz1; a.gact_part = gact_part;
    const dim3 grid((unsigned)((h1 + 16 * tiles - 1) / (16 * tiles)), (unsigned)(root->batch / 16 + 1), (unsigned)n_nets);
    const size_t lds = chain_lds_bytes(h2);
    const int nq = nq_for(h2, tiles);
    hipStream_t s = (hipStream_t)stream;
    CHAIN_DISPATCH(q_chain_bwd_kernel, nq, grid, lds, s, a);
    return (int)hipGetLastError();
}

extern "C" int cstr_sac_actor_chain_bwd_f32(const cstr_sac_actor_t *actor, const float *gact_part, int n_nets, int n_parts,
                                            const float *ent_coef, const float *x_pi, const float *params, const float *eps, const float *a_h1,
                                            const float *a_h2, float *g_params, float *dz2, float *dz1, int64_t batch, int tiles,
                                            cstr_stream_t stream)
{
    const int rc = check_actor(actor);
    if (rc) return rc;
    if (!gact_part || n_nets < 1 || n_nets > 2 || n_parts < 1 || !ent_coef || !x_pi || !params || !eps || !a_h1 || !a_h2 || !g_params || !dz2 || !dz1)
        return CSTR_E_BADARG;
    if (!chain_dims_ok(actor->h1, actor->h2, batch, tiles)) return CSTR_E_UNSUPPORTED;
    ActorBwdArgs a = {};
    a.net = *actor;
    a.gact_part = gact_part; a.n_nets = n_nets; a.n_parts = n_parts; a.ent_coef = ent_coef;
    a.x_pi = x_pi; a.params = params; a.eps = eps; a.a_h1 = a_h1; a.a_h2 = a_h2;
    a.g_params = g_params; a.dz2 = dz2; a.dz1 = dz1; a.batch = (int)batch; a.tiles = tiles;
    const dim3 grid((unsigned)((actor->h1 + 16 * tiles - 1) / (16 * tiles)), (unsigned)(batch / 16));
    const size_t lds = chain_lds_bytes(actor->h2);
    const int nq = nq_for(actor->h2, tiles);
    hipStream_t s = (hipStream_t)stream;
    CHAIN_DISPATCH(sac_actor_chain_bwd_kernel, nq, grid, lds, s, a);
    return (int)hipGetLastError();
}
