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
static size_t chain_lds_bytes(int k, int w1_pad) { return (size_t)(SM_PANEL + 16 * (k + 4) + w1_pad * k) * sizeof(float); }

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

// WX: the class-default hidden widths as compile-time constants -- 1 = 256 x 256 (SAC), 2 = 400 x 300 (TD3 / DDPG / MADDPG), 0 = run-time
// widths. With run-time widths and tiles the kernels' loop bounds, split-K shares and operand offsets are hoisted into dozens of
// scalars that overflow the SGPR file (v_readlane / v_writelane traffic) and every trip count is a compare-and-branch; the SAC iteration
// at the class defaults went 0.0662 -> 0.0613 ms with them constant (A/B on MI355X, profiles/r03_notes.md section 10). The column
// groups per workgroup follow from the per-wave share NQ the host picked: 256 / 16 = 16 chunks over S = 4 / T waves -> NQ = 4 T;
// 400 (25 chunks) or 300 (19 chunks) over S waves, rounded up to the instantiated share -> NQ = 8 T.
template <int WX> __device__ __forceinline__ constexpr int exact_h1() { return WX == 1 ? 256 : 400; }
template <int WX> __device__ __forceinline__ constexpr int exact_h2() { return WX == 1 ? 256 : 300; }
template <int NQ, int WX> __device__ __forceinline__ constexpr int exact_tiles() { return WX == 1 ? NQ / 4 : (NQ >= 8 ? NQ / 8 : 1); }  // (400 x 300 with a share of 4 is never dispatched)

#ifdef CSTR_CHAIN_STAMPS  // diagnostic build only (make diag): s_memtime per wave at the phase boundaries, read by tools/chain_stamps.py
__device__ unsigned long long chain_stamps[4 * 1024 * 4 * 8];  // [kernel][workgroup (linear, < 1024)][wave][stamp]
#define CH_STAMP(K, I) do { const unsigned wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; \
        if ((threadIdx.x & 63) == 0 && wg_ < 1024) chain_stamps[(((K) * 1024 + wg_) * 4 + (threadIdx.x >> 6)) * 8 + (I)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CH_STAMP(K, I) do { } while (0)
#endif

// LDS-only workgroup barrier: waits for this wave's LDS traffic (lgkmcnt), NOT for its outstanding global loads / stores -- the
// chain kernels keep operand loads in flight across their phase boundaries and issue their global stores at the very end.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// sum over the 16 lanes that share lane >> 4 (the 16 columns of an MFMA C/D tile row): four DPP row rotations (full-rate VALU, no
// LDS crossbar); every lane ends with the total of its row
__device__ __forceinline__ float rowsum16(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));  // row_ror:1
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
    // FOUR independent accumulator chains (a dependent MFMA waits ~its own issue time again for the previous result): the wave's
    // 4 NQ MFMAs are 4 chains of NQ instead of 2 of 2 NQ
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    float4 av[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {  // the A quads first: their LDS round trips overlap
        const int k = 16 * (ks + S * u) + 4 * h;
        av[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (k < kdim) av[u] = *reinterpret_cast<const float4 *>(panel + r * ld + k);  // kdim % 4 == 0: a quad is inside or outside
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        if (16 * (ks + S * u) < kdim) {  // wave-uniform
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bq[u].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bq[u].y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bq[u].z, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bq[u].w, acc3, 0, 0, 0);
        }
    }
    return (acc0 + acc1) + (acc2 + acc3);
}

// split-K combine: wave = tile + T * ks; the ks > 0 waves park their partial tile in LDS, the ks == 0 wave of the tile adds them in
// ks order. Every wave of the workgroup calls this (one LDS barrier).
__device__ __forceinline__ f32x4 combine_split_k(f32x4 acc, float *smem, const int T, const int S)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 *part = reinterpret_cast<f32x4 *>(smem + SM_PART);
    if (S > 1) {
        if (wave >= T) part[(wave - T) * 64 + lane] = acc;
        lds_barrier();
        if (wave < T) {
            for (int v = 1; v < S; ++v) acc += part[(wave + T * v - T) * 64 + lane];
        }
    }
    return acc;
}

// Sum of `n` partial values part[p * stride + at], p ascending: ALL loads are requested before the first add (a loop of load / wait /
// add would serialise n round trips to a freshly written buffer, the longest stretch of these kernels' prologues). n <= 16 takes the
// branch-free form (lanes beyond n read zeros through the descriptor's range check: x + 0 is exact).
constexpr int MAX_PARTS = 16;
template <int NP>
__device__ __forceinline__ float sum_parts_n(const __amdgpu_buffer_rsrc_t rs, const int n, const int64_t stride, const int64_t at)
{
    float v[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) v[p] = ld32(rs, p < n ? (int)(4 * (p * stride + at)) : BUF_OOB);
    float s = v[0];
#pragma unroll
    for (int p = 1; p < NP; ++p) s += v[p];
    return s;
}

// the same in two steps, so that a kernel can REQUEST its partials first thing, issue its other operand requests behind them and add
// the partials up when they are needed: n <= 8 in registers, more through sum_parts at `sum()` time
struct PartBatch {
    float v[8];
    const float *part; int n; int64_t stride, at, total;
    __device__ __forceinline__ void request(const float *part_, const int n_, const int64_t stride_, const int64_t at_, const int64_t total_)
    {
        part = part_; n = n_; stride = stride_; at = at_; total = total_;
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(part, total);
#pragma unroll
        for (int p = 0; p < 8; ++p) v[p] = ld32(rs, (p < n && n <= 8) ? (int)(4 * (p * stride + at)) : BUF_OOB);
    }
    __device__ __forceinline__ float sum() const;
};

__device__ __forceinline__ float sum_parts(const float *part, const int n, const int64_t stride, const int64_t at, const int64_t total_floats)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(part, total_floats);
    if (n <= 4) return sum_parts_n<4>(rs, n, stride, at);  // (n is launch-uniform)
    if (n <= 8) return sum_parts_n<8>(rs, n, stride, at);
    if (n <= MAX_PARTS) return sum_parts_n<MAX_PARTS>(rs, n, stride, at);
    float s = part[at];
    for (int p = 1; p < n; ++p) s += part[p * stride + at];
    return s;
}

__device__ __forceinline__ float PartBatch::sum() const
{
    if (n > 8) return sum_parts(part, n, stride, at, total);
    float s = v[0];
#pragma unroll
    for (int p = 1; p < 8; ++p) s += v[p];
    return s;
}

// The SAC head's sampling arithmetic for ONE (row, action) (gaussian_head_gemm_fwd_kernel's expressions): mean, raw log_std, eps ->
// action and this action's log-prob term (Normal log-pdf minus the tanh correction)
__device__ __forceinline__ void sample_action(const float mu, const float raw, const float e, float &a, float &lp_term)
{
    const float half_log_2pi = 0.91893853320467274178f;
    const float ls = fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX);
    const float sd = expf(ls);
    const float u = mu + sd * e;
    a = tanhf(u);
    const float d = u - mu, var = sd * sd;
    lp_term = (-(d * d) / (2.0f * var) - logf(sd) - half_log_2pi) - logf(1.0f - a * a + 1e-6f);
}

// ---- first layer of a chain on the matrix cores -----------------------------------------------------------------------------------
// h1 = relu(x W1^T + b1) for the row group's 16 rows, K = KIN <= 12 inputs: ONE 16-wide k chunk. The inputs sit in xs[16][16]
// (LDS; column KIN holds 1.0, the rest zeros) and W1 is staged as w1s[H1][16] with b1 in column KIN, so the bias rides in the k
// chain and a 16 x 16 tile is 4 MFMAs + 4 ReLU + 4 LDS stores; the wave takes the tiles wave, wave + 4, ...
// (inputs + the bias column that fit in 8 floats -- obs 4 / act 2 -- are staged 8 wide: half the LDS, so more workgroups per CU;
// the upper half of the 16-wide k chunk is zero in registers)
// Inputs whose rows are whole 16-byte quads (obs 4, obs 8, obs 8 + act 4 = 12) need no staging at all: the B operand quads come
// straight from W1 in L2 and the bias is added in the tile's epilogue (l1_pad() == 0: no LDS image -- 25 KB less per workgroup at
// H1 = 400, i.e. twice as many workgroups per CU).
// Round 3: rows of an EVEN number of inputs that is not a multiple of 4 (obs 4 + act 2 = 6: every critic of SAC / TD3; 10) are read
// straight from W1 as well, as 8-byte pairs, with b1 in the k slot behind the last input exactly as the staged image had it (same MFMA
// operands, same bits): no staging pass through LDS in the prologue (global load -> LDS store -> barrier -> LDS load), no LDS image.
template <int KIN>
constexpr int l1_pad() { return (KIN % 2 == 0 && KIN + 1 <= 16) ? 0 : (KIN + 1 <= 8 ? 8 : 16); }
template <int KIN>
constexpr bool l1_bias_in_k() { return KIN % 4 != 0; }  // the bias rides in the k chain (column KIN of the chunk; x holds 1.0 there)

// lane (r, h)'s quad k = 4h .. 4h + 3 of W1 row n for the direct form, and the bias the tile's epilogue adds (0 when it rides in k)
template <int KIN>
__device__ __forceinline__ float4 l1_direct_quad(const float *w1, const float *b1, const int n, const int h, float &bias)
{
    if (KIN % 4 == 0) {
        bias = b1[n];
        return 4 * h < KIN ? *reinterpret_cast<const float4 *>(w1 + (int64_t)n * KIN + 4 * h) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    bias = 0.0f;
    const int k0 = 4 * h;
    float2 lo = make_float2(0.0f, 0.0f), hi = lo;
    if (k0 + 1 < KIN) lo = *reinterpret_cast<const float2 *>(w1 + (int64_t)n * KIN + k0);
    else if (k0 == KIN) lo = make_float2(b1[n], 0.0f);
    if (k0 + 3 < KIN) hi = *reinterpret_cast<const float2 *>(w1 + (int64_t)n * KIN + k0 + 2);
    else if (k0 + 2 == KIN) hi = make_float2(b1[n], 0.0f);
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

template <int KIN>
__device__ __forceinline__ void stage_w1(float *w1s, const float *w1, const float *b1, const int h1)
{
    constexpr int KP = l1_pad<KIN>();
    if (KP == 0) return;  // direct form: nothing to stage
    for (int c = threadIdx.x; c < h1; c += CH_THREADS) {
        float w[KP > 0 ? KP : 4];
#pragma unroll
        for (int k = 0; k < KP; ++k) w[k] = k < KIN ? w1[(int64_t)c * KIN + k] : 0.0f;
        w[KP > 0 ? KIN : 0] = b1[c];
#pragma unroll
        for (int q = 0; q < KP / 4; ++q) *reinterpret_cast<float4 *>(w1s + c * KP + 4 * q) = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    }
}

// direct form (KP == 0): the wave's first four tiles' B quads and biases, requested at kernel entry
template <int KIN>
struct L1Direct {
    float4 wb[4];
    float bias[4];
    __device__ __forceinline__ void request(const float *w1, const float *b1, const int h1)
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, h = lane >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) wb[i] = l1_direct_quad<KIN>(w1, b1, min(16 * (wave + CH_WAVES * i) + r, h1 - 1), h, bias[i]);
    }
};

template <int KP, int KIN>
__device__ __forceinline__ void layer1_mfma(const float *xs, const float *w1s, float *panel, const int ld, const int h1,
                                            const L1Direct<KIN> *dir = nullptr, const float *w1 = nullptr, const float *b1 = nullptr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, h = lane >> 4;
    const bool hk = KP == 0 ? 4 * h < KIN + (l1_bias_in_k<KIN>() ? 1 : 0) : 4 * h < KP;  // this lane's quad of the k chunk exists
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 xa = hk ? *reinterpret_cast<const float4 *>(xs + r * 16 + 4 * h) : zero4;
    const int n_tiles = (h1 + 15) / 16;
    for (int t0 = wave; t0 < n_tiles; t0 += 4 * CH_WAVES) {  // four tiles in flight: their LDS reads, MFMA chains and stores interleave
        float4 wb[4];
        float bb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KP == 0) {  // direct: requested at entry (first group) or here (wider layers)
                if (t0 == wave) { wb[i] = dir->wb[i]; bb[i] = dir->bias[i]; }
                else wb[i] = l1_direct_quad<KIN>(w1, b1, min(16 * (t0 + CH_WAVES * i) + r, h1 - 1), h, bb[i]);
            } else {
                wb[i] = hk ? *reinterpret_cast<const float4 *>(w1s + min(16 * (t0 + CH_WAVES * i) + r, h1 - 1) * KP + 4 * h) : zero4;
            }
        }
        f32x4 acc0[4], acc1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc0[i] = {0.0f, 0.0f, 0.0f, 0.0f};
            acc1[i] = acc0[i];
            acc0[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, wb[i].x, acc0[i], 0, 0, 0);
            acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, wb[i].y, acc1[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc0[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, wb[i].z, acc0[i], 0, 0, 0);
            acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, wb[i].w, acc1[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tl = t0 + CH_WAVES * i;
            const f32x4 acc = acc0[i] + acc1[i];
            if (tl < n_tiles && 16 * tl + r < h1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) panel[(4 * h + e) * ld + 16 * tl + r] = fmaxf(acc[e] + bb[i], 0.0f);
            }
        }
    }
}

// the 16 x width panel (LDS) -> out[(m0 + row)][c]: the chain's "keep for the backward" store, issued at the END (16-byte quads)
__device__ __forceinline__ void store_panel(const float *panel, const int ld, float *out, const int width, const int m0)
{
    const int qpr = width / 4;
    for (int i = threadIdx.x; i < 16 * qpr; i += CH_THREADS) {
        const int row = i / qpr, q = i % qpr;
        *reinterpret_cast<float4 *>(out + (int64_t)(m0 + row) * width + 4 * q) = *reinterpret_cast<const float4 *>(panel + row * ld + 4 * q);
    }
}

// ---- SAC actor, forward chain -------------------------------------------------------------------------------------------------
struct ActorFwdArgs {
    cstr_sac_actor_t net;
    cstr_ring_t ring; int64_t *ring_ctl; int advance_ring;
    const int32_t *idx; int batch, tiles;
    float *x_data, *x_pi, *x_next, *out_done, *out_rew, *a_h1, *a_h2, *head_part;
    const uint64_t *head_rng_ctl; uint64_t head_rng_offset; float *eps_all;
    int rows_mode, head_n;  // CSTR_CHAIN_ROWS_*; head outputs: 2A (mu | log_std) or A (deterministic actor)
};

template <int D, int A, int NQ, int WX>
__global__ __launch_bounds__(CH_THREADS) void sac_actor_chain_fwd_kernel(const ActorFwdArgs a)
{
    extern __shared__ __align__(16) float smem[];
    constexpr int W = D + A, QPR = D / 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, h = lane >> 4;
    const int T = WX ? exact_tiles<NQ, WX>() : a.tiles, S = CH_WAVES / T, tile = wave % T, ks = wave / T;
    const int H1 = WX ? exact_h1<WX>() : a.net.h1, H2 = WX ? exact_h2<WX>() : a.net.h2, B = a.batch, mode = a.rows_mode, HN = a.head_n;
    const int M = mode == CSTR_CHAIN_ROWS_PAIR ? 2 * B : B;
    const int n0 = (blockIdx.x * T + tile) * 16, m0 = blockIdx.y * 16;
    float *xs = smem + SM_XS, *red = smem + SM_RED, *panel = smem + SM_PANEL;
    const int ld = H1 + 4;
    float *w1s = panel + 16 * ld;
    const bool col_ok = n0 + r < H2;
    CH_STAMP(0, 0);
    // (a) everything that depends on nothing is requested NOW: the wide layer's B operand, the epilogue's bias and head weights, W1
    float4 bq[NQ];
    load_b_fwd<NQ>(bq, a.net.w2, H2, H1, n0, ks, S);
    const int col = min(n0 + r, H2 - 1);
    const float bv = a.net.b2[col];
    float hwv[2 * A];
#pragma unroll
    for (int j = 0; j < 2 * A; ++j) hwv[j] = j < HN ? a.net.hw[(int64_t)j * H2 + col] : 0.0f;
    stage_w1<D>(w1s, a.net.w1, a.net.b1, H1);
    L1Direct<D> l1d;
    if (l1_pad<D>() == 0) l1d.request(a.net.w1, a.net.b1, H1);
    // (b) the row group's 16 input rows: ReplayBuffer.sample's gather (buffers.py:316-323) or the already packed observation columns
    const bool mat = a.idx != nullptr && blockIdx.x == 0;  // this workgroup also materialises the packed batch (stores at the end)
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f), vo = v;
    float actv[A], dn = 0.0f, to = 0.0f, rw = 0.0f;
#pragma unroll
    for (int j = 0; j < A; ++j) actv[j] = 0.0f;
    int b = 0;
    bool next = false;
    // rows of the pass: PAIR = [obs rows | next_obs rows] (SAC), NEXT = next_obs rows (a target actor), OBS = obs rows (the actor loss)
    const bool data_row = mode != CSTR_CHAIN_ROWS_PAIR;  // (with a gather) every row also writes its transition's x_data / reward / done
    if (t < 16 * QPR) {
        const int row = t / QPR, qd = t % QPR, m = m0 + row;
        next = mode == CSTR_CHAIN_ROWS_PAIR ? m >= B : mode == CSTR_CHAIN_ROWS_NEXT;
        b = (mode == CSTR_CHAIN_ROWS_PAIR && next) ? m - B : m;
        if (a.idx) {
            const int64_t o = (int64_t)a.idx[b] * a.ring.n_envs + a.idx[B + b];
            v = *reinterpret_cast<const float4 *>((next ? a.ring.next_obs : a.ring.obs) + o * D + 4 * qd);
            if (mat && data_row && next) vo = *reinterpret_cast<const float4 *>(a.ring.obs + o * D + 4 * qd);
            if (mat && (!next || data_row) && qd == 0) {
#pragma unroll
                for (int j = 0; j < A; ++j) actv[j] = a.ring.act[o * A + j];
                dn = a.ring.done[o]; to = a.ring.timeout[o]; rw = a.ring.rew[o];
            }
        } else {
            const float2 *src = reinterpret_cast<const float2 *>((next ? a.x_next : a.x_pi) + (int64_t)b * W + 4 * qd);
            const float2 lo = src[0], hi = src[1];
            v = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        *reinterpret_cast<float4 *>(xs + row * 16 + 4 * qd) = v;
    } else if (t >= 64 && t < 64 + 16 * (4 - QPR)) {  // the padding quads of xs: 1.0 in column D (the bias input), zeros behind it
        const int i = t - 64, row = i / (4 - QPR), qd = QPR + i % (4 - QPR);
        *reinterpret_cast<float4 *>(xs + row * 16 + 4 * qd) = make_float4(qd == QPR ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f);
    }
    // the noise of the row group's sampling head: counter-based, independent of the network -- drawn by an otherwise idle wave while the
    // gather is in flight, consumed by the Q chain launch that finalises the head (counter = offset + row of the 2B-row pass)
    constexpr int PAIRS = (A + 1) / 2;
    float e0 = 0.0f, e1 = 0.0f;
    const bool noise = a.eps_all != nullptr && blockIdx.x == 0 && wave == 3 && lane < 16 * PAIRS;
    if (noise) {
        const int row = lane / PAIRS, pr = lane % PAIRS;
        const uint64_t seed = a.head_rng_ctl[0], ctr = a.head_rng_ctl[1] + a.head_rng_offset + (uint64_t)(m0 + row);
        uint32_t rr[4];
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)pr, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rr);
        box_muller(rr[0], rr[1], e0, e1);
    }
    CH_STAMP(0, 1);
    lds_barrier();
    CH_STAMP(0, 2);
    // (c) layer 1, recomputed by every workgroup of the row group -> panel
    layer1_mfma<l1_pad<D>(), D>(xs, w1s, panel, ld, H1, &l1d, a.net.w1, a.net.b1);
    lds_barrier();
    CH_STAMP(0, 3);
    // (d) layer 2: this wave's 16 x 16 tile (its share of K), (e) split-K combine
    f32x4 acc = tile_mma<NQ>(panel, ld, H1, bq, ks, S);
    CH_STAMP(0, 4);
    acc = combine_split_k(acc, smem, T, S);
    CH_STAMP(0, 5);
    // (f) epilogue of the tile's first wave: bias + ReLU, head partials over these 16 columns
    float hv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (wave < T) {
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = col_ok ? fmaxf(acc[e] + bv, 0.0f) : 0.0f;
#pragma unroll
        for (int j = 0; j < 2 * A; ++j) {
            if (j < HN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = rowsum16(hv[e] * hwv[j]);
                    if (r == 0) red[(tile * 16 + 4 * h + e) * 8 + j] = s;
                }
            }
        }
    }
    lds_barrier();
    CH_STAMP(0, 6);
    if (t < 16 * HN) {
        const int row = t / HN, j = t % HN;
        float s = red[row * 8 + j];
        for (int tl = 1; tl < T; ++tl) s += red[(tl * 16 + row) * 8 + j];
        a.head_part[((int64_t)blockIdx.x * M + m0 + row) * HN + j] = s;
    }
    // ---- global stores, all behind the last barrier ----
    const bool keep_rows = mode == CSTR_CHAIN_ROWS_OBS || (mode == CSTR_CHAIN_ROWS_PAIR && m0 < B);  // the rows a backward follows
    if (wave < T && col_ok && a.a_h2 && keep_rows) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a.a_h2[(int64_t)(m0 + 4 * h + e) * H2 + n0 + r] = hv[e];
    }
    if (blockIdx.x == 0 && a.a_h1 && keep_rows) store_panel(panel, ld, a.a_h1, H1, m0);
    if (noise) {
        const int row = lane / PAIRS, pr = lane % PAIRS;
        float *eo = a.eps_all + (int64_t)(m0 + row) * A + 2 * pr;
        eo[0] = e0;
        if (2 * pr + 1 < A) eo[1] = e1;
    }
    if (mat && t < 16 * QPR) {  // the packed batch for the launches behind this one (cstr_linear_act_fwd_gather_f32's contract)
        const int qd = t % QPR;
        float *xo = (next ? a.x_next : a.x_pi) + (int64_t)b * W + 4 * qd;
        reinterpret_cast<float2 *>(xo)[0] = make_float2(v.x, v.y); reinterpret_cast<float2 *>(xo)[1] = make_float2(v.z, v.w);
        if (data_row && next) {  // a next_obs-only pass: this row's observation goes to x_data and x_pi as well
            if (a.x_pi) {
                float *xp = a.x_pi + (int64_t)b * W + 4 * qd;
                reinterpret_cast<float2 *>(xp)[0] = make_float2(vo.x, vo.y); reinterpret_cast<float2 *>(xp)[1] = make_float2(vo.z, vo.w);
            }
            v = vo;
        }
        if (!next || data_row) {
            float *xd = a.x_data + (int64_t)b * W;
            reinterpret_cast<float2 *>(xd + 4 * qd)[0] = make_float2(v.x, v.y); reinterpret_cast<float2 *>(xd + 4 * qd)[1] = make_float2(v.z, v.w);
            if (qd == 0) {
#pragma unroll
                for (int j = 0; j < A; ++j) xd[D + j] = actv[j];
                a.out_done[b] = dn * (1.0f - to);  // buffers.py:322
                a.out_rew[b] = rw;
            }
        }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0 && a.idx) {  // nobody reads these words in this launch
        if (a.advance_ring) {  // ReplayBuffer.add's epilogue (buffers.py:280-283), left over by the rollout launch
            int64_t pos = a.ring_ctl[0] + 1;
            if (pos == a.ring.rows) { a.ring_ctl[1] = 1; pos = 0; }
            a.ring_ctl[0] = pos;
            a.ring_ctl[3] += 1;
        }
    }
    CH_STAMP(0, 7);
}

// ---- Q networks, forward chain ------------------------------------------------------------------------------------------------
struct QFwdArgs {
    cstr_chain_net_t nets[CSTR_CHAIN_MAX_NETS];
    cstr_sac_head_fin_t fin;
    int n_nets, h1, h2, batch, tiles, has_fin;
};

template <int D, int A, int NQ, int WX>
__global__ __launch_bounds__(CH_THREADS) void q_chain_fwd_kernel(const QFwdArgs a)
{
    extern __shared__ __align__(16) float smem[];
    constexpr int W = D + A;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, h = lane >> 4;
    const int T = WX ? exact_tiles<NQ, WX>() : a.tiles, S = CH_WAVES / T, tile = wave % T, ks = wave / T;
    const cstr_chain_net_t &net = a.nets[blockIdx.z];
    const int H1 = WX ? exact_h1<WX>() : a.h1, H2 = WX ? exact_h2<WX>() : a.h2, B = a.batch;
    const int n0 = (blockIdx.x * T + tile) * 16, m0 = blockIdx.y * 16;
    float *xs = smem + SM_XS, *red = smem + SM_RED, *panel = smem + SM_PANEL;
    const int ld = H1 + 4;
    float *w1s = panel + 16 * ld;
    const bool col_ok = n0 + r < H2;
    CH_STAMP(1, 0);
    const int role = a.has_fin ? net.role : CSTR_CHAIN_ROLE_PLAIN;
    // the pending actor head's partial sums, noise and biases of this lane's (row, action): the longest dependent chain of the launch
    // starts with these loads, so they are requested before everything else
    const bool nxt = role == CSTR_CHAIN_ROLE_NEXT || role == CSTR_CHAIN_ROLE_NEXT_STORE;
    const bool own = nxt || role == CSTR_CHAIN_ROLE_PI;  // the finalised actions are this network's own input columns
    const bool fin_rows = own || (role == CSTR_CHAIN_ROLE_STORE_PI && blockIdx.x == 0);
    const bool fin_lane = fin_rows && t < 16 * A;  // a lane per (row, action) in wave 0
    const int f_row = t / A, f_j = t % A;
    PartBatch p_mu, p_raw;
    float f_ev = 0.0f, f_hb_mu = 0.0f, f_hb_raw = 0.0f;
    bool f_det = false;
    if (fin_lane) {
        f_det = a.fin.kind == CSTR_CHAIN_HEAD_DETERMINISTIC;
        const int HN = f_det ? A : 2 * A;
        const int64_t i = (nxt ? a.fin.next_offset : 0) + m0 + f_row;  // row of the actor pass
        const int64_t pstride = (int64_t)a.fin.part_rows * HN, ptotal = pstride * a.fin.n_parts;
        p_mu.request(a.fin.head_part, a.fin.n_parts, pstride, i * HN + f_j, ptotal);
        if (!f_det) p_raw.request(a.fin.head_part, a.fin.n_parts, pstride, i * HN + A + f_j, ptotal);
        f_ev = a.fin.eps ? a.fin.eps[i * A + f_j] : 0.0f;
        f_hb_mu = a.fin.hb[f_j];
        f_hb_raw = f_det ? 0.0f : a.fin.hb[A + f_j];
    }
    // (a) requests that depend on nothing
    float4 bq[NQ];
    load_b_fwd<NQ>(bq, net.w2, H2, H1, n0, ks, S);
    const int col = min(n0 + r, H2 - 1);
    const float bv = net.b2[col];
    const float w3v = col_ok ? net.w3[col] : 0.0f;
    stage_w1<W>(w1s, net.w1, net.b1, H1);
    L1Direct<W> l1d;
    if (l1_pad<W>() == 0) l1d.request(net.w1, net.b1, H1);
    // (b) input rows (16 floats per row in LDS: the inputs, 1.0 = the bias input, zeros); with a pending actor head the action columns
    //     of the pi(next_obs) rows are finalised HERE (every workgroup of the row group for itself: a lane per (row, action)), and the
    //     column-group-0 workgroups store what later launches need
    // NEXT / NEXT_STORE: the network reads x_next whose action columns are finalised here; PI: the same for x_pi; STORE_PI: the input
    // is complete and the column-group-0 workgroups finalise the pi(obs) rows as a side job
    {
        const int row = t >> 4, k = t & 15;  // 256 threads = 16 x 16
        float xv = k == W ? 1.0f : 0.0f;
        if (k < W && !(own && k >= D)) xv = net.x[(int64_t)(m0 + row) * W + k];
        if (!(own && k >= D && k < W)) xs[row * 16 + k] = xv;
    }
    float f_mu = 0.0f, f_raw = 0.0f, f_a = 0.0f, f_lp = 0.0f;
    if (fin_lane) {
        const bool det = f_det;
        const float ev = f_ev;
        f_mu = p_mu.sum() + f_hb_mu;
        if (det) {
            // deterministic actor (core/td3/policies.py:57-83): a = tanh(.); target actions get the clipped smoothing noise
            // (core/td3/td3.py:167-171; target_smooth_kernel's expressions)
            f_a = tanhf(f_mu);
            if (nxt) f_a = fminf(fmaxf(f_a + fminf(fmaxf(ev * a.fin.sigma, -a.fin.clip), a.fin.clip), -1.0f), 1.0f);
        } else {
            f_raw = p_raw.sum() + f_hb_raw;
            float term;
            sample_action(f_mu, f_raw, ev, f_a, term);
            // log-prob of the row = sum of its actions' terms: the A lanes of a row are adjacent
            f_lp = term;
            if (A >= 2) f_lp += __shfl_xor(f_lp, 1, 64);
            if (A == 4) f_lp += __shfl_xor(f_lp, 2, 64);
        }
        if (own) xs[f_row * 16 + D + f_j] = f_a;
    }
    CH_STAMP(1, 1);
    lds_barrier();
    CH_STAMP(1, 2);
    // (c) layer 1 recomputed on the matrix cores
    layer1_mfma<l1_pad<W>(), W>(xs, w1s, panel, ld, H1, &l1d, net.w1, net.b1);
    lds_barrier();
    CH_STAMP(1, 3);
    f32x4 acc = tile_mma<NQ>(panel, ld, H1, bq, ks, S);
    CH_STAMP(1, 4);
    acc = combine_split_k(acc, smem, T, S);
    CH_STAMP(1, 5);
    float hv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (wave < T) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hv[e] = col_ok ? fmaxf(acc[e] + bv, 0.0f) : 0.0f;
#ifdef CSTR_HEAD_F64  // experiment: the head's dot product over the workgroup's columns accumulated in f64
            double sd = (double)hv[e] * (double)w3v;
            sd += __shfl_xor(sd, 8, 64); sd += __shfl_xor(sd, 4, 64); sd += __shfl_xor(sd, 2, 64); sd += __shfl_xor(sd, 1, 64);
            const float s = (float)sd;
#else
            const float s = rowsum16(hv[e] * w3v);
#endif
            if (r == 0) red[(tile * 16 + 4 * h + e) * 8] = s;
        }
    }
    lds_barrier();
    CH_STAMP(1, 6);
    if (t < 16) {
        float s = red[t * 8];
        for (int tl = 1; tl < T; ++tl) s += red[(tl * 16 + t) * 8];
        net.q_part[(int64_t)blockIdx.x * B + m0 + t] = s;
    }
    // ---- global stores, all behind the last barrier ----
    if (wave < T && col_ok && net.h2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) net.h2[(int64_t)(m0 + 4 * h + e) * H2 + n0 + r] = hv[e];
    }
    if (blockIdx.x == 0 && net.h1) store_panel(panel, ld, net.h1, H1, m0);
    if (fin_lane && blockIdx.x == 0 && role != CSTR_CHAIN_ROLE_NEXT) {
        const int bb = m0 + f_row;
        const bool det = a.fin.kind == CSTR_CHAIN_HEAD_DETERMINISTIC;
        (nxt ? a.fin.x_next : a.fin.x_pi)[(int64_t)bb * W + D + f_j] = f_a;
        if (!nxt && !det) {
            a.fin.params[(int64_t)bb * 2 * A + f_j] = f_mu;
            a.fin.params[(int64_t)bb * 2 * A + A + f_j] = f_raw;
        }
        if (f_j == 0 && !det) (nxt ? a.fin.logp_next : a.fin.logp_pi)[bb] = f_lp;
    }
    CH_STAMP(1, 7);
}

// ---- Q networks, loss root + backward chain -------------------------------------------------------------------------------------
struct QBwdArgs {
    cstr_chain_net_t nets[2];
    cstr_chain_root_t root;
    int n_nets, h1, h2, tiles;
    float *dz2, *dz1, *gact_part;
};

__device__ __forceinline__ float q_from_parts(const float *part, const float *b3, const int n_parts, const int batch, const int row)
{
    const float b = b3[0];
    return sum_parts(part, n_parts, batch, row, (int64_t)n_parts * batch) + b;
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
    if (tid < 2 && rt.adam_advance[tid]) {  // state["step"] += 1 for the optimiser launch behind this one (adam_body's epilogue)
        int64_t *ctl = rt.adam_advance[tid];
        double *pw = reinterpret_cast<double *>(ctl + 2);
        ctl[0] += 1;
        pw[0] *= rt.adam_beta1[tid];
        pw[1] *= rt.adam_beta2[tid];
    }
}

template <int D, int A, int NQ, int WX>
__global__ __launch_bounds__(CH_THREADS) void q_chain_bwd_kernel(const QBwdArgs a)
{
    extern __shared__ __align__(16) float smem[];
    constexpr int W = D + A;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, h = lane >> 4;
    const cstr_chain_root_t &rt = a.root;
    const int B = rt.batch, H1 = WX ? exact_h1<WX>() : a.h1, H2 = WX ? exact_h2<WX>() : a.h2;
    const int g = blockIdx.z;
    if ((int)blockIdx.y == B / 16) {  // the loss workgroup
        if (g == 0 && blockIdx.x == 0) chain_loss_workgroup(rt, smem + SM_RED);
        return;
    }
    const int T = WX ? exact_tiles<NQ, WX>() : a.tiles, S = CH_WAVES / T, tile = wave % T, ks = wave / T;
    const cstr_chain_net_t &net = a.nets[g];
    const int k0 = (blockIdx.x * T + tile) * 16, m0 = blockIdx.y * 16;
    const bool col_ok = k0 + r < H1;
    const int col = min(k0 + r, H1 - 1);
    float *red = smem + SM_RED, *gs = smem + SM_GS, *panel = smem + SM_PANEL;
    const int ld = H2 + 4;
    CH_STAMP(2, 0);
    // the row group's Q partials (freshly written by the launch in front of this one) start the prologue's dependent chain: requested
    // before everything else
    PartBatch pq[4];
    float r_nl = 0.0f, r_rw = 0.0f, r_dn = 0.0f, r_b3[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (t < 16) {
        const int row = m0 + t, P = rt.n_parts, need = rt.mode == 1 ? 4 : a.n_nets;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < need) {
                pq[q].request(rt.q_part[q], P, B, row, (int64_t)P * B);
                r_b3[q] = rt.b3[q][0];
            }
        }
        if (rt.mode == 1) {
            r_nl = rt.next_logp ? rt.next_logp[row] : 0.0f;
            r_rw = rt.rew[row];
            r_dn = rt.done[row];
        }
    }
    // (a) operands that depend on nothing: the tile's B operand (W2 read along n), the epilogue's relu'(h1) mask values and first-layer
    //     action columns, and this thread's columns of the h2 panel + w3 for the dz2 recompute
    float4 bq[NQ];
    load_b_bwd<NQ>(bq, net.w2, H2, H1, k0, ks, S);
    float ty[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ty[e] = net.h1[(int64_t)(m0 + 4 * h + e) * H1 + col];
    float w1a[A];
#pragma unroll
    for (int j = 0; j < A; ++j) w1a[j] = a.gact_part ? net.w1[(int64_t)col * W + D + j] : 0.0f;
    float y[2][16], w3c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = min(t + CH_THREADS * i, H2 - 1);
        w3c[i] = 0.0f;
#pragma unroll
        for (int row = 0; row < 16; ++row) y[i][row] = 0.0f;
        if (i == 0 || H2 > CH_THREADS) {  // (uniform)
            w3c[i] = net.w3[c];
#pragma unroll
            for (int row = 0; row < 16; ++row) y[i][row] = net.h2[(int64_t)(m0 + row) * H2 + c];
        }
    }
    // (b) d(loss)/dq of this network for the row group's 16 rows (cstr_head_root_t's expressions)
    float o_q = 0.0f, o_gq = 0.0f, o_tq = 0.0f;
    if (t < 16) {
        const bool with_alpha = rt.mode == 1 && rt.alpha.log_alpha != nullptr;
        const float ec = with_alpha ? expf(rt.alpha.log_alpha[0]) : (rt.ent_coef ? rt.ent_coef[0] : 0.0f);
        const float kq = rt.scale * 2.0f / (float)B, inv = 1.0f / (float)B;
        if (rt.mode == 1) {
            const float qa = pq[2].sum() + r_b3[2], qb = pq[3].sum() + r_b3[3];
            o_q = (g == 0 ? pq[0].sum() + r_b3[0] : pq[1].sum() + r_b3[1]);
            float q = fminf(qa, qb);
            if (rt.next_logp) q = q - ec * r_nl;
            o_tq = r_rw + (1.0f - r_dn) * rt.gamma * q;
            o_gq = kq * (o_q - o_tq);
        } else if (rt.mode == 3) {
            o_q = pq[0].sum() + r_b3[0];
            o_gq = -inv;
        } else {
            const float q1 = pq[0].sum() + r_b3[0], q2 = pq[1].sum() + r_b3[1];
            const bool first = q1 <= q2;
            o_q = g == 0 ? q1 : q2;
            o_gq = (first == (g == 0)) ? -inv : 0.0f;
        }
        gs[t] = o_gq;
    }
    CH_STAMP(2, 1);
    lds_barrier();
    CH_STAMP(2, 2);
    // (c) dz2 = dq * w3 * relu'(h2), recomputed by every workgroup of the row group into the panel (values kept for the store at the end)
    {
        float gq[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 g4 = *reinterpret_cast<const float4 *>(gs + 4 * q);
            gq[4 * q] = g4.x; gq[4 * q + 1] = g4.y; gq[4 * q + 2] = g4.z; gq[4 * q + 3] = g4.w;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = t + CH_THREADS * i;
            if (c < H2) {
#pragma unroll
                for (int row = 0; row < 16; ++row) {
                    y[i][row] = y[i][row] > 0.0f ? gq[row] * w3c[i] : 0.0f;
                    panel[row * ld + c] = y[i][row];
                }
            }
        }
    }
    lds_barrier();
    CH_STAMP(2, 3);
    // (d) dz1 tile = (dz2 W2)[rows][k0 .. k0 + 15], (e) split-K combine, (f) * relu'(h1); partial action gradient
    f32x4 acc = tile_mma<NQ>(panel, ld, H2, bq, ks, S);
    CH_STAMP(2, 4);
    acc = combine_split_k(acc, smem, T, S);
    CH_STAMP(2, 5);
    float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (wave < T) {
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = (col_ok && ty[e] > 0.0f) ? acc[e] : 0.0f;
        if (a.gact_part) {
#pragma unroll
            for (int j = 0; j < A; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = rowsum16(d[e] * w1a[j]);
                    if (r == 0) red[(tile * 16 + 4 * h + e) * 8 + j] = s;
                }
            }
        }
    }
    if (a.gact_part) {
        lds_barrier();
        if (t < 16 * A) {
            const int row = t / A, j = t % A;
            float s = red[row * 8 + j];
            for (int tl = 1; tl < T; ++tl) s += red[(tl * 16 + row) * 8 + j];
            a.gact_part[(((int64_t)g * gridDim.x + blockIdx.x) * B + m0 + row) * A + j] = s;
        }
    }
    CH_STAMP(2, 6);
    // ---- global stores, all behind the last barrier ----
    if (a.dz1 && wave < T && col_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a.dz1[((int64_t)g * B + m0 + 4 * h + e) * H1 + k0 + r] = d[e];
    }
    if (a.dz2) {  // each workgroup of the row group stores its share of the columns
        const int cpg = (H2 + gridDim.x - 1) / gridDim.x;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = t + CH_THREADS * i;
            if (c < H2 && c / cpg == (int)blockIdx.x) {
#pragma unroll
                for (int row = 0; row < 16; ++row) a.dz2[((int64_t)g * B + m0 + row) * H2 + c] = y[i][row];
            }
        }
    }
    if (t < 16 && blockIdx.x == 0) {
        const int row = m0 + t;
        if (rt.q_out) rt.q_out[(int64_t)g * B + row] = o_q;
        if (rt.gq_out) rt.gq_out[(int64_t)g * B + row] = o_gq;
        if (rt.mode == 1 && g == 0 && rt.target_out) rt.target_out[row] = o_tq;
    }
    CH_STAMP(2, 7);
}

// ---- SAC actor, backward chain -------------------------------------------------------------------------------------------------
struct ActorBwdArgs {
    cstr_sac_actor_t net;
    const float *gact_part; int n_nets, n_parts; const float *ent_coef;
    const float *x_pi, *params, *eps, *a_h1, *a_h2;
    float *g_params, *dz2, *dz1;
    int batch, tiles, kind;  // CSTR_CHAIN_HEAD_*
};

template <int D, int A, int NQ, int WX>
__global__ __launch_bounds__(CH_THREADS) void sac_actor_chain_bwd_kernel(const ActorBwdArgs a)
{
    extern __shared__ __align__(16) float smem[];
    constexpr int W = D + A;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, h = lane >> 4;
    const int T = WX ? exact_tiles<NQ, WX>() : a.tiles, S = CH_WAVES / T, tile = wave % T, ks = wave / T;
    const int H1 = WX ? exact_h1<WX>() : a.net.h1, H2 = WX ? exact_h2<WX>() : a.net.h2, B = a.batch;
    const int k0 = (blockIdx.x * T + tile) * 16, m0 = blockIdx.y * 16;
    const bool col_ok = k0 + r < H1, det = a.kind == CSTR_CHAIN_HEAD_DETERMINISTIC;
    const int col = min(k0 + r, H1 - 1), HN = det ? A : 2 * A;
    float *gs = smem + SM_GS, *panel = smem + SM_PANEL;
    const int ld = H2 + 4;
    CH_STAMP(3, 0);
    float4 bq[NQ];
    load_b_bwd<NQ>(bq, a.net.w2, H2, H1, k0, ks, S);
    float ty[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ty[e] = a.a_h1[(int64_t)(m0 + 4 * h + e) * H1 + col];
    // the dz2 recompute's operands of this wave's first four tiles (tiles wave, wave + 4, ...): head weights and relu'(a_h2) mask values
    constexpr int ZT = 4;
    const int n_ztiles = (H2 + 15) / 16;
    float zb0[ZT], zb1[ZT], zm[ZT][4];
#pragma unroll
    for (int i = 0; i < ZT; ++i) {
        const int cc = min(16 * (wave + CH_WAVES * i) + r, H2 - 1);
        zb0[i] = h < HN ? a.net.hw[(int64_t)h * H2 + cc] : 0.0f;
        zb1[i] = HN > 4 ? a.net.hw[(int64_t)(4 + h) * H2 + cc] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) zm[i][e] = a.a_h2[(int64_t)(m0 + 4 * h + e) * H2 + cc];
    }
    // (b) d(loss)/d(action) from the critic's partial sums, then the squashed-Gaussian head's backward (gaussian_head_bwd_kernel's
    //     expressions): g_params[row] = (d/d mean | d/d log_std) -> gs[16][8] (zero-padded: the k chunk of the dz2 MFMA)
    float o_gu = 0.0f, o_gls = 0.0f;
    if (t < 16 * A) {
        const int row = t / A, j = t % A, b = m0 + row;
        const float av = a.x_pi[(int64_t)b * W + D + j];
        const float ecv = det ? 0.0f : a.ent_coef[0], raw = det ? 0.0f : a.params[(int64_t)b * 2 * A + A + j];
        const float epv = det ? 0.0f : a.eps[(int64_t)b * A + j];
        // networks x column groups are ONE run of partials (stride B * A): summed in (network, group) order
        const int np = a.n_nets * a.n_parts;
        const float ga = sum_parts(a.gact_part, np, (int64_t)B * A, (int64_t)b * A + j, (int64_t)np * B * A);
        const float one_m = 1.0f - av * av;
        if (det) {  // a = tanh(z): d/dz = d/da * (1 - a^2)   (bias_act_bwd's tanh expression)
            o_gu = ga * one_m;
        } else {
            const float gl = ecv * (1.0f / (float)B);  // d(loss)/d(logp) (sac_actor_loss_kernel)
            const float s = expf(fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX));
            o_gu = ga * one_m + gl * (2.0f * av * one_m / (one_m + 1e-6f));
            o_gls = (raw >= LOG_STD_MIN && raw <= LOG_STD_MAX) ? o_gu * epv * s - gl : 0.0f;
        }
        gs[row * 8 + j] = o_gu;
        gs[row * 8 + A + j] = o_gls;  // (deterministic head: zero -- the k chunk of the dz2 MFMA is 4 wide)
    }
    CH_STAMP(3, 1);
    lds_barrier();
    CH_STAMP(3, 2);
    // (c) dz2 = (g_params hw) * relu'(a_h2) on the matrix cores: K = 2A is one (A = 2) or two (A = 4) MFMA steps per 16 x 16 tile;
    //     A operand = g_params[r][h] (LDS), B operand = hw[h][column] straight from L2; the wave takes the tiles wave, wave + 4, ...
    {
        const bool two = HN > 4;  // 8 head outputs: two k steps
        const float ga0 = gs[r * 8 + h], ga1 = two ? gs[r * 8 + 4 + h] : 0.0f;
        const int cpg = (H2 + gridDim.x - 1) / gridDim.x;
        auto dz2_tile = [&](const int tl, const float hb0, const float hb1, const float (&m)[4]) {
            const int c = 16 * tl + r;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga0, hb0, acc, 0, 0, 0);
            if (two) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga1, hb1, acc, 0, 0, 0);
            const bool mine = c / cpg == (int)blockIdx.x;  // this workgroup's share of the dz2 columns (stored now: nothing waits for it)
            if (tl < n_ztiles && c < H2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dv = m[e] > 0.0f ? acc[e] : 0.0f;
                    panel[(4 * h + e) * ld + c] = dv;
                    if (mine) a.dz2[(int64_t)(m0 + 4 * h + e) * H2 + c] = dv;
                }
            }
        };
#pragma unroll
        for (int i = 0; i < ZT; ++i) dz2_tile(wave + CH_WAVES * i, zb0[i], zb1[i], zm[i]);
        for (int tl = wave + CH_WAVES * ZT; tl < n_ztiles; tl += CH_WAVES) {  // layers wider than 256: the remaining tiles
            const int cc = min(16 * tl + r, H2 - 1);
            const float hb0 = h < HN ? a.net.hw[(int64_t)h * H2 + cc] : 0.0f, hb1 = two ? a.net.hw[(int64_t)(4 + h) * H2 + cc] : 0.0f;
            float m[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = a.a_h2[(int64_t)(m0 + 4 * h + e) * H2 + cc];
            dz2_tile(tl, hb0, hb1, m);
        }
    }
    lds_barrier();
    CH_STAMP(3, 3);
    f32x4 acc = tile_mma<NQ>(panel, ld, H2, bq, ks, S);
    CH_STAMP(3, 4);
    acc = combine_split_k(acc, smem, T, S);
    CH_STAMP(3, 5);
    // ---- global stores ----
    if (wave < T && col_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a.dz1[(int64_t)(m0 + 4 * h + e) * H1 + k0 + r] = ty[e] > 0.0f ? acc[e] : 0.0f;
    }
    if (t < 16 * A && blockIdx.x == 0) {
        const int row = t / A, j = t % A, b = m0 + row;
        a.g_params[(int64_t)b * HN + j] = o_gu;
        if (!det) a.g_params[(int64_t)b * HN + A + j] = o_gls;
    }
    CH_STAMP(3, 7);
}

// out[row][col] = sum over parts of part[p][row][col]: the consumer-side finalisation of a chain launch's partial sums for a consumer that
// is NOT a chain kernel (MADDPG's per-layer actor backward reads d(loss)/d(action) from the action columns of a critic-input gradient)
__global__ __launch_bounds__(256) void chain_sum_parts_kernel(const float *__restrict__ part, const int n_parts, const int64_t rows, const int cols,
                                                              float *__restrict__ out, const int64_t out_stride)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, n = rows * cols;
    if (i >= n) return;
    out[(i / cols) * out_stride + i % cols] = sum_parts(part, n_parts, n, i, (int64_t)n_parts * n);
}

// chunks of 16 along the reduction a wave owns, rounded up to an instantiated size (0: not covered)
static int nq_for(int kdim, int tiles, int nq_max = 16)
{
    const int s = CH_WAVES / tiles, nch = (kdim + 15) / 16, per = (nch + s - 1) / s;
    return per <= 4 ? 4 : per <= 8 ? 8 : per <= 16 ? 16 : (per <= 32 && nq_max >= 32) ? 32 : 0;
}

static bool chain_dims_ok(int h1, int h2, int64_t batch, int tiles)
{
    return h1 >= 16 && h2 >= 16 && h1 <= CSTR_CHAIN_MAX_WIDTH && h2 <= CSTR_CHAIN_MAX_WIDTH && h1 % 4 == 0 && h2 % 4 == 0 && batch >= 16 &&
           batch <= 1024 && batch % 16 == 0 && (tiles == 1 || tiles == 2 || tiles == 4);
}

static int chain_layout(int obs_dim, int act_dim)
{
    return (obs_dim == 4 && act_dim == 2) ? 0 : (obs_dim == 8 && act_dim == 2) ? 1 : (obs_dim == 8 && act_dim == 4) ? 2 : -1;
}

#define CHAIN_NQ(KERNEL, D_, A_, WX_, NQV, GRID, LDS, STREAM, ARGS)                                \
    do {                                                                                           \
        if ((NQV) == 4) KERNEL<D_, A_, 4, WX_><<<GRID, CH_THREADS, LDS, STREAM>>>(ARGS);           \
        else if ((NQV) == 8) KERNEL<D_, A_, 8, WX_><<<GRID, CH_THREADS, LDS, STREAM>>>(ARGS);      \
        else if ((NQV) == 16) KERNEL<D_, A_, 16, WX_><<<GRID, CH_THREADS, LDS, STREAM>>>(ARGS);    \
        else KERNEL<D_, A_, NQ_TOP_##KERNEL, 0><<<GRID, CH_THREADS, LDS, STREAM>>>(ARGS);          \
    } while (0)
// the widest per-wave share instantiated per kernel: the forward chains (B operand = 16-byte quads) also exist with 32 chunks (wide
// first hidden layers with 4 tiles per workgroup: TD3's / MADDPG's 400); the backward chains stop at 16
#define NQ_TOP_sac_actor_chain_fwd_kernel 32
#define NQ_TOP_q_chain_fwd_kernel 32
#define NQ_TOP_q_chain_bwd_kernel 16
#define NQ_TOP_sac_actor_chain_bwd_kernel 16
// WXV (chain_wx): exact-width instantiations exist where a class default uses them -- 256 x 256 with the 4 / 2 and 8 / 2 layouts
// (SAC on the 4- and 8-wide observation), 400 x 300 with 4 / 2 (TD3, DDPG) and 8 / 4 (MADDPG's twin train); everything else runs
// the run-time-width kernels
#define CHAIN_DISPATCH(KERNEL, LAY, WXV, NQV, GRID, LDS, STREAM, ARGS)                             \
    do {                                                                                           \
        if ((LAY) == 0 && (WXV) == 1) CHAIN_NQ(KERNEL, 4, 2, 1, NQV, GRID, LDS, STREAM, ARGS);     \
        else if ((LAY) == 0 && (WXV) == 2) CHAIN_NQ(KERNEL, 4, 2, 2, NQV, GRID, LDS, STREAM, ARGS); \
        else if ((LAY) == 0) CHAIN_NQ(KERNEL, 4, 2, 0, NQV, GRID, LDS, STREAM, ARGS);              \
        else if ((LAY) == 1 && (WXV) == 1) CHAIN_NQ(KERNEL, 8, 2, 1, NQV, GRID, LDS, STREAM, ARGS); \
        else if ((LAY) == 1) CHAIN_NQ(KERNEL, 8, 2, 0, NQV, GRID, LDS, STREAM, ARGS);              \
        else if ((WXV) == 2) CHAIN_NQ(KERNEL, 8, 4, 2, NQV, GRID, LDS, STREAM, ARGS);              \
        else CHAIN_NQ(KERNEL, 8, 4, 0, NQV, GRID, LDS, STREAM, ARGS);                              \
    } while (0)

// which exact-width instantiation (0: none) serves widths (h1, h2) with `tiles` column groups per workgroup and per-wave share nq
static int chain_wx(int h1, int h2, int tiles, int nq)
{
    // CSTR_EXACT_SHAPES=0: run the run-time-width kernels everywhere (tests compare the two bit for bit)
    static const bool off = getenv("CSTR_EXACT_SHAPES") && atoi(getenv("CSTR_EXACT_SHAPES")) == 0;
    if (off || (nq != 4 && nq != 8 && nq != 16)) return 0;
    if (h1 == 256 && h2 == 256 && nq == 4 * tiles) return 1;
    if (h1 == 400 && h2 == 300 && nq == 8 * tiles) return 2;
    return 0;
}

constexpr size_t CHAIN_LDS_LIMIT = 64 * 1024;

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------------------------------------
#ifdef CSTR_CHAIN_STAMPS
extern "C" int cstr_diag_chain_stamps(unsigned long long *host_out, int64_t words)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(chain_stamps), (size_t)words * 8);
}
#endif

static int check_actor(const cstr_sac_actor_t *n)
{
    if (!n || !n->w1 || !n->b1 || !n->w2 || !n->b2 || !n->hw || !n->hb) return CSTR_E_BADARG;
    if (!((n->obs_dim == 4 && n->act_dim == 2) || (n->obs_dim == 8 && (n->act_dim == 2 || n->act_dim == 4)))) return CSTR_E_UNSUPPORTED;
    if (!aligned16(n->w1) || !aligned16(n->w2)) return CSTR_E_BADARG;
    return CSTR_OK;
}

extern "C" int cstr_sac_actor_chain_fwd_f32(const cstr_sac_actor_t *actor, const cstr_ring_t *ring, int64_t *ring_ctl, int advance_ring,
                                            const int32_t *sample_idx, int64_t batch, float *x_data, float *x_pi, float *x_next,
                                            float *out_done, float *out_rew, float *a_h1, float *a_h2, float *head_part,
                                            const uint64_t *head_rng_ctl, uint64_t head_rng_offset, float *eps_all, int rows_mode,
                                            int head_n, int tiles, cstr_stream_t stream)
{
    const int rc = check_actor(actor);
    if (rc) return rc;
    if (!head_part || (eps_all && !head_rng_ctl)) return CSTR_E_BADARG;
    if (rows_mode < CSTR_CHAIN_ROWS_PAIR || rows_mode > CSTR_CHAIN_ROWS_OBS || (head_n != actor->act_dim && head_n != 2 * actor->act_dim)) return CSTR_E_BADARG;
    if ((rows_mode != CSTR_CHAIN_ROWS_OBS && !x_next) || (rows_mode != CSTR_CHAIN_ROWS_NEXT && !x_pi)) return CSTR_E_BADARG;
    if (!chain_dims_ok(actor->h1, actor->h2, batch, tiles)) return CSTR_E_UNSUPPORTED;
    if ((x_pi && !aligned8(x_pi)) || (x_next && !aligned8(x_next))) return CSTR_E_BADARG;
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
    a.ring_ctl = ring_ctl; a.advance_ring = advance_ring;
    a.idx = sample_idx; a.batch = (int)batch; a.tiles = tiles;
    a.x_data = x_data; a.x_pi = x_pi; a.x_next = x_next; a.out_done = out_done; a.out_rew = out_rew;
    a.a_h1 = a_h1; a.a_h2 = a_h2; a.head_part = head_part; a.head_rng_ctl = head_rng_ctl; a.head_rng_offset = head_rng_offset; a.eps_all = eps_all;
    a.rows_mode = rows_mode; a.head_n = head_n;
    const dim3 grid((unsigned)((actor->h2 + 16 * tiles - 1) / (16 * tiles)), (unsigned)((rows_mode == CSTR_CHAIN_ROWS_PAIR ? 2 : 1) * batch / 16));
    const size_t lds = chain_lds_bytes(actor->h1, 0);  // obs rows are whole quads: layer 1's operand is read directly
    const int nq = nq_for(actor->h1, tiles, 32), lay = chain_layout(actor->obs_dim, actor->act_dim);
    if (!nq || lds > CHAIN_LDS_LIMIT) return CSTR_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    CHAIN_DISPATCH(sac_actor_chain_fwd_kernel, lay, chain_wx(actor->h1, actor->h2, tiles, nq), nq, grid, lds, s, a);
    return (int)hipGetLastError();
}

extern "C" int cstr_q_chain_fwd_f32(const cstr_chain_net_t *nets, int n_nets, int w_in, int obs_dim, int h1, int h2, int64_t batch,
                                    const cstr_sac_head_fin_t *fin, int tiles, cstr_stream_t stream)
{
    if (!nets || n_nets < 1 || n_nets > CSTR_CHAIN_MAX_NETS || w_in < 1 || w_in > 12 || obs_dim < 1 || obs_dim >= w_in) return CSTR_E_BADARG;
    if (!chain_dims_ok(h1, h2, batch, tiles)) return CSTR_E_UNSUPPORTED;
    QFwdArgs a = {};
    for (int g = 0; g < n_nets; ++g) {
        const cstr_chain_net_t &n = nets[g];
        if (!n.w1 || !n.b1 || !n.w2 || !n.b2 || !n.w3 || !n.b3 || !n.x || !n.q_part || !aligned16(n.w2)) return CSTR_E_BADARG;
        if (n.role < 0 || n.role > CSTR_CHAIN_ROLE_PI || (n.role != 0 && !fin)) return CSTR_E_BADARG;
        a.nets[g] = n;
    }
    if (fin) {
        if (!fin->head_part || !fin->hb || fin->n_parts < 1 || fin->act_dim < 1 || fin->act_dim > MAX_A || fin->obs_dim != obs_dim || fin->obs_dim + fin->act_dim != w_in)
            return CSTR_E_BADARG;
        const bool det = fin->kind == CSTR_CHAIN_HEAD_DETERMINISTIC;
        if (fin->kind != CSTR_CHAIN_HEAD_GAUSSIAN && !det) return CSTR_E_BADARG;
        if (fin->part_rows < batch || fin->next_offset < 0 || fin->next_offset + batch > fin->part_rows) return CSTR_E_BADARG;
        if (!det && (!fin->eps || !fin->x_pi || !fin->x_next || !fin->params || !fin->logp_pi || !fin->logp_next)) return CSTR_E_BADARG;
        for (int g = 0; g < n_nets; ++g) {  // the buffers the roles in use write
            const int role = nets[g].role;
            if ((role == CSTR_CHAIN_ROLE_NEXT_STORE && !fin->x_next) || ((role == CSTR_CHAIN_ROLE_STORE_PI || role == CSTR_CHAIN_ROLE_PI) && !fin->x_pi))
                return CSTR_E_BADARG;
        }
        a.fin = *fin;
        a.has_fin = 1;
    }
    a.n_nets = n_nets; a.h1 = h1; a.h2 = h2; a.batch = (int)batch; a.tiles = tiles;
    const dim3 grid((unsigned)((h2 + 16 * tiles - 1) / (16 * tiles)), (unsigned)(batch / 16), (unsigned)n_nets);
    const size_t lds = chain_lds_bytes(h1, (w_in % 2 == 0 && w_in + 1 <= 16) ? 0 : (w_in + 1 <= 8 ? 8 : 16));  // l1_pad<w_in>()
    const int nq = nq_for(h1, tiles, 32), lay = chain_layout(obs_dim, w_in - obs_dim);
    if (!nq || lay < 0 || lds > CHAIN_LDS_LIMIT) return CSTR_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    CHAIN_DISPATCH(q_chain_fwd_kernel, lay, chain_wx(h1, h2, tiles, nq), nq, grid, lds, s, a);
    return (int)hipGetLastError();
}

extern "C" int cstr_q_chain_bwd_f32(const cstr_chain_net_t *nets, int n_nets, const cstr_chain_root_t *root, int w_in, int obs_dim, int h1,
                                    int h2, float *dz2, float *dz1, float *gact_part, int tiles, cstr_stream_t stream)
{
    if (!nets || !root || n_nets < 1 || n_nets > 2 || w_in < 1 || w_in > 12 || obs_dim < 0 || obs_dim > w_in) return CSTR_E_BADARG;
    if (root->mode < 1 || root->mode > 3 || root->n_parts < 1) return CSTR_E_BADARG;
    if ((root->mode == 3) != (n_nets == 1)) return CSTR_E_BADARG;
    if (!chain_dims_ok(h1, h2, root->batch, tiles)) return CSTR_E_UNSUPPORTED;
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
    a.n_nets = n_nets; a.h1 = h1; a.h2 = h2; a.tiles = tiles;
    a.dz2 = dz2; a.dz1 = dz1; a.gact_part = gact_part;
    const dim3 grid((unsigned)((h1 + 16 * tiles - 1) / (16 * tiles)), (unsigned)(root->batch / 16 + 1), (unsigned)n_nets);
    const size_t lds = chain_lds_bytes(h2, 0);
    const int nq = nq_for(h2, tiles), lay = chain_layout(obs_dim, w_in - obs_dim);
    if (!nq || lay < 0 || lds > CHAIN_LDS_LIMIT) return CSTR_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    CHAIN_DISPATCH(q_chain_bwd_kernel, lay, chain_wx(h1, h2, tiles, nq), nq, grid, lds, s, a);
    return (int)hipGetLastError();
}

extern "C" int cstr_chain_sum_parts_f32(const float *part, int n_parts, int64_t rows, int cols, float *out, int64_t out_stride, cstr_stream_t stream)
{
    if (!part || !out || n_parts < 1 || rows < 1 || cols < 1 || out_stride < cols) return CSTR_E_BADARG;
    if ((int64_t)n_parts * rows * cols >= (1 << 28)) return CSTR_E_UNSUPPORTED;
    const int64_t n = rows * cols;
    chain_sum_parts_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(part, n_parts, rows, cols, out, out_stride);
    return (int)hipGetLastError();
}

extern "C" int cstr_sac_actor_chain_bwd_f32(const cstr_sac_actor_t *actor, const float *gact_part, int n_nets, int n_parts,
                                            const float *ent_coef, const float *x_pi, const float *params, const float *eps, const float *a_h1,
                                            const float *a_h2, float *g_params, float *dz2, float *dz1, int64_t batch, int kind, int tiles,
                                            cstr_stream_t stream)
{
    const int rc = check_actor(actor);
    if (rc) return rc;
    if (kind != CSTR_CHAIN_HEAD_GAUSSIAN && kind != CSTR_CHAIN_HEAD_DETERMINISTIC) return CSTR_E_BADARG;
    if (!gact_part || n_nets < 1 || n_nets > 2 || n_parts < 1 || !x_pi || !a_h1 || !a_h2 || !g_params || !dz2 || !dz1) return CSTR_E_BADARG;
    if (kind == CSTR_CHAIN_HEAD_GAUSSIAN && (!ent_coef || !params || !eps)) return CSTR_E_BADARG;
    if (!chain_dims_ok(actor->h1, actor->h2, batch, tiles)) return CSTR_E_UNSUPPORTED;
    ActorBwdArgs a = {};
    a.net = *actor;
    a.gact_part = gact_part; a.n_nets = n_nets; a.n_parts = n_parts; a.ent_coef = ent_coef;
    a.x_pi = x_pi; a.params = params; a.eps = eps; a.a_h1 = a_h1; a.a_h2 = a_h2;
    a.g_params = g_params; a.dz2 = dz2; a.dz1 = dz1; a.batch = (int)batch; a.tiles = tiles; a.kind = kind;
    const dim3 grid((unsigned)((actor->h1 + 16 * tiles - 1) / (16 * tiles)), (unsigned)(batch / 16));
    const size_t lds = chain_lds_bytes(actor->h2, 0);
    const int nq = nq_for(actor->h2, tiles), lay = chain_layout(actor->obs_dim, actor->act_dim);
    if (!nq || lds > CHAIN_LDS_LIMIT) return CSTR_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    CHAIN_DISPATCH(sac_actor_chain_bwd_kernel, lay, chain_wx(actor->h1, actor->h2, tiles, nq), nq, grid, lds, s, a);
    return (int)hipGetLastError();
}
