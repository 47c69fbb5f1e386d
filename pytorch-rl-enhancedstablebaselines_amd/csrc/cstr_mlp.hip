// cstr_mlp.hip -- the learners' MLP kernels for gfx950 (f32 matrix cores: v_mfma_f32_16x16x4_f32, exact f32 fma chains):
//   * Linear + bias + activation forward, input gradient x activation gradient of the layer below, dW + db: one launch each,
//     grouped (stacked twin critics) and pointer-table ("sets") forms;
//   * the rollout's WHOLE policy network + sampling in one launch (policy_rows_v2_kernel: role-split waves, register-resident
//     or register-pipelined B operand from a tile-major weight copy, split-K MFMA head, per-(row, action) sampling tail);
//   * SAC's squashed-Gaussian head (in-kernel Philox noise): rsample -> tanh -> log-prob forward, analytic backward, also carried
//     through the head's Linear;
//   * a Q network's last hidden layer epilogue + scalar head, forward and backward; the backward form with the LOSS ROOT inside
//     (TD critic loss, SAC actor loss: the loss launch and the backward's first launch are one);
//   * loss heads that are backward ROOTS (upstream gradient == 1): the kernels emit the loss value AND d(loss)/d(Q),
//     d(loss)/d(logp) directly, so no autograd nodes exist for the losses; SAC's entropy-coefficient loss / gradient;
//   * epilogue kernels around rocBLAS GEMMs for the CSTR_FUSED_LINEAR=0 configuration.
// All latency-bound at batch 256; 64-wide waves, LDS tree reductions, deterministic (no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/cstr_rl_hip.h"
#include "cstr_device.h"
#include "cstr_env_device.h"
#include "cstr_mt_device.h"
#include "cstr_rng_device.h"
#include "cstr_adam_device.h"

namespace {

constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2;
#ifndef CSTR_BWD_FLY
#define CSTR_BWD_FLY 4
#endif

// ---- Linear epilogues --------------------------------------------------------------------------------

// y[m][n] = act(y[m][n] + bias[n])   (nn.Linear + nn.ReLU / nn.Tanh of create_mlp, core/common/torch_layers.py:110-183)
// Grouped form: y is [G][m][n] and bias [G][n] (a batched GEMM's output: twin critics, merged heads); `gsz` = m * n.
template <int ACT>
__global__ void bias_act_fwd_kernel(float *__restrict__ y, const float *__restrict__ bias, const int64_t total, const int n,
                                    const int64_t gsz)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = y[i] + bias[(i / gsz) * n + i % n];
        if (ACT == ACT_RELU) v = v > 0.0f ? v : 0.0f;
        if (ACT == ACT_TANH) v = tanhf(v);
        y[i] = v;
    }
}

template <int ACT>
__global__ void bias_act_fwd_vec4_kernel(float4 *__restrict__ y, const float4 *__restrict__ bias, const int64_t total4, const int n4,
                                         const int64_t gsz4)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = y[i];
        const float4 b = bias[(i / gsz4) * n4 + i % n4];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (ACT == ACT_RELU) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        if (ACT == ACT_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
        y[i] = v;
    }
}

// gz[m][n] = gy[m][n] * act'(y[m][n]);  gbias[n] = sum_m gz[m][n]   (threshold_backward / tanh_backward + the bias
// gradient's sum over the batch). One workgroup owns 64 columns: lanes run along columns (coalesced 256-B row
// segments), its WAVES waves stride over the rows with four rows of loads in flight per wave (the kernel is pure
// latency at batch 256), LDS combines the waves; every gbias element is written exactly once -> deterministic.
// `ldg` / `ldy`: row strides of gy / y (n for contiguous tensors; a column block of a wider row-major matrix otherwise, one group)
template <int ACT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bias_act_bwd_kernel(const float *gy, const float *__restrict__ y, float *gz,
                                                                  float *__restrict__ gbias, const int m, const int n,
                                                                  const int64_t ldg, const int64_t ldy)
{
    __shared__ float part[WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int64_t goff = (int64_t)blockIdx.y * m * n;  // group (batched-GEMM member): [G][m][n] tensors, gbias [G][n]
    gy += goff;
    gz += goff;
    if (ACT != ACT_NONE) y += goff;
    if (gbias) gbias += (int64_t)blockIdx.y * n;
    float acc = 0.0f;
    if (col < n) {
        constexpr int FLY = CSTR_BWD_FLY;  // rows in flight per wave; A/B on MI355X at batch 256: 4 beats 8 and 16 (0.199-0.202 vs 0.203-0.208 / 0.207 ms per SAC iteration)
        for (int r0 = wave; r0 < m; r0 += FLY * WAVES) {
            float g[FLY], t[FLY];
#pragma unroll
            for (int k = 0; k < FLY; ++k) {
                const int r = r0 + k * WAVES;
                g[k] = r < m ? gy[(int64_t)r * ldg + col] : 0.0f;
                t[k] = (ACT != ACT_NONE && r < m) ? y[(int64_t)r * ldy + col] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < FLY; ++k) {
                const int r = r0 + k * WAVES;
                if (ACT == ACT_RELU) g[k] = t[k] > 0.0f ? g[k] : 0.0f;
                if (ACT == ACT_TANH) g[k] = g[k] * (1.0f - t[k] * t[k]);
                if (r < m && (ACT != ACT_NONE || gz != gy)) gz[(int64_t)r * n + col] = g[k];
                acc += g[k];
            }
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && col < n && gbias) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += part[w][lane];
        gbias[col] = s;
    }
}

// ---- block reduction helper ---------------------------------------------------------------------------

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ float block_sum_256(float v, float *sm)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    const float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}

// ---- small-batch Linear + bias + activation forward on the f32 matrix cores ----------------------------------------
// y[m][n] = act(sum_k x[m][k] * W[n][k] + b[n]) for the learners' shapes (M = 256..4096 rows, N, K <= a few hundred):
// a rocBLAS GEMM followed by the bias/activation epilogue is two ~3 us launches for ~30 MFLOP; this is one. One wave owns
// a 16 x 16 output tile and runs v_mfma_f32_16x16x4_f32 (exact f32 fma chains, the guide's "FP32-input MFMA") over K with
// two independent accumulators. Operands go straight from L2 to registers: lane (r = lane & 15, h = lane >> 4) loads
// x[m0 + r][16c + 4h .. +3] and W[n0 + r][16c + 4h .. +3] as one 16-byte vector per 16-wide K chunk and feeds element e to
// MFMA step e -- the k order inside a chunk is permuted identically for both operands, which a sum over k does not see.
// No LDS: at these sizes the tile re-reads hit L2 and the launch is latency-, not bandwidth-bound.
// Grouped (stacked twin critics): x [G][M][K] (group stride may be 0: shared input), W [G][N][K], b [G][N], y [G][M][N].
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <bool VEC>
__device__ __forceinline__ float4 load_k4(const float *__restrict__ row, const int k, const int K, const bool valid)
{
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!valid || k >= K) return v;
    if (VEC) return *reinterpret_cast<const float4 *>(row + k);  // K % 4 == 0 and 16-byte aligned rows
    v.x = row[k];
    if (k + 1 < K) v.y = row[k + 1];
    if (k + 2 < K) v.z = row[k + 2];
    if (k + 3 < K) v.w = row[k + 3];
    return v;
}

// The same values from an UNCONDITIONAL 16-byte load at a clamped address (VEC: K % 4 == 0, K >= 4), zeroed afterwards: hipcc
// scalarises a float4 load under a condition into four branchy dword loads.
template <bool VEC>
__device__ __forceinline__ float4 load_k4_clamped(const float *__restrict__ row, const int k, const int K, const bool valid)
{
    if (!VEC) return load_k4<false>(row, k, K, valid);
    float4 v = *reinterpret_cast<const float4 *>(row + min(k, K - 4));
    const bool ok = valid && k < K;
    v.x = ok ? v.x : 0.0f; v.y = ok ? v.y : 0.0f; v.z = ok ? v.z : 0.0f; v.w = ok ? v.w : 0.0f;
    return v;
}

// Operand rows through BUFFER loads (VEC kernels): a 4-SGPR descriptor over the matrix + a 32-bit byte offset per lane. A quad outside
// the matrix (row >= rows, or the offset the caller substitutes for k >= K) reads as zeros by the descriptor's range check -- no
// branch around the load, no 64-bit address arithmetic per lane, no select on the loaded value. bytes < 2^30 (checked by the callers).
constexpr int BUF_OOB = 0x40000000;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t operand_rsrc(const float *base, const int64_t floats)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)(floats * 4), 0x00020000);
}

// rows whose length is not a multiple of 4 floats (the critics' first layer: K = obs_dim + act_dim = 6): four dword loads
__device__ __forceinline__ float4 load_k4_buf_scalar(const __amdgpu_buffer_rsrc_t rsrc, const int row_byte, const int k, const int K)
{
    float4 v;
    v.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, k < K ? row_byte + 4 * k : BUF_OOB, 0, 0));
    v.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, k + 1 < K ? row_byte + 4 * k + 4 : BUF_OOB, 0, 0));
    v.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, k + 2 < K ? row_byte + 4 * k + 8 : BUF_OOB, 0, 0));
    v.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, k + 3 < K ? row_byte + 4 * k + 12 : BUF_OOB, 0, 0));
    return v;
}

__device__ __forceinline__ float4 load_k4_buf(const __amdgpu_buffer_rsrc_t rsrc, const int row_byte, const int k, const int K)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, k < K ? row_byte + 4 * k : BUF_OOB, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// WAVES > 1: split-K -- wave s takes the 16-wide K chunks s, s + WAVES, ... (all of its loads in flight at once: one L2
// round trip for K = 256 with four waves) and the partial tiles meet in LDS.
template <int ACT, bool VEC, int WAVES, bool BUF = true>
__global__ __launch_bounds__(64 * WAVES) void linear_act_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                    const int64_t x_group_stride, const int ldx, const int M, const int N,
                                                                    const int K, const float *__restrict__ bias, float *__restrict__ y)
{
    // argument order: what the operand addresses need comes first -- the first 14 dwords of a kernel's arguments arrive preloaded
    // in SGPRs (Makefile: PRELOAD); a 15th (K, in the order x, stride, ldx, w, bias, y, M, N, K) cost a scalar load and its wait
    __shared__ f32x4 part[WAVES > 1 ? WAVES - 1 : 1][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, h = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int64_t g = blockIdx.z;
    const float *xr = x + g * x_group_stride + (int64_t)(m0 + r) * ldx;
    const float *wr = w + (g * N + n0 + r) * (int64_t)K;
    const bool row_ok = m0 + r < M, col_ok = n0 + r < N;
    // BUF: buffer loads (rows >= M / N and k >= K read as zeros by the range check); matrices of 1 GiB and more: plain loads
    const __amdgpu_buffer_rsrc_t rx = operand_rsrc(x + g * x_group_stride, (int64_t)(M - 1) * ldx + K);
    const __amdgpu_buffer_rsrc_t rw = operand_rsrc(w + g * N * (int64_t)K, (int64_t)N * K);
    const int xo = 4 * (m0 + r) * ldx, wo = 4 * (n0 + r) * K;
    const float bv = bias[g * N + min(n0 + r, N - 1)];  // the epilogue's bias: requested with the operands, not behind the reduction
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int UNROLL = 4;  // 8 vector loads in flight per lane
    for (int c0 = 16 * wave; c0 < K; c0 += 16 * WAVES * UNROLL) {
        float4 a[UNROLL], b[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = c0 + 16 * WAVES * u + 4 * h;
            a[u] = BUF ? (VEC ? load_k4_buf(rx, xo, k, K) : load_k4_buf_scalar(rx, xo, k, K)) : load_k4<VEC>(xr, k, K, row_ok);
            b[u] = BUF ? (VEC ? load_k4_buf(rw, wo, k, K) : load_k4_buf_scalar(rw, wo, k, K)) : load_k4<VEC>(wr, k, K, col_ok);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc1, 0, 0, 0);
        }
    }
    f32x4 acc = acc0 + acc1;
    if (WAVES > 1) {
        if (wave > 0) part[wave - 1][lane] = acc;
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int v = 0; v < WAVES - 1; ++v) acc += part[v][lane];
    }
    // C/D map of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + register
    const int col = n0 + r;
    if (col < N) {
        float *yo = y + (g * M + m0 + 4 * h) * (int64_t)N + col;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (m0 + 4 * h + e < M) {
                float v = acc[e] + bv;
                if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
                if (ACT == ACT_TANH) v = tanhf(v);
                yo[(int64_t)e * N] = v;
            }
        }
    }
}

// The FIRST layer behind ReplayBuffer.sample with the gather inside (cstr_linear_act_fwd_gather_f32): its input rows are the
// sampled transitions themselves, fetched from the replay ring by the (row, env) index pairs an earlier launch drew
// (cstr_rollout_step_f32), so the gather launch and one dependent launch boundary disappear. Rows [0, B) = observations of the
// batch, rows [B, 2B) = next observations (SAC's 2B-row actor pass; `both` = 0: B rows of next observations only, TD3's target
// actor). K = D <= 8 is ONE k chunk: one wave per 16 x 16 output tile, the same MFMA sequence as linear_act_fwd_kernel (bit-
// identical). The workgroups of the first column tile also MATERIALISE the packed batch for the launches behind this one
// (x_data = (obs | act), the observation columns of x_pi / x_next, rewards, dones*(1 - timeouts): ReplayBuffer._get_samples,
// core/common/buffers.py:316-323), and thread 0 of the launch performs the control-word updates of the gather launch (nobody
// reads them here).
struct GatherArgs {
    cstr_ring_t ring; int64_t *ring_ctl; int advance_ring; uint64_t *rng_ctl; uint64_t rng_advance;
    const int32_t *idx; int batch, both;
    float *x_data, *x_pi, *x_next, *out_done, *out_rew;
};

template <int ACT, int D, int A>
__global__ __launch_bounds__(64) void gather_linear_act_fwd_kernel(const GatherArgs g, const float *__restrict__ w, const float *__restrict__ bias,
                                                                   float *__restrict__ y, const int N)
{
    constexpr int W = D + A;
    const int lane = threadIdx.x, r = lane & 15, h = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16, B = g.batch, M = g.both ? 2 * B : B;
    const bool row_ok = m0 + r < M, col_ok = n0 + r < N;
    const int m = min(m0 + r, M - 1);
    const bool next = !g.both || m >= B;  // which half of the transition this row reads
    const int b = m >= B ? m - B : m;
    const int64_t o = (int64_t)g.idx[b] * g.ring.n_envs + g.idx[B + b];
    const float *src = (next ? g.ring.next_obs : g.ring.obs) + o * D;
    float4 a = *reinterpret_cast<const float4 *>(src + min(4 * h, D - 4));
    float4 bw = *reinterpret_cast<const float4 *>(w + (int64_t)min(n0 + r, N - 1) * D + min(4 * h, D - 4));
    const float bv = bias[min(n0 + r, N - 1)];  // (the epilogue's bias, requested with the operands)
    const bool first = blockIdx.x == 0 && row_ok;  // this workgroup also writes the packed batch rows of its 16 samples
    float4 obs = a;
    float2 act[A / 2];
    float dn = 0.0f, to = 0.0f, rw = 0.0f;
    if (first && 4 * h < D) {
        if (!g.both) obs = *reinterpret_cast<const float4 *>(g.ring.obs + o * D + 4 * h);
        if (!next || !g.both) {
            if (h == 0) {
#pragma unroll
                for (int jj = 0; jj < A / 2; ++jj) act[jj] = *reinterpret_cast<const float2 *>(g.ring.act + o * A + 2 * jj);
                dn = g.ring.done[o]; to = g.ring.timeout[o]; rw = g.ring.rew[o];
            }
        }
    }
    const bool k_ok = 4 * h < D;
    if (!k_ok || !row_ok) a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!k_ok || !col_ok) bw = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bw.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bw.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bw.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bw.w, acc1, 0, 0, 0);
    const f32x4 acc = acc0 + acc1;
    const int col = n0 + r;
    if (col < N) {
        float *yo = y + (int64_t)(m0 + 4 * h) * N + col;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (m0 + 4 * h + e < M) {
                float v = acc[e] + bv;
                if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
                if (ACT == ACT_TANH) v = tanhf(v);
                yo[(int64_t)e * N] = v;
            }
        }
    }
    if (first && k_ok) {
        const float2 lo = make_float2(a.x, a.y), hi = make_float2(a.z, a.w);  // this row's own half (row_ok, k_ok: `a` is the loaded value)
        if (g.both) {
            float *base = next ? g.x_next : g.x_pi;
            if (base) {
                float *xo = base + (int64_t)b * W + 4 * h;
                reinterpret_cast<float2 *>(xo)[0] = lo; reinterpret_cast<float2 *>(xo)[1] = hi;
            }
        } else {
            float *xn = g.x_next + (int64_t)b * W + 4 * h;
            reinterpret_cast<float2 *>(xn)[0] = lo; reinterpret_cast<float2 *>(xn)[1] = hi;
            if (g.x_pi) {
                float *xp = g.x_pi + (int64_t)b * W + 4 * h;
                reinterpret_cast<float2 *>(xp)[0] = make_float2(obs.x, obs.y); reinterpret_cast<float2 *>(xp)[1] = make_float2(obs.z, obs.w);
            }
        }
        if (!next || !g.both) {
            float *xd = g.x_data + (int64_t)b * W;
            reinterpret_cast<float2 *>(xd + 4 * h)[0] = make_float2(obs.x, obs.y);
            reinterpret_cast<float2 *>(xd + 4 * h)[1] = make_float2(obs.z, obs.w);
            if (h == 0) {
#pragma unroll
                for (int jj = 0; jj < A / 2; ++jj) *reinterpret_cast<float2 *>(xd + D + 2 * jj) = act[jj];
                g.out_done[b] = dn * (1.0f - to);  // buffers.py:322
                g.out_rew[b] = rw;
            }
        }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
        if (g.advance_ring) {  // ReplayBuffer.add's epilogue (core/common/buffers.py:280-283), left over by the rollout launch
            int64_t pos = g.ring_ctl[0] + 1;
            if (pos == g.ring.rows) { g.ring_ctl[1] = 1; pos = 0; }
            g.ring_ctl[0] = pos;
            g.ring_ctl[3] += 1;
        }
        if (g.rng_ctl) g.rng_ctl[1] += g.rng_advance;
    }
}

// Pointer-table form: up to CSTR_MAX_LINEAR_SETS independent Linear layers of one shape -- every agent's actor layer in
// MADDPG (core/maddpg/policies.py: one MLP per agent, parameters in per-agent arena slices) -- in ONE launch. Each set has
// its own input, weight, bias and output; the output may be a column block of a wider row (the joint action).
struct LinearSets { cstr_linear_set_t s[CSTR_MAX_LINEAR_SETS]; };

template <int ACT, bool VEC, int WAVES, bool BUF = true>
__global__ __launch_bounds__(64 * WAVES) void linear_act_fwd_sets_kernel(const LinearSets sets, const int M, const int N, const int K)
{
    __shared__ f32x4 part[WAVES > 1 ? WAVES - 1 : 1][64];
    const cstr_linear_set_t &st = sets.s[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, h = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const float *xr = st.x + (int64_t)(m0 + r) * st.ldx;
    const float *wr = st.w + (int64_t)(n0 + r) * K;
    const bool row_ok = m0 + r < M, col_ok = n0 + r < N;
    const __amdgpu_buffer_rsrc_t rx = operand_rsrc(st.x, (int64_t)(M - 1) * st.ldx + K);  // BUF: see linear_act_fwd_kernel
    const __amdgpu_buffer_rsrc_t rw = operand_rsrc(st.w, (int64_t)N * K);
    const int xo = 4 * (m0 + r) * (int)st.ldx, wo = 4 * (n0 + r) * K;
    const float bv = st.bias[min(n0 + r, N - 1)];  // the epilogue's bias: requested with the operands, not behind the reduction
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int UNROLL = 4;
    for (int c0 = 16 * wave; c0 < K; c0 += 16 * WAVES * UNROLL) {
        float4 a[UNROLL], b[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = c0 + 16 * WAVES * u + 4 * h;
            a[u] = BUF ? (VEC ? load_k4_buf(rx, xo, k, K) : load_k4_buf_scalar(rx, xo, k, K)) : load_k4<VEC>(xr, k, K, row_ok);
            b[u] = BUF ? (VEC ? load_k4_buf(rw, wo, k, K) : load_k4_buf_scalar(rw, wo, k, K)) : load_k4<VEC>(wr, k, K, col_ok);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc1, 0, 0, 0);
        }
    }
    f32x4 acc = acc0 + acc1;
    if (WAVES > 1) {
        if (wave > 0) part[wave - 1][lane] = acc;
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int v = 0; v < WAVES - 1; ++v) acc += part[v][lane];
    }
    const int col = n0 + r;
    if (col < N) {
        float *yo = st.y + (int64_t)(m0 + 4 * h) * st.ldy + col;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (m0 + 4 * h + e < M) {
                float v = acc[e] + bv;
                if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
                if (ACT == ACT_TANH) v = tanhf(v);
                yo[(int64_t)e * st.ldy] = v;
            }
        }
    }
}

// Input gradient of a Linear fused with the activation gradient of the layer BELOW it:
//   dz[m][k] = (sum_n gz[m][n] * W[n][k]) * act'(y[m][k])
// where y = act(...) is the lower layer's output (= this layer's input): the rocBLAS dX GEMM and the element-wise pass of
// bias_act_bwd in one launch. Same tiling as the forward kernel: one workgroup per 16 x 16 tile of dz, WAVES waves split
// the reduction over n (A = gz rows as 16-byte vectors along n, B = W[n][k] read as 64-byte row segments), partial tiles
// meet in LDS. (The lower layer's bias gradient = column sums of dz over ALL row tiles stays a separate launch: folding it
// in needs a cross-workgroup hand-over -- agent-scope fences cost more than the launch they would save.)
template <int ACT, bool VEC, int WAVES, bool BUF = true>
__global__ __launch_bounds__(64 * WAVES) void linear_bwd_input_kernel(const float *__restrict__ gz, const float *__restrict__ w,
                                                                      const float *__restrict__ y, float *__restrict__ dz,
                                                                      const int M, const int N, const int K, const int n_sum)
{
    // n_sum > 0: the n_sum groups share ONE input (stacked critics on the same rows): dz[m][k] = sum_g gz[g][m] . W[g][:, k]
    // -- the reduction simply continues over the groups, which replaces the batched GEMM + the sum over groups.
    __shared__ f32x4 part[WAVES > 1 ? WAVES - 1 : 1][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, h = lane >> 4;
    const int k0 = blockIdx.x * 16, m0 = blockIdx.y * 16, col = k0 + r;
    const int64_t g0 = n_sum > 0 ? 0 : blockIdx.z;
    const int n_groups = n_sum > 0 ? n_sum : 1;
    if (n_sum <= 0) {
        y += g0 * (int64_t)M * K;
        dz += g0 * (int64_t)M * K;
    }
    const bool col_ok = col < K, row_ok = m0 + r < M;
    // the lower layer's outputs the epilogue multiplies by (activation gradient) do not depend on the reduction: requested NOW
    // (column = lane & 15, rows 4 * (lane >> 4) + e of the tile), not behind the MFMA chain and the split-K combine
    float ty[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ACT != ACT_NONE && BUF) {
        const __amdgpu_buffer_rsrc_t ry = operand_rsrc(y, (int64_t)M * K);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            ty[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, col_ok ? 4 * ((m0 + 4 * h + e) * K + col) : BUF_OOB, 0, 0));
    }
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int UNROLL = 4;
    for (int gi = 0; gi < n_groups; ++gi) {
        const float *gzg = gz + (g0 + gi) * (int64_t)M * N, *wg = w + (g0 + gi) * (int64_t)N * K;
        const float *gr = gzg + (int64_t)(m0 + r) * N;
        // BUF: buffer loads -- gz rows >= M and W rows n >= N read as zeros by the descriptors' range check (lanes of columns >= K
        // may read other columns' values: they only feed output columns that are not stored)
        const __amdgpu_buffer_rsrc_t rg = operand_rsrc(gzg, BUF ? (int64_t)M * N : 0), rwt = operand_rsrc(wg, BUF ? (int64_t)N * K : 0);
        const int go = 4 * (m0 + r) * N;
        for (int c0 = 16 * wave; c0 < N; c0 += 16 * WAVES * UNROLL) {
            float4 a[UNROLL], b[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int n = c0 + 16 * WAVES * u + 4 * h;
                if (BUF) {
                    a[u] = VEC ? load_k4_buf(rg, go, n, N) : load_k4_buf_scalar(rg, go, n, N);
                    const int wo = 4 * (n * K + col);
                    b[u].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwt, wo, 0, 0));
                    b[u].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwt, wo + 4 * K, 0, 0));
                    b[u].z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwt, wo + 8 * K, 0, 0));
                    b[u].w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwt, wo + 12 * K, 0, 0));
                    continue;
                }
                a[u] = load_k4<VEC>(gr, n, N, row_ok);
                const float *wn = wg + (int64_t)n * K + col;
                b[u].x = (col_ok && n < N) ? wn[0] : 0.0f;
                b[u].y = (col_ok && n + 1 < N) ? wn[K] : 0.0f;
                b[u].z = (col_ok && n + 2 < N) ? wn[2 * (int64_t)K] : 0.0f;
                b[u].w = (col_ok && n + 3 < N) ? wn[3 * (int64_t)K] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc1, 0, 0, 0);
            }
        }
    }
    f32x4 acc = acc0 + acc1;
    if (WAVES > 1) {
        if (wave > 0) part[wave - 1][lane] = acc;
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int v = 0; v < WAVES - 1; ++v) acc += part[v][lane];
    }
    if (!col_ok) return;  // column = lane & 15, row = 4 * (lane >> 4) + register
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m = m0 + 4 * h + e;
        if (m < M) {
            const int64_t i = (int64_t)m * K + col;
            float d = acc[e];
            if (ACT != ACT_NONE) {
                const float t = BUF ? ty[e] : y[i];
                if (ACT == ACT_RELU) d = t > 0.0f ? d : 0.0f;
                if (ACT == ACT_TANH) d = d * (1.0f - t * t);
            }
            dz[i] = d;
        }
    }
}

// Weight + bias gradient of a Linear in one launch:
//   dW[n][k] = sum_m dz[m][n] * x[m][k],   db[n] = sum_m dz[m][n]
// (the rocBLAS dW GEMM + the column-sum launch). The reduction runs over the BATCH: one workgroup per 16 x 16 tile of dW,
// WAVES waves split the rows m; A = dz^T and B = x are both read as 64-byte row segments (lane r takes column n0 + r /
// k0 + r of row 16c + 4h + e). The workgroups of the first k strip also add up the dz values they load anyway: db needs no
// extra pass and, being reduced inside one workgroup in a fixed order, stays deterministic.
// ADAM: the workgroup that has reduced a tile of dW (and db) also applies the optimiser step to exactly those parameters
// (cstr_linear_bwd_weight_adam_sets_f32): parameter and moment quads are requested at entry, thread 0 turns the (pre-advanced)
// control words into the step's scalars beside the reduction, the tile's first wave updates p / m / v after the split-M combine.
#ifdef CSTR_POLICY_STAMPS  // diagnostic build only: phase stamps of the dW + Adam tile in the policy kernel's stamp buffer (tools/wa_stamps.py)
extern __device__ unsigned long long policy_stamps[4096 * 16 * 8];
#define WA_STAMP(i) do { if (ADAM && lane == 0 && blockIdx.x < 4096) policy_stamps[(blockIdx.x * 16 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WA_STAMP(i) do { } while (0)
#endif

struct AdamTile {
    float *w, *w_m, *w_v, *b, *b_m, *b_v, *shadow;
    const int64_t *adam_ctl; const double *lr; double beta1, beta2, eps; float gscale;
    float *w_target, *b_target; float tau;  // soft update of these parameters' own target with the values just computed, or NULL
};

template <int WAVES, bool BUF = false, bool ADAM = false>
__device__ __forceinline__ void linear_bwd_weight_tile(const float *__restrict__ dz, const float *__restrict__ x, const int ldx,
                                                       float *__restrict__ dw, float *__restrict__ db, const int M, const int N,
                                                       const int K, const int64_t g, const AdamTile *ad = nullptr,
                                                       const int tile_k = blockIdx.x, const int tile_n = blockIdx.y)
{
    __shared__ f32x4 part[WAVES > 1 ? WAVES - 1 : 1][64];
    __shared__ float colpart[WAVES][64];
    __shared__ AdamScalars adam_sc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, h = lane >> 4;
    const int k0 = tile_k * 16, n0 = tile_n * 16;
    const bool n_ok = n0 + r < N, k_ok = k0 + r < K;
    const bool want_db = db != nullptr && tile_k == 0;
    WA_STAMP(0);
    float pw[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pm[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, bw = 0.0f, bm = 0.0f, bvv = 0.0f;
    float pt[4] = {0.0f, 0.0f, 0.0f, 0.0f}, bt = 0.0f;  // the parameters' own target (soft update in the same pass)
    // The optimiser's operands (wave 0: parameter, moments, own target of the tile's 256 elements; the last wave: the step's scalars)
    // are requested BEHIND the wave's first batch of dz / x rows, not in front of it: their addresses come out of a chain of
    // argument reads (~1,300 cycles on wave 0, in-kernel stamps tools/wa_stamps.py) that used to delay wave 0's row requests -- and with
    // them the split-M barrier -- by as much; they are not needed before that barrier.
    bool adam_requested = false;
    auto adam_request = [&]() {
        adam_requested = true;
        if (wave == 0 && k_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n0 + 4 * h + e < N) {
                    const int64_t i = (int64_t)(n0 + 4 * h + e) * K + k0 + r;
                    pw[e] = ad->w[i]; pm[e] = ad->w_m[i]; pv[e] = ad->w_v[i];
                    // requested HERE with the parameter and its moments: read where it is used -- behind the stores of p / m / v, which
                    // the compiler cannot move it across -- it was one more cold round trip at the end of the launch
                    if (ad->w_target) pt[e] = ad->w_target[i];
                }
            }
        }
        if (wave == 0 && want_db && lane < 16 && n_ok) {
            bw = ad->b[n0 + lane]; bm = ad->b_m[n0 + lane]; bvv = ad->b_v[n0 + lane];
            if (ad->b_target) bt = ad->b_target[n0 + lane];
        }
        if (threadIdx.x == 64 * (WAVES - 1)) adam_sc = adam_scalars_advanced(ad->adam_ctl, ad->lr, ad->beta1, ad->beta2, ad->eps, ad->gscale);
    };
    WA_STAMP(1);
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    float colsum = 0.0f;
    constexpr int UNROLL = 4;
    // BUF: buffer loads -- rows m >= M read as zeros by the descriptors' range check, no branch per element (lanes of rows n >= N /
    // columns k >= K may read neighbouring values: they only feed outputs that are not stored and column sums that are not used)
    const __amdgpu_buffer_rsrc_t rdz = operand_rsrc(dz, BUF ? (int64_t)M * N : 0), rxx = operand_rsrc(x, BUF ? (int64_t)(M - 1) * ldx + K : 0);
    for (int c0 = 16 * wave; c0 < M; c0 += 16 * WAVES * UNROLL) {
        float4 a[UNROLL], b[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int m = c0 + 16 * WAVES * u + 4 * h;
            if (BUF) {
                const int ao = 4 * (m * N + n0 + r), bo = 4 * (m * ldx + k0 + r);
                a[u].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdz, ao, 0, 0));
                a[u].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdz, ao + 4 * N, 0, 0));
                a[u].z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdz, ao + 8 * N, 0, 0));
                a[u].w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdz, ao + 12 * N, 0, 0));
                b[u].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxx, bo, 0, 0));
                b[u].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxx, bo + 4 * ldx, 0, 0));
                b[u].z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxx, bo + 8 * ldx, 0, 0));
                b[u].w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxx, bo + 12 * ldx, 0, 0));
                continue;
            }
            const float *dr = dz + (int64_t)m * N + n0 + r;
            const float *xr = x + (int64_t)m * ldx + k0 + r;
            a[u].x = (n_ok && m < M) ? dr[0] : 0.0f;
            a[u].y = (n_ok && m + 1 < M) ? dr[N] : 0.0f;
            a[u].z = (n_ok && m + 2 < M) ? dr[2 * (int64_t)N] : 0.0f;
            a[u].w = (n_ok && m + 3 < M) ? dr[3 * (int64_t)N] : 0.0f;
            b[u].x = (k_ok && m < M) ? xr[0] : 0.0f;
            b[u].y = (k_ok && m + 1 < M) ? xr[ldx] : 0.0f;
            b[u].z = (k_ok && m + 2 < M) ? xr[2 * (int64_t)ldx] : 0.0f;
            b[u].w = (k_ok && m + 3 < M) ? xr[3 * (int64_t)ldx] : 0.0f;
        }
        if (ADAM && !adam_requested) adam_request();
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc1, 0, 0, 0);
            colsum += (a[u].x + a[u].y) + (a[u].z + a[u].w);
        }
    }
    if (ADAM && !adam_requested) adam_request();  // (a wave without rows: M <= 16 * wave)
    if (want_db) colpart[wave][lane] = colsum;
    f32x4 acc = acc0 + acc1;
    WA_STAMP(2);
    if (WAVES > 1) {
        if (wave > 0) part[wave - 1][lane] = acc;
        __syncthreads();
        WA_STAMP(3);
        if (wave > 0) return;
#pragma unroll
        for (int v = 0; v < WAVES - 1; ++v) acc += part[v][lane];
    } else if (ADAM) {
        __syncthreads();
    }
    // tile of dW: column (k) = lane & 15, row (n) = 4 * (lane >> 4) + register
    if (k_ok) {
        float *out = dw + (g * N + n0 + 4 * h) * (int64_t)K + k0 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n0 + 4 * h + e < N) out[(int64_t)e * K] = acc[e];
        if (ADAM) {
            const AdamScalars a = adam_sc;
            WA_STAMP(4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = n0 + 4 * h + e, colk = k0 + r;
                if (row < N) {
                    const int64_t i = (int64_t)row * K + colk;
                    adam1(pw[e], acc[e], pm[e], pv[e], a);
                    ad->w[i] = pw[e]; ad->w_m[i] = pm[e]; ad->w_v[i] = pv[e];
                    if (ad->w_target) ad->w_target[i] = polyak1(pw[e], pt[e], ad->tau, 1.0f - ad->tau);
                    if (ad->shadow)  // the tile-major copy the rollout kernel reads (cstr_policy_swizzle_f32's layout)
                        ad->shadow[(((int64_t)(row >> 4) * ((K + 15) >> 4) + (colk >> 4)) * 64 + (row & 15) + 16 * ((colk & 15) >> 2)) * 4 + (colk & 3)] = pw[e];
                }
            }
        }
    }
    if (want_db && lane < 16 && n_ok) {
        float sum = 0.0f;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) sum += (colpart[v][lane] + colpart[v][lane + 16]) + (colpart[v][lane + 32] + colpart[v][lane + 48]);
        db[g * N + n0 + lane] = sum;
        if (ADAM) {
            adam1(bw, sum, bm, bvv, adam_sc);
            ad->b[n0 + lane] = bw; ad->b_m[n0 + lane] = bm; ad->b_v[n0 + lane] = bvv;
            if (ad->b_target) ad->b_target[n0 + lane] = polyak1(bw, bt, ad->tau, 1.0f - ad->tau);
        }
    }
    WA_STAMP(5);
}

template <int WAVES, bool BUF>
__global__ __launch_bounds__(64 * WAVES) void linear_bwd_weight_kernel(const float *__restrict__ dz, const float *__restrict__ x,
                                                                       const int64_t x_group_stride, const int ldx,
                                                                       float *__restrict__ dw, float *__restrict__ db, const int M,
                                                                       const int N, const int K)
{
    const int64_t g = blockIdx.z;
    linear_bwd_weight_tile<WAVES, BUF>(dz + g * (int64_t)M * N, x + g * x_group_stride, ldx, dw, db, M, N, K, g);
}

// Several Linears' weight + bias gradients in ONE launch (an MLP's layers after its backward chain has produced every dz:
// the parameter gradients are leaves nobody waits for): blockIdx.z selects the operand set, every set has its own shape.
struct WgradSets { cstr_wgrad_set_t s[CSTR_MAX_LINEAR_SETS]; };

template <int WAVES, bool BUF>
__global__ __launch_bounds__(64 * WAVES) void linear_bwd_weight_sets_kernel(const WgradSets sets)
{
    const cstr_wgrad_set_t &q = sets.s[blockIdx.z];
    if ((int64_t)blockIdx.x * 16 >= q.k || (int64_t)blockIdx.y * 16 >= q.n) return;  // the grid covers the largest set
    linear_bwd_weight_tile<WAVES, BUF>(q.dz, q.x, (int)q.ldx, q.dw, q.db, (int)q.m, (int)q.n, (int)q.k, 0);
}

// The same launch with the OPTIMISER STEP inside (single-GPU training: nothing sits between a gradient and its Adam step): slices
// z < n_sets reduce one Linear's dW / db tile and update exactly those parameters; slices z >= n_sets are flat segments
// (cstr_adam_seg_t: parameters without a weight-gradient tile -- SAC's entropy coefficient -- and soft target updates). Step counters
// are pre-advanced by an earlier launch (cstr_chain_root_t.adam_advance): no control word is written here, no ticket.
// The grid is ONE dimension of exactly the workgroups that have work: `first[i]` = first workgroup of slice i (sets, then flat
// segments), first[n_sets + n_flat] = grid size; a set's workgroups walk its tiles k-major.
struct WgradAdamSets {
    cstr_wgrad_adam_set_t s[CSTR_MAX_LINEAR_SETS];
    cstr_adam_opt_t o[CSTR_MAX_ADAM_SEGS];
    cstr_adam_seg_t f[CSTR_MAX_ADAM_SEGS];
    int first[CSTR_MAX_LINEAR_SETS + CSTR_MAX_ADAM_SEGS + 1];
    int n_sets, n_flat;
};

template <int WAVES, bool BUF>
__global__ __launch_bounds__(64 * WAVES) void linear_bwd_weight_adam_sets_kernel(const WgradAdamSets sets)
{
    const int b = blockIdx.x, n_slices = sets.n_sets + sets.n_flat;
    int i = 0;
    while (i + 1 < n_slices && b >= sets.first[i + 1]) ++i;  // (uniform: a handful of scalar compares)
    const int local = b - sets.first[i], count = sets.first[i + 1] - sets.first[i];
    if (i >= sets.n_sets) {
        const cstr_adam_seg_t &f = sets.f[i - sets.n_sets];
        if (f.polyak_source) { polyak_body(f.polyak_source, f.param, (float)f.tau, (float)(1.0 - f.tau), f.n, local, count); return; }
        adam_body(f.param, f.grad, f.exp_avg, f.exp_avg_sq, f.adam_ctl, f.lr, f.beta1, f.beta2, f.eps, f.grad_scale, f.n, AdamShadow{nullptr, 0, 0, 4, 1},
                  f.own_target, (float)f.tau, (float)(1.0 - f.tau), true, local, count);
        return;
    }
    const cstr_wgrad_adam_set_t &q = sets.s[i];
    const int kt = (int)((q.g.k + 15) >> 4);
    const cstr_adam_opt_t &o = sets.o[q.opt];
    const AdamTile ad = {q.w, q.w_m, q.w_v, q.b, q.b_m, q.b_v, q.shadow, o.adam_ctl, o.lr, o.beta1, o.beta2, o.eps, o.grad_scale,
                         q.w_target, q.b_target, q.tau};
    linear_bwd_weight_tile<WAVES, BUF, true>(q.g.dz, q.g.x, (int)q.g.ldx, q.g.dw, q.g.db, (int)q.g.m, (int)q.g.n, (int)q.g.k, 0, &ad, local % kt,
                                             local / kt);
}

// ---- last hidden layer + scalar head of a Q network ------------------------------------------------------------
// create_mlp(..., output_dim = 1) ends in  y = act(z + b1);  q = y . w2 + b2  (core/common/torch_layers.py:110-183,
// ContinuousCritic: core/common/policies.py:960-987). The head is a matrix-VECTOR product (n = 1), which rocBLAS runs
// as an 8 us degenerate GEMM; here it rides the epilogue of the previous layer: one wave per row applies bias +
// activation to the GEMM output z IN PLACE and reduces the dot product with w2 in the same pass.
// Grouped: z [G][m][k], b1 [G][k], w2 [G][k], b2 [G], q [G][m].
template <int ACT, bool VEC4>
__global__ __launch_bounds__(256) void hidden_head_fwd_kernel(float *__restrict__ z, const float *__restrict__ b1,
                                                              const float *__restrict__ w2, const float *__restrict__ b2,
                                                              float *__restrict__ q, const int64_t rows, const int m, const int k)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;  // whole waves leave together; no barrier below
    const int64_t g = row / m;
    float *zr = z + row * k;
    const float *br = b1 + g * k, *wr = w2 + g * k;
    float acc = 0.0f;
    if (VEC4) {
        for (int c = lane * 4; c < k; c += 256) {
            float4 v = *reinterpret_cast<const float4 *>(zr + c);
            const float4 b = *reinterpret_cast<const float4 *>(br + c), w = *reinterpret_cast<const float4 *>(wr + c);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            if (ACT == ACT_RELU) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
            if (ACT == ACT_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
            *reinterpret_cast<float4 *>(zr + c) = v;
            acc += (v.x * w.x + v.y * w.y) + (v.z * w.z + v.w * w.w);
        }
    } else {
        for (int c = lane; c < k; c += 64) {
            float v = zr[c] + br[c];
            if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
            if (ACT == ACT_TANH) v = tanhf(v);
            zr[c] = v;
            acc += v * wr[c];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) q[row] = acc + b2[g];
}

// Backward of the pair, given gq = d(loss)/dq [G][m]:
//   dz[m][c]  = gq[m] * w2[c] * act'(y[m][c])      (head's input gradient folded into the activation gradient)
//   gb1[c]    = sum_m dz[m][c]                      gw2[c] = sum_m gq[m] * y[m][c]       gb2 = sum_m gq[m]
// Same ownership as bias_act_bwd_kernel: a workgroup owns 64 columns of one group, WAVES waves stride over the rows with
// eight rows of loads in flight; every output is written exactly once (deterministic, no atomics).
template <int ACT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void hidden_head_bwd_kernel(const float *__restrict__ gq, const float *__restrict__ y,
                                                                     const float *__restrict__ w2, float *__restrict__ dz,
                                                                     float *__restrict__ gb1, float *__restrict__ gw2,
                                                                     float *__restrict__ gb2, const int m, const int k)
{
    __shared__ float part[3][WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int64_t g = blockIdx.y, goff = g * (int64_t)m * k;
    gq += g * m;
    y += goff;
    dz += goff;
    const float w = col < k ? w2[g * k + col] : 0.0f;
    float s_b1 = 0.0f, s_w2 = 0.0f, s_b2 = 0.0f;
    constexpr int FLY = 8;
    for (int r0 = wave; r0 < m; r0 += FLY * WAVES) {
        float t[FLY], u[FLY];
#pragma unroll
        for (int j = 0; j < FLY; ++j) {
            const int r = r0 + j * WAVES;
            t[j] = (r < m && col < k) ? y[(int64_t)r * k + col] : 0.0f;
            u[j] = r < m ? gq[r] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < FLY; ++j) {
            const int r = r0 + j * WAVES;
            float d = u[j] * w;
            if (ACT == ACT_RELU) d = t[j] > 0.0f ? d : 0.0f;
            if (ACT == ACT_TANH) d = d * (1.0f - t[j] * t[j]);
            if (r < m && col < k) dz[(int64_t)r * k + col] = d;
            s_b1 += d;
            s_w2 += u[j] * t[j];
            s_b2 += u[j];
        }
    }
    part[0][wave][lane] = s_b1;
    part[1][wave][lane] = s_w2;
    part[2][wave][lane] = s_b2;
    __syncthreads();
    if (wave == 0 && gw2) {
        float a = 0.0f, b = 0.0f, c = 0.0f;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) { a += part[0][v][lane]; b += part[1][v][lane]; c += part[2][v][lane]; }
        if (col < k) {
            gb1[g * k + col] = a;
            gw2[g * k + col] = b;
        }
        if (blockIdx.x == 0 && lane == 0) gb2[g] = c;  // every lane of a wave saw the same gq rows
    }
}

// ---- squashed diagonal Gaussian (core/common/distributions.py:161-260) ----------------------------------

constexpr float LOG_STD_MIN = -20.0f, LOG_STD_MAX = 2.0f;  // core/sac/policies.py:20-22

// u = mean + exp(clamp(log_std)) * eps; a = tanh(u); logp = sum_j Normal.log_prob(u_j) - sum_j log(1 - a_j^2 + 1e-6)
// `in_stride`: row stride of mean / log_std_raw (act_dim when they are separate tensors, 2 * act_dim when they are the two
// halves of one merged-head GEMM output [B][2A]).
__global__ void squashed_gaussian_fwd_kernel(const float *__restrict__ mean, const float *__restrict__ log_std_raw,
                                             const float *__restrict__ eps, float *__restrict__ action,
                                             float *__restrict__ logp, const int64_t batch, const int act_dim, const int in_stride)
{
    const float half_log_2pi = 0.91893853320467274178f;  // math.log(math.sqrt(2 * math.pi))
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < batch; b += (int64_t)gridDim.x * blockDim.x) {
        float lp = 0.0f, corr = 0.0f;
        for (int j = 0; j < act_dim; ++j) {
            const int64_t i = b * act_dim + j, k = b * in_stride + j;
            const float mu = mean[k];
            const float ls = fminf(fmaxf(log_std_raw[k], LOG_STD_MIN), LOG_STD_MAX);
            const float s = expf(ls);
            const float u = mu + s * eps[i];  // Normal.rsample
            const float a = tanhf(u);
            action[i] = a;
            const float d = u - mu, var = s * s;
            lp += -(d * d) / (2.0f * var) - logf(s) - half_log_2pi;  // torch Normal.log_prob
            corr += logf(1.0f - a * a + 1e-6f);                      // distributions.py:232
        }
        if (logp) logp[b] = lp - corr;
    }
}

// Analytic backward. With d = s*eps the Gaussian term is -eps^2/2 - ls - c, so
//   dlogp/du_j = 2 a (1 - a^2) / (1 - a^2 + 1e-6),  da/du = 1 - a^2
//   g_mean = G_u,  g_ls = G_u * eps * s - g_logp,  g_raw = g_ls inside the clamp range, 0 outside
// (autograd through the reference's expression adds terms that cancel to rounding noise, ~1e-7 relative).
__global__ void squashed_gaussian_bwd_kernel(const float *__restrict__ g_action, const float *__restrict__ g_logp,
                                             const float *__restrict__ action, const float *__restrict__ log_std_raw,
                                             const float *__restrict__ eps, float *__restrict__ g_mean,
                                             float *__restrict__ g_log_std_raw, const int64_t batch, const int act_dim,
                                             const int in_stride, const int ga_stride)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < batch * act_dim; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / act_dim, j = i % act_dim, k = b * in_stride + j;
        const float a = action[i], raw = log_std_raw[k];
        const float ls = fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX);
        const float s = expf(ls);
        const float one_m = 1.0f - a * a;
        const float gl = g_logp ? g_logp[b] : 0.0f;
        const float ga = g_action ? g_action[b * ga_stride + j] : 0.0f;  // may be a column slice of a wider gradient
        const float gu = ga * one_m + gl * (2.0f * a * one_m / (one_m + 1e-6f));
        g_mean[k] = gu;
        const float gls = gu * eps[i] * s - gl;
        g_log_std_raw[k] = (raw >= LOG_STD_MIN && raw <= LOG_STD_MAX) ? gls : 0.0f;
    }
}

// ---- SAC actor head: bias + rsample + tanh + log-prob in one pass, noise from an in-kernel counter RNG ----------------
// The merged mu / log_std GEMM output z [B][2A] gets its bias here (no separate epilogue launch); eps ~ N(0, 1) is either
// given (teacher-forced tests) or drawn in the kernel: Philox4x32-10 keyed by rng_ctl[0], counter = (rng_ctl[1] + row, pair)
// -> Box-Muller, so no ATen generator launch (and none of the two generator-state fills PyTorch issues before every replay
// of a graph that captured torch.randn). rng_ctl = {seed, offset, ticket, -} lives in HBM; the last workgroup advances the
// offset, so graph replays continue the stream. `action` may be a column block of a wider row (the critic's input buffer).
__global__ __launch_bounds__(64) void gaussian_head_fwd_kernel(float *__restrict__ params, const float *__restrict__ bias,
                                                               float *__restrict__ eps, uint64_t *__restrict__ rng_ctl,
                                                               float *__restrict__ action, const int64_t action_stride,
                                                               float *__restrict__ logp, const int64_t batch, const int act_dim)
{
    const float half_log_2pi = 0.91893853320467274178f;
    const uint64_t seed = rng_ctl ? rng_ctl[0] : 0ull, base = rng_ctl ? rng_ctl[1] : 0ull;
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < batch; b += (int64_t)gridDim.x * blockDim.x) {
        float *pr = params + b * 2 * act_dim;
        float lp = 0.0f, corr = 0.0f;
        for (int j0 = 0; j0 < act_dim; j0 += 2) {
            float e[2];
            if (rng_ctl) {
                const uint64_t ctr = base + (uint64_t)b;
                uint32_t r[4];
                philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)(j0 >> 1), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
                box_muller(r[0], r[1], e[0], e[1]);
            }
            for (int jj = 0; jj < 2 && j0 + jj < act_dim; ++jj) {
                const int j = j0 + jj;
                if (rng_ctl) eps[b * act_dim + j] = e[jj];
                else e[jj] = eps[b * act_dim + j];
                float mu = pr[j], raw = pr[act_dim + j];
                if (bias) { mu += bias[j]; raw += bias[act_dim + j]; pr[j] = mu; pr[act_dim + j] = raw; }
                const float ls = fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX);  // core/sac/policies.py:173
                const float s = expf(ls);
                const float u = mu + s * e[jj];  // Normal.rsample (distributions.py:183)
                const float a = tanhf(u);
                action[b * action_stride + j] = a;
                const float d = u - mu, var = s * s;
                lp += -(d * d) / (2.0f * var) - logf(s) - half_log_2pi;
                corr += logf(1.0f - a * a + 1e-6f);  // distributions.py:232
            }
        }
        if (logp) logp[b] = lp - corr;
    }
    if (rng_ctl && last_block_ticket(reinterpret_cast<unsigned long long *>(rng_ctl + 2)) && threadIdx.x == 0)
        rng_ctl[1] = base + (uint64_t)batch;
}

// TD3 / MADDPG target policy smoothing (core/td3/td3.py:167-173; core/maddpg/maddpg.py:131-142):
//   noise = clamp(N(0, sigma), -clip, clip);  out = clamp(action + noise, -1, 1)
// in one launch (the reference's clone + normal_ + clamp + add + clamp). The noise is read when given (teacher-forced tests)
// and drawn from the same in-kernel Philox stream as the SAC head otherwise. `out` may be a column block of a wider row.
__global__ __launch_bounds__(64) void target_smooth_kernel(const float *__restrict__ action, const float *__restrict__ noise,
                                                           uint64_t *__restrict__ rng_ctl, const float sigma, const float clip,
                                                           float *__restrict__ out, const int64_t out_stride, const int64_t batch,
                                                           const int act_dim)
{
    const uint64_t seed = rng_ctl ? rng_ctl[0] : 0ull, base = rng_ctl ? rng_ctl[1] : 0ull;
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < batch; b += (int64_t)gridDim.x * blockDim.x) {
        for (int j0 = 0; j0 < act_dim; j0 += 2) {
            float e[2] = {0.0f, 0.0f};
            if (rng_ctl) {
                const uint64_t ctr = base + (uint64_t)b;
                uint32_t r[4];
                philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)(j0 >> 1), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
                box_muller(r[0], r[1], e[0], e[1]);
                e[0] *= sigma;
                e[1] *= sigma;
            }
            for (int jj = 0; jj < 2 && j0 + jj < act_dim; ++jj) {
                const int j = j0 + jj;
                const float z = rng_ctl ? e[jj] : noise[b * act_dim + j];
                const float nz = fminf(fmaxf(z, -clip), clip);
                out[b * out_stride + j] = fminf(fmaxf(action[b * act_dim + j] + nz, -1.0f), 1.0f);
            }
        }
    }
    if (rng_ctl && last_block_ticket(reinterpret_cast<unsigned long long *>(rng_ctl + 2)) && threadIdx.x == 0)
        rng_ctl[1] = base + (uint64_t)batch;
}

// A deterministic TARGET actor's last layer with the target policy smoothing inside (cstr_linear_smooth_fwd_f32): out = clamp(
// act(x W^T + b) + clamp(N(0, sigma), -clip, clip), -1, 1) written into the action columns of the target critic's input -- the
// Linear launch and the smoothing launch (cstr_target_smooth_f32) are one. n <= 16 (one column tile), k > 32: a workgroup per 16
// rows, four-way split-K exactly like linear_act_fwd_kernel<., ., 4> (same MFMA sequence, same LDS combine: bit-identical), the
// noise of the tile's rows drawn while the operand loads are in flight (same Philox counters as target_smooth_kernel).
template <int ACT, bool VEC>
__global__ __launch_bounds__(256) void linear_smooth_fwd_kernel(const float *__restrict__ x, const int ldx, const float *__restrict__ w,
                                                                const float *__restrict__ bias, const int M, const int N, const int K,
                                                                const float *__restrict__ noise, uint64_t *__restrict__ rng_ctl,
                                                                const float sigma, const float clip, float *__restrict__ out,
                                                                const int64_t out_stride)
{
    constexpr int WAVES = 4;
    __shared__ f32x4 part[WAVES - 1][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, h = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const bool col_ok = r < N;
    // operands through buffer descriptors (rows >= M / N and k >= K read as zeros: see linear_act_fwd_kernel); below 1 GiB (entry point)
    const __amdgpu_buffer_rsrc_t rx = operand_rsrc(x, (int64_t)(M - 1) * ldx + K), rw = operand_rsrc(w, (int64_t)N * K);
    const int xo = 4 * (m0 + r) * ldx, wo = 4 * r * K;
    const float bv = bias[min(r, N - 1)];  // (the epilogue's bias, requested with the operands)
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int UNROLL = 4;
    float4 a[UNROLL], b[UNROLL];
    const int c_first = 16 * wave;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {  // the first batch of operand loads ...
        const int k = c_first + 16 * WAVES * u + 4 * h;
        a[u] = VEC ? load_k4_buf(rx, xo, k, K) : load_k4_buf_scalar(rx, xo, k, K);
        b[u] = VEC ? load_k4_buf(rw, wo, k, K) : load_k4_buf_scalar(rw, wo, k, K);
    }
    // ... and, beside them, the smoothing noise of this lane's four rows (column r): target_smooth_kernel's counters
    float nz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint64_t seed = rng_ctl ? rng_ctl[0] : 0ull, base = rng_ctl ? rng_ctl[1] : 0ull;
    if (wave == 0 && col_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t row = m0 + 4 * h + e;
            float z = 0.0f;
            if (rng_ctl) {
                const uint64_t ctr = base + (uint64_t)row;
                uint32_t rr[4];
                float e0, e1;
                philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)(r >> 1), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rr);
                box_muller(rr[0], rr[1], e0, e1);
                e0 *= sigma;
                e1 *= sigma;
                z = (r & 1) ? e1 : e0;
            } else if (row < M) {
                z = noise[row * N + r];
            }
            nz[e] = fminf(fmaxf(z, -clip), clip);
        }
    }
    for (int c0 = c_first; c0 < K; c0 += 16 * WAVES * UNROLL) {
        if (c0 != c_first) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int k = c0 + 16 * WAVES * u + 4 * h;
                a[u] = VEC ? load_k4_buf(rx, xo, k, K) : load_k4_buf_scalar(rx, xo, k, K);
                b[u] = VEC ? load_k4_buf(rw, wo, k, K) : load_k4_buf_scalar(rw, wo, k, K);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc1, 0, 0, 0);
        }
    }
    f32x4 acc = acc0 + acc1;
    if (wave > 0) part[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int v = 0; v < WAVES - 1; ++v) acc += part[v][lane];
        if (col_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = m0 + 4 * h + e;
                if (row < M) {
                    float v = acc[e] + bv;
                    if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
                    if (ACT == ACT_TANH) v = tanhf(v);
                    out[row * out_stride + r] = fminf(fmaxf(v + nz[e], -1.0f), 1.0f);
                }
            }
        }
    }
    if (rng_ctl && last_block_ticket(reinterpret_cast<unsigned long long *>(rng_ctl + 2)) && threadIdx.x == 0)
        rng_ctl[1] = base + (uint64_t)M;
}


// The same head INCLUDING its GEMM: the merged (mu | log_std) Linear has 2A <= 8 outputs, i.e. it is 2A dot products per
// row, not a matrix-matrix product. One wave per row: 16-byte loads of the latent row and of the 2A weight rows, xor-shuffle
// reductions, then the sampling arithmetic of gaussian_head_fwd_kernel on the reduced values (computed redundantly by every
// lane, stored by lane 0). params [B][2A] is written for the backward.
__global__ __launch_bounds__(1024) void gaussian_head_gemm_fwd_kernel(const float *__restrict__ hid, const float *__restrict__ w,
                                                                     const int ldh, const int K, const int64_t batch, const int act_dim,
                                                                     const float *__restrict__ bias, uint64_t *__restrict__ rng_ctl,
                                                                     float *__restrict__ params, float *__restrict__ eps,
                                                                     float *__restrict__ action, const int64_t action_stride,
                                                                     float *__restrict__ logp)
{   // (argument order: the dot products' operands first -- see linear_act_fwd_kernel)
    const float half_log_2pi = 0.91893853320467274178f;
    const uint64_t seed = rng_ctl ? rng_ctl[0] : 0ull, base = rng_ctl ? rng_ctl[1] : 0ull;
    const int lane = threadIdx.x & 63;
    const int64_t b = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b < batch) {  // (no early return: every wave reaches the ticket barrier below)
        const float *hr = hid + b * ldh;
        float p[2 * CSTR_MAX_HEAD_ACT];
#pragma unroll
        for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) p[j] = 0.0f;
        for (int c = lane * 4; c < K; c += 256) {
            const float4 hv = *reinterpret_cast<const float4 *>(hr + c);
#pragma unroll
            for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) {
                if (j >= 2 * act_dim) break;
                const float4 wv = *reinterpret_cast<const float4 *>(w + (int64_t)j * K + c);
                p[j] += (hv.x * wv.x + hv.y * wv.y) + (hv.z * wv.z + hv.w * wv.w);
            }
        }
#pragma unroll
        for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) {
            if (j >= 2 * act_dim) break;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) p[j] += __shfl_xor(p[j], o, 64);
            p[j] += bias[j];
        }
        float lp = 0.0f, corr = 0.0f;
        for (int j0 = 0; j0 < act_dim; j0 += 2) {
            float e[2] = {0.0f, 0.0f};
            if (rng_ctl) {
                const uint64_t ctr = base + (uint64_t)b;
                uint32_t r[4];
                philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)(j0 >> 1), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
                box_muller(r[0], r[1], e[0], e[1]);
            }
            for (int jj = 0; jj < 2 && j0 + jj < act_dim; ++jj) {
                const int j = j0 + jj;
                if (!rng_ctl) e[jj] = eps[b * act_dim + j];
                float mu = 0.0f, raw = 0.0f;
#pragma unroll
                for (int q = 0; q < 2 * CSTR_MAX_HEAD_ACT; ++q) {  // selects with static register indices
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
                if (lane == 0) {
                    if (rng_ctl) eps[b * act_dim + j] = e[jj];
                    params[b * 2 * act_dim + j] = mu;
                    params[b * 2 * act_dim + act_dim + j] = raw;
                    action[b * action_stride + j] = a;
                }
            }
        }
        if (logp && lane == 0) logp[b] = lp - corr;
    }
    if (rng_ctl && last_block_ticket(reinterpret_cast<unsigned long long *>(rng_ctl + 2)) && threadIdx.x == 0)
        rng_ctl[1] = base + (uint64_t)batch;
}

// Backward (same algebra as squashed_gaussian_bwd_kernel) + the merged head's bias gradient (column sums over the batch).
// One workgroup: the batch is a few hundred rows.
__global__ __launch_bounds__(256) void gaussian_head_bwd_kernel(const float *__restrict__ g_action, const int64_t ga_stride,
                                                                const float *__restrict__ g_logp, const float *__restrict__ action,
                                                                const int64_t action_stride, const float *__restrict__ params,
                                                                const float *__restrict__ eps, float *__restrict__ g_params,
                                                                float *__restrict__ g_bias, const int64_t batch, const int act_dim)
{
    __shared__ float sm[4];
    float col[2 * CSTR_MAX_HEAD_ACT];
#pragma unroll
    for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) col[j] = 0.0f;
    for (int64_t b = threadIdx.x; b < batch; b += 256) {
        const float gl = g_logp ? g_logp[b] : 0.0f;
#pragma unroll
        for (int j = 0; j < CSTR_MAX_HEAD_ACT; ++j) {
            if (j >= act_dim) break;
            const float a = action[b * action_stride + j], raw = params[b * 2 * act_dim + act_dim + j];
            const float s = expf(fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX));
            const float one_m = 1.0f - a * a;
            const float ga = g_action ? g_action[b * ga_stride + j] : 0.0f;
            const float gu = ga * one_m + gl * (2.0f * a * one_m / (one_m + 1e-6f));
            const float gls = (raw >= LOG_STD_MIN && raw <= LOG_STD_MAX) ? gu * eps[b * act_dim + j] * s - gl : 0.0f;
            g_params[b * 2 * act_dim + j] = gu;
            g_params[b * 2 * act_dim + act_dim + j] = gls;
            col[j] += gu;
            col[CSTR_MAX_HEAD_ACT + j] += gls;
        }
    }
    if (g_bias) {
#pragma unroll
        for (int j = 0; j < CSTR_MAX_HEAD_ACT; ++j) {
            if (j >= act_dim) break;
            const float m = block_sum_256(col[j], sm), l = block_sum_256(col[CSTR_MAX_HEAD_ACT + j], sm);
            if (threadIdx.x == 0) { g_bias[j] = m; g_bias[act_dim + j] = l; }
        }
    }
}

// The same backward when the head's Linear sits on a fused hidden layer: besides g_params, the workgroup (4 rows) carries the
// gradient through the head's weights and the hidden layer's activation,
//   dz[b][c] = (sum_j g_params[b][j] * w[j][c]) * act'(hidden[b][c])      (2A <= 8 terms per element)
// i.e. d(loss)/d(pre-activation) of the layer below -- what the Linear backward kernels take as input. The head's own
// dW / db come from cstr_linear_bwd_weight_f32(g_params, hidden).
constexpr int HEAD_BWD_ROWS = 4;

template <int ACT>
__global__ __launch_bounds__(256) void gaussian_head_bwd_input_kernel(const float *__restrict__ g_action, const int64_t ga_stride,
                                                                      const float *__restrict__ g_logp, const float *__restrict__ action,
                                                                      const int64_t action_stride, const float *__restrict__ params,
                                                                      const float *__restrict__ eps, const float *__restrict__ w,
                                                                      const float *__restrict__ hidden, const int64_t ldh,
                                                                      float *__restrict__ g_params, float *__restrict__ dz,
                                                                      const int64_t batch, const int act_dim, const int width)
{
    __shared__ float gp[HEAD_BWD_ROWS][2 * CSTR_MAX_HEAD_ACT];
    const int64_t row0 = (int64_t)blockIdx.x * HEAD_BWD_ROWS;
    const int t = threadIdx.x;
    // Phase 2's operands (the head's weight rows and this row's hidden activations) do not depend on phase 1: with 16-byte rows
    // every lane requests its first quad of each NOW, so their round trip runs beside phase 1's dependent loads instead of behind the
    // barrier, and a 256-wide layer is ONE pass per lane instead of four dependent ones.
    const bool vec = (width & 3) == 0 && (ldh & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(hidden) |
                                                               reinterpret_cast<uintptr_t>(dz)) & 15) == 0;
    const int c4 = 4 * (t & 63);
    const int64_t b2 = row0 + (t >> 6);
    float4 wq[2 * CSTR_MAX_HEAD_ACT], hq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const bool pre = vec && b2 < batch && c4 < width;
    if (pre) {
#pragma unroll
        for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) wq[j] = *reinterpret_cast<const float4 *>(w + (int64_t)min(j, 2 * act_dim - 1) * width + c4);
        if (ACT != ACT_NONE) hq = *reinterpret_cast<const float4 *>(hidden + b2 * ldh + c4);
    }
    if (t < HEAD_BWD_ROWS * CSTR_MAX_HEAD_ACT) {
        const int r = t / CSTR_MAX_HEAD_ACT, j = t % CSTR_MAX_HEAD_ACT;
        const int64_t b = row0 + r;
        if (b < batch && j < act_dim) {
            const float gl = g_logp ? g_logp[b] : 0.0f;
            const float a = action[b * action_stride + j], raw = params[b * 2 * act_dim + act_dim + j];
            const float s = expf(fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX));
            const float one_m = 1.0f - a * a;
            const float ga = g_action ? g_action[b * ga_stride + j] : 0.0f;
            const float gu = ga * one_m + gl * (2.0f * a * one_m / (one_m + 1e-6f));
            const float gls = (raw >= LOG_STD_MIN && raw <= LOG_STD_MAX) ? gu * eps[b * act_dim + j] * s - gl : 0.0f;
            g_params[b * 2 * act_dim + j] = gu;
            g_params[b * 2 * act_dim + act_dim + j] = gls;
            gp[r][j] = gu;
            gp[r][act_dim + j] = gls;
        }
    }
    __syncthreads();
    const int r = t >> 6, lane = t & 63;  // one wave per row
    const int64_t b = row0 + r;
    if (b >= batch) return;
    float g[2 * CSTR_MAX_HEAD_ACT];
#pragma unroll
    for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) g[j] = j < 2 * act_dim ? gp[r][j] : 0.0f;
    if (vec) {  // the same fma chain per column (j ascending), four columns per lane
        for (int c = c4; c < width; c += 256) {
            if (c != c4) {
#pragma unroll
                for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) wq[j] = *reinterpret_cast<const float4 *>(w + (int64_t)min(j, 2 * act_dim - 1) * width + c);
                if (ACT != ACT_NONE) hq = *reinterpret_cast<const float4 *>(hidden + b * ldh + c);
            }
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) {
                if (j >= 2 * act_dim) break;
                acc.x = __fmaf_rn(g[j], wq[j].x, acc.x); acc.y = __fmaf_rn(g[j], wq[j].y, acc.y);
                acc.z = __fmaf_rn(g[j], wq[j].z, acc.z); acc.w = __fmaf_rn(g[j], wq[j].w, acc.w);
            }
            if (ACT == ACT_RELU) {
                acc.x = hq.x > 0.0f ? acc.x : 0.0f; acc.y = hq.y > 0.0f ? acc.y : 0.0f;
                acc.z = hq.z > 0.0f ? acc.z : 0.0f; acc.w = hq.w > 0.0f ? acc.w : 0.0f;
            }
            if (ACT == ACT_TANH) {
                acc.x = acc.x * (1.0f - hq.x * hq.x); acc.y = acc.y * (1.0f - hq.y * hq.y);
                acc.z = acc.z * (1.0f - hq.z * hq.z); acc.w = acc.w * (1.0f - hq.w * hq.w);
            }
            *reinterpret_cast<float4 *>(dz + b * width + c) = acc;
        }
        return;
    }
    for (int c = lane; c < width; c += 64) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) {
            if (j >= 2 * act_dim) break;
            acc = __fmaf_rn(g[j], w[(int64_t)j * width + c], acc);
        }
        if (ACT != ACT_NONE) {
            const float y = hidden[b * ldh + c];
            if (ACT == ACT_RELU) acc = y > 0.0f ? acc : 0.0f;
            if (ACT == ACT_TANH) acc = acc * (1.0f - y * y);
        }
        dz[b * width + c] = acc;
    }
}

// ---- whole policy network, inference only ------------------------------------------------------------------------
// create_mlp(obs_dim, ., [H1, H2]) + the action head for MANY rows and no backward (the rollout's 4096-row policy pass, the
// target's pi(next_obs) pass): ONE launch instead of three to six. A workgroup owns 16 rows and keeps their activations
// in LDS the whole way:
//   phase 1  h1 = act(x W1^T + b1)          K0 <= 64 inputs: plain FMAs, x staged in LDS
//   phase 2  h2 = act(h1 W2^T + b2)         f32 matrix cores: 8 waves take the 16-column tiles round robin, A = h1 from LDS
//                                           (row stride H1 + 4 floats: conflict-free 16-byte reads), B = W2 rows from L2
//   phase 3  head: n_out <= 8 dot products per row (one wave per row, like gaussian_head_gemm_fwd_kernel), then either the
//            squashed-Gaussian sample (HEAD 0: noise from the Philox stream or given, tanh, optional log-prob) or the
//            deterministic output activation (HEAD 1: TD3 / DDPG actors).
// Every workgroup streams all of W2 (H1 x H2 x 4 bytes from L2) for 16 x H1 x H2 MACs: at 4096 rows that is one workgroup
// per CU and the matrix-core time of 16 x 256 x 256 (about 3.4 us) bounds the launch, not the weight traffic.
struct PolicyArgs {
    const float *x; int64_t ldx; int k0;
    const float *w1, *b1; int h1;
    const float *w2, *b2; int h2;
    const float *w3, *b3; int act_dim, out_act;
    const float *eps_in; uint64_t *rng_ctl;
    float *action; int64_t action_stride; float *logp; int64_t m; const float *w2s; int flags;
};

constexpr int POLICY_ROWS = 16, POLICY_WAVES = 8;

// "Last workgroup out" with a two-level ticket for large grids: 256 same-address atomics serialise to ~4 us; here the
// workgroups draw from 8 sub-counters (blockIdx & 7: different L2 channels, in parallel) and only the last of each group
// touches the top counter. Same contract as last_block_ticket (cstr_device.h): counters self-reset for the next launch.
__device__ __forceinline__ bool last_block_ticket_tree(unsigned long long *top, unsigned long long *sub)
{
    __shared__ int tree_is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & 7u, G = gridDim.x;
        const unsigned long long members = (G >> 3) + (g < (G & 7u) ? 1u : 0u), groups = G < 8u ? G : 8u;
        int last = 0;
        if (atomicAdd(sub + g, 1ULL) == members - 1ULL) {
            sub[g] = 0ULL;
            if (atomicAdd(top, 1ULL) == groups - 1ULL) {
                *top = 0ULL;
                last = 1;
            }
        }
        tree_is_last = last;
    }
    __syncthreads();
    return tree_is_last != 0;
}

// One hidden layer for the workgroup's 16 rows: out[16][N] (LDS, row stride so) = act(in[16][K] W^T + b). The waves take the
// 16-column tiles round robin, TWO per pass: every lane issues ALL of its loads for both tiles (up to 256 k values each) at
// once (A = input rows, shared by the two tiles: global memory for the first layer, LDS after it; B = weight rows from L2), so
// a layer of up to 16 x POLICY_WAVES columns costs one memory round trip.
template <int ACT, bool A_GLOBAL, bool VEC, bool SWZ = false, int UNROLL = 16>
__device__ __forceinline__ void policy_layer(const float *__restrict__ in, const int64_t in_stride, const bool in_row_ok, const int K,
                                             const float *__restrict__ w, const float *__restrict__ bias, const int N,
                                             float *__restrict__ out, const int so, const int wave = threadIdx.x >> 6,
                                             const int n_waves = POLICY_WAVES)
{
    const int lane = threadIdx.x & 63, r = lane & 15, h = lane >> 4;
    const int tiles = (N + 15) >> 4;
    const float *ar = in + r * in_stride;
    const float4 *wsw = reinterpret_cast<const float4 *>(w);
    const int kc = (K + 15) >> 4;
    for (int t = wave; t < tiles; t += 2 * n_waves) {
        const int n0 = t * 16, n1 = n0 + 16 * n_waves;
        const bool ok0 = n0 + r < N, ok1 = n1 + r < N;
        const bool second = t + n_waves < tiles;  // wave-uniform
        const float *wr0 = w + (int64_t)(n0 + r) * K, *wr1 = w + (int64_t)(n1 + r) * K;
        f32x4 c00 = {0.0f, 0.0f, 0.0f, 0.0f}, c01 = c00, c10 = c00, c11 = c00;
        for (int c0 = 0; c0 < K; c0 += 16 * UNROLL) {
            float4 av[UNROLL], b0[UNROLL], b1[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int k = c0 + 16 * u + 4 * h;
                // SWZ: `w` is the tile-major copy [tile][k chunk][lane] of float4 (cstr_policy_swizzle_f32): a wave's load
                // instruction reads 1 KB of consecutive bytes instead of sixteen 64-byte row pieces (7.2 -> ~2 us per 256 KB)
                b0[u] = SWZ ? (k < K ? wsw[((int64_t)t * kc + (c0 >> 4) + u) * 64 + lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f))
                            : load_k4<VEC>(wr0, k, K, ok0);
                av[u] = A_GLOBAL ? load_k4<VEC>(ar, k, K, in_row_ok) : load_k4<true>(ar, k, K, true);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                b1[u] = SWZ ? ((second && c0 + 16 * u + 4 * h < K) ? wsw[((int64_t)(t + n_waves) * kc + (c0 >> 4) + u) * 64 + lane]
                                                                    : make_float4(0.0f, 0.0f, 0.0f, 0.0f))
                            : load_k4<VEC>(wr1, c0 + 16 * u + 4 * h, K, ok1 && second);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (c0 + 16 * u >= K) break;  // wave-uniform: no matrix-core passes on all-zero chunks (K = 4: one chunk)
                c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, b0[u].x, c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, b0[u].y, c01, 0, 0, 0);
                c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, b0[u].z, c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, b0[u].w, c01, 0, 0, 0);
            }
            if (second) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    if (c0 + 16 * u >= K) break;
                    c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, b1[u].x, c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, b1[u].y, c11, 0, 0, 0);
                    c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, b1[u].z, c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, b1[u].w, c11, 0, 0, 0);
                }
            }
        }
        // column = lane & 15, row = 4 * (lane >> 4) + register
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int col = (half ? n1 : n0) + r;
            if (half ? (ok1 && second) : ok0) {
                const f32x4 acc = half ? c10 + c11 : c00 + c01;
                const float bb = bias[col];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[e] + bb;
                    if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
                    if (ACT == ACT_TANH) v = tanhf(v);
                    out[(4 * h + e) * so + col] = v;
                }
            }
        }
    }
}

// w [n][k] -> tile-major float4 entries (see cstr_policy_swizzle_f32 in the header); one thread per entry
__global__ void policy_swizzle_kernel(const float *__restrict__ w, const int n, const int k, float4 *__restrict__ out, const int64_t entries)
{
    const int kc = (k + 15) >> 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < entries; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int64_t tc = i >> 6;
        const int tile = (int)(tc / kc), chunk = (int)(tc - (int64_t)tile * kc);
        const int row = 16 * tile + (lane & 15), col = 16 * chunk + 4 * (lane >> 4);
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (row < n && col < k) v = *reinterpret_cast<const float4 *>(w + (int64_t)row * k + col);  // k % 4 == 0
        out[i] = v;
    }
}

template <int ACT, int HEAD, bool VEC0>
__global__ __launch_bounds__(64 * POLICY_WAVES) void policy_rows_fwd_kernel(const PolicyArgs a)
{
    extern __shared__ float policy_lds[];
    const int H1 = a.h1, H2 = a.h2, S1 = H1 + 4, S2 = H2 + 4;
    float *h1s = policy_lds, *h2s = h1s + POLICY_ROWS * S1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * POLICY_ROWS;
    const uint64_t seed = a.rng_ctl ? a.rng_ctl[0] : 0ull, base = a.rng_ctl ? a.rng_ctl[1] : 0ull;
    const int n_out = HEAD == 0 ? 2 * a.act_dim : a.act_dim;
    policy_layer<ACT, true, VEC0>(a.x + m0 * a.ldx, a.ldx, m0 + (lane & 15) < a.m, a.k0, a.w1, a.b1, H1, h1s, S1);
    __syncthreads();
    if (a.w2s) policy_layer<ACT, false, true, true>(h1s, S1, true, H1, a.w2s, a.b2, H2, h2s, S2);
    else policy_layer<ACT, false, true>(h1s, S1, true, H1, a.w2, a.b2, H2, h2s, S2);
    __syncthreads();

    // head: POLICY_ROWS / POLICY_WAVES = 2 rows per wave, evaluated together (one pass over the head's weights)
    constexpr int RPW = POLICY_ROWS / POLICY_WAVES;
    float p[RPW][2 * CSTR_MAX_HEAD_ACT];
#pragma unroll
    for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) p[q][j] = 0.0f;
    for (int c = lane * 4; c < H2; c += 256) {
        float4 hv[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) hv[q] = *reinterpret_cast<const float4 *>(h2s + (wave + q * POLICY_WAVES) * S2 + c);
#pragma unroll
        for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) {
            if (j >= n_out) break;
            const float4 wv = *reinterpret_cast<const float4 *>(a.w3 + (int64_t)j * H2 + c);
#pragma unroll
            for (int q = 0; q < RPW; ++q) p[q][j] += (hv[q].x * wv.x + hv[q].y * wv.y) + (hv[q].z * wv.z + hv[q].w * wv.w);
        }
    }
    // the reduced head outputs meet in LDS (h1s is free again); then ONE wave samples all 16 rows, a lane per row: the
    // Philox / Box-Muller / tanh / log chain is ~1.5 k instructions, run once instead of once per row
    float *ps = h1s;  // [POLICY_ROWS][2 * CSTR_MAX_HEAD_ACT]
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
#pragma unroll
        for (int j = 0; j < 2 * CSTR_MAX_HEAD_ACT; ++j) {
            if (j >= n_out) break;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) p[q][j] += __shfl_xor(p[q][j], o, 64);
            if (lane == 0) ps[(wave + q * POLICY_WAVES) * 2 * CSTR_MAX_HEAD_ACT + j] = p[q][j] + a.b3[j];
        }
    }
    __syncthreads();
    const int64_t row = m0 + lane;
    if (wave == 0 && lane < POLICY_ROWS && row < a.m) {
        const float *pr = ps + lane * 2 * CSTR_MAX_HEAD_ACT;
        if (HEAD == 1) {
            for (int j = 0; j < n_out; ++j) {
                float v = pr[j];
                if (a.out_act == ACT_RELU) v = fmaxf(v, 0.0f);
                if (a.out_act == ACT_TANH) v = tanhf(v);
                a.action[row * a.action_stride + j] = v;
            }
        } else {
            const float half_log_2pi = 0.91893853320467274178f;
            float lp = 0.0f, corr = 0.0f;
            for (int j0 = 0; j0 < a.act_dim; j0 += 2) {
                float e[2] = {0.0f, 0.0f};
                if (!a.eps_in) {
                    const uint64_t ctr = base + (uint64_t)row;
                    uint32_t rnd[4];
                    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)(j0 >> 1), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
                    box_muller(rnd[0], rnd[1], e[0], e[1]);
                }
                for (int jj = 0; jj < 2 && j0 + jj < a.act_dim; ++jj) {
                    const int j = j0 + jj;
                    if (a.eps_in) e[jj] = a.eps_in[row * a.act_dim + j];
                    const float mu = pr[j], raw = pr[a.act_dim + j];
                    const float ls = fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX);
                    const float sd = expf(ls);
                    const float u = mu + sd * e[jj];
                    const float act = tanhf(u);
                    const float d = u - mu, var = sd * sd;
                    lp += -(d * d) / (2.0f * var) - logf(sd) - half_log_2pi;
                    corr += logf(1.0f - act * act + 1e-6f);
                    a.action[row * a.action_stride + j] = act;
                }
            }
            if (a.logp) a.logp[row] = lp - corr;
        }
    }
    if (a.rng_ctl && !(a.flags & 1) && last_block_ticket_tree(reinterpret_cast<unsigned long long *>(a.rng_ctl + 2),
                                                              reinterpret_cast<unsigned long long *>(a.rng_ctl + 4)) && tid == 0)
        a.rng_ctl[1] = base + (uint64_t)a.m;
}

// ---- the same network, software-pipelined (needs the tile-major copy of W2; h1, h2 <= 512) ---------------------------------
// What the first version leaves on the table at 4096 rows (r01: 17.8 us = 0.20 of the f32 matrix-core peak): per wave ALL of
// layer 2's loads are issued, waited for, and only then do the 128 MFMAs run (no load / MFMA overlap); the head is 48 cross-lane
// shuffles per wave followed by a ~1.5 k-instruction Philox / Box-Muller / tanh / log chain on 16 lanes of ONE wave while
// the other seven idle; a 256-workgroup ticket closes the launch. Here:
//   * layer 2 runs as a 2-stage register pipeline over K: stage s + 1's operands (4 k-chunks of 16: A from LDS, B = 1 KB full-
//     line loads from the tile-major copy) are in flight while stage s's 32 MFMAs issue; stage 0 of B and the head's weights
//     are requested BEFORE layer 1 (they do not depend on activations);
//   * the Gaussian noise does not depend on the network either: wave 7 draws it (Philox + Box-Muller, a lane per row) while
//     waves 0-6 compute layer 1, whose time is the latency of its observation / weight loads anyway;
//   * the head is one 16 x 16 MFMA tile with K split over the 8 waves (2 k-chunks each at H2 = 256), partial sums combined in a
//     fixed order from LDS, and the sampling tail runs on 16 x A lanes, a lane per (row, action);
//   * bit 0 of cstr_policy_mlp_t.reserved: the caller advances the Philox offset (cstr_collect_step_f32 does it in its own
//     last-workgroup epilogue), no 256-workgroup ticket at the end of this launch.
// Layer 1 / layer 2 values are bit-identical to the first version (same k order, same accumulator assignment); the head's
// summation order differs (split-K MFMA instead of per-lane partial dots + shuffle tree).
constexpr int V2_CH = 4, V2_MAX_WIDTH = 512;

// 1 KB piece `piece` of the tile-major weight copy: lane's 16 bytes through a buffer load whose per-lane offset (16 * lane) is ONE
// register for every request and whose piece offset is scalar -- no 64-bit vector address arithmetic between the MFMAs
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned int;
__device__ __forceinline__ float4 v2_piece(const __amdgpu_buffer_rsrc_t rs, const int lane16, const int piece)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, piece << 10, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

#ifdef CSTR_POLICY_STAMPS  // diagnostic build only (make diag): s_memtime per wave at the phase boundaries, read by tools/policy_stamps.py
__device__ unsigned long long policy_stamps[4096 * 16 * 8];
#define V2_STAMP(i) do { if (lane == 0 && blockIdx.x < 4096) policy_stamps[(blockIdx.x * 16 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// second bank (wave slots 8-15 of the workgroup's 16): stamps inside layer 2 of the h <= 256 instantiation
#define V2_STAMP2(i) do { if (lane == 0 && blockIdx.x < 4096) policy_stamps[(blockIdx.x * 16 + 8 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define V2_STAMP(i) do { } while (0)
#define V2_STAMP2(i) do { } while (0)
#endif

// FULL: every chunk of the stage is inside K (all stages but the last): no checks between the MFMAs
// Workgroup barrier of the pipelined policy / rollout kernel: the waves hand activations over in LDS only, so the barrier waits for
// LDS traffic (lgkmcnt) and NOT for the weight stream the waves have in flight -- `__syncthreads()` is a workgroup release fence,
// i.e. s_waitcnt vmcnt(0): every wave sat out its outstanding global loads at each barrier (-DCSTR_POLICY_SYNC=1: the old form).
#if defined(CSTR_POLICY_SYNC) && CSTR_POLICY_SYNC
#define POLICY_BARRIER() __syncthreads()
#else
#define POLICY_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
#endif

template <bool SECOND, bool FULL>
__device__ __forceinline__ void v2_mfma_stage(const int c_begin, const int kc, const float4 (&av)[V2_CH], const float4 (&b0)[V2_CH],
                                              const float4 (&b1)[V2_CH], f32x4 &c00, f32x4 &c01, f32x4 &c10, f32x4 &c11)
{
#pragma unroll
    for (int u = 0; u < V2_CH; ++u) {
        if (!FULL && c_begin + u >= kc) break;  // wave-uniform
        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, b0[u].x, c00, 0, 0, 0);
        if (SECOND) c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, b1[u].x, c10, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, b0[u].y, c01, 0, 0, 0);
        if (SECOND) c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, b1[u].y, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, b0[u].z, c00, 0, 0, 0);
        if (SECOND) c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, b1[u].z, c10, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, b0[u].w, c01, 0, 0, 0);
        if (SECOND) c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, b1[u].w, c11, 0, 0, 0);
    }
}

// What the rollout launch (cstr_rollout_step_f32) does beyond the policy network: the fused collect step of the workgroup's 16 envs
// on the sampling tail's lanes, and -- on ONE otherwise idle wave of the last workgroup, beside that tail -- this iteration's
// replay index draw (numpy legacy MT19937, cstr_mt_device.h).
struct RolloutArgs {
    cstr_coef_t k; CollectArgs c; const int64_t *ring_ctl; int layout, integrator;
    uint32_t *mt_state; int32_t *sample_idx; int batch;
};

// SHAPE: 0 = widths <= 512, 1 = widths <= 256 (one tile pair per wave), 2 = exactly 256 x 256 and 3 = exactly 400 x 300 (the class-default
// networks of SAC and of TD3 / DDPG / MADDPG: every loop bound and piece index of layer 2 is a constant -- with run-time widths the
// hoisted bounds and indices overflow the scalar registers and come back through v_readlane in front of every request)
template <int ACT, int HEAD, bool VEC0, bool K0_SMALL, int SHAPE, bool FUSE, int ENV = -1>
__device__ __forceinline__ void policy_rows_v2_body(const PolicyArgs &a, const RolloutArgs *ro)
{
    constexpr bool SMALL = SHAPE == 1 || SHAPE == 2, EXACT = SHAPE >= 2;
    constexpr int H1C = SHAPE == 2 ? 256 : 400, H2C = SHAPE == 2 ? 256 : 300;
    constexpr int WAVES = POLICY_WAVES, L1_WAVES = 4;
    constexpr int MAXC = SMALL ? 16 : V2_MAX_WIDTH / 16;  // 16-wide k chunks / column tiles the instantiation is sized for (register budget)
    constexpr int V2_HEAD_Q = MAXC / WAVES;     // head k chunks per wave
    constexpr int L1_T = MAXC / L1_WAVES;       // layer-1 column tiles per layer-1 wave
#ifndef CSTR_L2_DEPTH
#define CSTR_L2_DEPTH 4
#endif
    constexpr int NB = SMALL ? CSTR_L2_DEPTH : CSTR_L2_DEPTH + 1;  // the ring of layer 2's per-chunk pipeline: B chunks per tile in flight (5 divides TD3's 400 = 25 chunks)
    extern __shared__ float policy_lds[];
    const int H1 = EXACT ? H1C : a.h1, H2 = EXACT ? H2C : a.h2, kc1 = (H1 + 15) >> 4, kc2 = (H2 + 15) >> 4, S1 = 16 * kc1 + 4, S2 = 16 * kc2 + 4;
    float *h1s = policy_lds, *h2s = h1s + POLICY_ROWS * S1, *part = h2s + POLICY_ROWS * S2;  // part [8 waves][16 rows][8]
    float *eps_s = part + WAVES * POLICY_ROWS * 8, *term = eps_s + POLICY_ROWS * 8;          // eps [16][8] (radius), terms [2][16][8]
    float *csn_s = term + 2 * POLICY_ROWS * 8;                                                 // [16][8]: cos | sin of the draw's angle
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, h = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * POLICY_ROWS;
    const float4 *w2s = reinterpret_cast<const float4 *>(a.w2s);
    const bool l1_wave = wave < L1_WAVES;
    V2_STAMP(0);
    // L2 warm-up: a launch starts with its XCD's L2 cold, and every line of the weight copy is first touched by ONE of the XCD's
    // 32 workgroups, whose wave then waits a fabric round trip in the middle of layer 2 (the slowest wave of a workgroup arrived
    // ~1.3 us after the typical one). So each wave first touches one distinct 1 KB piece (8 lines, lanes 0-7; per XCD, under the
    // observed round-robin placement, blocks b, b + 8, ... cover the whole copy; any other placement is only slower): the copy is
    // in every L2 one fabric round trip after launch. The value is kept live to the end so that the load is not eliminated.
    // Round 3 (in-kernel stamps): the touch is WAITED for before a wave's first operand requests of layer 2 -- without the wait they race
    // the warm-up into a cold L2 (first barrier 1,700 cycles later) -- but layer 1's own operand requests leave right behind it, in the
    // same cold round trip.
    constexpr int WARM_N = SMALL ? 1 : 4;  // pieces <= 256 (h <= 256) / <= 1024 (h <= 512) over 256 loaders
    float warm[WARM_N];
    auto warm_issue = [&]() {
        const int pieces = kc2 * kc1, loader = (int)((blockIdx.x >> 3) & 31u) * WAVES + wave;
        const float *wf = reinterpret_cast<const float *>(w2s);
#pragma unroll
        for (int i = 0; i < WARM_N; ++i) warm[i] = wf[(int64_t)min(loader + 32 * WAVES * i, pieces - 1) * 256 + (lane & 7) * 32];
    };
    auto warm_wait = [&]() {
#pragma unroll
        for (int i = 0; i < WARM_N; ++i) asm volatile("" ::"v"(warm[i]));  // the value has to be in its register here
    };
    // ---- the launch's first requests: everything up to the sched_barrier reads only what the rollout kernel's leading scalar parameters
    // carry (preloaded SGPRs); the first wait for the argument structs sits behind them ----
    float4 xa = make_float4(0.0f, 0.0f, 0.0f, 0.0f), w1v[L1_T];
    float b1v[L1_T];
    warm_issue();  // in front of layer 1's operands: their out-of-range zeroing below already waits for them
    if (K0_SMALL && l1_wave) {
        xa = load_k4_clamped<VEC0>(a.x + min(m0 + r, a.m - 1) * a.ldx, 4 * h, a.k0, m0 + r < a.m);
#pragma unroll
        for (int i = 0; i < L1_T; ++i) {
            const int n = 16 * (wave + L1_WAVES * i) + r;
            w1v[i] = load_k4_clamped<VEC0>(a.w1 + (int64_t)min(n, H1 - 1) * a.k0, 4 * h, a.k0, n < H1);
            b1v[i] = a.b1[min(n, H1 - 1)];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool draw = HEAD == 0 && a.rng_ctl != nullptr;
    const int n_out = HEAD == 0 ? 2 * a.act_dim : a.act_dim;
    // the lanes that step an env at the end (thread 8 * row of the sampling tail, waves 0-1 = layer-1 waves)
    const bool env_lane = FUSE && K0_SMALL && tid < POLICY_ROWS * 8 && (tid & 7) == 0 && m0 + (tid >> 3) < a.m;
    CollectIn env_in;   // layout 2 (two trains): a lane per env
    CollectInQ env_q;   // layouts 0, 1: four lanes per env (collect_env_quad)
    int64_t ring_pos = 0;

    // Roles before the first barrier (round 3; the round-2 form requested a wave's WHOLE layer-2 operand here, which serialised the
    // stream and the MFMAs -- see the layer-2 loop below):
    //   waves 0-3: layer 1 (its operand requests left at the top of the kernel; one k chunk for k0 <= 16: lane (r, h) holds
    //              x[row r][4h..4h+3] and W1[column tile row r][4h..4h+3]), then the first NB chunks of their layer-2 ring;
    //   waves 4-7: the first NB chunks of their ring; waves 6-7 then draw the Gaussian noise (Philox + one half of Box-Muller each, a
    //              lane per row: dependent VALU work that does not depend on the network).
    // Every load is UNCONDITIONAL on a clamped (always valid) address: hipcc scalarises a float4 load under a condition into four
    // branchy dword loads; out-of-range chunks are never fed to an MFMA, out-of-range head lanes are zeroed after the load.
    float4 bq0[NB], bq1[NB], w3q[V2_HEAD_Q];
    float b2q0, b2q1;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);  // the wave index as a SCALAR (piece offsets of the buffer loads)
    const int t0 = min(wave_s, kc2 - 1), t1 = min(wave_s + WAVES, kc2 - 1);
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w2s), 0, kc2 * kc1 * 1024, 0x00020000);
    const int lane16 = 16 * lane;
#define V2_REQUEST_B(FROM, TO) do { _Pragma("unroll") for (int u = (FROM); u < (TO); ++u) { const int c = min(u, kc1 - 1); \
        bq0[u] = v2_piece(w2rs, lane16, t0 * kc1 + c); bq1[u] = v2_piece(w2rs, lane16, t1 * kc1 + c); } } while (0)
    // History of this spot (tools/rollout_ab.py, graph-replayed launches): round 2 requested 12 of a wave's 16 chunks per tile ahead of
    // the first barrier and the rest behind it (12.93 us per launch), LDS-only barriers 12.81, all 16 ahead 12.56; the per-chunk ring
    // requests NB chunks here and one chunk per 8 MFMAs afterwards (9.4 us with everything else of round 3).
    constexpr int PRE_YOUNG = NB, PRE_OLD = NB;
    // Who draws the Gaussian noise (dependent VALU work that needs nothing from memory but the stream's control words): behind the
    // wave's ring requests (in front of them the requests queue behind everybody else's); on a layer-1 wave, between requesting its
    // operands and using them, the draw delayed the first barrier by as much as it saved (round 2).
    // Round 3: TWO waves, each with half of Box-Muller behind its own Philox call (same counters, same words): wave 7 the radius
    // sqrt(-2 log u1), wave 6 cos / sin of the angle; the sampling tail multiplies them (z = r * cos, r * sin: the same single product).
    const bool noise_wave = draw && wave >= WAVES - 2;
    const bool radius_wave = wave == WAVES - 1;  // wave-uniform
    auto draw_noise = [&]() {
        // the stream's control words are read HERE, by the one wave that needs them: a dependent scalar load at the top of the
        // kernel sits in front of every wave's first vector loads (the scalar-memory wait in front of them covers it too)
        const uint64_t seed = a.rng_ctl[0], base = a.rng_ctl[1];
        const int64_t row = m0 + lane;
        if (lane < POLICY_ROWS && row < a.m) {
            for (int j0 = 0; j0 < a.act_dim; j0 += 2) {
                const uint64_t ctr = base + (uint64_t)row;
                uint32_t rnd[4];
                philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)(j0 >> 1), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
                if (radius_wave) {
                    const float rad = box_muller_radius(rnd[0]);
                    eps_s[lane * 8 + j0] = rad;
                    eps_s[lane * 8 + j0 + 1] = rad;
                } else {
                    float cs, sn;
                    box_muller_angle(rnd[1], cs, sn);
                    csn_s[lane * 8 + j0] = cs;
                    csn_s[lane * 8 + j0 + 1] = sn;
                }
            }
        }
        V2_STAMP(7);
    };
    // zero the k padding of both activation images (widths that are not multiples of 16: the MFMA chunks read them)
    {
        const int p1 = 16 * kc1 - H1, p2 = 16 * kc2 - H2;
        if (p1 > 0 && tid < POLICY_ROWS * p1) h1s[(tid / max(p1, 1)) * S1 + H1 + tid % max(p1, 1)] = 0.0f;
        if (p2 > 0 && tid < POLICY_ROWS * p2) h2s[(tid / max(p2, 1)) * S2 + H2 + tid % max(p2, 1)] = 0.0f;
    }
    if (!l1_wave) {
        warm_wait();
        V2_REQUEST_B(0, PRE_YOUNG);
        if (noise_wave) draw_noise();  // behind its ring requests (8 loads, not the 32 of round 2: their issue is short)
    } else if (K0_SMALL) {
#pragma unroll
        for (int i = 0; i < L1_T; ++i) {
            const int t = wave + L1_WAVES * i;
            if (t >= kc1) break;  // wave-uniform
            // the first version's accumulator assignment (elements x, z -> one chain, y, w -> the other): same bits
            f32x4 c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = c0;
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, w1v[i].x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, w1v[i].y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, w1v[i].z, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, w1v[i].w, c1, 0, 0, 0);
            const f32x4 acc = c0 + c1;
            const int col = 16 * t + r;
            if (col < H1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[e] + b1v[i];
                    if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
                    if (ACT == ACT_TANH) v = tanhf(v);
                    h1s[(4 * h + e) * S1 + col] = v;
                }
            }
        }
        warm_wait();
        V2_REQUEST_B(0, PRE_OLD);
    } else {
        warm_wait();
        policy_layer<ACT, true, VEC0, false, 16>(a.x + m0 * a.ldx, a.ldx, m0 + r < a.m, a.k0, a.w1, a.b1, H1, h1s, S1, wave, L1_WAVES);
        V2_REQUEST_B(0, PRE_OLD);
    }
    // the tail's head biases (thread = 8 * row + j reads slots j and act_dim + j): requested now, used ~8 us later
    const float b3v0 = a.b3[min(tid & 7, n_out - 1)], b3v1 = a.b3[min((HEAD == 0 ? a.act_dim : 0) + (tid & 7), n_out - 1)];
    {   // needed last: the first tile pair's bias values and the head's rows
        b2q0 = a.b2[min(16 * t0 + r, H2 - 1)];
        b2q1 = a.b2[min(16 * t1 + r, H2 - 1)];
        const float *w3r = a.w3 + (int64_t)min(r, n_out - 1) * H2;
#pragma unroll
        for (int q = 0; q < V2_HEAD_Q; ++q) w3q[q] = load_k4_clamped<true>(w3r, 16 * (wave + WAVES * q) + 4 * h, H2, r < n_out);
    }
    V2_STAMP(1);
    POLICY_BARRIER();
    V2_STAMP(2);
#undef V2_REQUEST_B
    if (FUSE && K0_SMALL && wave < (POLICY_ROWS * 8) / 64) {
        // The collect step's operands of this lane's env, used by the sampling tail: requested HERE, behind the first barrier -- in
        // front of it the argument reads of this block (a chain of scalar-cache round trips) held waves 0-1, and with them the barrier,
        // ~1,100 cycles longer than waves 2-3; the loads land while the wave walks its first chunks of layer 2.
        const int64_t e = min(m0 + (tid >> 3), a.m - 1);
        // the ring position as a VECTOR load (per-lane zero offset the compiler cannot see through): a scalar load's cold round trip
        // counts in lgkmcnt, which every LDS wait of layer 2 would sit out
        int zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        ring_pos = ro->ring_ctl[zero];
        if (ENV == 0 || (ENV < 0 && ro->layout == 0)) collect_quad_load<0>(ro->c, e, tid & 7, env_q);
        else if (ENV == 3 || (ENV < 0 && ro->layout == 1)) collect_quad_load<1>(ro->c, e, tid & 7, env_q);
        else if (env_lane) collect_env_load<2>(ro->c, e, env_in);
    }

    // layer 2: tile pairs (t, t + 8); column = lane & 15, row = 4 * (lane >> 4) + register in the epilogue
    const float *ar = h1s + r * S1 + 4 * h;
#define V2_EPILOGUE(T, SECOND, FIRST) do { _Pragma("unroll") for (int half = 0; half < 2; ++half) { \
            const int col = 16 * (half ? (T) + WAVES : (T)) + r; \
            if ((half == 0 || (SECOND)) && col < H2) { \
                const f32x4 acc = half ? c10 + c11 : c00 + c01; \
                const float bb = (FIRST) ? (half ? b2q1 : b2q0) : a.b2[col]; \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) { \
                    float v = acc[e] + bb; \
                    if (ACT == ACT_RELU) v = fmaxf(v, 0.0f); \
                    if (ACT == ACT_TANH) v = tanhf(v); \
                    h2s[(4 * h + e) * S2 + col] = v; } } } } while (0)
    {
        // Tile pairs (t, t + 8) per wave, K walked chunk by chunk with a ring of NB B chunks per tile in registers: chunk c's 8 MFMAs, then
        // the request for chunk c + NB into the slot they have just read (beyond K: the first chunks of the wave's NEXT tile pair when
        // the ring divides K, else a harmless re-read and the next pair starts with fresh requests). Why per chunk and per wave
        // (tools/probes/stream_mfma_probe.hip, profiles/r03_stream_mfma_probe.txt; one workgroup per CU, 256 KB stream + 1024 MFMAs per CU):
        //   * requesting the whole operand ahead and then consuming it in k order costs 16,200 cycles (the waves' load issue and MFMA issue
        //     serialise: stream 5,000 + MFMAs 8,400 would be the sum, 8,400 the ideal); the round-2 form of this loop was that;
        //   * loader waves beside matrix waves on the same SIMD slow each other down (loads x 3.7, MFMAs x 1.5);
        //   * a wave that interleaves ONE load with the MFMAs of one chunk, 4 chunks ahead, does both in 9,900 cycles.
        const bool wrap = kc1 % NB == 0;  // wave-uniform
        int ring_t = wave_s;              // the tile pair whose first NB chunks the ring holds (requested before the first barrier)
        // Both loops are FULLY unrolled (<= 2 tile pairs, <= MAXC chunks): a ring carried around a back-edge is copied into fixed
        // registers there, behind s_waitcnt vmcnt(0) -- the pipeline would drain once per trip.
#pragma unroll
        for (int pair = 0; pair < MAXC / (2 * WAVES); ++pair) {
            const int t = wave_s + 2 * WAVES * pair;
            if (t >= kc2) break;                  // wave-uniform
            const bool second = t + WAVES < kc2;  // wave-uniform
            const int ta = t, tb = second ? t + WAVES : t, tn = t + 2 * WAVES;
            const int na = tn < kc2 ? tn : ta, nb = tn + WAVES < kc2 ? tn + WAVES : na;  // the next pair (clamped to a valid tile)
            const int base_a = ta * kc1, base_b = tb * kc1, base_na = (na - 1) * kc1, base_nb = (nb - 1) * kc1;  // piece = base + chunk
            if (ring_t != t) {  // (the ring does not divide K: no requests were carried over from the previous pair)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    bq0[u] = v2_piece(w2rs, lane16, ta * kc1 + min(u, kc1 - 1));
                    bq1[u] = v2_piece(w2rs, lane16, tb * kc1 + min(u, kc1 - 1));
                }
            }
            f32x4 c00 = {0.0f, 0.0f, 0.0f, 0.0f}, c01 = c00, c10 = c00, c11 = c00;
            if (t == wave_s) V2_STAMP2(0);
            float4 a_cur = *reinterpret_cast<const float4 *>(ar);  // the A operand (layer 1's activations, LDS) runs one chunk ahead
            auto walk = [&](auto second_c) {
                constexpr bool SECOND = decltype(second_c)::value;
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    {
                        if (c >= kc1) break;   // wave-uniform (a constant in the exact-shape instantiations)
                        const int u = c % NB;  // a constant once the loop is unrolled: the ring is registers
                        const float4 a4 = a_cur, b0 = bq0[u], b1 = bq1[u];
                        a_cur = *reinterpret_cast<const float4 *>(ar + 16 * min(c + 1, kc1 - 1));
                        // the first version's accumulator assignment (elements x, z -> one chain, y, w -> the other): same bits
                        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0.x, c00, 0, 0, 0);
                        if (SECOND) c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b1.x, c10, 0, 0, 0);
                        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b0.y, c01, 0, 0, 0);
                        if (SECOND) c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1.y, c11, 0, 0, 0);
                        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b0.z, c00, 0, 0, 0);
                        if (SECOND) c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b1.z, c10, 0, 0, 0);
                        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b0.w, c01, 0, 0, 0);
                        if (SECOND) c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b1.w, c11, 0, 0, 0);
                        // the scheduler must leave the request HERE: left alone it sinks every load down to its use (fewer registers),
                        // i.e. load / wait for it / 8 MFMAs -- no chunk in flight at all. Unconditional on a clamped (always valid) piece.
                        __builtin_amdgcn_sched_barrier(0);
                        const int cn = c + NB;
                        if (EXACT && SMALL && cn >= MAXC) continue;  // one pair per wave: nothing left to request
                        const bool next = wrap && cn >= kc1;  // wave-uniform: the request belongs to the next pair
                        // (readfirstlane: under scalar-register pressure the compiler moves this index arithmetic to the vector unit and
                        // then wraps every load in a first-lane loop)
                        bq0[u] = v2_piece(w2rs, lane16, __builtin_amdgcn_readfirstlane(next ? base_na + cn : base_a + min(cn, kc1 - 1)));
                        bq1[u] = v2_piece(w2rs, lane16, __builtin_amdgcn_readfirstlane(next ? base_nb + cn : base_b + min(cn, kc1 - 1)));
                        __builtin_amdgcn_sched_barrier(0);
                        if (SMALL && t == wave_s && (c & 3) == 3) V2_STAMP2(1 + (c >> 2));
                    }
                }
            };
            if (second) walk(std::true_type{});
            else walk(std::false_type{});
            if (wrap) ring_t = tn;
            V2_EPILOGUE(t, second, t == wave_s);
            if (t == wave_s) V2_STAMP2(5);
        }
    }
#undef V2_EPILOGUE
    V2_STAMP(3);
    // No workgroup barrier here: the head's k chunks of wave w are c = w + 8 q -- exactly the column tiles this wave has just
    // written (t = w, w + 8, w + 16, w + 24), so it reads back only its OWN LDS stores; a wave's LDS operations complete in order,
    // the fence keeps the compiler from moving the reads up and drains the stores.
    __threadfence_block();
    V2_STAMP(4);

    // head: ONE 16 x 16 tile (rows x outputs, outputs >= n_out are zero columns), K split over the waves
    {
        f32x4 p0 = {0.0f, 0.0f, 0.0f, 0.0f}, p1 = p0;
        const float *hr = h2s + r * S2 + 4 * h;
#pragma unroll
        for (int q = 0; q < V2_HEAD_Q; ++q) {
            const int c = wave + WAVES * q;
            if (c >= kc2) break;  // wave-uniform
            const float4 av = *reinterpret_cast<const float4 *>(hr + 16 * c);
            p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, w3q[q].x, p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, w3q[q].y, p1, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, w3q[q].z, p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, w3q[q].w, p1, 0, 0, 0);
        }
        p0 += p1;
        if (r < 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) part[(wave * POLICY_ROWS + 4 * h + e) * 8 + r] = p0[e];
        }
    }
    POLICY_BARRIER();
    V2_STAMP(5);

    // tail: a lane per (row, output slot): thread = 8 * row + j
    const int trow = tid >> 3, j = tid & 7;
    const int64_t row = m0 + trow;
    const bool live = tid < POLICY_ROWS * 8 && row < a.m;
    auto head_out = [&](const int jj, const float bias) {
        float sum = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sum += part[(w * POLICY_ROWS + trow) * 8 + jj];
        return sum + bias;
    };
    float act_out = 0.0f;  // this lane's action component (FUSE: handed to the row's collect lane)
    if (HEAD == 1) {
        if (live && j < n_out) {
            float v = head_out(j, b3v0);
            if (a.out_act == ACT_RELU) v = fmaxf(v, 0.0f);
            if (a.out_act == ACT_TANH) v = tanhf(v);
            if (!FUSE || a.action) a.action[row * a.action_stride + j] = v;
            act_out = v;
        }
    } else {
        if (live && j < a.act_dim) {
            const float half_log_2pi = 0.91893853320467274178f;
            const float mu = head_out(j, b3v0), raw = head_out(a.act_dim + j, b3v1);
            const float e = a.eps_in ? a.eps_in[row * a.act_dim + j] : eps_s[trow * 8 + j] * csn_s[trow * 8 + j];
            const float ls = fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX);
            const float sd = expf(ls);
            const float u = mu + sd * e;
            const float act = tanhf(u);
            const float d = u - mu, var = sd * sd;
            if (!FUSE || a.action) a.action[row * a.action_stride + j] = act;
            act_out = act;
            if (a.logp) {
                term[trow * 8 + j] = -(d * d) / (2.0f * var) - logf(sd) - half_log_2pi;
                term[POLICY_ROWS * 8 + trow * 8 + j] = logf(1.0f - act * act + 1e-6f);
            }
        }
        if (a.logp) {  // kernel-uniform
            __syncthreads();
            if (live && j == 0) {
                float lp = 0.0f, corr = 0.0f;
                for (int jj = 0; jj < a.act_dim; ++jj) {  // the reference's summation order over the action dimension
                    lp += term[trow * 8 + jj];
                    corr += term[POLICY_ROWS * 8 + trow * 8 + jj];
                }
                a.logp[row] = lp - corr;
            }
        }
    }
    if (FUSE && wave < (POLICY_ROWS * 8) / 64) {
        V2_STAMP(7);
        // the fused collect step (cstr_collect_step_f32) of the workgroup's 16 envs: lane (row, 0) of the tail gathers the row's
        // action components from its neighbours and steps env `row`; the ring position is only READ here (the gather launch behind
        // this one advances it: cstr_replay_gather_packed_f32), so no workgroup hands anything to another one
        const int64_t ring_row = ring_pos * ro->c.ring.n_envs, env = min(row, a.m - 1);
        // ENV >= 0: the environment variant is a template constant (2 * layout + integrator) -- the six variants of the collect step are
        // ~2.5 k instructions each, and only one of them runs
        const bool eu = ENV >= 0 ? (ENV & 1) == 0 : ro->integrator == CSTR_INTEGRATOR_EULER;
        if (ENV == 0 || (ENV < 0 && ro->layout == 0)) {
            if (eu) collect_env_quad<0, CSTR_INTEGRATOR_EULER>(ro->k, ro->c, ring_row, env, j, live, act_out, env_q);
            else collect_env_quad<0, CSTR_INTEGRATOR_RK4>(ro->k, ro->c, ring_row, env, j, live, act_out, env_q);
        } else if (ENV == 3 || (ENV < 0 && ro->layout == 1)) {
            if (eu) collect_env_quad<1, CSTR_INTEGRATOR_EULER>(ro->k, ro->c, ring_row, env, j, live, act_out, env_q);
            else collect_env_quad<1, CSTR_INTEGRATOR_RK4>(ro->k, ro->c, ring_row, env, j, live, act_out, env_q);
        } else {
            float u4[4];
            u4[0] = act_out; u4[1] = __shfl_down(act_out, 1); u4[2] = __shfl_down(act_out, 2); u4[3] = __shfl_down(act_out, 3);
            if (env_lane) {
                if (eu) collect_env_lane<2, CSTR_INTEGRATOR_EULER>(ro->k, ro->c, ring_row, row, u4, env_in);
                else collect_env_lane<2, CSTR_INTEGRATOR_RK4>(ro->k, ro->c, ring_row, row, u4, env_in);
            }
        }
    }
    V2_STAMP(6);
    if (!FUSE && a.rng_ctl && !(a.flags & 1) && last_block_ticket_tree(reinterpret_cast<unsigned long long *>(a.rng_ctl + 2),
                                                                       reinterpret_cast<unsigned long long *>(a.rng_ctl + 4)) && tid == 0)
        a.rng_ctl[1] += (uint64_t)a.m;
}

template <int ACT, int HEAD, bool VEC0, bool K0_SMALL, bool SMALL>
__global__ __launch_bounds__(64 * POLICY_WAVES) void policy_rows_v2_kernel(const PolicyArgs a)
{
    policy_rows_v2_body<ACT, HEAD, VEC0, K0_SMALL, SMALL ? 1 : 0, false>(a, nullptr);
}

// The rollout of one vec-step in ONE launch: policy network + sampling (policy_rows_v2_body) + fused collect step + the replay
// index draw (RolloutArgs). Layer-1 input width <= 16 (the CSTR observations), 16-byte aligned rows.
template <int ACT, int HEAD, int SHAPE, int ENV>
__global__ __launch_bounds__(64 * POLICY_WAVES) void rollout_step_kernel(const float *x, const int64_t ldx, const float *w1, const float *b1,
                                                                         const float *w2s, const int m, const int k0, const int h1, const int h2,
                                                                         const PolicyArgs a0, const RolloutArgs ro)
{
    // ReplayBuffer.sample's two index draws for the gather launch behind this one (buffers.py:112-113, :309) run on ONE wave of an extra
    // workgroup (the grid's last one, present when mt_state is given): it shares a CU with a policy workgroup, starts at t = 0 and is done
    // (~4 us of serial MT19937 work) long before the launch ends. On an idle wave of the last policy workgroup, beside its sampling tail,
    // the draw ended 0.6 us AFTER everybody else (tools/rollout_ab.py: 9.9 us without the draw, 10.5 with it).
    if (ro.mt_state != nullptr && blockIdx.x == gridDim.x - 1) {
        __shared__ uint32_t mt_lds[MT_N];
        const int lane = threadIdx.x & 63;
        if (threadIdx.x >= 64) return;
        constexpr int MT_Q = (MT_N + 63) / 64;
#pragma unroll
        for (int i = 0; i < MT_Q; ++i)
            if (lane + 64 * i < MT_N) mt_lds[lane + 64 * i] = ro.mt_state[lane + 64 * i];
        // the ring as ReplayBuffer.add leaves it after THIS launch's row: upper = rows if full else pos (buffers.py:280-283, :112)
        const int64_t rpos = ro.ring_ctl[0], rows = ro.c.ring.rows;
        const int64_t upper = (ro.ring_ctl[1] || rpos + 1 == rows) ? rows : rpos + 1;
        int pos = (int)ro.mt_state[MT_N];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        pos = mt_randint_fill_wave(mt_lds, pos, (uint32_t)(upper - 1), ro.batch, ro.sample_idx, lane);
        pos = mt_randint_fill_wave(mt_lds, pos, (uint32_t)(ro.c.ring.n_envs - 1), ro.batch, ro.sample_idx + ro.batch, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int i = 0; i < MT_Q; ++i)
            if (lane + 64 * i < MT_N) ro.mt_state[lane + 64 * i] = mt_lds[lane + 64 * i];
        if (lane == 0) ro.mt_state[MT_N] = (uint32_t)pos;
        return;
    }
    // The leading scalars (14 dwords: gfx950 preloads them into SGPRs with the wave; struct parameters are never preloaded) are what
    // layer 1's operand requests and the L2 warm-up touch need: those leave before the first scalar load of the argument structs
    // has returned. The same fields of a0 are not read.
    PolicyArgs a = a0;
    a.x = x; a.ldx = ldx; a.w1 = w1; a.b1 = b1; a.w2s = w2s; a.m = m; a.k0 = k0; a.h1 = h1; a.h2 = h2;
    policy_rows_v2_body<ACT, HEAD, true, true, SHAPE, true, ENV>(a, &ro);
}

// ---- loss heads (single workgroup; batch <= 16384) -----------------------------------------------------

// SAC entropy coefficient (core/sac/sac.py:230-243): ent_coef = exp(log_alpha); loss = -mean(log_alpha * (logp + H));
// d loss / d log_alpha = -mean(logp + H). Writes the gradient straight into the alpha arena's gradient word.
__global__ __launch_bounds__(256) void sac_alpha_kernel(const float *__restrict__ log_alpha, const float *__restrict__ logp,
                                                        const float target_entropy, float *__restrict__ grad_out,
                                                        float *__restrict__ ent_coef_out, float *__restrict__ loss_out,
                                                        float *__restrict__ loss_sum, float *__restrict__ ent_coef_sum, const int batch)
{
    __shared__ float sm[4];
    float acc = 0.0f;
    for (int b = threadIdx.x; b < batch; b += 256) acc += logp[b] + target_entropy;
    const float mean = block_sum_256(acc, sm) / (float)batch;
    if (threadIdx.x == 0) {
        const float la = log_alpha[0], ec = expf(la);
        grad_out[0] = -mean;
        ent_coef_out[0] = ec;
        if (loss_out) loss_out[0] = -(la * mean);
        if (loss_sum) loss_sum[0] += -(la * mean);
        if (ent_coef_sum) ent_coef_sum[0] += ec;
    }
}

// Twin-critic loss (SAC: scale 0.5, core/sac/sac.py:261; TD3/MADDPG: scale 1, core/td3/td3.py:182):
// loss = scale * (mse(q1,t) + mse(q2,t));  d loss / d q_k = scale * 2 (q_k - t) / B
__global__ __launch_bounds__(256) void twin_q_loss_kernel(const float *__restrict__ q1, const float *__restrict__ q2,
                                                          const float *__restrict__ target, const float scale,
                                                          float *__restrict__ gq1, float *__restrict__ gq2,
                                                          float *__restrict__ loss_out, float *__restrict__ loss_sum, const int batch)
{
    __shared__ float sm[4];
    const float k = scale * 2.0f / (float)batch;
    float a1 = 0.0f, a2 = 0.0f;
    for (int b = threadIdx.x; b < batch; b += 256) {
        const float t = target[b], d1 = q1[b] - t, d2 = q2[b] - t;
        gq1[b] = k * d1;
        gq2[b] = k * d2;
        a1 += d1 * d1;
        a2 += d2 * d2;
    }
    const float s1 = block_sum_256(a1, sm), s2 = block_sum_256(a2, sm);
    if (threadIdx.x == 0) {
        const float loss = scale * (s1 / (float)batch + s2 / (float)batch);
        if (loss_out) loss_out[0] = loss;
        if (loss_sum) loss_sum[0] += loss;
    }
}

// TD target + twin-critic loss (+ SAC's entropy-coefficient loss) in ONE single-workgroup launch: the three reductions over
// the batch that sit between the forward passes and the critic backward. Per element exactly td_target_min_kernel
// (cstr_learner.hip), twin_q_loss_kernel and sac_alpha_kernel above, so the results are those kernels' bit for bit.
//   t = rew + (1 - done) * gamma * (min(q1_t, q2_t) - alpha * next_logp);  loss = scale * (mse(q1, t) + mse(q2, t))
// alpha = exp(log_alpha) when the entropy-coefficient part rides along (its value BEFORE this step's update, sac.py:230),
// the given constant otherwise.
__global__ __launch_bounds__(256) void td_twin_q_loss_kernel(const float *__restrict__ q1_t, const float *__restrict__ q2_t,
                                                             const float *__restrict__ next_logp, const float *__restrict__ rew,
                                                             const float *__restrict__ done, const float *__restrict__ ent_coef,
                                                             const float gamma, const float *__restrict__ q1,
                                                             const float *__restrict__ q2, const float scale,
                                                             float *__restrict__ target_out, float *__restrict__ gq1,
                                                             float *__restrict__ gq2, float *__restrict__ loss_out,
                                                             float *__restrict__ loss_sum, const cstr_alpha_part_t ap, const int batch)
{
    __shared__ float sm[4];
    const bool with_alpha = ap.log_alpha != nullptr;
    const float la = with_alpha ? ap.log_alpha[0] : 0.0f;
    const float ec = with_alpha ? expf(la) : (ent_coef ? ent_coef[0] : 0.0f);
    const float k = scale * 2.0f / (float)batch;
    float a1 = 0.0f, a2 = 0.0f, aa = 0.0f;
    for (int b = threadIdx.x; b < batch; b += 256) {
        float q = fminf(q1_t[b], q2_t[b]);
        if (next_logp) q = q - ec * next_logp[b];
        const float t = rew[b] + (1.0f - done[b]) * gamma * q;
        if (target_out) target_out[b] = t;
        const float d1 = q1[b] - t, d2 = q2[b] - t;
        gq1[b] = k * d1;
        gq2[b] = k * d2;
        a1 += d1 * d1;
        a2 += d2 * d2;
        if (with_alpha) aa += ap.logp_pi[b] + ap.target_entropy;
    }
    const float s1 = block_sum_256(a1, sm), s2 = block_sum_256(a2, sm);
    const float mean = with_alpha ? block_sum_256(aa, sm) / (float)batch : 0.0f;
    if (threadIdx.x == 0) {
        const float loss = scale * (s1 / (float)batch + s2 / (float)batch);
        if (loss_out) loss_out[0] = loss;
        if (loss_sum) loss_sum[0] += loss;
        if (with_alpha) {
            ap.grad_out[0] = -mean;
            ap.ent_coef_out[0] = ec;
            if (ap.loss_out) ap.loss_out[0] = -(la * mean);
            if (ap.loss_sum) ap.loss_sum[0] += -(la * mean);
            if (ap.ent_coef_sum) ap.ent_coef_sum[0] += ec;
        }
    }
}

// SAC actor loss (core/sac/sac.py:273-275): loss = mean(ent_coef * logp - min(q1, q2));
// d/d logp = ent_coef / B;  d/d q_k = -1/B for the smaller one (first index on ties, like th.min), 0 for the other
__global__ __launch_bounds__(256) void sac_actor_loss_kernel(const float *__restrict__ logp, const float *__restrict__ q1,
                                                             const float *__restrict__ q2, const float *__restrict__ ent_coef,
                                                             float *__restrict__ g_logp, float *__restrict__ gq1,
                                                             float *__restrict__ gq2, float *__restrict__ loss_out,
                                                             float *__restrict__ loss_sum, const int batch)
{
    __shared__ float sm[4];
    const float ec = ent_coef[0], inv = 1.0f / (float)batch;
    float acc = 0.0f;
    for (int b = threadIdx.x; b < batch; b += 256) {
        const float a = q1[b], c = q2[b];
        const bool first = a <= c;
        acc += ec * logp[b] - (first ? a : c);
        g_logp[b] = ec * inv;
        gq1[b] = first ? -inv : 0.0f;
        gq2[b] = first ? 0.0f : -inv;
    }
    const float s = block_sum_256(acc, sm);
    if (threadIdx.x == 0) {
        const float loss = s * inv;
        if (loss_out) loss_out[0] = loss;
        if (loss_sum) loss_sum[0] += loss;
    }
}

constexpr int HEAD_ROOT_MAX_ROWS = 1024;  // batch rows of the loss-root form (its per-row gradients sit in LDS)

// block_sum_256 inside a larger workgroup: every thread calls it, waves 0-3 contribute (the same tree, the same bits)
__device__ __forceinline__ float block_sum_first4(float v, float *sm)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0 && wave < 4) sm[wave] = v;
    __syncthreads();
    const float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}

// hidden_head_bwd_kernel with the loss root inside (cstr_hidden_head_bwd_root_f32): d(loss)/dq is a per-row function of the
// batch's Q values, so every workgroup recomputes it for the rows it walks (the same expressions as td_twin_q_loss_kernel /
// sac_actor_loss_kernel: the same bits), and ONE extra workgroup (blockIdx.x == gridDim.x - 1, group 0) does what only the
// separate loss launch did: target_out / g_logp, the logged loss and the entropy-coefficient part, with that launch's 256-thread
// accumulation pattern and reduction tree. Two groups (the twin Q networks); MODE 3 (-mean(Q1), the deterministic actors' loss,
// core/td3/td3.py:194, core/maddpg/maddpg.py:177): one group.
template <int ACT, int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void hidden_head_bwd_root_kernel(const cstr_head_root_t rt, const float *__restrict__ y,
                                                                          const float *__restrict__ w2, float *__restrict__ dz,
                                                                          float *__restrict__ gb1, float *__restrict__ gw2,
                                                                          float *__restrict__ gb2, const int m, const int k)
{
    __shared__ float part[3][WAVES][64];
    __shared__ float sm[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t g = blockIdx.y;
    const bool with_alpha = MODE == 1 && rt.alpha.log_alpha != nullptr;
    const float la = with_alpha ? rt.alpha.log_alpha[0] : 0.0f;
    const float ec = with_alpha ? expf(la) : (rt.ent_coef ? rt.ent_coef[0] : 0.0f);
    const float kq = rt.scale * 2.0f / (float)rt.batch, inv = 1.0f / (float)rt.batch;
    if (blockIdx.x == gridDim.x - 1) {  // the loss workgroup
        if (g != 0) return;
        const int tid = threadIdx.x;
        if (MODE == 1) {
            float a1 = 0.0f, a2 = 0.0f, aa = 0.0f;
            if (tid < 256) {
                for (int b = tid; b < rt.batch; b += 256) {
                    float q = fminf(rt.q1_t[b], rt.q2_t[b]);
                    if (rt.next_logp) q = q - ec * rt.next_logp[b];
                    const float t = rt.rew[b] + (1.0f - rt.done[b]) * rt.gamma * q;
                    if (rt.target_out) rt.target_out[b] = t;
                    const float d1 = rt.q1[b] - t, d2 = rt.q2[b] - t;
                    a1 += d1 * d1;
                    a2 += d2 * d2;
                    if (with_alpha) aa += rt.alpha.logp_pi[b] + rt.alpha.target_entropy;
                }
            }
            const float s1 = block_sum_first4(a1, sm), s2 = block_sum_first4(a2, sm);
            const float mean = with_alpha ? block_sum_first4(aa, sm) / (float)rt.batch : 0.0f;
            if (tid == 0) {
                const float loss = rt.scale * (s1 / (float)rt.batch + s2 / (float)rt.batch);
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
        } else if (MODE == 3) {  // -mean(Q1): neg_mean_loss_kernel's accumulation and tree
            float acc = 0.0f;
            if (tid < 256) {
                for (int b = tid; b < rt.batch; b += 256) acc += rt.q1[b];
            }
            const float sum = block_sum_first4(acc, sm);
            if (tid == 0) {
                const float loss = -(sum * inv);
                if (rt.loss_out) rt.loss_out[0] = loss;
                if (rt.loss_sum) rt.loss_sum[0] += loss;
            }
        } else {
            float acc = 0.0f;
            if (tid < 256) {
                for (int b = tid; b < rt.batch; b += 256) {
                    const float a = rt.q1[b], c = rt.q2[b];
                    acc += ec * rt.logp[b] - (a <= c ? a : c);
                    rt.g_logp[b] = ec * inv;
                }
            }
            const float sum = block_sum_first4(acc, sm);
            if (tid == 0) {
                const float loss = sum * inv;
                if (rt.loss_out) rt.loss_out[0] = loss;
                if (rt.loss_sum) rt.loss_sum[0] += loss;
            }
        }
        return;
    }
    const int col = blockIdx.x * 64 + lane;
    const int64_t goff = g * (int64_t)m * k;
    y += goff;
    dz += goff;
    // this group's d(loss)/dq for every row, ONCE per workgroup (a thread per row, coalesced loads), then read from LDS
    __shared__ float gq_s[HEAD_ROOT_MAX_ROWS];
    {
        const float *qg = g == 0 ? rt.q1 : rt.q2;
        for (int r = threadIdx.x; r < m; r += WAVES * 64) {
            float gqv;
            if (MODE == 1) {
                float q = fminf(rt.q1_t[r], rt.q2_t[r]);
                if (rt.next_logp) q = q - ec * rt.next_logp[r];
                const float tq = rt.rew[r] + (1.0f - rt.done[r]) * rt.gamma * q;
                gqv = kq * (qg[r] - tq);
            } else if (MODE == 3) {
                gqv = -inv;
            } else {
                const bool first = rt.q1[r] <= rt.q2[r];
                gqv = (first == (g == 0)) ? -inv : 0.0f;
            }
            gq_s[r] = gqv;
        }
    }
    __syncthreads();
    const float w = col < k ? w2[g * k + col] : 0.0f;
    float s_b1 = 0.0f, s_w2 = 0.0f, s_b2 = 0.0f;
    constexpr int FLY = 8;
    for (int r0 = wave; r0 < m; r0 += FLY * WAVES) {
        float t[FLY], u[FLY];
#pragma unroll
        for (int j = 0; j < FLY; ++j) {
            const int r = r0 + j * WAVES;
            t[j] = (r < m && col < k) ? y[(int64_t)r * k + col] : 0.0f;
            u[j] = r < m ? gq_s[r] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < FLY; ++j) {
            const int r = r0 + j * WAVES;
            float d = u[j] * w;
            if (ACT == ACT_RELU) d = t[j] > 0.0f ? d : 0.0f;
            if (ACT == ACT_TANH) d = d * (1.0f - t[j] * t[j]);
            if (r < m && col < k) dz[(int64_t)r * k + col] = d;
            s_b1 += d;
            s_w2 += u[j] * t[j];
            s_b2 += u[j];
        }
    }
    part[0][wave][lane] = s_b1;
    part[1][wave][lane] = s_w2;
    part[2][wave][lane] = s_b2;
    __syncthreads();
    if (wave == 0 && gw2) {
        float a = 0.0f, b = 0.0f, c = 0.0f;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) { a += part[0][v][lane]; b += part[1][v][lane]; c += part[2][v][lane]; }
        if (col < k) {
            gb1[g * k + col] = a;
            gw2[g * k + col] = b;
        }
        if (blockIdx.x == 0 && lane == 0) gb2[g] = c;
    }
}

// Deterministic-policy actor loss (core/td3/td3.py:194, core/maddpg/maddpg.py:174): loss = -mean(q1); dq = -1/B
__global__ __launch_bounds__(256) void neg_mean_loss_kernel(const float *__restrict__ q, float *__restrict__ gq,
                                                            float *__restrict__ loss_out, float *__restrict__ loss_sum, const int batch)
{
    __shared__ float sm[4];
    const float inv = 1.0f / (float)batch;
    float acc = 0.0f;
    for (int b = threadIdx.x; b < batch; b += 256) {
        acc += q[b];
        gq[b] = -inv;
    }
    const float s = block_sum_256(acc, sm);
    if (threadIdx.x == 0) {
        const float loss = -(s * inv);
        if (loss_out) loss_out[0] = loss;
        if (loss_sum) loss_sum[0] += loss;
    }
}

}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------------

extern "C" int cstr_bias_act_fwd_f32(float *y, const float *bias, int act, int64_t groups, int64_t m, int64_t n, cstr_stream_t stream)
{
    if (!y || !bias || groups <= 0 || m <= 0 || n <= 0) return CSTR_E_BADARG;
    if (act < 0 || act > 2 || n > 0x7fffffff) return CSTR_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int64_t gsz = m * n, total = groups * gsz;
    int block, grid;
    if ((n & 3) == 0 && aligned16(y) && aligned16(bias)) {
        flat_launch_shape(total / 4, block, grid);
        float4 *y4 = reinterpret_cast<float4 *>(y);
        const float4 *b4 = reinterpret_cast<const float4 *>(bias);
        if (act == 0) bias_act_fwd_vec4_kernel<0><<<grid, block, 0, s>>>(y4, b4, total / 4, (int)(n / 4), gsz / 4);
        else if (act == 1) bias_act_fwd_vec4_kernel<1><<<grid, block, 0, s>>>(y4, b4, total / 4, (int)(n / 4), gsz / 4);
        else bias_act_fwd_vec4_kernel<2><<<grid, block, 0, s>>>(y4, b4, total / 4, (int)(n / 4), gsz / 4);
    } else {
        flat_launch_shape(total, block, grid);
        if (act == 0) bias_act_fwd_kernel<0><<<grid, block, 0, s>>>(y, bias, total, (int)n, gsz);
        else if (act == 1) bias_act_fwd_kernel<1><<<grid, block, 0, s>>>(y, bias, total, (int)n, gsz);
        else bias_act_fwd_kernel<2><<<grid, block, 0, s>>>(y, bias, total, (int)n, gsz);
    }
    return (int)hipGetLastError();
}

extern "C" int cstr_bias_act_bwd_f32(const float *gy, const float *y, int act, float *gz, float *gbias, int64_t groups, int64_t m,
                                     int64_t n, cstr_stream_t stream)
{
    if (groups <= 0) return CSTR_E_BADARG;
    if (groups == 1) return cstr_bias_act_bwd_rows_f32(gy, n, y, n, act, gz, gbias, m, n, stream);
    if (!gy || !gz || m <= 0 || n <= 0 || (act != 0 && !y)) return CSTR_E_BADARG;
    if (act < 0 || act > 2 || m > 0x7fffffff || n > 0x7fffffff || groups > 65535) return CSTR_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)groups);
#define BAB(A, W) bias_act_bwd_kernel<A, W><<<grid, 64 * W, 0, s>>>(gy, y, gz, gbias, (int)m, (int)n, n, n)
    if (m >= 64) { if (act == 0) BAB(0, 16); else if (act == 1) BAB(1, 16); else BAB(2, 16); }  // 16 rows per wave at batch 256
    else { if (act == 0) BAB(0, 4); else if (act == 1) BAB(1, 4); else BAB(2, 4); }
#undef BAB
    return (int)hipGetLastError();
}

extern "C" int cstr_bias_act_bwd_rows_f32(const float *gy, int64_t ldg, const float *y, int64_t ldy, int act, float *gz, float *gbias,
                                          int64_t m, int64_t n, cstr_stream_t stream)
{
    if (!gy || !gz || m <= 0 || n <= 0 || ldg < n || (act != 0 && (!y || ldy < n))) return CSTR_E_BADARG;
    if (act < 0 || act > 2 || m > 0x7fffffff || n > 0x7fffffff) return CSTR_E_UNSUPPORTED;
    if (gz == gy && ldg != n) return CSTR_E_BADARG;  // gz is contiguous [m][n]
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + 63) / 64), 1u);
#define BAB(A, W) bias_act_bwd_kernel<A, W><<<grid, 64 * W, 0, s>>>(gy, y, gz, gbias, (int)m, (int)n, ldg, ldy)
    if (m >= 64) { if (act == 0) BAB(0, 16); else if (act == 1) BAB(1, 16); else BAB(2, 16); }
    else { if (act == 0) BAB(0, 4); else if (act == 1) BAB(1, 4); else BAB(2, 4); }
#undef BAB
    return (int)hipGetLastError();
}

extern "C" int cstr_squashed_gaussian_fwd_f32(const float *mean, const float *log_std_raw, const float *eps, float *action,
                                              float *logp, int64_t batch, int act_dim, int in_stride, cstr_stream_t stream)
{
    if (!mean || !log_std_raw || !eps || !action || batch <= 0 || act_dim <= 0 || in_stride < act_dim) return CSTR_E_BADARG;
    int block, grid;
    flat_launch_shape(batch, block, grid);
    squashed_gaussian_fwd_kernel<<<grid, block, 0, (hipStream_t)stream>>>(mean, log_std_raw, eps, action, logp, batch, act_dim, in_stride);
    return (int)hipGetLastError();
}

extern "C" int cstr_squashed_gaussian_bwd_f32(const float *g_action, const float *g_logp, const float *action,
                                              const float *log_std_raw, const float *eps, float *g_mean, float *g_log_std_raw,
                                              int64_t batch, int act_dim, int in_stride, int g_action_stride, cstr_stream_t stream)
{
    if (!action || !log_std_raw || !eps || !g_mean || !g_log_std_raw || batch <= 0 || act_dim <= 0 || in_stride < act_dim ||
        (g_action && g_action_stride < act_dim))
        return CSTR_E_BADARG;
    int block, grid;
    flat_launch_shape(batch * act_dim, block, grid);
    squashed_gaussian_bwd_kernel<<<grid, block, 0, (hipStream_t)stream>>>(g_action, g_logp, action, log_std_raw, eps, g_mean,
                                                                          g_log_std_raw, batch, act_dim, in_stride, g_action_stride);
    return (int)hipGetLastError();
}

extern "C" int cstr_sac_alpha_f32(const float *log_alpha, const float *logp, float target_entropy, float *grad_out,
                                  float *ent_coef_out, float *loss_out, float *loss_sum, float *ent_coef_sum, int64_t batch,
                                  cstr_stream_t stream)
{
    if (!log_alpha || !logp || !grad_out || !ent_coef_out || batch <= 0) return CSTR_E_BADARG;
    if (batch > CSTR_MAX_SAMPLE_BATCH) return CSTR_E_UNSUPPORTED;
    sac_alpha_kernel<<<1, 256, 0, (hipStream_t)stream>>>(log_alpha, logp, target_entropy, grad_out, ent_coef_out, loss_out, loss_sum,
                                                         ent_coef_sum, (int)batch);
    return (int)hipGetLastError();
}

extern "C" int cstr_twin_q_loss_f32(const float *q1, const float *q2, const float *target, float scale, float *gq1, float *gq2,
                                    float *loss_out, float *loss_sum, int64_t batch, cstr_stream_t stream)
{
    if (!q1 || !q2 || !target || !gq1 || !gq2 || batch <= 0) return CSTR_E_BADARG;
    if (batch > CSTR_MAX_SAMPLE_BATCH) return CSTR_E_UNSUPPORTED;
    twin_q_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(q1, q2, target, scale, gq1, gq2, loss_out, loss_sum, (int)batch);
    return (int)hipGetLastError();
}

extern "C" int cstr_td_twin_q_loss_f32(const float *q1_t, const float *q2_t, const float *next_logp, const float *rew, const float *done,
                                       const float *ent_coef, float gamma, const float *q1, const float *q2, float scale,
                                       float *target_out, float *gq1, float *gq2, float *loss_out, float *loss_sum,
                                       const cstr_alpha_part_t *alpha, int64_t batch, cstr_stream_t stream)
{
    if (!q1_t || !q2_t || !rew || !done || !q1 || !q2 || !gq1 || !gq2 || batch <= 0) return CSTR_E_BADARG;
    cstr_alpha_part_t ap = {};
    if (alpha && alpha->log_alpha) {
        if (!alpha->logp_pi || !alpha->grad_out || !alpha->ent_coef_out) return CSTR_E_BADARG;
        ap = *alpha;
    }
    if (next_logp && !ap.log_alpha && !ent_coef) return CSTR_E_BADARG;  // an entropy term needs its coefficient
    if (batch > CSTR_MAX_SAMPLE_BATCH) return CSTR_E_UNSUPPORTED;
    td_twin_q_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(q1_t, q2_t, next_logp, rew, done, ent_coef, gamma, q1, q2, scale,
                                                              target_out, gq1, gq2, loss_out, loss_sum, ap, (int)batch);
    return (int)hipGetLastError();
}

extern "C" int cstr_sac_actor_loss_f32(const float *logp, const float *q1, const float *q2, const float *ent_coef, float *g_logp,
                                       float *gq1, float *gq2, float *loss_out, float *loss_sum, int64_t batch, cstr_stream_t stream)
{
    if (!logp || !q1 || !q2 || !ent_coef || !g_logp || !gq1 || !gq2 || batch <= 0) return CSTR_E_BADARG;
    if (batch > CSTR_MAX_SAMPLE_BATCH) return CSTR_E_UNSUPPORTED;
    sac_actor_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logp, q1, q2, ent_coef, g_logp, gq1, gq2, loss_out, loss_sum, (int)batch);
    return (int)hipGetLastError();
}

extern "C" int cstr_neg_mean_loss_f32(const float *q, float *gq, float *loss_out, float *loss_sum, int64_t batch, cstr_stream_t stream)
{
    if (!q || !gq || batch <= 0) return CSTR_E_BADARG;
    if (batch > CSTR_MAX_SAMPLE_BATCH) return CSTR_E_UNSUPPORTED;
    neg_mean_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(q, gq, loss_out, loss_sum, (int)batch);
    return (int)hipGetLastError();
}

extern "C" int cstr_hidden_head_fwd_f32(float *z, const float *b1, int act, const float *w2, const float *b2, float *q, int64_t groups,
                                        int64_t m, int64_t k, cstr_stream_t stream)
{
    if (!z || !b1 || !w2 || !b2 || !q || groups <= 0 || m <= 0 || k <= 0) return CSTR_E_BADARG;
    if (act < 0 || act > 2 || m > 0x7fffffff || k > 0x7fffffff) return CSTR_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = groups * m;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const bool v4 = (k & 3) == 0 && aligned16(z) && aligned16(b1) && aligned16(w2);
#define HH_FWD(A, V) hidden_head_fwd_kernel<A, V><<<grid, 256, 0, s>>>(z, b1, w2, b2, q, rows, (int)m, (int)k)
    if (v4) { if (act == 0) HH_FWD(0, true); else if (act == 1) HH_FWD(1, true); else HH_FWD(2, true); }
    else { if (act == 0) HH_FWD(0, false); else if (act == 1) HH_FWD(1, false); else HH_FWD(2, false); }
#undef HH_FWD
    return (int)hipGetLastError();
}

extern "C" int cstr_hidden_head_bwd_f32(const float *gq, const float *y, int act, const float *w2, float *dz, float *gb1, float *gw2,
                                        float *gb2, int64_t groups, int64_t m, int64_t k, cstr_stream_t stream)
{
    if (!gq || !y || !w2 || !dz || groups <= 0 || m <= 0 || k <= 0) return CSTR_E_BADARG;
    if ((gb1 == nullptr) != (gw2 == nullptr) || (gw2 == nullptr) != (gb2 == nullptr)) return CSTR_E_BADARG;  // all three or none
    if (act < 0 || act > 2 || m > 0x7fffffff || k > 0x7fffffff || groups > 65535) return CSTR_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((k + 63) / 64), (unsigned)groups);
#define HH_BWD(A, W) hidden_head_bwd_kernel<A, W><<<grid, W * 64, 0, s>>>(gq, y, w2, dz, gb1, gw2, gb2, (int)m, (int)k)
    if (m >= 64) { if (act == 0) HH_BWD(0, 16); else if (act == 1) HH_BWD(1, 16); else HH_BWD(2, 16); }
    else { if (act == 0) HH_BWD(0, 4); else if (act == 1) HH_BWD(1, 4); else HH_BWD(2, 4); }
#undef HH_BWD
    return (int)hipGetLastError();
}

extern "C" int cstr_hidden_head_bwd_root_f32(const cstr_head_root_t *root, const float *y, int act, const float *w2, float *dz,
                                             float *gb1, float *gw2, float *gb2, int64_t m, int64_t k, cstr_stream_t stream)
{
    if (!root || !y || !w2 || !dz || m <= 0 || k <= 0) return CSTR_E_BADARG;
    const cstr_head_root_t &r = *root;
    if ((gb1 == nullptr) != (gw2 == nullptr) || (gw2 == nullptr) != (gb2 == nullptr)) return CSTR_E_BADARG;  // all three or none
    if ((r.mode != 1 && r.mode != 2 && r.mode != 3) || r.batch != m || !r.q1 || (r.mode != 3 && !r.q2)) return CSTR_E_BADARG;
    if (r.mode == 1 && (!r.q1_t || !r.q2_t || !r.rew || !r.done || (r.next_logp && !r.alpha.log_alpha && !r.ent_coef))) return CSTR_E_BADARG;
    if (r.mode == 1 && r.alpha.log_alpha && (!r.alpha.logp_pi || !r.alpha.grad_out || !r.alpha.ent_coef_out)) return CSTR_E_BADARG;
    if (r.mode == 2 && (!r.logp || !r.g_logp || !r.ent_coef)) return CSTR_E_BADARG;
    if (act < 0 || act > 2 || m > HEAD_ROOT_MAX_ROWS || k > 0x7fffffff) return CSTR_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((k + 63) / 64) + 1u, r.mode == 3 ? 1u : 2u);  // + the loss workgroup; mode 3: ONE Q network
#define HH_ROOT(A, W) do { if (r.mode == 1) hidden_head_bwd_root_kernel<A, W, 1><<<grid, W * 64, 0, s>>>(r, y, w2, dz, gb1, gw2, gb2, (int)m, (int)k); \
                           else if (r.mode == 2) hidden_head_bwd_root_kernel<A, W, 2><<<grid, W * 64, 0, s>>>(r, y, w2, dz, gb1, gw2, gb2, (int)m, (int)k); \
                           else hidden_head_bwd_root_kernel<A, W, 3><<<grid, W * 64, 0, s>>>(r, y, w2, dz, gb1, gw2, gb2, (int)m, (int)k); } while (0)
    if (m >= 64) { if (act == 0) HH_ROOT(0, 16); else if (act == 1) HH_ROOT(1, 16); else HH_ROOT(2, 16); }
    else { if (act == 0) HH_ROOT(0, 4); else if (act == 1) HH_ROOT(1, 4); else HH_ROOT(2, 4); }
#undef HH_ROOT
    return (int)hipGetLastError();
}

extern "C" int cstr_gaussian_head_fwd_f32(float *params, const float *bias, float *eps, uint64_t *rng_ctl, float *action,
                                          int64_t action_stride, float *logp, int64_t batch, int act_dim, cstr_stream_t stream)
{
    if (!params || !eps || !action || batch <= 0 || act_dim <= 0 || action_stride < act_dim) return CSTR_E_BADARG;
    if (act_dim > CSTR_MAX_HEAD_ACT) return CSTR_E_UNSUPPORTED;
    const int64_t g = (batch + 63) / 64;
    gaussian_head_fwd_kernel<<<(unsigned)(g < 4096 ? g : 4096), 64, 0, (hipStream_t)stream>>>(params, bias, eps, rng_ctl, action,
                                                                                             action_stride, logp, batch, act_dim);
    return (int)hipGetLastError();
}

extern "C" int cstr_gaussian_head_bwd_f32(const float *g_action, int64_t ga_stride, const float *g_logp, const float *action,
                                          int64_t action_stride, const float *params, const float *eps, float *g_params,
                                          float *g_bias, int64_t batch, int act_dim, cstr_stream_t stream)
{
    if (!action || !params || !eps || !g_params || batch <= 0 || act_dim <= 0 || action_stride < act_dim) return CSTR_E_BADARG;
    if (g_action && ga_stride < act_dim) return CSTR_E_BADARG;
    if (act_dim > CSTR_MAX_HEAD_ACT || batch > 65536) return CSTR_E_UNSUPPORTED;
    gaussian_head_bwd_kernel<<<1, 256, 0, (hipStream_t)stream>>>(g_action, ga_stride, g_logp, action, action_stride, params, eps,
                                                                 g_params, g_bias, batch, act_dim);
    return (int)hipGetLastError();
}

extern "C" int cstr_gaussian_head_bwd_input_f32(const float *g_action, int64_t ga_stride, const float *g_logp, const float *action,
                                               int64_t action_stride, const float *params, const float *eps, const float *w,
                                               const float *hidden, int64_t ldh, int act, float *g_params, float *dz, int64_t batch,
                                               int act_dim, int64_t width, cstr_stream_t stream)
{
    if (!action || !params || !eps || !w || !hidden || !g_params || !dz || batch <= 0 || act_dim <= 0 || width <= 0) return CSTR_E_BADARG;
    if (action_stride < act_dim || ldh < width || (g_action && ga_stride < act_dim)) return CSTR_E_BADARG;
    if (act < 0 || act > 2 || act_dim > CSTR_MAX_HEAD_ACT || width > 0x7fffff || batch > (int64_t)0x7fffffff * HEAD_BWD_ROWS) return CSTR_E_UNSUPPORTED;
    const unsigned grid = (unsigned)((batch + HEAD_BWD_ROWS - 1) / HEAD_BWD_ROWS);
    hipStream_t s = (hipStream_t)stream;
#define HBI(A) gaussian_head_bwd_input_kernel<A><<<grid, 64 * HEAD_BWD_ROWS, 0, s>>>(g_action, ga_stride, g_logp, action, action_stride, \
        params, eps, w, hidden, ldh, g_params, dz, batch, act_dim, (int)width)
    if (act == 0) HBI(0); else if (act == 1) HBI(1); else HBI(2);
#undef HBI
    return (int)hipGetLastError();
}

#ifdef CSTR_POLICY_STAMPS
extern "C" int cstr_diag_policy_stamps(unsigned long long *host_out, int64_t words)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(policy_stamps), (size_t)words * 8);
}
#endif

extern "C" int cstr_policy_swizzle_f32(const float *w, int64_t n, int64_t k, float *out, cstr_stream_t stream)
{
    if (!w || !out || n <= 0 || k <= 0) return CSTR_E_BADARG;
    if ((k & 3) || !aligned16(w) || !aligned16(out) || n > 0x7ffffff || k > 0x7ffffff) return CSTR_E_UNSUPPORTED;
    const int64_t entries = ((n + 15) / 16) * ((k + 15) / 16) * 64;
    int block, grid;
    flat_launch_shape(entries, block, grid);
    policy_swizzle_kernel<<<grid, block, 0, (hipStream_t)stream>>>(w, (int)n, (int)k, reinterpret_cast<float4 *>(out), entries);
    return (int)hipGetLastError();
}

// cstr_policy_rows_fwd_f32's operand checks, shared with cstr_rollout_step_f32
static int check_policy_net(const cstr_policy_mlp_t *net, const float *x, int64_t ldx, const float *eps, uint64_t *rng_ctl, const float *logp,
                            int64_t action_stride, int64_t m)
{
    if (!net || !x || m <= 0) return CSTR_E_BADARG;
    const cstr_policy_mlp_t &n = *net;
    if (!n.w1 || !n.b1 || !n.w2 || !n.b2 || !n.w3 || !n.b3 || n.k0 <= 0 || n.h1 <= 0 || n.h2 <= 0 || n.act_dim <= 0) return CSTR_E_BADARG;
    if (ldx < n.k0 || action_stride < n.act_dim) return CSTR_E_BADARG;
    if (n.head == 0 && (eps == nullptr) == (rng_ctl == nullptr)) return CSTR_E_BADARG;  // exactly one noise source
    if (n.head != 0 && (eps || rng_ctl || logp)) return CSTR_E_BADARG;
    const size_t lds = (size_t)POLICY_ROWS * (n.h1 + 4 + n.h2 + 4) * sizeof(float);
    if (n.act < 0 || n.act > 2 || n.out_act < 0 || n.out_act > 2 || n.head < 0 || n.head > 1 || n.k0 > 256 || (n.h1 & 3) || (n.h2 & 3) ||
        lds > 64 * 1024 || !aligned16(n.w2) || !aligned16(n.w3) || (n.head == 0 ? 2 : 1) * n.act_dim > 2 * CSTR_MAX_HEAD_ACT ||
        (n.head == 0 && n.act_dim > CSTR_MAX_HEAD_ACT) || (m + POLICY_ROWS - 1) / POLICY_ROWS > 0x7fffffff)
        return CSTR_E_UNSUPPORTED;
    if (n.reserved & ~1) return CSTR_E_BADARG;  // bit 0: the caller advances the Philox offset (no ticket in this launch)
    if (n.w2_swizzled && !aligned16(n.w2_swizzled)) return CSTR_E_BADARG;
    return CSTR_OK;
}

static size_t policy_v2_lds(const cstr_policy_mlp_t &n)
{
    const int kc1 = (n.h1 + 15) / 16, kc2 = (n.h2 + 15) / 16;
    return (size_t)(POLICY_ROWS * (16 * kc1 + 4 + 16 * kc2 + 4) + POLICY_WAVES * POLICY_ROWS * 8 + 4 * POLICY_ROWS * 8) * sizeof(float);
}

extern "C" int cstr_policy_rows_fwd_f32(const cstr_policy_mlp_t *net, const float *x, int64_t ldx, const float *eps, uint64_t *rng_ctl,
                                       float *action, int64_t action_stride, float *logp, int64_t m, cstr_stream_t stream)
{
    if (!action) return CSTR_E_BADARG;
    const int rc = check_policy_net(net, x, ldx, eps, rng_ctl, logp, action_stride, m);
    if (rc) return rc;
    const cstr_policy_mlp_t &n = *net;
    const size_t lds = (size_t)POLICY_ROWS * (n.h1 + 4 + n.h2 + 4) * sizeof(float);
    PolicyArgs a = {x, ldx, n.k0, n.w1, n.b1, n.h1, n.w2, n.b2, n.h2, n.w3, n.b3, n.act_dim, n.out_act, eps, rng_ctl,
                    action, action_stride, logp, m, n.w2_swizzled, n.reserved};
    const unsigned grid = (unsigned)((m + POLICY_ROWS - 1) / POLICY_ROWS);
    const bool vec0 = (n.k0 & 3) == 0 && (ldx & 3) == 0 && aligned16(x) && aligned16(n.w1);
    hipStream_t s = (hipStream_t)stream;
    static const bool force_v1 = getenv("CSTR_POLICY_V1") != nullptr;  // development A/B knob (tools/policy_ab.py)
    if (n.w2_swizzled && n.h1 <= V2_MAX_WIDTH && n.h2 <= V2_MAX_WIDTH && policy_v2_lds(n) <= 64 * 1024 && !force_v1) {
        // the software-pipelined kernel (tile-major W2 required)
        const int kc1 = (n.h1 + 15) / 16, kc2 = (n.h2 + 15) / 16;
        const size_t lds2 = policy_v2_lds(n);
        const bool k0s = n.k0 <= 16, small = kc1 <= 16 && kc2 <= 16;
#define POLV4(A, H, V, K) do { if (small) policy_rows_v2_kernel<A, H, V, K, true><<<grid, 64 * POLICY_WAVES, lds2, s>>>(a); \
                               else policy_rows_v2_kernel<A, H, V, K, false><<<grid, 64 * POLICY_WAVES, lds2, s>>>(a); } while (0)
#define POLV2(A, H) do { if (vec0) { if (k0s) POLV4(A, H, true, true); else POLV4(A, H, true, false); } \
                         else { if (k0s) POLV4(A, H, false, true); else POLV4(A, H, false, false); } } while (0)
        if (n.head == 0) { if (n.act == 0) POLV2(0, 0); else if (n.act == 1) POLV2(1, 0); else POLV2(2, 0); }
        else { if (n.act == 0) POLV2(0, 1); else if (n.act == 1) POLV2(1, 1); else POLV2(2, 1); }
#undef POLV2
#undef POLV4
        return (int)hipGetLastError();
    }
#define POL2(A, H) do { if (vec0) policy_rows_fwd_kernel<A, H, true><<<grid, 64 * POLICY_WAVES, lds, s>>>(a); \
                        else policy_rows_fwd_kernel<A, H, false><<<grid, 64 * POLICY_WAVES, lds, s>>>(a); } while (0)
#define POL(A, H) POL2(A, H)
    if (n.head == 0) { if (n.act == 0) POL(0, 0); else if (n.act == 1) POL(1, 0); else POL(2, 0); }
    else { if (n.act == 0) POL(0, 1); else if (n.act == 1) POL(1, 1); else POL(2, 1); }
#undef POL2
#undef POL
    return (int)hipGetLastError();
}

extern "C" int cstr_rollout_step_f32(const cstr_policy_mlp_t *net, const float *x, int64_t ldx, uint64_t *rng_ctl, const cstr_coef_t *coef,
                                     int integrator, const cstr_ring_t *ring, const int64_t *ring_ctl, float *env_obs, int32_t *step_count,
                                     int squashed, const float *act_low, const float *act_high, const float *noise, const float *reset_obs,
                                     uint64_t *pcg_state, double *static_init, float *reward_out, float *done_out, float *ep_return,
                                     double *ep_stats, float *action_out, uint32_t *mt_state, int64_t batch, int32_t *sample_idx,
                                     cstr_stream_t stream)
{
    RolloutArgs ro;
    int rc = make_collect_args(coef, integrator, ring, env_obs, step_count, squashed, act_low, act_high, noise, reset_obs, pcg_state,
                               static_init, reward_out, done_out, ep_return, ep_stats, ro.c);
    if (rc) return rc;
    if (!ring_ctl) return CSTR_E_BADARG;
    const int64_t m = ring->n_envs;
    rc = check_policy_net(net, x, ldx, nullptr, rng_ctl, nullptr, net ? net->act_dim : 0, m);
    if (rc) return rc;
    const cstr_policy_mlp_t &n = *net;
    if (n.act_dim != ring->act_dim) return CSTR_E_BADARG;
    if ((mt_state == nullptr) != (sample_idx == nullptr) || (mt_state && (batch <= 0 || batch > CSTR_MAX_SAMPLE_BATCH))) return CSTR_E_BADARG;
    if (mt_state && (ring->rows >= 0xFFFFFFFFLL || ring->n_envs >= 0xFFFFFFFFLL)) return CSTR_E_UNSUPPORTED;
    // the software-pipelined kernel with a one-chunk first layer: tile-major W2, widths <= 512, k0 <= 16 in 16-byte rows
    const bool vec0 = (n.k0 & 3) == 0 && (ldx & 3) == 0 && aligned16(x) && aligned16(n.w1);
    if (!n.w2_swizzled || n.h1 > V2_MAX_WIDTH || n.h2 > V2_MAX_WIDTH || n.k0 > 16 || !vec0 || policy_v2_lds(n) > 64 * 1024) return CSTR_E_UNSUPPORTED;
    ro.k = *coef;
    ro.ring_ctl = ring_ctl;
    ro.layout = layout_of(ring->obs_dim, ring->act_dim);
    ro.integrator = integrator;
    ro.mt_state = mt_state;
    ro.sample_idx = sample_idx;
    ro.batch = (int)batch;
    // reserved bit 0 is implied: nobody advances a control word in this launch (the gather launch behind it does)
    PolicyArgs a = {x, ldx, n.k0, n.w1, n.b1, n.h1, n.w2, n.b2, n.h2, n.w3, n.b3, n.act_dim, n.out_act, nullptr, rng_ctl,
                    action_out, n.act_dim, nullptr, m, n.w2_swizzled, n.reserved | 1};
    const unsigned grid = (unsigned)((m + POLICY_ROWS - 1) / POLICY_ROWS) + (mt_state ? 1u : 0u);  // + the index-draw workgroup
    const size_t lds2 = policy_v2_lds(n);
    const bool small = (n.h1 + 15) / 16 <= 16 && (n.h2 + 15) / 16 <= 16;
    hipStream_t s = (hipStream_t)stream;
#define ROL4(A, H, S, E) rollout_step_kernel<A, H, S, E><<<grid, 64 * POLICY_WAVES, lds2, s>>>(a.x, a.ldx, a.w1, a.b1, a.w2s, (int)a.m, a.k0, a.h1, a.h2, a, ro)
    // the environment variant as a template constant where a class default runs it (ReLU networks at the exact shapes): the one-train env
    // with Euler steps (BASELINE configs 2-4) and with the raw state in the observation + RK4 (the north_star-literal variant)
    const int env_variant = 2 * ro.layout + (integrator == CSTR_INTEGRATOR_RK4 ? 1 : 0);
#define ROL3(A, H, S) do { if ((A) == 1 && (S) >= 2 && env_variant == 0) ROL4(1, H, (S) >= 2 ? (S) : 2, 0); \
                           else if ((A) == 1 && (S) == 2 && env_variant == 3) ROL4(1, H, 2, 3); else ROL4(A, H, S, -1); } while (0)
    // CSTR_EXACT_SHAPES=0: run the run-time-width instantiations everywhere (tests compare the two bit for bit)
    static const bool exact = !(getenv("CSTR_EXACT_SHAPES") && atoi(getenv("CSTR_EXACT_SHAPES")) == 0);
#define ROL2(A, H) do { if (exact && n.h1 == 256 && n.h2 == 256) ROL3(A, H, 2); else if (exact && n.h1 == 400 && n.h2 == 300) ROL3(A, H, 3); \
                        else if (small) ROL3(A, H, 1); else ROL3(A, H, 0); } while (0)
    if (n.head == 0) { if (n.act == 0) ROL2(0, 0); else if (n.act == 1) ROL2(1, 0); else ROL2(2, 0); }
    else { if (n.act == 0) ROL2(0, 1); else if (n.act == 1) ROL2(1, 1); else ROL2(2, 1); }
#undef ROL2
#undef ROL3
#undef ROL4
    return (int)hipGetLastError();
}

extern "C" int cstr_target_smooth_f32(const float *action, const float *noise, uint64_t *rng_ctl, float sigma, float clip, float *out,
                                      int64_t out_stride, int64_t batch, int act_dim, cstr_stream_t stream)
{
    if (!action || !out || batch <= 0 || act_dim <= 0 || out_stride < act_dim) return CSTR_E_BADARG;
    if ((noise == nullptr) == (rng_ctl == nullptr)) return CSTR_E_BADARG;  // exactly one noise source
    if (!(sigma >= 0.0f) || !(clip >= 0.0f)) return CSTR_E_BADARG;
    const int64_t g = (batch + 63) / 64;
    target_smooth_kernel<<<(unsigned)(g < 4096 ? g : 4096), 64, 0, (hipStream_t)stream>>>(action, noise, rng_ctl, sigma, clip, out,
                                                                                         out_stride, batch, act_dim);
    return (int)hipGetLastError();
}

extern "C" int cstr_linear_act_fwd_f32(const float *x, int64_t x_group_stride, int64_t ldx, const float *w, const float *bias, int act,
                                       float *y, int64_t groups, int64_t m, int64_t n, int64_t k, cstr_stream_t stream)
{
    if (!x || !w || !bias || !y || groups <= 0 || m <= 0 || n <= 0 || k <= 0 || ldx < k || x_group_stride < 0) return CSTR_E_BADARG;
    if (act < 0 || act > 2 || m > 0x7fffff || n > 0x7fffff || k > 0x7fffff || groups > 65535 || (m + 15) / 16 > 65535) return CSTR_E_UNSUPPORTED;
    const dim3 grid((unsigned)((n + 15) / 16), (unsigned)((m + 15) / 16), (unsigned)groups);
    const bool vec = (k & 3) == 0 && (ldx & 3) == 0 && (x_group_stride & 3) == 0 && aligned16(x) && aligned16(w);
    const bool buf = m * ldx < (1 << 28) && n * k < (1 << 28);  // operands through buffer descriptors with 32-bit byte offsets: below 1 GiB
    hipStream_t s = (hipStream_t)stream;
    // one wave per tile when K fits one chunk batch of a single wave's first round trip or the grid is already large;
    // four-way split-K otherwise (K = 256 on 256 tiles: each wave's loads are one round trip)
    const bool split = k > 32 && (int64_t)grid.x * grid.y * grid.z <= 2048;
#define LIN(A, V, W) do { if (buf) linear_act_fwd_kernel<A, V, W, true><<<grid, 64 * W, 0, s>>>(x, w, x_group_stride, (int)ldx, (int)m, (int)n, (int)k, bias, y); \
                          else linear_act_fwd_kernel<A, V, W, false><<<grid, 64 * W, 0, s>>>(x, w, x_group_stride, (int)ldx, (int)m, (int)n, (int)k, bias, y); } while (0)
#define LIN_ACT(V, W) do { if (act == 0) LIN(0, V, W); else if (act == 1) LIN(1, V, W); else LIN(2, V, W); } while (0)
    if (vec) { if (split) LIN_ACT(true, 4); else LIN_ACT(true, 1); }
    else { if (split) LIN_ACT(false, 4); else LIN_ACT(false, 1); }
#undef LIN_ACT
#undef LIN
    return (int)hipGetLastError();
}

extern "C" int cstr_linear_act_fwd_gather_f32(const cstr_ring_t *ring, int64_t *ring_ctl, int advance_ring, uint64_t *rng_ctl,
                                              uint64_t rng_advance, const int32_t *sample_idx, int64_t batch, int both, const float *w,
                                              const float *bias, int act, float *y, int64_t n, float *x_data, float *x_next, float *x_pi,
                                              float *out_done, float *out_rew, cstr_stream_t stream)
{
    if (!ring || !ring->obs || !ring->next_obs || !ring->act || !ring->rew || !ring->done || !ring->timeout) return CSTR_E_BADARG;
    if (!sample_idx || !w || !bias || !y || !x_data || !x_next || !out_done || !out_rew || batch <= 0 || n <= 0) return CSTR_E_BADARG;
    if (advance_ring && !ring_ctl) return CSTR_E_BADARG;
    if (act < 0 || act > 2) return CSTR_E_BADARG;
    const int lay = layout_of(ring->obs_dim, ring->act_dim);
    if (lay < 0 || batch > CSTR_MAX_SAMPLE_BATCH || n > 0x7ffffff || ring->rows >= 0xFFFFFFFFLL || ring->n_envs >= 0xFFFFFFFFLL) return CSTR_E_UNSUPPORTED;
    if (!aligned16(ring->obs) || !aligned16(ring->next_obs) || !aligned8(ring->act) || !aligned16(w) || !aligned8(x_data) || !aligned8(x_next) ||
        (x_pi && !aligned8(x_pi)))
        return CSTR_E_BADARG;
    const GatherArgs g = {*ring, ring_ctl, advance_ring, rng_ctl, rng_advance, sample_idx, (int)batch, both ? 1 : 0, x_data, x_pi, x_next, out_done, out_rew};
    const int64_t m = both ? 2 * batch : batch;
    const dim3 grid((unsigned)((n + 15) / 16), (unsigned)((m + 15) / 16));
    hipStream_t s = (hipStream_t)stream;
#define GL2(A_, D_, AD_) gather_linear_act_fwd_kernel<A_, D_, AD_><<<grid, 64, 0, s>>>(g, w, bias, y, (int)n)
#define GL(A_) do { if (lay == 0) GL2(A_, 4, 2); else if (lay == 1) GL2(A_, 8, 2); else GL2(A_, 8, 4); } while (0)
    if (act == 0) GL(0); else if (act == 1) GL(1); else GL(2);
#undef GL
#undef GL2
    return (int)hipGetLastError();
}

extern "C" int cstr_linear_smooth_fwd_f32(const float *x, int64_t ldx, const float *w, const float *bias, int act, const float *noise,
                                          uint64_t *rng_ctl, float sigma, float clip, float *out, int64_t out_stride, int64_t m, int64_t n,
                                          int64_t k, cstr_stream_t stream)
{
    if (!x || !w || !bias || !out || m <= 0 || n <= 0 || k <= 0 || ldx < k || out_stride < n) return CSTR_E_BADARG;
    if ((noise == nullptr) == (rng_ctl == nullptr)) return CSTR_E_BADARG;  // exactly one noise source
    if (!(sigma >= 0.0f) || !(clip >= 0.0f) || act < 0 || act > 2) return CSTR_E_BADARG;
    // the shapes for which cstr_linear_act_fwd_f32 takes its four-way split-K form (bit-identical results)
    if (n > 16 || k <= 32 || k > 0x7fffff || (m + 15) / 16 > 2048 || m * ldx >= (1 << 28)) return CSTR_E_UNSUPPORTED;
    const unsigned grid = (unsigned)((m + 15) / 16);
    const bool vec = (k & 3) == 0 && (ldx & 3) == 0 && aligned16(x) && aligned16(w);
    hipStream_t s = (hipStream_t)stream;
#define LS(A, V) linear_smooth_fwd_kernel<A, V><<<grid, 256, 0, s>>>(x, (int)ldx, w, bias, (int)m, (int)n, (int)k, noise, rng_ctl, sigma, clip, out, out_stride)
#define LS_ACT(V) do { if (act == 0) LS(0, V); else if (act == 1) LS(1, V); else LS(2, V); } while (0)
    if (vec) LS_ACT(true); else LS_ACT(false);
#undef LS_ACT
#undef LS
    return (int)hipGetLastError();
}

extern "C" int cstr_linear_bwd_input_f32(const float *gz, const float *w, const float *y, int act, float *dz, int64_t groups,
                                         int sum_groups, int64_t m, int64_t n, int64_t k, cstr_stream_t stream)
{
    if (!gz || !w || !dz || groups <= 0 || m <= 0 || n <= 0 || k <= 0 || (act != 0 && !y)) return CSTR_E_BADARG;
    if (act < 0 || act > 2 || m > 0x7fffff || n > 0x7fffff || k > 0x7fffff || groups > 65535 || (m + 15) / 16 > 65535) return CSTR_E_UNSUPPORTED;
    const dim3 grid((unsigned)((k + 15) / 16), (unsigned)((m + 15) / 16), (unsigned)(sum_groups ? 1 : groups));
    const int n_sum = sum_groups ? (int)groups : 0;
    const bool vec = (n & 3) == 0 && aligned16(gz);
    const bool buf = m * n < (1 << 28) && n * k < (1 << 28) && m * k < (1 << 28);  // buffer descriptors with 32-bit byte offsets: below 1 GiB
    hipStream_t s = (hipStream_t)stream;
#define LBI(A, V, W) do { if (buf) linear_bwd_input_kernel<A, V, W, true><<<grid, 64 * W, 0, s>>>(gz, w, y, dz, (int)m, (int)n, (int)k, n_sum); \
                          else linear_bwd_input_kernel<A, V, W, false><<<grid, 64 * W, 0, s>>>(gz, w, y, dz, (int)m, (int)n, (int)k, n_sum); } while (0)
#define LBI_ACT(V, W) do { if (act == 0) LBI(0, V, W); else if (act == 1) LBI(1, V, W); else LBI(2, V, W); } while (0)
    if (n > 32) { if (vec) LBI_ACT(true, 4); else LBI_ACT(false, 4); }
    else { if (vec) LBI_ACT(true, 1); else LBI_ACT(false, 1); }
#undef LBI_ACT
#undef LBI
    return (int)hipGetLastError();
}

extern "C" int cstr_linear_bwd_weight_f32(const float *dz, const float *x, int64_t x_group_stride, int64_t ldx, float *dw, float *db,
                                          int64_t groups, int64_t m, int64_t n, int64_t k, cstr_stream_t stream)
{
    if (!dz || !x || !dw || groups <= 0 || m <= 0 || n <= 0 || k <= 0 || ldx < k || x_group_stride < 0) return CSTR_E_BADARG;
    if (m > 0x7fffff || n > 0x7fffff || k > 0x7fffff || groups > 65535 || (n + 15) / 16 > 65535) return CSTR_E_UNSUPPORTED;
    const dim3 grid((unsigned)((k + 15) / 16), (unsigned)((n + 15) / 16), (unsigned)groups);
    hipStream_t s = (hipStream_t)stream;
    const bool buf = m * n < (1 << 28) && m * ldx < (1 << 28);  // buffer descriptors with 32-bit byte offsets: below 1 GiB
#define LBW(W, B) linear_bwd_weight_kernel<W, B><<<grid, 64 * W, 0, s>>>(dz, x, x_group_stride, (int)ldx, dw, db, (int)m, (int)n, (int)k)
    if (m > 32) { if (buf) LBW(4, true); else LBW(4, false); }
    else { if (buf) LBW(1, true); else LBW(1, false); }
#undef LBW
    return (int)hipGetLastError();
}

extern "C" int cstr_linear_bwd_weight_sets_f32(const cstr_wgrad_set_t *sets, int n_sets, cstr_stream_t stream)
{
    if (!sets || n_sets <= 0) return CSTR_E_BADARG;
    if (n_sets > CSTR_MAX_LINEAR_SETS) return CSTR_E_UNSUPPORTED;
    WgradSets t;
    int64_t kt = 1, nt = 1, m_min = INT64_MAX;
    for (int i = 0; i < n_sets; ++i) {
        const cstr_wgrad_set_t &q = sets[i];
        if (!q.dz || !q.x || !q.dw || q.m <= 0 || q.n <= 0 || q.k <= 0 || q.ldx < q.k) return CSTR_E_BADARG;
        if (q.m > 0x7fffff || q.n > 0x7fffff || q.k > 0x7fffff || (q.n + 15) / 16 > 65535) return CSTR_E_UNSUPPORTED;
        kt = kt > (q.k + 15) / 16 ? kt : (q.k + 15) / 16;
        nt = nt > (q.n + 15) / 16 ? nt : (q.n + 15) / 16;
        m_min = m_min < q.m ? m_min : q.m;
        t.s[i] = q;
    }
    const dim3 grid((unsigned)kt, (unsigned)nt, (unsigned)n_sets);
    hipStream_t s = (hipStream_t)stream;
    bool buf = true;  // operands through buffer descriptors with 32-bit byte offsets: every matrix below 1 GiB
    for (int i = 0; i < n_sets; ++i) buf = buf && sets[i].m * sets[i].n < (1 << 28) && sets[i].m * sets[i].ldx < (1 << 28);
    if (m_min > 32) { if (buf) linear_bwd_weight_sets_kernel<4, true><<<grid, 256, 0, s>>>(t); else linear_bwd_weight_sets_kernel<4, false><<<grid, 256, 0, s>>>(t); }
    else { if (buf) linear_bwd_weight_sets_kernel<1, true><<<grid, 64, 0, s>>>(t); else linear_bwd_weight_sets_kernel<1, false><<<grid, 64, 0, s>>>(t); }
    return (int)hipGetLastError();
}

extern "C" int cstr_linear_bwd_weight_adam_sets_f32(const cstr_wgrad_adam_set_t *sets, int n_sets, const cstr_adam_opt_t *opts, int n_opts,
                                                    const cstr_adam_seg_t *flat, int n_flat, cstr_stream_t stream)
{
    if (!sets || n_sets <= 0 || !opts || n_opts <= 0 || n_flat < 0 || (n_flat > 0 && !flat)) return CSTR_E_BADARG;
    if (n_sets > CSTR_MAX_LINEAR_SETS || n_opts > CSTR_MAX_ADAM_SEGS || n_flat > CSTR_MAX_ADAM_SEGS) return CSTR_E_UNSUPPORTED;
    WgradAdamSets t;
    t.n_sets = n_sets;
    t.n_flat = n_flat;
    int64_t total = 0;
    for (int i = 0; i < n_opts; ++i) {
        if (!opts[i].adam_ctl || !opts[i].lr) return CSTR_E_BADARG;
        t.o[i] = opts[i];
    }
    for (int i = 0; i < n_sets; ++i) {
        const cstr_wgrad_adam_set_t &q = sets[i];
        if (!q.g.dz || !q.g.x || !q.g.dw || q.g.m <= 32 || q.g.n <= 0 || q.g.k <= 0 || q.g.ldx < q.g.k) return CSTR_E_BADARG;
        if (!q.w || !q.w_m || !q.w_v || q.opt < 0 || q.opt >= n_opts || (q.g.db && (!q.b || !q.b_m || !q.b_v))) return CSTR_E_BADARG;
        if ((q.w_target != nullptr) != (q.b_target != nullptr && q.g.db != nullptr) && q.g.db) return CSTR_E_BADARG;
        if (q.g.m * q.g.n >= (1 << 28) || q.g.m * q.g.ldx >= (1 << 28) || (q.g.n + 15) / 16 > 65535) return CSTR_E_UNSUPPORTED;
        t.first[i] = (int)total;
        total += ((q.g.k + 15) / 16) * ((q.g.n + 15) / 16);
        t.s[i] = q;
    }
    for (int i = 0; i < n_flat; ++i) {
        const cstr_adam_seg_t &f = flat[i];
        if (!f.param || f.n <= 0) return CSTR_E_BADARG;
        if (!f.polyak_source && (!f.grad || !f.exp_avg || !f.exp_avg_sq || !f.adam_ctl || !f.lr)) return CSTR_E_BADARG;
        if (f.shadow) return CSTR_E_UNSUPPORTED;  // a shadowed matrix is updated by its weight-gradient tiles
        if (!aligned16(f.param) || (f.polyak_source && !aligned16(f.polyak_source))) return CSTR_E_BADARG;
        t.first[n_sets + i] = (int)total;
        int64_t wgs = ((f.n + 3) / 4 + 255) / 256;  // one 16-byte quad per lane
        total += wgs < 1 ? 1 : (wgs > 256 ? 256 : wgs);
        t.f[i] = f;
    }
    t.first[n_sets + n_flat] = (int)total;
    if (total > 0x7fffffff) return CSTR_E_UNSUPPORTED;
    linear_bwd_weight_adam_sets_kernel<4, true><<<dim3((unsigned)total), 256, 0, (hipStream_t)stream>>>(t);
    return (int)hipGetLastError();
}

extern "C" int cstr_gaussian_head_gemm_fwd_f32(const float *hidden, int64_t ldh, const float *w, const float *bias, float *params,
                                               float *eps, uint64_t *rng_ctl, float *action, int64_t action_stride, float *logp,
                                               int64_t batch, int act_dim, int64_t k, cstr_stream_t stream)
{
    if (!hidden || !w || !bias || !params || !eps || !action || batch <= 0 || act_dim <= 0 || k <= 0 || ldh < k || action_stride < act_dim)
        return CSTR_E_BADARG;
    if (act_dim > CSTR_MAX_HEAD_ACT || (k & 3) || (ldh & 3) || !aligned16(hidden) || !aligned16(w) || k > 0x7fffff) return CSTR_E_UNSUPPORTED;
    // rows (waves) per workgroup: 4 at batch size (spread over many CUs), 16 for large batches (fewer tickets on the RNG
    // control word: they serialise on one address)
    const int rows_per_wg = batch <= 1024 ? 4 : 16;
    const int64_t g = (batch + rows_per_wg - 1) / rows_per_wg;
    if (g > 0x7fffffff) return CSTR_E_UNSUPPORTED;
    gaussian_head_gemm_fwd_kernel<<<(unsigned)g, 64 * rows_per_wg, 0, (hipStream_t)stream>>>(hidden, w, (int)ldh, (int)k, batch, act_dim, bias, rng_ctl,
                                                                                params, eps, action, action_stride, logp);
    return (int)hipGetLastError();
}

extern "C" int cstr_linear_act_fwd_sets_f32(const cstr_linear_set_t *sets, int n_sets, int act, int64_t m, int64_t n, int64_t k,
                                            cstr_stream_t stream)
{
    if (!sets || n_sets <= 0 || m <= 0 || n <= 0 || k <= 0) return CSTR_E_BADARG;
    if (n_sets > CSTR_MAX_LINEAR_SETS || act < 0 || act > 2 || m > 0x7fffff || n > 0x7fffff || k > 0x7fffff || (m + 15) / 16 > 65535)
        return CSTR_E_UNSUPPORTED;
    LinearSets t;
    bool vec = (k & 3) == 0, buf = true;  // buf: operands through buffer descriptors with 32-bit byte offsets (below 1 GiB)
    for (int i = 0; i < n_sets; ++i) {
        const cstr_linear_set_t &q = sets[i];
        if (!q.x || !q.w || !q.bias || !q.y || q.ldx < k || q.ldy < n) return CSTR_E_BADARG;
        vec = vec && (q.ldx & 3) == 0 && aligned16(q.x) && aligned16(q.w);
        buf = buf && m * q.ldx < (1 << 28) && n * k < (1 << 28);
        t.s[i] = q;
    }
    const dim3 grid((unsigned)((n + 15) / 16), (unsigned)((m + 15) / 16), (unsigned)n_sets);
    // split-K when the tile grid is small; decided as if there were at most four sets, so that a launch carrying several
    // independent four-set chains (fused.twin_pair_forward_many) sums in the same order as one launch per chain
    const bool split = k > 32 && (int64_t)grid.x * grid.y * (grid.z < 4 ? grid.z : 4) <= 2048;
    hipStream_t s = (hipStream_t)stream;
#define LIN(A, V, W) do { if (buf) linear_act_fwd_sets_kernel<A, V, W, true><<<grid, 64 * W, 0, s>>>(t, (int)m, (int)n, (int)k); \
                          else linear_act_fwd_sets_kernel<A, V, W, false><<<grid, 64 * W, 0, s>>>(t, (int)m, (int)n, (int)k); } while (0)
#define LIN_ACT(V, W) do { if (act == 0) LIN(0, V, W); else if (act == 1) LIN(1, V, W); else LIN(2, V, W); } while (0)
    if (vec) { if (split) LIN_ACT(true, 4); else LIN_ACT(true, 1); }
    else { if (split) LIN_ACT(false, 4); else LIN_ACT(false, 1); }
#undef LIN_ACT
#undef LIN
    return (int)hipGetLastError();
}
