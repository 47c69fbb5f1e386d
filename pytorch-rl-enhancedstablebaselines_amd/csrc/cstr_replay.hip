// cstr_replay.hip -- HBM replay ring sampler for gfx950: NumPy-legacy MT19937 index draw (bit-exact)
// + masked-rejection compaction + SoA gather in one launch (core/common/buffers.py:106-115, :285-325).
//
// One workgroup of 256 lanes (4 waves). The 624-word MT19937 state is staged in LDS; a twist is three
// 227/227/170-lane phases (each output word only needs words that are either old or produced by an
// earlier phase); tempering, masking, the accept test and a ballot/popcount prefix sum run 256 words
// per round, so the data-dependent consumption of `randint` (rejection) is reproduced word for word and
// the second draw starts exactly where the first stopped. The gather then reads 16-byte (D=4) rows of
// the field arrays by (row, env) index and writes the five output tensors coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cstr_rl_hip.h"
#include "cstr_device.h"

#include "cstr_mt_device.h"

namespace {

constexpr int TPB = 256;

// mt19937_gen (numpy/random/src/mt19937/mt19937.c), parallel over one workgroup. All lanes call it.
__device__ void mt_twist(uint32_t *mt)
{
    const int t = threadIdx.x;
    uint32_t v = 0;
    // phase A: kk in [0, 227): needs old mt[kk], old mt[kk+1], old mt[kk+397]
    if (t < MT_N - MT_M) v = mt_mix(mt[t], mt[t + 1], mt[t + MT_M]);
    __syncthreads();
    if (t < MT_N - MT_M) mt[t] = v;
    __syncthreads();
    // phase B: kk in [227, 454): needs old mt[kk], old mt[kk+1], NEW mt[kk-227] (phase A)
    const int kb = t + (MT_N - MT_M);
    if (t < MT_N - MT_M) v = mt_mix(mt[kb], mt[kb + 1], mt[kb - (MT_N - MT_M)]);
    __syncthreads();
    if (t < MT_N - MT_M) mt[kb] = v;
    __syncthreads();
    // phase C: kk in [454, 624): NEW mt[kk-227] (phase B); kk = 623 pairs with NEW mt[0]
    const int kc = t + 2 * (MT_N - MT_M);
    if (kc < MT_N) v = mt_mix(mt[kc], (kc == MT_N - 1) ? mt[0] : mt[kc + 1], mt[kc - (MT_N - MT_M)]);
    __syncthreads();
    if (kc < MT_N) mt[kc] = v;
    __syncthreads();
}

struct SampleShared {
    uint32_t mt[MT_N];
    int wave_tot[TPB / 64];
    int consumed;
};

// random_bounded_uint64_fill(off=0, rng=high-1, use_masked) for rng < 2^32 - 1 (numpy distributions.c):
// `count` accepted values into out[], consuming words from (mt, pos) in order. Returns the new pos.
__device__ int mt_randint_fill(SampleShared &sh, int pos, uint32_t rng, int count, int32_t *out)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (rng == 0u) {  // high == 1: zeros, consumes nothing
        for (int i = t; i < count; i += TPB) out[i] = 0;
        __syncthreads();
        return pos;
    }
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    int filled = 0;
    while (filled < count) {  // every quantity in the loop condition is workgroup-uniform
        if (pos == MT_N) { mt_twist(sh.mt); pos = 0; }
        const int chunk = min(MT_N - pos, TPB);
        uint32_t w = 0;
        bool acc = false;
        if (t < chunk) {
            w = mt_temper(sh.mt[pos + t]) & mask;
            acc = (w <= rng);
        }
        const unsigned long long bal = __ballot(acc);
        const int in_wave = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) sh.wave_tot[wave] = __popcll(bal);
        if (t == 0) sh.consumed = chunk;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int wv = 0; wv < TPB / 64; ++wv) {
            const int c = sh.wave_tot[wv];
            if (wv < wave) before += c;
            total += c;
        }
        const int rank = before + in_wave, need = count - filled;
        if (acc && rank < need) {
            out[filled + rank] = (int32_t)w;
            if (rank == need - 1) sh.consumed = t + 1;  // the draw stops right after the need-th accepted word
        }
        __syncthreads();
        pos += sh.consumed;
        filled += min(total, need);
        __syncthreads();  // sh.consumed / wave_tot are rewritten next round
    }
    return pos;
}

// The gather of ReplayBuffer._get_samples (buffers.py:316-323) for `batch` (row, env) index pairs, a lane per sampled row.
// PACKED: the batch is gathered straight into the critics' input rows (ContinuousCritic.forward's th.cat([obs, actions], 1),
// core/common/policies.py:975-981, without the cat launches): out_obs = x_data [B][D+A] <- (obs | action),
// out_next_obs = x_next [B][D+A] <- (next_obs | .), out_act = x_pi [B][D+A] or NULL <- (obs | .); the '.' columns belong
// to the actor head kernel. Rows are (D+A)*4 bytes apart (8-byte aligned), so the stores are float2.
template <int D, int A, bool PACKED>
__device__ __forceinline__ void gather_rows(const cstr_ring_t &ring, const int32_t *row_idx, const int32_t *env_idx, const int batch,
                                            float *__restrict__ out_obs, float *__restrict__ out_act, float *__restrict__ out_next_obs,
                                            float *__restrict__ out_done, float *__restrict__ out_rew, int64_t *__restrict__ out_row_idx,
                                            int64_t *__restrict__ out_env_idx)
{
    const int t = threadIdx.x;
    const int64_t n = ring.n_envs;
    for (int b = t; b < batch; b += TPB) {  // buffers.py:316-323
        const int64_t r = row_idx[b], e = env_idx[b], o = r * n + e;
        const float4 x0 = *reinterpret_cast<const float4 *>(ring.obs + o * D);
        const float4 y0 = *reinterpret_cast<const float4 *>(ring.next_obs + o * D);
        const float dn = ring.done[o], to = ring.timeout[o], rw = ring.rew[o];
        if (PACKED) {
            constexpr int W = D + A;
            float *xd = out_obs + (int64_t)b * W, *xn = out_next_obs + (int64_t)b * W, *xp = out_act ? out_act + (int64_t)b * W : nullptr;
            const float2 x01 = make_float2(x0.x, x0.y), x23 = make_float2(x0.z, x0.w);
            reinterpret_cast<float2 *>(xd)[0] = x01; reinterpret_cast<float2 *>(xd)[1] = x23;
            reinterpret_cast<float2 *>(xn)[0] = make_float2(y0.x, y0.y); reinterpret_cast<float2 *>(xn)[1] = make_float2(y0.z, y0.w);
            if (xp) { reinterpret_cast<float2 *>(xp)[0] = x01; reinterpret_cast<float2 *>(xp)[1] = x23; }
            if (D == 8) {
                const float4 x1 = *reinterpret_cast<const float4 *>(ring.obs + o * D + 4), y1 = *reinterpret_cast<const float4 *>(ring.next_obs + o * D + 4);
                reinterpret_cast<float2 *>(xd)[2] = make_float2(x1.x, x1.y); reinterpret_cast<float2 *>(xd)[3] = make_float2(x1.z, x1.w);
                reinterpret_cast<float2 *>(xn)[2] = make_float2(y1.x, y1.y); reinterpret_cast<float2 *>(xn)[3] = make_float2(y1.z, y1.w);
                if (xp) { reinterpret_cast<float2 *>(xp)[2] = make_float2(x1.x, x1.y); reinterpret_cast<float2 *>(xp)[3] = make_float2(x1.z, x1.w); }
            }
#pragma unroll
            for (int j = 0; j < A; j += 2)
                *reinterpret_cast<float2 *>(xd + D + j) = *reinterpret_cast<const float2 *>(ring.act + o * A + j);
            out_done[b] = dn * (1.0f - to);  // buffers.py:322
            out_rew[b] = rw;
            if (out_row_idx) out_row_idx[b] = r;
            if (out_env_idx) out_env_idx[b] = e;
            continue;
        }
        *reinterpret_cast<float4 *>(out_obs + (int64_t)b * D) = x0;
        *reinterpret_cast<float4 *>(out_next_obs + (int64_t)b * D) = y0;
        if (D == 8) {
            *reinterpret_cast<float4 *>(out_obs + (int64_t)b * D + 4) = *reinterpret_cast<const float4 *>(ring.obs + o * D + 4);
            *reinterpret_cast<float4 *>(out_next_obs + (int64_t)b * D + 4) = *reinterpret_cast<const float4 *>(ring.next_obs + o * D + 4);
        }
        if (A == 2) *reinterpret_cast<float2 *>(out_act + (int64_t)b * 2) = *reinterpret_cast<const float2 *>(ring.act + o * 2);
        else *reinterpret_cast<float4 *>(out_act + (int64_t)b * 4) = *reinterpret_cast<const float4 *>(ring.act + o * 4);
        out_done[b] = dn * (1.0f - to);  // buffers.py:322
        out_rew[b] = rw;
        if (out_row_idx) out_row_idx[b] = r;
        if (out_env_idx) out_env_idx[b] = e;
    }
}

template <int D, int A, bool PACKED = false>
__global__ __launch_bounds__(TPB) void replay_sample_kernel(const cstr_ring_t ring, const int64_t *__restrict__ ring_ctl,
                                                            uint32_t *__restrict__ mt_state, const int batch,
                                                            float *__restrict__ out_obs, float *__restrict__ out_act,
                                                            float *__restrict__ out_next_obs, float *__restrict__ out_done,
                                                            float *__restrict__ out_rew, int64_t *__restrict__ out_row_idx,
                                                            int64_t *__restrict__ out_env_idx)
{
    __shared__ SampleShared sh;
    extern __shared__ __align__(16) int32_t idx[];  // [2][batch]
    int32_t *row_idx = idx, *env_idx = idx + batch;
    const int t = threadIdx.x;
    for (int i = t; i < MT_N; i += TPB) sh.mt[i] = mt_state[i];
    int pos = (int)mt_state[MT_N];
    __syncthreads();

    const int64_t upper = ring_ctl[1] ? ring.rows : ring_ctl[0];  // buffers.py:112
    pos = mt_randint_fill(sh, pos, (uint32_t)(upper - 1), batch, row_idx);          // buffers.py:113
    pos = mt_randint_fill(sh, pos, (uint32_t)(ring.n_envs - 1), batch, env_idx);    // buffers.py:309
    __syncthreads();

    for (int i = t; i < MT_N; i += TPB) mt_state[i] = sh.mt[i];
    if (t == 0) mt_state[MT_N] = (uint32_t)pos;

    gather_rows<D, A, PACKED>(ring, row_idx, env_idx, batch, out_obs, out_act, out_next_obs, out_done, out_rew, out_row_idx, out_env_idx);
}

// The gather alone, for index pairs drawn earlier in the iteration by the rollout launch (cstr_rollout_step_f32: idx[0..batch) rows,
// idx[batch..2 batch) envs), plus the control-word updates that launch left to its successor: ReplayBuffer.add's epilogue
// (core/common/buffers.py:280-283) and the rollout policy's Philox offset. One workgroup: nobody else reads the control words.
template <int D, int A>
__global__ __launch_bounds__(TPB) void replay_gather_kernel(const cstr_ring_t ring, int64_t *__restrict__ ring_ctl, const int advance_ring,
                                                            uint64_t *__restrict__ rng_ctl, const uint64_t rng_advance,
                                                            const int32_t *__restrict__ idx, const int batch,
                                                            float *__restrict__ x_data, float *__restrict__ x_pi, float *__restrict__ x_next,
                                                            float *__restrict__ out_done, float *__restrict__ out_rew,
                                                            int64_t *__restrict__ out_row_idx, int64_t *__restrict__ out_env_idx)
{
    gather_rows<D, A, true>(ring, idx, idx + batch, batch, x_data, x_pi, x_next, out_done, out_rew, out_row_idx, out_env_idx);
    if (threadIdx.x == 0) {
        if (advance_ring) {
            int64_t pos = ring_ctl[0] + 1;
            if (pos == ring.rows) { ring_ctl[1] = 1; pos = 0; }
            ring_ctl[0] = pos;
            ring_ctl[3] += 1;
        }
        if (rng_ctl) rng_ctl[1] += rng_advance;
    }
}

// ---- np.random.normal on the same stream (exploration noise, core/common/noise.py:44-45, :141-142) -----------------
// numpy's legacy_gauss is the polar Box-Muller method: an attempt eats two 53-bit doubles (four 32-bit words), is
// rejected with probability 1 - pi/4, and an accepted attempt yields TWO deviates (f*x2 first, f*x1 cached for the next
// call). Attempts are independent, so one round evaluates every 4-word group of the words left in the current 624-word
// block in parallel (<= 156 lanes), and the same ballot/popcount prefix sum as the index draw assigns accepted attempts
// to output slots in stream order. Groups that straddle a twist are handled by carrying the <= 3 left-over words to the
// front of the window. The f64 arithmetic of the accept test is exact IEEE (the TU is built with -ffp-contract=off), so
// the stream position follows numpy word for word; log() is ocml's (<= 1 ulp in f64, invisible after the f32 cast).
struct NormalParams { double loc[CSTR_MAX_NOISE_PERIOD], scale[CSTR_MAX_NOISE_PERIOD]; int period; };

struct NormalShared {
    uint32_t mt[MT_N];
    uint32_t win[MT_N + 4];
    int wave_tot[TPB / 64];
    int consumed, has_gauss;
    double gauss;
};

template <typename T>  // float: NormalActionNoise's .astype(float32); double: the raw deviates (Ornstein-Uhlenbeck noise)
__global__ __launch_bounds__(TPB) void mt_normal_kernel(uint32_t *__restrict__ mt_state, const NormalParams prm,
                                                        T *__restrict__ out, const int64_t count)
{
    __shared__ NormalShared sh;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, P = prm.period;
    for (int i = t; i < MT_N; i += TPB) sh.mt[i] = mt_state[i];
    int pos = (int)mt_state[MT_N];
    const int has0 = (int)mt_state[MT_N + 1];
    if (t == 0) {
        sh.has_gauss = has0;
        sh.gauss = __hiloint2double((int)mt_state[MT_N + 3], (int)mt_state[MT_N + 2]);
    }
    __syncthreads();
    int64_t j = 0;
    if (has0 && count > 0) {  // the cached deviate of the previous call comes first
        if (t == 0) {
            out[0] = (T)(prm.loc[0] + prm.scale[0] * sh.gauss);
            sh.has_gauss = 0;
            sh.gauss = 0.0;
        }
        j = 1;
    }
    int64_t pairs = (count - j + 1) >> 1;
    int nwin = 0, wpos = 0;
    bool filled = false;
    while (pairs > 0) {  // every quantity in the loop condition is workgroup-uniform
        const int rem = nwin - wpos;
        if (rem < 4) {
            uint32_t c = 0;
            if (t < rem) c = sh.win[wpos + t];
            __syncthreads();
            if (t < rem) sh.win[t] = c;
            if (pos == MT_N) { mt_twist(sh.mt); pos = 0; }
            const int blk = MT_N - pos;
            for (int i = t; i < blk; i += TPB) sh.win[rem + i] = mt_temper(sh.mt[pos + i]);
            nwin = rem + blk; wpos = 0; pos = MT_N; filled = true;
            __syncthreads();
        }
        const int natt = (nwin - wpos) >> 2;
        bool acc = false;
        double x1 = 0.0, x2 = 0.0, r2 = 1.0;
        if (t < natt) {
            const uint32_t *w = sh.win + wpos + 4 * t;
            // mt19937_next_double: (a * 2^26 + b) / 2^53 with a = w0 >> 5, b = w1 >> 6
            const double d1 = ((double)(int)(w[0] >> 5) * 67108864.0 + (double)(int)(w[1] >> 6)) / 9007199254740992.0;
            const double d2 = ((double)(int)(w[2] >> 5) * 67108864.0 + (double)(int)(w[3] >> 6)) / 9007199254740992.0;
            x1 = 2.0 * d1 - 1.0;
            x2 = 2.0 * d2 - 1.0;
            r2 = x1 * x1 + x2 * x2;
            acc = !(r2 >= 1.0 || r2 == 0.0);
        }
        const unsigned long long bal = __ballot(acc);
        const int in_wave = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) sh.wave_tot[wave] = __popcll(bal);
        if (t == 0) sh.consumed = natt;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int wv = 0; wv < TPB / 64; ++wv) {
            const int c = sh.wave_tot[wv];
            if (wv < wave) before += c;
            total += c;
        }
        const int rank = before + in_wave;
        if (acc && rank < pairs) {
            const double f = sqrt(-2.0 * log(r2) / r2);
            const int64_t e0 = j + 2 * (int64_t)rank;
            const int p0 = (int)(e0 % P), p1 = (int)((e0 + 1) % P);
            out[e0] = (T)(prm.loc[p0] + prm.scale[p0] * (f * x2));
            if (e0 + 1 < count) out[e0 + 1] = (T)(prm.loc[p1] + prm.scale[p1] * (f * x1));
            else { sh.has_gauss = 1; sh.gauss = f * x1; }  // odd tail: keep the first deviate for the next call
            if (rank == pairs - 1) sh.consumed = t + 1;
        }
        __syncthreads();
        wpos += 4 * sh.consumed;
        const int took = (int)(total < pairs ? total : pairs);
        j += 2 * (int64_t)took;
        pairs -= took;
        __syncthreads();  // sh.consumed / wave_tot are rewritten next round
    }
    if (filled) pos = MT_N - (nwin - wpos);  // unconsumed window words are the tail of the current block
    for (int i = t; i < MT_N; i += TPB) mt_state[i] = sh.mt[i];
    if (t == 0) {
        mt_state[MT_N] = (uint32_t)pos;
        mt_state[MT_N + 1] = (uint32_t)sh.has_gauss;
        mt_state[MT_N + 2] = (uint32_t)__double2loint(sh.gauss);
        mt_state[MT_N + 3] = (uint32_t)__double2hiint(sh.gauss);
    }
}

// init_genrand (mt19937_seed): serial recurrence, 624 steps, one lane; runs once per (re)seed.
__global__ void mt_seed_kernel(uint32_t *__restrict__ mt_state, uint32_t seed)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t x = seed;
    mt_state[0] = x;
    for (int i = 1; i < MT_N; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        mt_state[i] = x;
    }
    mt_state[MT_N] = MT_N;
    mt_state[MT_N + 1] = 0;  // has_gauss, gauss (f64) of the legacy stream
    mt_state[MT_N + 2] = 0;
    mt_state[MT_N + 3] = 0;
}

}  // namespace

extern "C" int cstr_mt19937_seed(uint32_t *mt_state, uint32_t seed, cstr_stream_t stream)
{
    if (!mt_state) return CSTR_E_BADARG;
    mt_seed_kernel<<<1, 64, 0, (hipStream_t)stream>>>(mt_state, seed);
    return (int)hipGetLastError();
}

template <typename T>
static int mt_normal_launch(uint32_t *mt_state, const double *loc, const double *scale, int32_t period, T *out, int64_t count,
                            cstr_stream_t stream)
{
    if (!mt_state || !loc || !scale || !out || count < 0) return CSTR_E_BADARG;
    if (period < 1 || period > CSTR_MAX_NOISE_PERIOD) return CSTR_E_UNSUPPORTED;
    if (count == 0) return CSTR_OK;
    NormalParams prm;
    for (int i = 0; i < CSTR_MAX_NOISE_PERIOD; ++i) {
        prm.loc[i] = loc[i % period];
        prm.scale[i] = scale[i % period];
        if (i < period && !(scale[i] >= 0.0)) return CSTR_E_BADARG;  // numpy: "scale < 0"
    }
    prm.period = period;
    mt_normal_kernel<T><<<1, TPB, 0, (hipStream_t)stream>>>(mt_state, prm, out, count);
    return (int)hipGetLastError();
}

extern "C" int cstr_mt19937_normal_f32(uint32_t *mt_state, const double *loc, const double *scale, int32_t period, float *out,
                                       int64_t count, cstr_stream_t stream)
{
    return mt_normal_launch<float>(mt_state, loc, scale, period, out, count, stream);
}

extern "C" int cstr_mt19937_normal_f64(uint32_t *mt_state, const double *loc, const double *scale, int32_t period, double *out,
                                       int64_t count, cstr_stream_t stream)
{
    return mt_normal_launch<double>(mt_state, loc, scale, period, out, count, stream);
}

extern "C" int cstr_replay_sample_mt19937_f32(const cstr_ring_t *ring, const int64_t *ring_ctl, uint32_t *mt_state,
                                              int64_t batch, float *out_obs, float *out_act, float *out_next_obs,
                                              float *out_done, float *out_rew, int64_t *out_row_idx,
                                              int64_t *out_env_idx, cstr_stream_t stream)
{
    if (!ring || !ring->obs || !ring->next_obs || !ring->act || !ring->rew || !ring->done || !ring->timeout) return CSTR_E_BADARG;
    if (!ring_ctl || !mt_state || !out_obs || !out_act || !out_next_obs || !out_done || !out_rew || batch <= 0) return CSTR_E_BADARG;
    const bool lay_ok = (ring->obs_dim == 4 && ring->act_dim == 2) || (ring->obs_dim == 8 && (ring->act_dim == 2 || ring->act_dim == 4));
    if (!lay_ok) return CSTR_E_UNSUPPORTED;
    // 32-bit masked-rejection path only (numpy switches to 64-bit words above 2^32 - 1; rng == 2^32 - 1 is unmasked)
    if (batch > CSTR_MAX_SAMPLE_BATCH || ring->rows >= 0xFFFFFFFFLL || ring->n_envs >= 0xFFFFFFFFLL) return CSTR_E_UNSUPPORTED;
    const bool a4 = ring->act_dim == 4;
    if (!aligned16(ring->obs) || !aligned16(ring->next_obs) || !(a4 ? aligned16(ring->act) : aligned8(ring->act)) || !aligned16(out_obs) ||
        !aligned16(out_next_obs) || !(a4 ? aligned16(out_act) : aligned8(out_act)))
        return CSTR_E_BADARG;
    const size_t dyn = sizeof(int32_t) * 2 * (size_t)batch;
    hipStream_t s = (hipStream_t)stream;
#define SAMPLE_ARGS *ring, ring_ctl, mt_state, (int)batch, out_obs, out_act, out_next_obs, out_done, out_rew, out_row_idx, out_env_idx
    if (ring->obs_dim == 4) replay_sample_kernel<4, 2><<<1, TPB, dyn, s>>>(SAMPLE_ARGS);
    else if (!a4) replay_sample_kernel<8, 2><<<1, TPB, dyn, s>>>(SAMPLE_ARGS);
    else replay_sample_kernel<8, 4><<<1, TPB, dyn, s>>>(SAMPLE_ARGS);
#undef SAMPLE_ARGS
    return (int)hipGetLastError();
}

extern "C" int cstr_replay_sample_packed_mt19937_f32(const cstr_ring_t *ring, const int64_t *ring_ctl, uint32_t *mt_state,
                                                     int64_t batch, float *x_data, float *x_next, float *x_pi, float *out_done,
                                                     float *out_rew, int64_t *out_row_idx, int64_t *out_env_idx, cstr_stream_t stream)
{
    if (!ring || !ring->obs || !ring->next_obs || !ring->act || !ring->rew || !ring->done || !ring->timeout) return CSTR_E_BADARG;
    if (!ring_ctl || !mt_state || !x_data || !x_next || !out_done || !out_rew || batch <= 0) return CSTR_E_BADARG;
    const bool lay_ok = (ring->obs_dim == 4 && ring->act_dim == 2) || (ring->obs_dim == 8 && (ring->act_dim == 2 || ring->act_dim == 4));
    if (!lay_ok) return CSTR_E_UNSUPPORTED;
    if (batch > CSTR_MAX_SAMPLE_BATCH || ring->rows >= 0xFFFFFFFFLL || ring->n_envs >= 0xFFFFFFFFLL) return CSTR_E_UNSUPPORTED;
    if (!aligned16(ring->obs) || !aligned16(ring->next_obs) || !aligned8(ring->act) || !aligned8(x_data) || !aligned8(x_next) ||
        (x_pi && !aligned8(x_pi)))
        return CSTR_E_BADARG;
    const size_t dyn = sizeof(int32_t) * 2 * (size_t)batch;
    hipStream_t s = (hipStream_t)stream;
#define SAMPLE_ARGS *ring, ring_ctl, mt_state, (int)batch, x_data, x_pi, x_next, out_done, out_rew, out_row_idx, out_env_idx
    if (ring->obs_dim == 4) replay_sample_kernel<4, 2, true><<<1, TPB, dyn, s>>>(SAMPLE_ARGS);
    else if (ring->act_dim == 2) replay_sample_kernel<8, 2, true><<<1, TPB, dyn, s>>>(SAMPLE_ARGS);
    else replay_sample_kernel<8, 4, true><<<1, TPB, dyn, s>>>(SAMPLE_ARGS);
#undef SAMPLE_ARGS
    return (int)hipGetLastError();
}

extern "C" int cstr_replay_gather_packed_f32(const cstr_ring_t *ring, int64_t *ring_ctl, int advance_ring, uint64_t *rng_ctl,
                                             uint64_t rng_advance, const int32_t *sample_idx, int64_t batch, float *x_data, float *x_next,
                                             float *x_pi, float *out_done, float *out_rew, int64_t *out_row_idx, int64_t *out_env_idx,
                                             cstr_stream_t stream)
{
    if (!ring || !ring->obs || !ring->next_obs || !ring->act || !ring->rew || !ring->done || !ring->timeout) return CSTR_E_BADARG;
    if (!sample_idx || !x_data || !x_next || !out_done || !out_rew || batch <= 0) return CSTR_E_BADARG;
    if (advance_ring && !ring_ctl) return CSTR_E_BADARG;
    const bool lay_ok = (ring->obs_dim == 4 && ring->act_dim == 2) || (ring->obs_dim == 8 && (ring->act_dim == 2 || ring->act_dim == 4));
    if (!lay_ok) return CSTR_E_UNSUPPORTED;
    if (batch > CSTR_MAX_SAMPLE_BATCH || ring->rows >= 0xFFFFFFFFLL || ring->n_envs >= 0xFFFFFFFFLL) return CSTR_E_UNSUPPORTED;
    if (!aligned16(ring->obs) || !aligned16(ring->next_obs) || !aligned8(ring->act) || !aligned8(x_data) || !aligned8(x_next) ||
        (x_pi && !aligned8(x_pi)))
        return CSTR_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
#define GATHER_ARGS *ring, ring_ctl, advance_ring, rng_ctl, rng_advance, sample_idx, (int)batch, x_data, x_pi, x_next, out_done, out_rew, out_row_idx, out_env_idx
    if (ring->obs_dim == 4) replay_gather_kernel<4, 2><<<1, TPB, 0, s>>>(GATHER_ARGS);
    else if (ring->act_dim == 2) replay_gather_kernel<8, 2><<<1, TPB, 0, s>>>(GATHER_ARGS);
    else replay_gather_kernel<8, 4><<<1, TPB, 0, s>>>(GATHER_ARGS);
#undef GATHER_ARGS
    return (int)hipGetLastError();
}
