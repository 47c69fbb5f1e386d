// cstr_env.hip -- batched two-series CSTR environment kernels for gfx950 (MI355X).
//
// One lane = one reactor train (independent trajectory). State/ring rows are [N][D] f32 with D = 4
// (or 8), so a lane's observation is one 16-byte (two 16-byte) coalesced access: a wave moves 1 KiB
// per instruction, the widest the memory pipeline takes. The reaction coefficients are wave-uniform:
// they arrive by value in the kernarg segment and live in SGPRs (scalar loads, zero VGPR/LDS cost);
// staging them through LDS would only add a ds_read per use. HBM-bound streaming work, no MFMA.
//
// Numerics follow the reference's NumPy-f32 evaluation order (twoseriescstr.py:456-503) exactly;
// the TU is built with -ffp-contract=off so nothing is fused behind its back.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cstr_rl_hip.h"
#include "cstr_device.h"

namespace {

// ---- dynamics ------------------------------------------------------------------------------------

// safe_exp (twoseriescstr.py:476-477)
__device__ __forceinline__ float safe_expf(float x) { return expf(fminf(fmaxf(x, -100.0f), 100.0f)); }

// One reactor's RHS (twoseriescstr.py:479-484 / :486-491). c_in/t_in: feed (Cf,Tf | pre-step C1,T1).
__device__ __forceinline__ void reactor_rhs(const cstr_coef_t &k, float q_v, float cool, float neg_ua, float c_in,
                                            float t_in, float C, float T, float F, float &dC, float &dT)
{
    const float arr = safe_expf(k.neg_e / (k.r_gas * T));
    dC = q_v * (c_in - C) - k.k0 * C * arr;
    const float heat = ((k.hk * C) / k.rho_cp) * arr;
    // Jacket term: (1 - exp(-(U A)/(F rho_c c_pc))). For every admissible coolant flow (F in [30, 250] after the
    // action clip) the exponent is <= -98.9, exp() < 2^-25 and 1 - exp() == 1.0f EXACTLY, so the second expf of
    // the reference is skipped whenever the argument is below -18 (exp(-18) < 2^-25): same bits, half the
    // transcendental work. The guarded branch keeps the full expression for out-of-box flows (F up to 1e5).
    const float jarg = neg_ua / (F * k.rho_c * k.c_pc);
    const float one_minus = (jarg < -18.0f) ? 1.0f : (1.0f - safe_expf(jarg));
    const float jacket = cool * F * one_minus * (k.tcf - T);
    dT = q_v * (t_in - T) + heat + jacket;
}

__device__ __forceinline__ void cstr_rhs(const cstr_coef_t &k, const float s[4], float F1, float F2, float d[4])
{
    const float T1 = fmaxf(s[1], 273.15f), T2 = fmaxf(s[3], 273.15f);  // :470-471
    F1 = fminf(fmaxf(F1, 1e-5f), 1e5f);                                // :472-473
    F2 = fminf(fmaxf(F2, 1e-5f), 1e5f);
    reactor_rhs(k, k.q_v1, k.cool1, k.neg_ua1, k.cf, k.tf, s[0], T1, F1, d[0], d[1]);
    reactor_rhs(k, k.q_v2, k.cool2, k.neg_ua2, s[0], T1, s[2], T2, F2, d[2], d[3]);
}

// compute_reward, effective terms (weights twoseriescstr.py:369-377); raw state re-derived from the NEW
// normalised observation like :283 does.
__device__ __forceinline__ float cstr_reward(const cstr_coef_t &k, const float o[4])
{
    const float C2 = k.s_lo[2] + (o[2] + 1.0f) * k.s_span[2] / 2.0f;
    const float T1 = k.s_lo[1] + (o[1] + 1.0f) * k.s_span[1] / 2.0f;
    const float T2 = k.s_lo[3] + (o[3] + 1.0f) * k.s_span[3] / 2.0f;
    const float ne = fabsf(C2 - k.target_c2) / k.conc_span;  // :288-290
    const float conc = -5.0f * (ne * ne) - 2.0f * ne;        // :291
    float tp = 0.0f;                                         // :331-341
    if (T1 < 280.0f) tp -= 0.2f * ((280.0f - T1) / 280.0f);
    else if (T1 > 350.0f) tp -= 0.5f * ((T1 - 350.0f) / 350.0f);
    if (T2 < 280.0f) tp -= 0.2f * ((280.0f - T2) / 280.0f);
    else if (T2 > 350.0f) tp -= 0.5f * ((T2 - 350.0f) / 350.0f);
    return 1.0f * conc + 0.5f * tp;
}

// TwoSeriesCSTREnv.step for the lane's env (twoseriescstr.py:394-454). Returns `truncated`.
// o_new[4]: new normalised state; raw_new[4]: clipped raw state (info["original_state"], :446).
template <int INTEG>
__device__ __forceinline__ bool cstr_step_lane(const cstr_coef_t &k, const float o[4], float a0, float a1, int32_t &step,
                                               float o_new[4], float raw_new[4], float &reward)
{
    step += 1;  // :396
    // np.clip propagates NaN; fminf/fmaxf would drop it, so NaN is tested on the inputs first
    bool bad = (a0 != a0) || (a1 != a1);
    const float an0 = fminf(fmaxf(a0, -1.0f), 1.0f), an1 = fminf(fmaxf(a1, -1.0f), 1.0f);  // :399
    const float F1 = k.a_lo[0] + (an0 + 1.0f) * k.a_span[0] / 2.0f;                         // :148-149
    const float F2 = k.a_lo[1] + (an1 + 1.0f) * k.a_span[1] / 2.0f;
    float s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bad |= (o[i] != o[i]);
        const float r = k.s_lo[i] + (o[i] + 1.0f) * k.s_span[i] / 2.0f;  // :404
        s[i] = fminf(fmaxf(r, k.s_lo[i]), k.s_hi[i]);                    // :406-410
    }
    if (bad) {  // _dynamics raises (:466-467) -> step returns the old state, -10, truncated (:415-421)
#pragma unroll
        for (int i = 0; i < 4; ++i) { o_new[i] = o[i]; raw_new[i] = s[i]; }
        reward = -10.0f;
        return true;
    }
    float n[4];
    if (INTEG == CSTR_INTEGRATOR_EULER) {
        float d[4];
        cstr_rhs(k, s, F1, F2, d);
#pragma unroll
        for (int i = 0; i < 4; ++i) n[i] = s[i] + d[i] * k.dt;  // :493-496
    } else {
        float k1[4], k2[4], k3[4], k4[4], t[4];
        const float h = k.dt, h2 = 0.5f * k.dt;
        cstr_rhs(k, s, F1, F2, k1);
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = s[i] + h2 * k1[i];
        cstr_rhs(k, t, F1, F2, k2);
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = s[i] + h2 * k2[i];
        cstr_rhs(k, t, F1, F2, k3);
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = s[i] + h * k3[i];
        cstr_rhs(k, t, F1, F2, k4);
#pragma unroll
        for (int i = 0; i < 4; ++i) n[i] = s[i] + (h / 6.0f) * (k1[i] + 2.0f * k2[i] + 2.0f * k3[i] + k4[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        raw_new[i] = fminf(fmaxf(n[i], k.s_lo[i]), k.s_hi[i]);            // :499-503, :424-428
        o_new[i] = 2.0f * (raw_new[i] - k.s_lo[i]) / k.s_span[i] - 1.0f;  // :131, :429
    }
    reward = cstr_reward(k, o_new);  // :432
    return step >= k.max_steps;      // :438
}

// ---- per-env reset draw (numpy PCG64 + Generator.uniform) -------------------------------------------

__device__ __forceinline__ double pcg64_next_double(uint64_t st[4])
{
    const unsigned __int128 mult = ((unsigned __int128)2549297995355413924ULL << 64) | 4865540595714422341ULL;
    unsigned __int128 s = ((unsigned __int128)st[0] << 64) | st[1];
    const unsigned __int128 inc = ((unsigned __int128)st[2] << 64) | st[3];
    s = s * mult + inc;  // pcg_setseq_128_step_r
    st[0] = (uint64_t)(s >> 64);
    st[1] = (uint64_t)s;
    const uint64_t x = st[0] ^ st[1];  // XSL-RR
    const unsigned rot = (unsigned)(st[0] >> 58);
    const uint64_t r = (x >> rot) | (x << ((0u - rot) & 63u));
    return (double)(r >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double pcg64_uniform(uint64_t st[4], double lo, double hi) { return lo + (hi - lo) * pcg64_next_double(st); }

// generate_initial_state + _normalize_state (twoseriescstr.py:187-224, :267): f64 draws, swaps, clip, f64
// normalisation against the f32 box, cast to f32.
__device__ __forceinline__ void cstr_reset_draw_lane(uint64_t st[4], float o[4])
{
    double s[4];
    s[0] = pcg64_uniform(st, 0.05, 0.45);
    s[1] = pcg64_uniform(st, 280.0, 380.0);
    s[2] = pcg64_uniform(st, 0.05, 0.45 * 0.8);
    s[3] = pcg64_uniform(st, 280.0, 380.0);
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] += pcg64_uniform(st, -0.05, 0.05);  // :202-207
    if (s[1] < s[3]) { const double t = s[1]; s[1] = s[3]; s[3] = t; }    // :211-212
    if (s[0] < s[2]) { const double t = s[0]; s[0] = s[2]; s[2] = t; }    // :214-215
    const float lo[4] = {0.0f, 273.15f, 0.0f, 273.15f}, hi[4] = {0.7f, 400.0f, 0.7f, 400.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double c = fmin(fmax(s[i], (double)lo[i]), (double)hi[i]);  // :218-222
        const float span = hi[i] - lo[i];
        o[i] = (float)(2.0 * (c - (double)lo[i]) / (double)span - 1.0);   // :131-132
    }
}

// init_mode="static" (twoseriescstr.py:94-96, :246-255): the env's f64 `init_state` ([0.45, 310, 0.25, 290] at construction)
// is perturbed IN PLACE by Generator.uniform([-0.05,-10,-0.05,-10], [0.05,10,0.05,10]) at every reset -- a per-env random
// walk that is never clipped -- and then normalised in f64 against the f32 box.
__device__ __forceinline__ void cstr_reset_static_lane(uint64_t st[4], double *init, float o[4])
{
    const float lo[4] = {0.0f, 273.15f, 0.0f, 273.15f}, hi[4] = {0.7f, 400.0f, 0.7f, 400.0f};
    const double nlo[4] = {-0.05, -10.0, -0.05, -10.0}, nhi[4] = {0.05, 10.0, 0.05, 10.0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double s = init[i] + pcg64_uniform(st, nlo[i], nhi[i]);  // initial_state += noise (:255)
        init[i] = s;
        const float span = hi[i] - lo[i];
        o[i] = (float)(2.0 * (s - (double)lo[i]) / (double)span - 1.0);  // :131-132
    }
}

// raw half of the 8-dim observation of a freshly reset env: _denormalize_state of the normalised half
__device__ __forceinline__ void denorm4(const cstr_coef_t &k, const float o[4], float raw[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) raw[i] = k.s_lo[i] + (o[i] + 1.0f) * k.s_span[i] / 2.0f;
}

// ---- observation layouts ----------------------------------------------------------------------------
// L = 0: D=4, A=2  the reference's observation [C1,T1,C2,T2] normalised (twoseriescstr.py:74-85)
// L = 1: D=8, A=2  [normalised | raw] (both halves are what the reference's `info` carries)
// L = 2: D=8, A=4  TWO reactor trains side by side, [train A normalised | train B normalised], actions
//                  [F1A, F2A, F1B, F2B], reward = rA + rB, one step counter / one reset stream per env: the 8-obs/4-act
//                  environment MADDPG's 4-agent configuration needs (no such env exists in the reference).
template <int L>
struct Lay {
    static constexpr int D = (L == 0) ? 4 : 8, A = (L == 2) ? 4 : 2, TR = (L == 2) ? 2 : 1;
};

template <int L>
__device__ __forceinline__ void load_obs(const float *p, int64_t i, float o[2][4])
{
    const float4 v = *reinterpret_cast<const float4 *>(p + i * Lay<L>::D);
    o[0][0] = v.x; o[0][1] = v.y; o[0][2] = v.z; o[0][3] = v.w;
    if (L != 0) {  // L=1: raw half (carried through to the ring); L=2: train B
        const float4 w = *reinterpret_cast<const float4 *>(p + i * 8 + 4);
        o[1][0] = w.x; o[1][1] = w.y; o[1][2] = w.z; o[1][3] = w.w;
    }
}

template <int L>
__device__ __forceinline__ void store_obs(float *p, int64_t i, const float o[2][4])
{
    *reinterpret_cast<float4 *>(p + i * Lay<L>::D) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
    if (L != 0) *reinterpret_cast<float4 *>(p + i * 8 + 4) = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
}

// Ring rows are written once and not read again until some later sample(): streaming (non-temporal) stores keep them
// from displacing the env state and the parameters in L2 / Infinity Cache. Measured on MI355X at N = 2^22 (A/B, 4 interleaved
// rounds, profiles/r01_notes.md): 89.7 -> 70.3 us per launch, 4.86 -> 6.2 TB/s. -DCSTR_NT_STORES=0 restores plain stores.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int L>
__device__ __forceinline__ void store_ring_obs(float *p, int64_t i, const float o[2][4])
{
#if CSTR_NT_STORES
    v4f a = {o[0][0], o[0][1], o[0][2], o[0][3]};
    __builtin_nontemporal_store(a, reinterpret_cast<v4f *>(p + i * Lay<L>::D));
    if (L != 0) {
        v4f b = {o[1][0], o[1][1], o[1][2], o[1][3]};
        __builtin_nontemporal_store(b, reinterpret_cast<v4f *>(p + i * 8 + 4));
    }
#else
    store_obs<L>(p, i, o);
#endif
}

template <int A>
__device__ __forceinline__ void store_ring_act(float *p, int64_t i, const float a[4])
{
#if CSTR_NT_STORES
    if (A == 2) {
        v2f v = {a[0], a[1]};
        __builtin_nontemporal_store(v, reinterpret_cast<v2f *>(p + 2 * i));
    } else {
        v4f v = {a[0], a[1], a[2], a[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(p + 4 * i));
    }
#else
    if (A == 2) *reinterpret_cast<float2 *>(p + 2 * i) = make_float2(a[0], a[1]);
    else *reinterpret_cast<float4 *>(p + 4 * i) = make_float4(a[0], a[1], a[2], a[3]);
#endif
}

__device__ __forceinline__ void store_ring_f32(float *p, int64_t i, float v)
{
#if CSTR_NT_STORES
    __builtin_nontemporal_store(v, p + i);
#else
    p[i] = v;
#endif
}

template <int L>
__device__ __forceinline__ void copy_obs(float *dst, int64_t di, const float *src, int64_t si)
{
    constexpr int D = Lay<L>::D;
    *reinterpret_cast<float4 *>(dst + di * D) = *reinterpret_cast<const float4 *>(src + si * D);
    if (D == 8) *reinterpret_cast<float4 *>(dst + di * D + 4) = *reinterpret_cast<const float4 *>(src + si * D + 4);
}

// All trains of one env advance one step. on[][]: next observation in the layout's register image.
template <int L, int INTEG>
__device__ __forceinline__ bool env_step_lane(const cstr_coef_t &k, const float o[2][4], const float a[4], int32_t &step,
                                              float on[2][4], float &reward)
{
    float raw[4], r0;
    int32_t st = step;
    bool trunc = cstr_step_lane<INTEG>(k, o[0], a[0], a[1], st, on[0], raw, r0);
    reward = r0;
    if (L == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) on[1][j] = raw[j];
    }
    if (L == 2) {
        int32_t st2 = step;
        float r1;
        trunc |= cstr_step_lane<INTEG>(k, o[1], a[2], a[3], st2, on[1], raw, r1);
        reward = r0 + r1;
    }
    step = st;
    return trunc;
}

template <int L>
__device__ __forceinline__ void reset_draw_env(const cstr_coef_t &k, uint64_t st[4], double *static_init, int64_t i, float o[2][4])
{
    constexpr int TR = Lay<L>::TR;
    if (static_init) cstr_reset_static_lane(st, static_init + 4 * TR * i, o[0]);
    else cstr_reset_draw_lane(st, o[0]);
    if (L == 1) denorm4(k, o[0], o[1]);
    if (L == 2) {  // train B continues the env's stream
        if (static_init) cstr_reset_static_lane(st, static_init + 4 * TR * i + 4, o[1]);
        else cstr_reset_draw_lane(st, o[1]);
    }
}

__device__ __forceinline__ void load_pcg(const uint64_t *pcg, int64_t i, uint64_t st[4])
{
    const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(pcg + 4 * i), b = *reinterpret_cast<const ulonglong2 *>(pcg + 4 * i + 2);
    st[0] = a.x; st[1] = a.y; st[2] = b.x; st[3] = b.y;
}

template <int A>
__device__ __forceinline__ void load_act(const float *p, int64_t i, float a[4])
{
    if (A == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(p + 2 * i);
        a[0] = v.x; a[1] = v.y; a[2] = 0.0f; a[3] = 0.0f;
    } else {
        const float4 v = *reinterpret_cast<const float4 *>(p + 4 * i);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    }
}

template <int A>
__device__ __forceinline__ void store_act(float *p, int64_t i, const float a[4])
{
    if (A == 2) *reinterpret_cast<float2 *>(p + 2 * i) = make_float2(a[0], a[1]);
    else *reinterpret_cast<float4 *>(p + 4 * i) = make_float4(a[0], a[1], a[2], a[3]);
}

// ---- kernels --------------------------------------------------------------------------------------

template <int L, int INTEG>
__global__ void vec_step_kernel(const cstr_coef_t k, const float *__restrict__ obs, const float *__restrict__ act,
                                int32_t *__restrict__ step_count, const float *__restrict__ reset_obs,
                                float *__restrict__ next_obs, float *obs_after, float *__restrict__ reward,
                                float *__restrict__ done, float *__restrict__ timeout, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float o[2][4], on[2][4], a[4], r;
        load_obs<L>(obs, i, o);
        load_act<Lay<L>::A>(act, i, a);
        int32_t st = step_count[i];
        const bool trunc = env_step_lane<L, INTEG>(k, o, a, st, on, r);
        const bool d = trunc;  // terminated is always False (:435); done = terminated or truncated (dummy_vec_env.py:63)
        store_obs<L>(next_obs, i, on);
        if (d) copy_obs<L>(obs_after, i, reset_obs, i);  // dummy_vec_env.py:68-72
        else store_obs<L>(obs_after, i, on);
        reward[i] = r;
        done[i] = d ? 1.0f : 0.0f;
        timeout[i] = trunc ? 1.0f : 0.0f;  // dummy_vec_env.py:66
        step_count[i] = d ? 0 : st;
    }
}

template <int L>
__global__ void reset_draw_kernel(const cstr_coef_t k, uint64_t *__restrict__ pcg, const uint8_t *__restrict__ mask,
                                  double *__restrict__ static_init, float *__restrict__ obs_out, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (mask && !mask[i]) continue;
        uint64_t st[4];
        load_pcg(pcg, i, st);
        float o[2][4];
        reset_draw_env<L>(k, st, static_init, i, o);
        store_obs<L>(obs_out, i, o);
        *reinterpret_cast<ulonglong2 *>(pcg + 4 * i) = make_ulonglong2(st[0], st[1]);
    }
}

struct ActBounds { float lo[4], hi[4]; };

// Fused collect step: action scaling chain + env step + auto-reset + ring row write, one pass.
// Algorithmic HBM traffic per env (D = 4): 16 B state + 8 B action + 4 B step read; 52 B ring row
// + 16 B state + 4 B step written  => 104 B / env-step.
template <int L, int INTEG>
__global__ void collect_step_kernel(const cstr_coef_t k, const cstr_ring_t ring, int64_t *__restrict__ ring_ctl,
                                    float *__restrict__ env_obs, int32_t *__restrict__ step_count,
                                    const float *__restrict__ policy_out, const int squashed, const ActBounds ab,
                                    const float *__restrict__ noise, const float *__restrict__ reset_obs,
                                    uint64_t *__restrict__ pcg, double *__restrict__ static_init, float *__restrict__ reward_out,
                                    float *__restrict__ done_out, float *__restrict__ ep_return, double *__restrict__ ep_stats,
                                    uint64_t *__restrict__ policy_rng_ctl, const uint64_t policy_rng_advance)
{
    constexpr int A = Lay<L>::A;
    const int64_t n = ring.n_envs;
    const int64_t pos = ring_ctl[0];  // wave-uniform scalar load
    const int64_t row = pos * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float o[2][4], on[2][4], u[4], sa[4], ea[4], z[4], r;
        load_obs<L>(env_obs, i, o);
        load_act<A>(policy_out, i, u);
        if (noise) load_act<A>(noise, i, z);
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float lo = ab.lo[j], hi = ab.hi[j];
            float v = u[j];
            if (squashed & 1) v = lo + (0.5f * (v + 1.0f) * (hi - lo));  // predict(): unscale_action (policies.py:375, :413)
            if (squashed & 2) {
                // multi-agent algorithms: `isinstance(any(...), spaces.Box)` is always False in the reference, so neither
                // scaling nor action noise is applied and buffer_action = action = predict() output
                // (core/common/multiagent_policy_algorithm.py:369, :391-392)
                sa[j] = ea[j] = v;
            } else {
                float sc = 2.0f * ((v - lo) / (hi - lo)) - 1.0f;        // scale_action (policies.py:402)
                if (noise) sc = fminf(fmaxf(sc + z[j], -1.0f), 1.0f);   // off_policy_algorithm.py:401-402
                sa[j] = sc;                                              // buffer_action (:405)
                ea[j] = lo + (0.5f * (sc + 1.0f) * (hi - lo));           // unscale_action (:406)
            }
        }
        int32_t st = step_count[i];
        const bool trunc = env_step_lane<L, INTEG>(k, o, ea, st, on, r);
        const bool d = trunc;

        // ring row: obs = _last_obs, next_obs = terminal observation (off_policy_algorithm.py:477-496)
        store_ring_obs<L>(ring.obs, row + i, o);
        store_ring_obs<L>(ring.next_obs, row + i, on);
        store_ring_act<A>(ring.act, row + i, sa);
        store_ring_f32(ring.rew, row + i, r);
        store_ring_f32(ring.done, row + i, d ? 1.0f : 0.0f);
        store_ring_f32(ring.timeout, row + i, trunc ? 1.0f : 0.0f);
        if (reward_out) reward_out[i] = r;
        if (done_out) done_out[i] = d ? 1.0f : 0.0f;
        if (ep_return) {  // Monitor semantics: return/length of the episode that ends here
            const float ret = ep_return[i] + r;
            ep_return[i] = d ? 0.0f : ret;
            if (d) {
                atomicAdd(ep_stats + 0, 1.0);
                atomicAdd(ep_stats + 1, (double)ret);
                atomicAdd(ep_stats + 2, (double)st);
            }
        }

        // env state for the next iteration (dummy_vec_env.py:68-72)
        if (d) {
            if (reset_obs) {
                copy_obs<L>(env_obs, i, reset_obs, i);
            } else {
                uint64_t pst[4];
                load_pcg(pcg, i, pst);
                float ro[2][4];
                reset_draw_env<L>(k, pst, static_init, i, ro);
                store_obs<L>(env_obs, i, ro);
                *reinterpret_cast<ulonglong2 *>(pcg + 4 * i) = make_ulonglong2(pst[0], pst[1]);
            }
            st = 0;
        } else {
            store_obs<L>(env_obs, i, on);
        }
        step_count[i] = st;
    }
    ring_advance_last_block(ring_ctl, ring.rows, policy_rng_ctl, policy_rng_advance);
}

// ReplayBuffer.add: six row copies in one launch
template <int L>
__global__ void replay_add_kernel(const cstr_ring_t ring, int64_t *__restrict__ ring_ctl, const float *__restrict__ obs,
                                  const float *__restrict__ next_obs, const float *__restrict__ act,
                                  const float *__restrict__ rew, const float *__restrict__ done,
                                  const float *__restrict__ timeout)
{
    constexpr int A = Lay<L>::A;
    const int64_t n = ring.n_envs, row = ring_ctl[0] * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float a[4];
        copy_obs<L>(ring.obs, row + i, obs, i);
        copy_obs<L>(ring.next_obs, row + i, next_obs, i);
        load_act<A>(act, i, a);
        store_act<A>(ring.act, row + i, a);
        ring.rew[row + i] = rew[i];
        ring.done[row + i] = done[i];
        ring.timeout[row + i] = timeout[i];
    }
    ring_advance_last_block(ring_ctl, ring.rows);
}

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------

static int layout_of(int obs_dim, int act_dim)
{
    if (obs_dim == 4 && act_dim == 2) return 0;
    if (obs_dim == 8 && act_dim == 2) return 1;
    if (obs_dim == 8 && act_dim == 4) return 2;
    return -1;
}

#define DISPATCH_L_INTEG(KERNEL, ...)                                                                       \
    do {                                                                                                    \
        const bool eu = integrator == CSTR_INTEGRATOR_EULER;                                               \
        if (layout == 0 && eu) KERNEL<0, CSTR_INTEGRATOR_EULER><<<grid, block, 0, s>>>(__VA_ARGS__);        \
        else if (layout == 0) KERNEL<0, CSTR_INTEGRATOR_RK4><<<grid, block, 0, s>>>(__VA_ARGS__);           \
        else if (layout == 1 && eu) KERNEL<1, CSTR_INTEGRATOR_EULER><<<grid, block, 0, s>>>(__VA_ARGS__);   \
        else if (layout == 1) KERNEL<1, CSTR_INTEGRATOR_RK4><<<grid, block, 0, s>>>(__VA_ARGS__);           \
        else if (eu) KERNEL<2, CSTR_INTEGRATOR_EULER><<<grid, block, 0, s>>>(__VA_ARGS__);                  \
        else KERNEL<2, CSTR_INTEGRATOR_RK4><<<grid, block, 0, s>>>(__VA_ARGS__);                            \
    } while (0)

extern "C" int cstr_vec_step_f32(const cstr_coef_t *coef, int integrator, int obs_dim, int act_dim, const float *obs,
                                 const float *act, int32_t *step_count, const float *reset_obs, float *next_obs,
                                 float *obs_after, float *reward, float *done, float *timeout, int64_t n_envs,
                                 cstr_stream_t stream)
{
    if (!coef || !obs || !act || !step_count || !reset_obs || !next_obs || !obs_after || !reward || !done || !timeout || n_envs <= 0)
        return CSTR_E_BADARG;
    const int layout = layout_of(obs_dim, act_dim);
    if (layout < 0 || (integrator != CSTR_INTEGRATOR_EULER && integrator != CSTR_INTEGRATOR_RK4)) return CSTR_E_UNSUPPORTED;
    if (!aligned16(obs) || !aligned16(reset_obs) || !aligned16(next_obs) || !aligned16(obs_after) ||
        !(act_dim == 4 ? aligned16(act) : aligned8(act)))
        return CSTR_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    int block, grid;
    env_launch_shape(n_envs, block, grid);
    DISPATCH_L_INTEG(vec_step_kernel, *coef, obs, act, step_count, reset_obs, next_obs, obs_after, reward, done, timeout, n_envs);
    return (int)hipGetLastError();
}

extern "C" int cstr_reset_draw_f32(uint64_t *pcg_state, const uint8_t *mask, double *static_init, int obs_dim, int act_dim,
                                   float *obs_out, int64_t n_envs, cstr_stream_t stream)
{
    if (!pcg_state || !obs_out || n_envs <= 0 || !aligned16(obs_out) || !aligned16(pcg_state)) return CSTR_E_BADARG;
    const int layout = layout_of(obs_dim, act_dim);
    if (layout < 0) return CSTR_E_UNSUPPORTED;
    cstr_coef_t k;
    cstr_default_coef(&k, 0.2, 0.05, 0.45, 400);
    hipStream_t s = (hipStream_t)stream;
    int block, grid;
    env_launch_shape(n_envs, block, grid);
    if (layout == 0) reset_draw_kernel<0><<<grid, block, 0, s>>>(k, pcg_state, mask, static_init, obs_out, n_envs);
    else if (layout == 1) reset_draw_kernel<1><<<grid, block, 0, s>>>(k, pcg_state, mask, static_init, obs_out, n_envs);
    else reset_draw_kernel<2><<<grid, block, 0, s>>>(k, pcg_state, mask, static_init, obs_out, n_envs);
    return (int)hipGetLastError();
}

static int check_ring(const cstr_ring_t *r)
{
    if (!r || !r->obs || !r->next_obs || !r->act || !r->rew || !r->done || !r->timeout || r->rows <= 0 || r->n_envs <= 0)
        return CSTR_E_BADARG;
    if (layout_of(r->obs_dim, r->act_dim) < 0) return CSTR_E_UNSUPPORTED;
    if (!aligned16(r->obs) || !aligned16(r->next_obs) || !(r->act_dim == 4 ? aligned16(r->act) : aligned8(r->act))) return CSTR_E_BADARG;
    return CSTR_OK;
}

extern "C" int cstr_replay_add_f32(const cstr_ring_t *ring, int64_t *ring_ctl, const float *obs, const float *next_obs,
                                   const float *act, const float *rew, const float *done, const float *timeout,
                                   cstr_stream_t stream)
{
    int rc = check_ring(ring);
    if (rc) return rc;
    if (!ring_ctl || !obs || !next_obs || !act || !rew || !done || !timeout) return CSTR_E_BADARG;
    if (!aligned16(obs) || !aligned16(next_obs) || !(ring->act_dim == 4 ? aligned16(act) : aligned8(act))) return CSTR_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    int block, grid;
    env_launch_shape(ring->n_envs, block, grid);
    const int layout = layout_of(ring->obs_dim, ring->act_dim);
    if (layout == 0) replay_add_kernel<0><<<grid, block, 0, s>>>(*ring, ring_ctl, obs, next_obs, act, rew, done, timeout);
    else if (layout == 1) replay_add_kernel<1><<<grid, block, 0, s>>>(*ring, ring_ctl, obs, next_obs, act, rew, done, timeout);
    else replay_add_kernel<2><<<grid, block, 0, s>>>(*ring, ring_ctl, obs, next_obs, act, rew, done, timeout);
    return (int)hipGetLastError();
}

extern "C" int cstr_collect_step_f32(const cstr_coef_t *coef, int integrator, const cstr_ring_t *ring, int64_t *ring_ctl,
                                     float *env_obs, int32_t *step_count, const float *policy_out, int squashed,
                                     const float *act_low, const float *act_high, const float *noise,
                                     const float *reset_obs, uint64_t *pcg_state, double *static_init, float *reward_out,
                                     float *done_out, float *ep_return, double *ep_stats, cstr_stream_t stream)
{
    return cstr_collect_step_rng_f32(coef, integrator, ring, ring_ctl, env_obs, step_count, policy_out, squashed, act_low, act_high, noise,
                                     reset_obs, pcg_state, static_init, reward_out, done_out, ep_return, ep_stats, nullptr, 0, stream);
}

extern "C" int cstr_collect_step_rng_f32(const cstr_coef_t *coef, int integrator, const cstr_ring_t *ring, int64_t *ring_ctl,
                                         float *env_obs, int32_t *step_count, const float *policy_out, int squashed,
                                         const float *act_low, const float *act_high, const float *noise,
                                         const float *reset_obs, uint64_t *pcg_state, double *static_init, float *reward_out,
                                         float *done_out, float *ep_return, double *ep_stats, uint64_t *policy_rng_ctl,
                                         uint64_t policy_rng_advance, cstr_stream_t stream)
{
    int rc = check_ring(ring);
    if (rc) return rc;
    if (!coef || !ring_ctl || !env_obs || !step_count || !policy_out || !act_low || !act_high) return CSTR_E_BADARG;
    if ((reset_obs == nullptr) == (pcg_state == nullptr)) return CSTR_E_BADARG;  // exactly one reset source
    if (static_init && !pcg_state) return CSTR_E_BADARG;
    if ((ep_return == nullptr) != (ep_stats == nullptr)) return CSTR_E_BADARG;
    if (integrator != CSTR_INTEGRATOR_EULER && integrator != CSTR_INTEGRATOR_RK4) return CSTR_E_UNSUPPORTED;
    const int A = ring->act_dim;
    const bool a_ok = A == 4 ? (aligned16(policy_out) && (!noise || aligned16(noise))) : (aligned8(policy_out) && (!noise || aligned8(noise)));
    if (!aligned16(env_obs) || !a_ok || (reset_obs && !aligned16(reset_obs)) || (pcg_state && !aligned16(pcg_state))) return CSTR_E_BADARG;
    ActBounds ab;
    for (int j = 0; j < 4; ++j) {
        ab.lo[j] = j < A ? act_low[j] : -1.0f;
        ab.hi[j] = j < A ? act_high[j] : 1.0f;
        if (!(ab.hi[j] > ab.lo[j])) return CSTR_E_BADARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int layout = layout_of(ring->obs_dim, A);
    int block, grid;
    env_launch_shape(ring->n_envs, block, grid);
    DISPATCH_L_INTEG(collect_step_kernel, *coef, *ring, ring_ctl, env_obs, step_count, policy_out, squashed, ab, noise, reset_obs,
                     pcg_state, static_init, reward_out, done_out, ep_return, ep_stats, policy_rng_ctl, policy_rng_advance);
    return (int)hipGetLastError();
}
