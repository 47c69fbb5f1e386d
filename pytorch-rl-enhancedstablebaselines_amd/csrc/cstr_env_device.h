// cstr_env_device.h -- device functions of the two-series CSTR environment (internal, included by cstr_env.hip and by the
// rollout kernel in cstr_mlp.hip): dynamics, reward, step, PCG64 reset draws, observation layouts, ring-row stores and the
// per-env body of the fused collect step. Numerics follow the reference's NumPy-f32 evaluation order
// (twoseriescstr.py:456-503); every TU that includes this is built with -ffp-contract=off.
#pragma once
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


struct ActBounds { float lo[4], hi[4]; };

// What one env of the fused collect step needs besides its own action (cstr_collect_step_f32): everything is wave-uniform.
struct CollectArgs {
    cstr_ring_t ring;
    float *env_obs; int32_t *step_count;
    int squashed; ActBounds ab;
    const float *noise, *reset_obs;
    uint64_t *pcg; double *static_init;
    float *reward_out, *done_out, *ep_return; double *ep_stats;
};

// What env i's collect step reads besides its action: requested early by the rollout kernel (its sampling tail then starts
// from registers), right in front of the arithmetic by collect_step_kernel.
struct CollectIn { float o[2][4], z[4], ep_ret; int32_t st; };

template <int L>
__device__ __forceinline__ void collect_env_load(const CollectArgs &c, const int64_t i, CollectIn &in)
{
    load_obs<L>(c.env_obs, i, in.o);
    if (c.noise) load_act<Lay<L>::A>(c.noise, i, in.z);
    in.st = c.step_count[i];
    in.ep_ret = c.ep_return ? c.ep_return[i] : 0.0f;
}

// Env i of the fused collect step: action scaling chain (off_policy_algorithm.py:364-411) on the policy output u[], env step,
// ring row `row` (= pos * n_envs), Monitor statistics, auto-reset (dummy_vec_env.py:68-72). Shared by collect_step_kernel (a
// lane per env, u[] loaded from the policy launch's output) and the rollout kernel (u[] straight from the sampling tail).
template <int L, int INTEG>
__device__ __forceinline__ void collect_env_lane(const cstr_coef_t &k, const CollectArgs &c, const int64_t row, const int64_t i,
                                                 const float u[4], const CollectIn &in)
{
    constexpr int A = Lay<L>::A;
    const cstr_ring_t &ring = c.ring;
    float on[2][4], sa[4], ea[4], r;
#pragma unroll
    for (int j = 0; j < A; ++j) {
        const float lo = c.ab.lo[j], hi = c.ab.hi[j];
        float v = u[j];
        if (c.squashed & 1) v = lo + (0.5f * (v + 1.0f) * (hi - lo));  // predict(): unscale_action (policies.py:375, :413)
        if (c.squashed & 2) {
            // multi-agent algorithms: `isinstance(any(...), spaces.Box)` is always False in the reference, so neither
            // scaling nor action noise is applied and buffer_action = action = predict() output
            // (core/common/multiagent_policy_algorithm.py:369, :391-392)
            sa[j] = ea[j] = v;
        } else {
            float sc = 2.0f * ((v - lo) / (hi - lo)) - 1.0f;            // scale_action (policies.py:402)
            if (c.noise) sc = fminf(fmaxf(sc + in.z[j], -1.0f), 1.0f);  // off_policy_algorithm.py:401-402
            sa[j] = sc;                                                  // buffer_action (:405)
            ea[j] = lo + (0.5f * (sc + 1.0f) * (hi - lo));               // unscale_action (:406)
        }
    }
    int32_t st = in.st;
    const bool trunc = env_step_lane<L, INTEG>(k, in.o, ea, st, on, r);
    const bool d = trunc;

    // ring row: obs = _last_obs, next_obs = terminal observation (off_policy_algorithm.py:477-496)
    store_ring_obs<L>(ring.obs, row + i, in.o);
    store_ring_obs<L>(ring.next_obs, row + i, on);
    store_ring_act<A>(ring.act, row + i, sa);
    store_ring_f32(ring.rew, row + i, r);
    store_ring_f32(ring.done, row + i, d ? 1.0f : 0.0f);
    store_ring_f32(ring.timeout, row + i, trunc ? 1.0f : 0.0f);
    if (c.reward_out) c.reward_out[i] = r;
    if (c.done_out) c.done_out[i] = d ? 1.0f : 0.0f;
    if (c.ep_return) {  // Monitor semantics: return/length of the episode that ends here
        const float ret = in.ep_ret + r;
        c.ep_return[i] = d ? 0.0f : ret;
        if (d) {
            atomicAdd(c.ep_stats + 0, 1.0);
            atomicAdd(c.ep_stats + 1, (double)ret);
            atomicAdd(c.ep_stats + 2, (double)st);
        }
    }

    // env state for the next iteration (dummy_vec_env.py:68-72)
    if (d) {
        if (c.reset_obs) {
            copy_obs<L>(c.env_obs, i, c.reset_obs, i);
        } else {
            uint64_t pst[4];
            load_pcg(c.pcg, i, pst);
            float ro[2][4];
            reset_draw_env<L>(k, pst, c.static_init, i, ro);
            store_obs<L>(c.env_obs, i, ro);
            *reinterpret_cast<ulonglong2 *>(c.pcg + 4 * i) = make_ulonglong2(pst[0], pst[1]);
        }
        st = 0;
    } else {
        store_obs<L>(c.env_obs, i, on);
    }
    c.step_count[i] = st;
}

// ---- the same collect step spread over the FOUR lanes of a quad (layouts with one reactor train, A = 2) --------------------------
// The rollout kernel's sampling tail has eight lanes per env (thread = 8 * row + j) and nothing else to run: with a lane per
// env the ~600-instruction step is a 2.8 us dependent chain on 8 active lanes of a wave (in-kernel stamps). Here lane j < 4 of
// the env's group owns state component j (C1, T1 | C2, T2) and lanes 0, 1 own action component j: the per-component work
// (denormalise, clip, update, normalise) is ONE instruction stream for four components, the two reactors' right-hand sides are
// ONE stream for two reactors (lanes 0-1: reactor 1, lanes 2-3: reactor 2; each lane keeps dC or dT), the reward's terms are
// evaluated beside each other. Every value is produced by the same IEEE operations in the same order as in cstr_step_lane /
// collect_env_lane -- the lanes only exchange finished values (DPP quad permutes) -- so results are bit-identical.
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ float quad_perm(const float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true));
}

__device__ __forceinline__ float sel4(const int j, const float a[4]) { return j == 0 ? a[0] : (j == 1 ? a[1] : (j == 2 ? a[2] : a[3])); }

struct CollectInQ { float o, oraw, z, ep_ret; int32_t st; };

template <int L>
__device__ __forceinline__ void collect_quad_load(const CollectArgs &c, const int64_t i, const int j, CollectInQ &in)
{
    constexpr int D = Lay<L>::D;
    in.o = c.env_obs[i * D + (j & 3)];
    in.oraw = (L == 1) ? c.env_obs[i * D + 4 + (j & 3)] : 0.0f;
    in.z = c.noise ? c.noise[i * 2 + (j & 1)] : 0.0f;
    in.st = c.step_count[i];
    in.ep_ret = c.ep_return ? c.ep_return[i] : 0.0f;
}

// one lane's share of cstr_rhs: d(component j)/dt of the state whose component j this lane holds in `sv`
__device__ __forceinline__ float quad_rhs(const cstr_coef_t &k, const int j, const float sv, const float F_lane)
{
    const bool r2 = (j & 2) != 0;
    const float C = quad_perm<0, 0, 2, 2>(sv), T = fmaxf(quad_perm<1, 1, 3, 3>(sv), 273.15f);              // :470-471
    const float s0 = quad_perm<0, 0, 0, 0>(sv), T1 = fmaxf(quad_perm<1, 1, 1, 1>(sv), 273.15f);
    const float F = fminf(fmaxf(quad_perm<0, 0, 1, 1>(F_lane), 1e-5f), 1e5f);                               // :472-473
    float dC, dT;
    reactor_rhs(k, r2 ? k.q_v2 : k.q_v1, r2 ? k.cool2 : k.cool1, r2 ? k.neg_ua2 : k.neg_ua1, r2 ? s0 : k.cf, r2 ? T1 : k.tf, C, T, F, dC, dT);
    return (j & 1) ? dT : dC;
}

// All eight lanes of an env's group call this together (j = lane & 7; `live`: the env exists); lanes j >= 4 only keep the quad
// permutes of their own quad well defined. u = the policy's action component j (lanes 0, 1).
template <int L, int INTEG>
__device__ __forceinline__ void collect_env_quad(const cstr_coef_t &k, const CollectArgs &c, const int64_t row, const int64_t i, const int j,
                                                 const bool live, const float u, const CollectInQ &in)
{
    static_assert(L == 0 || L == 1, "one reactor train, two actions");
    constexpr int D = Lay<L>::D;
    const cstr_ring_t &ring = c.ring;
    const int jj = j & 3, ja = j & 1;
    const bool own = live && j < 4, lane0 = live && j == 0;
    // action scaling chain of component ja (lanes 0, 1)
    const float lo = ja ? c.ab.lo[1] : c.ab.lo[0], hi = ja ? c.ab.hi[1] : c.ab.hi[0];
    float v = u, sa, ea;
    if (c.squashed & 1) v = lo + (0.5f * (v + 1.0f) * (hi - lo));  // predict(): unscale_action (policies.py:375, :413)
    if (c.squashed & 2) {
        sa = ea = v;
    } else {
        float sc = 2.0f * ((v - lo) / (hi - lo)) - 1.0f;         // scale_action (policies.py:402)
        if (c.noise) sc = fminf(fmaxf(sc + in.z, -1.0f), 1.0f);  // off_policy_algorithm.py:401-402
        sa = sc;                                                  // buffer_action (:405)
        ea = lo + (0.5f * (sc + 1.0f) * (hi - lo));               // unscale_action (:406)
    }
    // cstr_step_lane, component jj
    const float s_lo = sel4(jj, k.s_lo), s_hi = sel4(jj, k.s_hi), s_span = sel4(jj, k.s_span);
    const float an = fminf(fmaxf(ea, -1.0f), 1.0f);                                      // :399
    const float F_lane = (ja ? k.a_lo[1] : k.a_lo[0]) + (an + 1.0f) * (ja ? k.a_span[1] : k.a_span[0]) / 2.0f;  // :148-149
    const float o = in.o;
    const float rr = s_lo + (o + 1.0f) * s_span / 2.0f;                                  // :404
    const float s = fminf(fmaxf(rr, s_lo), s_hi);                                        // :406-410
    const bool bad_l = j < 4 && ((o != o) || (j < 2 && ea != ea));
    const unsigned long long bal = __ballot(bad_l);
    const bool bad = ((bal >> (threadIdx.x & 56)) & 0xFull) != 0;                        // any lane of this env's quad
    float n;
    if (INTEG == CSTR_INTEGRATOR_EULER) {
        n = s + quad_rhs(k, j, s, F_lane) * k.dt;                                        // :493-496
    } else {
        const float h = k.dt, h2 = 0.5f * k.dt;
        const float k1 = quad_rhs(k, j, s, F_lane);
        const float k2 = quad_rhs(k, j, s + h2 * k1, F_lane);
        const float k3 = quad_rhs(k, j, s + h2 * k2, F_lane);
        const float k4 = quad_rhs(k, j, s + h * k3, F_lane);
        n = s + (h / 6.0f) * (k1 + 2.0f * k2 + 2.0f * k3 + k4);
    }
    float raw_new = fminf(fmaxf(n, s_lo), s_hi);                                         // :499-503, :424-428
    float o_new = 2.0f * (raw_new - s_lo) / s_span - 1.0f;                               // :131, :429
    // cstr_reward: every lane evaluates both kinds of term on ITS component, lane 0 picks C2's and T1's, T2's (:288-291, :331-341)
    const float X = s_lo + (o_new + 1.0f) * s_span / 2.0f;
    const float ne = fabsf(X - k.target_c2) / k.conc_span;
    const float conc_l = -5.0f * (ne * ne) - 2.0f * ne;
    float pen_l = 0.0f;
    if (X < 280.0f) pen_l = 0.2f * ((280.0f - X) / 280.0f);
    else if (X > 350.0f) pen_l = 0.5f * ((X - 350.0f) / 350.0f);
    const float conc = quad_perm<2, 2, 2, 2>(conc_l), pen1 = quad_perm<1, 1, 1, 1>(pen_l), pen2 = quad_perm<3, 3, 3, 3>(pen_l);
    float tp = 0.0f;
    tp -= pen1;
    tp -= pen2;
    float r = 1.0f * conc + 0.5f * tp;                                                   // :432
    int32_t st = in.st + 1;                                                              // :396
    bool trunc = st >= k.max_steps;                                                      // :438
    if (bad) {  // _dynamics raises (:466-467) -> step returns the old state, -10, truncated (:415-421)
        o_new = o; raw_new = s; r = -10.0f; trunc = true;
    }
    const bool d = trunc;

    // ring row (off_policy_algorithm.py:477-496): a lane per component
    const int64_t e = row + i;
    if (own) {
        store_ring_f32(ring.obs, e * D + jj, o);
        store_ring_f32(ring.next_obs, e * D + jj, o_new);
        if (L == 1) {
            store_ring_f32(ring.obs, e * D + 4 + jj, in.oraw);
            store_ring_f32(ring.next_obs, e * D + 4 + jj, raw_new);
        }
        if (j < 2) store_ring_f32(ring.act, e * 2 + j, sa);
    }
    if (lane0) {
        store_ring_f32(ring.rew, e, r);
        store_ring_f32(ring.done, e, d ? 1.0f : 0.0f);
        store_ring_f32(ring.timeout, e, trunc ? 1.0f : 0.0f);
        if (c.reward_out) c.reward_out[i] = r;
        if (c.done_out) c.done_out[i] = d ? 1.0f : 0.0f;
        if (c.ep_return) {  // Monitor semantics: return/length of the episode that ends here
            const float ret = in.ep_ret + r;
            c.ep_return[i] = d ? 0.0f : ret;
            if (d) {
                atomicAdd(c.ep_stats + 0, 1.0);
                atomicAdd(c.ep_stats + 1, (double)ret);
                atomicAdd(c.ep_stats + 2, (double)st);
            }
        }
        c.step_count[i] = d ? 0 : st;
    }
    // env state for the next iteration (dummy_vec_env.py:68-72)
    if (d) {
        if (lane0) {
            if (c.reset_obs) {
                copy_obs<L>(c.env_obs, i, c.reset_obs, i);
            } else {
                uint64_t pst[4];
                load_pcg(c.pcg, i, pst);
                float ro[2][4];
                reset_draw_env<L>(k, pst, c.static_init, i, ro);
                store_obs<L>(c.env_obs, i, ro);
                *reinterpret_cast<ulonglong2 *>(c.pcg + 4 * i) = make_ulonglong2(pst[0], pst[1]);
            }
        }
    } else if (own) {
        c.env_obs[i * D + jj] = o_new;
        if (L == 1) c.env_obs[i * D + 4 + jj] = raw_new;
    }
}

}  // namespace

// ---- host-side argument checks shared by the entry points that run the collect step ---------------------------------

static int layout_of(int obs_dim, int act_dim)
{
    if (obs_dim == 4 && act_dim == 2) return 0;
    if (obs_dim == 8 && act_dim == 2) return 1;
    if (obs_dim == 8 && act_dim == 4) return 2;
    return -1;
}

static int check_ring(const cstr_ring_t *r)
{
    if (!r || !r->obs || !r->next_obs || !r->act || !r->rew || !r->done || !r->timeout || r->rows <= 0 || r->n_envs <= 0)
        return CSTR_E_BADARG;
    if (layout_of(r->obs_dim, r->act_dim) < 0) return CSTR_E_UNSUPPORTED;
    if (!aligned16(r->obs) || !aligned16(r->next_obs) || !(r->act_dim == 4 ? aligned16(r->act) : aligned8(r->act))) return CSTR_E_BADARG;
    return CSTR_OK;
}

// Validates the operands of the fused collect step (cstr_collect_step_f32's contract) and fills `c`.
static inline int make_collect_args(const cstr_coef_t *coef, int integrator, const cstr_ring_t *ring, float *env_obs, int32_t *step_count,
                                    int squashed, const float *act_low, const float *act_high, const float *noise, const float *reset_obs,
                                    uint64_t *pcg_state, double *static_init, float *reward_out, float *done_out, float *ep_return,
                                    double *ep_stats, CollectArgs &c)
{
    int rc = check_ring(ring);
    if (rc) return rc;
    if (!coef || !env_obs || !step_count || !act_low || !act_high) return CSTR_E_BADARG;
    if ((reset_obs == nullptr) == (pcg_state == nullptr)) return CSTR_E_BADARG;  // exactly one reset source
    if (static_init && !pcg_state) return CSTR_E_BADARG;
    if ((ep_return == nullptr) != (ep_stats == nullptr)) return CSTR_E_BADARG;
    if (integrator != CSTR_INTEGRATOR_EULER && integrator != CSTR_INTEGRATOR_RK4) return CSTR_E_UNSUPPORTED;
    const int A = ring->act_dim;
    const bool n_ok = !noise || (A == 4 ? aligned16(noise) : aligned8(noise));
    if (!aligned16(env_obs) || !n_ok || (reset_obs && !aligned16(reset_obs)) || (pcg_state && !aligned16(pcg_state))) return CSTR_E_BADARG;
    ActBounds ab;
    for (int j = 0; j < 4; ++j) {
        ab.lo[j] = j < A ? act_low[j] : -1.0f;
        ab.hi[j] = j < A ? act_high[j] : 1.0f;
        if (!(ab.hi[j] > ab.lo[j])) return CSTR_E_BADARG;
    }
    c = CollectArgs{*ring, env_obs, step_count, squashed, ab, noise, reset_obs, pcg_state, static_init, reward_out, done_out, ep_return,
                    ep_stats};
    return CSTR_OK;
}
