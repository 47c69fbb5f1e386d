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

#include "cstr_env_device.h"

namespace {

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

// Fused collect step: action scaling chain + env step + auto-reset + ring row write, one pass (collect_env_lane).
// Algorithmic HBM traffic per env (D = 4): 16 B state + 8 B action + 4 B step read; 52 B ring row
// + 16 B state + 4 B step written  => 104 B / env-step.
template <int L, int INTEG>
__global__ void collect_step_kernel(const cstr_coef_t k, const CollectArgs c, int64_t *__restrict__ ring_ctl,
                                    const float *__restrict__ policy_out, uint64_t *__restrict__ policy_rng_ctl,
                                    const uint64_t policy_rng_advance)
{
    const int64_t n = c.ring.n_envs;
    const int64_t pos = ring_ctl[0];  // wave-uniform scalar load
    const int64_t row = pos * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float u[4];
        CollectIn in;
        load_act<Lay<L>::A>(policy_out, i, u);
        collect_env_load<L>(c, i, in);
        collect_env_lane<L, INTEG>(k, c, row, i, u, in);
    }
    ring_advance_last_block(ring_ctl, c.ring.rows, policy_rng_ctl, policy_rng_advance);
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
    CollectArgs c;
    int rc = make_collect_args(coef, integrator, ring, env_obs, step_count, squashed, act_low, act_high, noise, reset_obs, pcg_state,
                               static_init, reward_out, done_out, ep_return, ep_stats, c);
    if (rc) return rc;
    if (!ring_ctl || !policy_out || !(ring->act_dim == 4 ? aligned16(policy_out) : aligned8(policy_out))) return CSTR_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int layout = layout_of(ring->obs_dim, ring->act_dim);
    int block, grid;
    env_launch_shape(ring->n_envs, block, grid);
    DISPATCH_L_INTEG(collect_step_kernel, *coef, c, ring_ctl, policy_out, policy_rng_ctl, policy_rng_advance);
    return (int)hipGetLastError();
}
