/*
 * include/cstr_rl_hip.h -- C ABI of libcstr_rl_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the hot path of CHAINNEVERLIU/Pytorch-RL-EnhancedStableBaselines named by
 * BASELINE.json:north_star: core.{SAC,TD3,MADDPG}("MlpPolicy", env).learn() on the two-series
 * CSTR environment. The reference is 100 % Python and has no FFI of its own, so
 * each entry point below names the Python statements it replaces (file:line relative to the
 * reference root). INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers (HBM) + sizes; no torch / C++ types. `stream` is a hipStream_t
 *     passed as void* (torch: torch.cuda.current_stream().cuda_stream).
 *   - every function only enqueues work on `stream`: no allocation, no host sync, graph-capturable.
 *   - return value: 0 = ok; < 0 = cstr error (CSTR_E_*); > 0 = hipError_t of the failed launch.
 *   - all floating point is IEEE fp32, compiled with -ffp-contract=off; fused ops are explicit.
 *   - control words that change every call (ring position, Adam step) live in HBM (`*_ctl`), so a
 *     captured hipGraph replays correctly; the last workgroup of a launch advances them.
 */
#ifndef CSTR_RL_HIP_H
#define CSTR_RL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSTR_ABI_VERSION 5

#define CSTR_OK 0
#define CSTR_E_BADARG (-1)      /* null pointer / non-positive size / misaligned buffer */
#define CSTR_E_UNSUPPORTED (-2) /* (obs_dim, act_dim) not in {(4,2), (8,2), (8,4)}, batch too large, 64-bit index range */

#define CSTR_INTEGRATOR_EULER 0 /* the reference: forward Euler, dt = 0.1 (twoseriescstr.py:493-496) */
#define CSTR_INTEGRATOR_RK4 1   /* north_star's ask; no reference counterpart */

typedef void *cstr_stream_t; /* hipStream_t */

/* f32-folded coefficients of TwoSeriesCSTREnv (twoseriescstr.py:37-61, :99, :103-106), folded with
 * NumPy >= 2 (NEP 50) semantics: Python-float products are formed in double, then rounded to f32
 * at their first contact with an f32 operand. Passed BY VALUE to the kernels (kernarg -> SGPRs). */
typedef struct cstr_coef {
    float q_v1, q_v2, cf, tf, tcf, k0, neg_e, r_gas, hk, rho_cp, cool1, cool2, neg_ua1, neg_ua2, rho_c, c_pc, dt;
    float s_lo[4], s_hi[4], s_span[4]; /* raw_state_low/high, high-low    (:56-57) */
    float a_lo[2], a_hi[2], a_span[2]; /* raw_action_low/high, high-low   (:60-61) */
    float target_c2, conc_span;        /* target_C2, max_conc - min_conc  (:103-106, :290) */
    int32_t max_steps;                 /* 400 (:99) */
} cstr_coef_t;

/* Replay ring in HBM, field-level SoA with the reference's array shapes (core/common/buffers.py:212-234):
 * observations/next_observations [rows][n_envs][obs_dim], actions [rows][n_envs][act_dim],
 * rewards/dones/timeouts [rows][n_envs]; all f32; obs rows 16-byte aligned. */
typedef struct cstr_ring {
    float *obs, *next_obs, *act, *rew, *done, *timeout;
    int64_t rows, n_envs;
    int32_t obs_dim, act_dim; /* (4,2) | (8,2) | (8,4) */
} cstr_ring_t;

/* ring_ctl: int64[4] in HBM = { pos, full, ticket, adds }  (buffers.py:101-104 `pos`, `full`) */
#define CSTR_RING_CTL_WORDS 4
/* adam_ctl: 4 x 64-bit words in HBM = { int64 step, int64 ticket, double beta1^step, double beta2^step }
 * (torch.optim.Adam state["step"]; the powers are the running products the bias corrections need) */
#define CSTR_ADAM_CTL_WORDS 4
/* mt_state: uint32[628] in HBM = { key[624], pos, has_gauss, gauss (f64, lo word first) } (numpy legacy RandomState) */
#define CSTR_MT_STATE_WORDS 628
#define CSTR_MAX_NOISE_PERIOD 8
/* pcg_state: uint64[4] per env in HBM = { state_hi, state_lo, inc_hi, inc_lo } (numpy PCG64) */
#define CSTR_PCG_STATE_WORDS 4
#define CSTR_MAX_SAMPLE_BATCH 16384

int cstr_abi_version(void);
const char *cstr_error_string(int code);

/* Host helper: fold the class constants of TwoSeriesCSTREnv (twoseriescstr.py:37-61) and the ctor
 * arguments default_target / min_concentration / max_concentration (:63-68) into f32 coefficients. */
void cstr_default_coef(cstr_coef_t *coef, double target_c2, double min_conc, double max_conc, int32_t max_steps);

/* VecEnv.step for N CSTR envs in one launch. Replaces the Python loop of
 * DummyVecEnv.step_wait (core/common/vec_env/dummy_vec_env.py:56-73) over
 * TwoSeriesCSTREnv.step/_dynamics/compute_reward (twoseriescstr.py:394-503, :271-392).
 *   (obs_dim, act_dim) selects the layout: (4,2) the reference's observation; (8,2) [normalised | raw];
 *   (8,4) TWO reactor trains side by side -- [train A | train B] normalised, actions [F1A,F2A,F1B,F2B], reward rA + rB,
 *   one step counter and one reset stream per env: the 8-obs / 4-act environment a 4-agent MADDPG needs.
 *   obs        [N][obs_dim] in : current observations
 *   act        [N][act_dim] in : env actions (normalised, clipped to [-1,1] inside, :399)
 *   step_count [N]          i/o: TwoSeriesCSTREnv.current_step
 *   reset_obs  [N][obs_dim] in : observation an env restarts from when it finishes this step
 *   next_obs   [N][obs_dim] out: true next observation (= infos[i]["terminal_observation"] when done)
 *   obs_after  [N][obs_dim] out: what VecEnv.step returns (reset obs when done); may alias obs
 *   reward/done/timeout [N] out: f32; timeout = infos[i]["TimeLimit.truncated"] */
int cstr_vec_step_f32(const cstr_coef_t *coef, int integrator, int obs_dim, int act_dim, const float *obs, const float *act,
                      int32_t *step_count, const float *reset_obs, float *next_obs, float *obs_after, float *reward,
                      float *done, float *timeout, int64_t n_envs, cstr_stream_t stream);

/* TwoSeriesCSTREnv.reset draws for envs with mask[i] != 0 (mask NULL = all) from per-env numpy-PCG64 states that the
 * host seeds exactly like gymnasium.utils.seeding.np_random(seed + i) (twoseriescstr.py:162):
 *   static_init NULL: init_mode="random", generate_initial_state (twoseriescstr.py:187-224, :267);
 *   static_init double[N][4 per train] i/o: init_mode="static" (:94-96, :246-255) -- the env's f64 `init_state`
 *   (constructed as {0.45, 310, 0.25, 290}) takes an in-place uniform step at every reset, like the reference's. */
int cstr_reset_draw_f32(uint64_t *pcg_state, const uint8_t *mask, double *static_init, int obs_dim, int act_dim,
                        float *obs_out, int64_t n_envs, cstr_stream_t stream);

/* ReplayBuffer.add (core/common/buffers.py:247-283) at the device-resident ring position; the last
 * workgroup advances ring_ctl (pos, full). */
int cstr_replay_add_f32(const cstr_ring_t *ring, int64_t *ring_ctl, const float *obs, const float *next_obs,
                        const float *act, const float *rew, const float *done, const float *timeout,
                        cstr_stream_t stream);

/* One fused collect step = _sample_action's scaling chain + VecEnv.step + _store_transition + ReplayBuffer.add
 * (core/common/off_policy_algorithm.py:396-406, :564, :477-496; buffers.py:247-283) in ONE pass over the envs:
 * reads state + policy output once, writes the ring row once, updates the env state in place.
 *   env_obs    [N][obs_dim] i/o: VecEnv state (_last_obs); replaced by the post-reset observation
 *   policy_out [N][act_dim] in : actor output. `squashed` is a bit field: bit 0 set = tanh output in [-1,1] that
 *                                predict() first unscales (core/common/policies.py:375), clear = an action already in
 *                                [low, high] (warm-up sample); bit 1 set = the multi-agent algorithms' behaviour
 *                                (core/common/multiagent_policy_algorithm.py:369, :391-392): no scale/unscale round
 *                                trip and no noise, buffer_action = env action = that value
 *   act_low/act_high [act_dim] in : HOST pointers, bounds of the algorithm-facing action space
 *   noise [N][act_dim] or NULL : added to the scaled action, then clip [-1,1] (off_policy_algorithm.py:401-402)
 *   reset_obs  [N][obs_dim] or NULL, pcg_state [N][4] or NULL: reset source (exactly one non-NULL)
 *   static_init double[N][4 per train] or NULL: with pcg_state, init_mode="static" (see cstr_reset_draw_f32)
 *   reward_out/done_out [N] or NULL: per-env copies (what VecEnv.step would have returned)
 *   ep_return [N] + ep_stats double[4] = {episodes, sum of returns, sum of lengths, -} or both NULL: device-side
 *                                episode statistics (Monitor's info["episode"], core/common/monitor.py:96-109; the
 *                                counters behind _episode_num / rollout/ep_rew_mean) accumulated without a host sync */
int cstr_collect_step_f32(const cstr_coef_t *coef, int integrator, const cstr_ring_t *ring, int64_t *ring_ctl,
                          float *env_obs, int32_t *step_count, const float *policy_out, int squashed,
                          const float *act_low, const float *act_high, const float *noise, const float *reset_obs,
                          uint64_t *pcg_state, double *static_init, float *reward_out, float *done_out, float *ep_return,
                          double *ep_stats, cstr_stream_t stream);
/* The same launch additionally advancing a Philox stream offset: policy_rng_ctl (may be NULL) is the control block of the
 * cstr_policy_rows_fwd_f32 launch that produced policy_out with cstr_policy_mlp_t.reserved bit 0 set ("the caller advances the
 * offset"); this launch's last workgroup adds policy_rng_advance (= that launch's row count) to its offset word -- the rollout's
 * _sample_action (core/common/off_policy_algorithm.py:364-411) and env step then share ONE control-word ticket. */
int cstr_collect_step_rng_f32(const cstr_coef_t *coef, int integrator, const cstr_ring_t *ring, int64_t *ring_ctl,
                              float *env_obs, int32_t *step_count, const float *policy_out, int squashed,
                              const float *act_low, const float *act_high, const float *noise, const float *reset_obs,
                              uint64_t *pcg_state, double *static_init, float *reward_out, float *done_out, float *ep_return,
                              double *ep_stats, uint64_t *policy_rng_ctl, uint64_t policy_rng_advance, cstr_stream_t stream);

/* np.random.seed(seed) for the device-resident legacy MT19937 state (core/common/utils.py:46;
 * twoseriescstr.py:164 reseeds the same global stream). */
int cstr_mt19937_seed(uint32_t *mt_state, uint32_t seed, cstr_stream_t stream);

/* ---- VecNormalize (core/common/vec_env/vec_normalize.py:174-290; core/common/running_mean_std.py:34-55) ----
 * vn_state: double[CSTR_VECNORM_STATE_WORDS] in HBM = { obs mean[8], obs var[8], obs count, ret mean, ret var, ret count }
 * (the reference's obs_rms / ret_rms, f64); returns: double[N] discounted returns (VecNormalize.returns). */
#define CSTR_VECNORM_STATE_WORDS 20
typedef struct {
    int32_t training, norm_obs, norm_reward, obs_dim; /* obs_dim <= 8 */
    double clip_obs, clip_reward, gamma, epsilon;
} cstr_vecnorm_cfg_t;

/* RunningMeanStd.__init__ for both statistics: mean 0, var 1, count 1e-4 (running_mean_std.py:5-15). */
int cstr_vecnorm_init_f64(double *vn_state, cstr_stream_t stream);

/* VecNormalize.step_wait (vec_normalize.py:174-204) on the raw outputs of the inner VecEnv, in the reference's order:
 * obs_rms.update(obs) [training && norm_obs], norm_obs_out = clip((obs - mean) / sqrt(var + eps)), returns = returns*gamma
 * + reward and ret_rms.update(returns) [training], norm_reward_out = clip(reward / sqrt(ret var + eps)), returns[done] = 0.
 * reward == NULL is VecNormalize.reset (:291-307): observation part only and returns = 0.
 *   obs [N][obs_dim] f32 raw; reward/done [N] f32 or NULL; norm_obs_out [N][obs_dim] / norm_reward_out [N] or NULL. */
int cstr_vecnorm_step_f64(const cstr_vecnorm_cfg_t *cfg, double *vn_state, double *returns, const float *obs,
                          const float *reward, const float *done, float *norm_obs_out, float *norm_reward_out,
                          int64_t n_envs, cstr_stream_t stream);

/* ReplayBuffer._get_samples(env=VecNormalize) (core/common/buffers.py:143-155, :312-323): normalize_obs on the sampled
 * observations / next_observations [B][obs_dim] and normalize_reward on rewards [B], in place, with the current
 * statistics. Each of obs / next_obs / reward may be NULL (at least one is not). */
int cstr_vecnorm_apply_f32(const cstr_vecnorm_cfg_t *cfg, const double *vn_state, float *obs, float *next_obs,
                           float *reward, int64_t batch, cstr_stream_t stream);

/* Exploration noise from the SAME legacy stream as the replay sampler: the reference's
 * VectorizedActionNoise.__call__ -> n_envs x NormalActionNoise.__call__ = np.random.normal(mu, sigma).astype(float32)
 * (core/common/noise.py:44-45, :141-142; OU noise draws np.random.normal(size=) the same way, :85-89).
 * out[j] = (float)(loc[j % period] + scale[j % period] * legacy_gauss()), j < count, in numpy's draw order (polar
 * Box-Muller, second deviate first, odd tail cached in mt_state). loc / scale: HOST double[period], period <= 8. */
int cstr_mt19937_normal_f32(uint32_t *mt_state, const double *loc, const double *scale, int32_t period, float *out,
                            int64_t count, cstr_stream_t stream);
/* The same draws kept in double precision: np.random.normal(size=...) as OrnsteinUhlenbeckActionNoise.__call__ uses it
 * (core/common/noise.py:85-89) before its own float64 arithmetic. */
int cstr_mt19937_normal_f64(uint32_t *mt_state, const double *loc, const double *scale, int32_t period, double *out,
                            int64_t count, cstr_stream_t stream);

/* ReplayBuffer.sample (core/common/buffers.py:106-115, :285-325): upper = rows if full else pos;
 * batch_inds = np.random.randint(0, upper, batch); env_indices = np.random.randint(0, n_envs, batch)
 * (legacy MT19937, masked rejection, bit-exact) fused with the gather of the five fields.
 *   out_obs [B][obs_dim], out_act [B][act_dim], out_next_obs [B][obs_dim], out_done [B] (= dones*(1-timeouts)),
 *   out_rew [B]; out_row_idx/out_env_idx int64 [B] or NULL. */
int cstr_replay_sample_mt19937_f32(const cstr_ring_t *ring, const int64_t *ring_ctl, uint32_t *mt_state, int64_t batch,
                                   float *out_obs, float *out_act, float *out_next_obs, float *out_done,
                                   float *out_rew, int64_t *out_row_idx, int64_t *out_env_idx, cstr_stream_t stream);

/* The same draw, gathered straight into the critics' input rows -- ReplayBuffer.sample + the th.cat([obs, actions], dim=1)
 * of ContinuousCritic.forward (core/common/policies.py:975-981) without the cat launches. Row width W = obs_dim + act_dim:
 *   x_data [B][W] <- (observations | actions)          the critic input of the Bellman error
 *   x_next [B][W] <- (next_observations | untouched)   the target critic's input; the actor head writes the action columns
 *   x_pi   [B][W] <- (observations | untouched)        the critic input of the actor loss; may be NULL */
int cstr_replay_sample_packed_mt19937_f32(const cstr_ring_t *ring, const int64_t *ring_ctl, uint32_t *mt_state, int64_t batch,
                                          float *x_data, float *x_next, float *x_pi, float *out_done, float *out_rew,
                                          int64_t *out_row_idx, int64_t *out_env_idx, cstr_stream_t stream);

/* The gather half of that call for index pairs drawn earlier in the iteration (cstr_rollout_step_f32): sample_idx int32
 * [2][batch] = { batch_inds, env_indices } (core/common/buffers.py:113, :309). ONE workgroup, which also performs the
 * control-word updates the rollout launch leaves to its successor: advance_ring != 0 = ReplayBuffer.add's epilogue on
 * ring_ctl (pos, full, adds; core/common/buffers.py:280-283), rng_ctl (may be NULL) = the rollout policy's Philox control
 * block, whose offset grows by rng_advance. Both happen after the gather, so the gather reads the ring by the given indices only. */
int cstr_replay_gather_packed_f32(const cstr_ring_t *ring, int64_t *ring_ctl, int advance_ring, uint64_t *rng_ctl,
                                  uint64_t rng_advance, const int32_t *sample_idx, int64_t batch, float *x_data, float *x_next,
                                  float *x_pi, float *out_done, float *out_rew, int64_t *out_row_idx, int64_t *out_env_idx,
                                  cstr_stream_t stream);

/* ReplayBuffer.sample's gather FUSED INTO ITS FIRST CONSUMER: y [M][n] = act(x W^T + b) where the input rows x are the sampled
 * transitions themselves, read from the ring by the index pairs of cstr_rollout_step_f32 (sample_idx as in
 * cstr_replay_gather_packed_f32). both != 0: M = 2 batch, rows [0, batch) = observations, rows [batch, 2 batch) = next
 * observations (SAC's two actor passes of a gradient step as one 2B-row pass, core/sac/sac.py:222, :247); both == 0: M = batch
 * rows of next observations (a target actor's first layer, core/td3/td3.py:173). w [n][obs_dim], bias [n]; act 0 none / 1 ReLU /
 * 2 Tanh. The launch also writes the packed batch of cstr_replay_sample_packed_mt19937_f32 (x_data, x_next, x_pi or NULL,
 * out_done, out_rew) for the launches behind it and performs the control-word updates of cstr_replay_gather_packed_f32
 * (advance_ring, rng_ctl / rng_advance): one launch and one dependent launch boundary less per gradient step. Results are
 * bit-identical to cstr_replay_gather_packed_f32 followed by cstr_linear_act_fwd_f32. */
int cstr_linear_act_fwd_gather_f32(const cstr_ring_t *ring, int64_t *ring_ctl, int advance_ring, uint64_t *rng_ctl, uint64_t rng_advance,
                                   const int32_t *sample_idx, int64_t batch, int both, const float *w, const float *bias, int act, float *y,
                                   int64_t n, float *x_data, float *x_next, float *x_pi, float *out_done, float *out_rew,
                                   cstr_stream_t stream);

/* Target-Q: SAC core/sac/sac.py:250-254 (logp, ent_coef non-NULL), TD3 core/td3/td3.py:174-176 (both NULL):
 * out = rew + (1 - done) * gamma * (min(q1, q2) - ent_coef[0] * logp). ent_coef is a DEVICE scalar. */
int cstr_td_target_min_f32(const float *q1, const float *q2, const float *logp, const float *rew, const float *done,
                           const float *ent_coef, float gamma, float *out, int64_t n, cstr_stream_t stream);

/* polyak_update over one flat parameter arena (core/common/utils.py:457-481):
 * target = fma(tau, param, target * (1 - tau)), bit-identical to torch's mul_ + add(alpha=). */
int cstr_polyak_f32(const float *param, float *target, double tau, int64_t n, cstr_stream_t stream);

/* torch.optim.Adam step (defaults of core/common/policies.py:96-117) over one flat arena; the step
 * counter lives in adam_ctl and is advanced by the last workgroup. lr is a DEVICE scalar (f64) so
 * _update_learning_rate (core/common/base_class.py:303-317) needs no re-capture. */
int cstr_adam_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t *adam_ctl,
                  const double *lr, double beta1, double beta2, double eps, float grad_scale, int64_t n,
                  cstr_stream_t stream);

/* Several flat-arena updates in ONE launch (SAC: the entropy coefficient -- one parameter -- next to the critic,
 * core/sac/sac.py:240-243 and :266-268; the actor's step next to the critic's soft target update, :281 and :284-287). A
 * segment is a cstr_adam_f32 call's argument list, or -- polyak_source != NULL -- a cstr_polyak_f32 call with param = the
 * TARGET arena (the other Adam fields are ignored). Segments must not depend on each other. */
#define CSTR_MAX_ADAM_SEGS 4
typedef struct {
    float *param; const float *grad; float *exp_avg; float *exp_avg_sq; int64_t *adam_ctl; const double *lr;
    double beta1, beta2, eps; float grad_scale; int64_t n;
    const float *polyak_source; double tau;
    /* optional (NULL = none): a tile-major shadow copy (cstr_policy_swizzle_f32) of the [shadow_n][shadow_k] weight matrix that
     * starts shadow_begin floats into `param` -- rewritten by the same threads that update those weights, so the policy
     * kernel's copy never lags the parameters. shadow_begin and shadow_k are multiples of 4. */
    float *shadow; int64_t shadow_begin, shadow_n, shadow_k;
    /* optional (NULL = none), Adam segments only: the soft-update target OF THESE PARAMETERS. The thread that has just computed
     * param_new[i] also writes own_target[i] = fma(tau, param_new[i], own_target[i] * (1 - tau)) -- a cstr_polyak_f32(param,
     * own_target, tau) that would otherwise need its own launch BEHIND this one (TD3 / MADDPG: the actor's step followed by the
     * actor target's soft update, core/td3/td3.py:199-205, core/maddpg/maddpg.py:179-185). Uses the segment's `tau`. */
    float *own_target;
} cstr_adam_seg_t;
int cstr_adam_multi_f32(const cstr_adam_seg_t *segs, int n_segs, cstr_stream_t stream);

/* ---- learner glue around the PyTorch-ROCm GEMMs (csrc/cstr_mlp.hip) --------------------------------------------- */

#define CSTR_ACT_NONE 0
#define CSTR_ACT_RELU 1
#define CSTR_ACT_TANH 2

/* nn.Linear's bias add + the activation create_mlp puts behind it (core/common/torch_layers.py:110-183), in place on
 * the GEMM output. Grouped form for batched GEMMs (twin critics): y [groups][m][n], bias [groups][n]. */
int cstr_bias_act_fwd_f32(float *y, const float *bias, int act, int64_t groups, int64_t m, int64_t n, cstr_stream_t stream);

/* Backward of the same: gz = gy * act'(y) and gbias[n] = sum_m gz[m][n] (autograd's threshold/tanh backward + the
 * bias gradient's batch sum). gbias may be NULL; with act == NONE and gz == gy nothing is copied. */
int cstr_bias_act_bwd_f32(const float *gy, const float *y, int act, float *gz, float *gbias, int64_t groups, int64_t m, int64_t n,
                          cstr_stream_t stream);
/* The same for ONE group whose gy / y are column blocks of wider row-major matrices (row strides ldg / ldy >= n): the action columns of
 * d(loss)/d(critic input) and of the critic input itself in the deterministic actors' backward (core/td3/td3.py:194-200,
 * core/maddpg/maddpg.py:167-185) -- no gather copies. gz is contiguous [m][n]. */
int cstr_bias_act_bwd_rows_f32(const float *gy, int64_t ldg, const float *y, int64_t ldy, int act, float *gz, float *gbias, int64_t m,
                               int64_t n, cstr_stream_t stream);

/* nn.Linear + the activation create_mlp puts behind it (core/common/torch_layers.py:110-183), forward, in ONE launch for the
 * learners' small shapes: y[g][m][n] = act(sum_k x[g][m][k] * w[g][n][k] + bias[g][n]) on the f32 matrix cores (exact f32 fma
 * chains). x rows are ldx floats apart and groups x_group_stride floats apart (0 = every group reads the same input); w
 * [groups][n][k], bias [groups][n], y [groups][m][n] contiguous. The backward keeps the rocBLAS GEMMs + cstr_bias_act_bwd_f32. */
int cstr_linear_act_fwd_f32(const float *x, int64_t x_group_stride, int64_t ldx, const float *w, const float *bias, int act, float *y,
                            int64_t groups, int64_t m, int64_t n, int64_t k, cstr_stream_t stream);

/* cstr_linear_act_fwd_f32 for up to 8 INDEPENDENT layers of one shape in one launch (every agent's actor layer in MADDPG /
 * IDDPG: core/maddpg/policies.py builds one MLP per agent; the reference evaluates them one after the other). Each set:
 * y[m][n] = act(x[m] . w[n] + bias[n]) with x rows ldx apart and y rows ldy apart (y may be the agent's column block of the
 * joint action). */
#define CSTR_MAX_LINEAR_SETS 16
typedef struct { const float *x; int64_t ldx; const float *w; const float *bias; float *y; int64_t ldy; } cstr_linear_set_t;
int cstr_linear_act_fwd_sets_f32(const cstr_linear_set_t *sets, int n_sets, int act, int64_t m, int64_t n, int64_t k,
                                 cstr_stream_t stream);

/* Backward of a Linear w.r.t. its input, fused with the activation gradient of the layer BELOW it (autograd's mm backward +
 * threshold/tanh backward of core/common/torch_layers.py:110-183's Linear -> ReLU -> Linear):
 *   dz[g][m][k] = (sum_n gz[g][m][n] * w[g][n][k]) * act'(y[g][m][k])
 * gz: gradient w.r.t. this layer's pre-activation [groups][m][n]; w [groups][n][k]; y [groups][m][k]: the lower layer's
 * OUTPUT (this layer's input), NULL with act == NONE; all contiguous. sum_groups != 0: the groups share ONE input
 * (stacked critics): dz [m][k] = sum_g gz[g] @ w[g] (and y [m][k]) -- the batched product and the sum over groups at once. */
int cstr_linear_bwd_input_f32(const float *gz, const float *w, const float *y, int act, float *dz, int64_t groups, int sum_groups,
                              int64_t m, int64_t n, int64_t k, cstr_stream_t stream);

/* Weight and bias gradient of a Linear in one launch (autograd's mm backward for the weight + the bias sum over the batch):
 *   dw[g][n][k] = sum_m dz[g][m][n] * x[g][m][k],   db[g][n] = sum_m dz[g][m][n]   (db may be NULL)
 * dz [groups][m][n] contiguous (gradient w.r.t. the pre-activation); x rows ldx floats apart, groups x_group_stride apart
 * (0 = shared input); dw [groups][n][k] and db [groups][n] contiguous -- views of the gradient arena. */
int cstr_linear_bwd_weight_f32(const float *dz, const float *x, int64_t x_group_stride, int64_t ldx, float *dw, float *db,
                               int64_t groups, int64_t m, int64_t n, int64_t k, cstr_stream_t stream);

/* The same for up to CSTR_MAX_LINEAR_SETS independent Linears (each with its own shape) in ONE launch: the parameter gradients
 * of an MLP's layers once its backward chain has produced every dz. db may be NULL per set. */
typedef struct cstr_wgrad_set {
    const float *dz; /* [m][n] contiguous */
    const float *x;  /* [m][k], rows ldx floats apart */
    int64_t ldx;
    float *dw;       /* [n][k] */
    float *db;       /* [n] or NULL */
    int64_t m, n, k;
} cstr_wgrad_set_t;
int cstr_linear_bwd_weight_sets_f32(const cstr_wgrad_set_t *sets, int n_sets, cstr_stream_t stream);

/* Last hidden layer + scalar head of a Q network: create_mlp(..., output_dim = 1) (core/common/torch_layers.py:110-183;
 * ContinuousCritic.forward, core/common/policies.py:960-987) ends in y = act(z + b1), q = y . w2 + b2. The head is a
 * matrix-vector product, done in the epilogue of the previous GEMM: z [groups][m][k] is replaced by y IN PLACE and
 * q [groups][m] is written. b1, w2 [groups][k]; b2 [groups]. */
int cstr_hidden_head_fwd_f32(float *z, const float *b1, int act, const float *w2, const float *b2, float *q, int64_t groups,
                             int64_t m, int64_t k, cstr_stream_t stream);

/* Backward of the pair given gq = d(loss)/dq [groups][m]: dz = gq * w2 * act'(y) [groups][m][k] and, when the three
 * parameter-gradient pointers are given (all or none), gb1[k] = sum_m dz, gw2[k] = sum_m gq * y, gb2 = sum_m gq. */
int cstr_hidden_head_bwd_f32(const float *gq, const float *y, int act, const float *w2, float *dz, float *gb1, float *gw2,
                             float *gb2, int64_t groups, int64_t m, int64_t k, cstr_stream_t stream);

/* SquashedDiagGaussianDistribution (core/common/distributions.py:161-260) with SAC's log_std clamp
 * (core/sac/policies.py:20-22, :162-164): u = mean + exp(clamp(log_std_raw, -20, 2)) * eps, action = tanh(u),
 * logp = sum_j Normal.log_prob(u_j) - sum_j log(1 - action_j^2 + 1e-6). logp may be NULL (acting only).
 * in_stride = row stride of mean / log_std_raw: act_dim for separate tensors, 2*act_dim when they are the two halves of
 * one merged mu|log_std GEMM output. */
int cstr_squashed_gaussian_fwd_f32(const float *mean, const float *log_std_raw, const float *eps, float *action, float *logp,
                                   int64_t batch, int act_dim, int in_stride, cstr_stream_t stream);
/* Its analytic backward: (g_action rows of stride g_action_stride or NULL, g_logp [B] or NULL) -> g_mean, g_log_std_raw
 * (rows of stride in_stride). */
int cstr_squashed_gaussian_bwd_f32(const float *g_action, const float *g_logp, const float *action, const float *log_std_raw,
                                   const float *eps, float *g_mean, float *g_log_std_raw, int64_t batch, int act_dim,
                                   int in_stride, int g_action_stride, cstr_stream_t stream);

/* The SAC actor's merged (mu | log_std) head in one pass (core/sac/policies.py:162-175; core/common/distributions.py:161-260):
 * params [B][2A] = the head GEMM's output, gets `bias` [2A] added in place (NULL: already biased); eps [B][A] ~ N(0,1) is READ
 * when rng_ctl is NULL and DRAWN (and written, for the backward) when rng_ctl is given; action = tanh(mean + exp(clamp(
 * log_std)) * eps) with row stride `action_stride` (>= A: it may be a column block of the critic's input); logp [B] or NULL.
 * rng_ctl: uint64[CSTR_RNG_CTL_WORDS] in HBM = { seed, offset, ticket, -, sub-tickets[8] (cstr_policy_rows_fwd_f32), - }:
 * Philox4x32-10 counter RNG + Box-Muller, the offset advances by B per launch on the device (graph-replay safe). */
#define CSTR_RNG_CTL_WORDS 16
#define CSTR_MAX_HEAD_ACT 4
int cstr_gaussian_head_fwd_f32(float *params, const float *bias, float *eps, uint64_t *rng_ctl, float *action,
                               int64_t action_stride, float *logp, int64_t batch, int act_dim, cstr_stream_t stream);

/* The same head INCLUDING its Linear: params[b][j] = hidden[b] . w[j] + bias[j] (2A <= 8 dot products per row -- a matrix-
 * vector shaped product, one wave per row) followed by the sampling above. hidden rows ldh floats apart (k and ldh multiples
 * of 4, 16-byte aligned); w [2A][k], bias [2A]; params [B][2A] is an OUTPUT here (kept for the backward). */
int cstr_gaussian_head_gemm_fwd_f32(const float *hidden, int64_t ldh, const float *w, const float *bias, float *params, float *eps,
                                    uint64_t *rng_ctl, float *action, int64_t action_stride, float *logp, int64_t batch, int act_dim,
                                    int64_t k, cstr_stream_t stream);

/* Backward: g_params [B][2A] = d/d(mean | log_std_raw) from g_action (row stride ga_stride, or NULL) and g_logp ([B] or
 * NULL); g_bias [2A] = column sums over the batch (or NULL). */
int cstr_gaussian_head_bwd_f32(const float *g_action, int64_t ga_stride, const float *g_logp, const float *action,
                               int64_t action_stride, const float *params, const float *eps, float *g_params, float *g_bias,
                               int64_t batch, int act_dim, cstr_stream_t stream);

/* The same backward carried one layer further when the head's Linear (w [2A][width], core/sac/policies.py:100-104) sits on
 * a hidden layer with activation `act` (0 none, 1 ReLU, 2 Tanh) whose OUTPUT is `hidden` (rows ldh floats apart):
 * g_params as above and dz [B][width] = (g_params . w) * act'(hidden) = d(loss)/d(that layer's pre-activation), the input
 * of cstr_linear_bwd_weight_f32 / cstr_linear_bwd_input_f32. The head's dW / db: cstr_linear_bwd_weight_f32(g_params, hidden). */
int cstr_gaussian_head_bwd_input_f32(const float *g_action, int64_t ga_stride, const float *g_logp, const float *action,
                                     int64_t action_stride, const float *params, const float *eps, const float *w,
                                     const float *hidden, int64_t ldh, int act, float *g_params, float *dz, int64_t batch,
                                     int act_dim, int64_t width, cstr_stream_t stream);

/* A whole policy network for MANY rows, inference only (no activations kept, no backward): create_mlp(k0, ., [h1, h2]) with
 * activation `act` on both hidden layers (core/common/torch_layers.py:110-183) + the action head, ONE launch.
 * head 0: SAC's squashed Gaussian (core/sac/policies.py:147-175): w3 [2A][h2] = (mu | log_std) weights, b3 [2A]; the action
 *         (rows action_stride apart, e.g. the action columns of a critic input) = tanh(mu + exp(clamp(log_std)) * eps), eps
 *         read ([m][A]) or drawn from the Philox stream `rng_ctl` (same stream positions as cstr_gaussian_head_fwd_f32:
 *         counter = offset + row; the offset advances by m); logp [m] optional.
 * head 1: deterministic actor (core/td3/policies.py:75-78): w3 [A][h2], b3 [A], action = out_act(h2 w3^T + b3); eps, rng_ctl
 *         and logp must be NULL.
 * h1, h2 multiples of 4, k0 <= 256, 16 * (h1 + h2 + 8) floats of LDS <= 64 KB.
 * reserved: bit 0 set = the caller advances the Philox offset by m before the stream's next consumer
 * (cstr_collect_step_rng_f32 does): the launch then ends without its 256-workgroup "last one out" ticket; other bits 0.
 * With w2_swizzled and h1, h2 <= 512 the software-pipelined kernel runs. */
typedef struct cstr_policy_mlp {
    int32_t k0, h1, h2, act_dim;
    int32_t act, head, out_act, reserved;
    const float *w1, *b1; /* [h1][k0], [h1] */
    const float *w2, *b2; /* [h2][h1], [h2] */
    const float *w3, *b3; /* head */
    const float *w2_swizzled; /* NULL, or w2 in the tile-major layout of cstr_policy_swizzle_f32 (full-line loads) */
} cstr_policy_mlp_t;
/* Tile-major copy of a weight matrix w [n][k] (k a multiple of 4) for the matrix-core operand loads of
 * cstr_policy_rows_fwd_f32: out[((tile * ceil(k/16) + chunk) * 64 + lane) * 4 + e] = w[16 tile + (lane & 15)][16 chunk +
 * 4 (lane >> 4) + e], zeros outside the matrix; out holds ceil(n/16) * ceil(k/16) * 256 floats. A wave's operand load then
 * reads 1 KB of consecutive bytes instead of sixteen 64-byte row pieces. */
int cstr_policy_swizzle_f32(const float *w, int64_t n, int64_t k, float *out, cstr_stream_t stream);
int cstr_policy_rows_fwd_f32(const cstr_policy_mlp_t *net, const float *x, int64_t ldx, const float *eps, uint64_t *rng_ctl,
                             float *action, int64_t action_stride, float *logp, int64_t m, cstr_stream_t stream);

/* One vec-step of collect_rollouts in ONE launch (core/common/off_policy_algorithm.py:510-605 for a device-resident vec-env):
 * the policy network + sampling of cstr_policy_rows_fwd_f32 on x [n_envs][k0] (the policy's view of the observations: env_obs
 * itself, or its normalised image), then -- on the sampling tail's lanes, the action never leaving registers -- the fused
 * collect step of cstr_collect_step_f32 for the workgroup's 16 envs (same operands, same arithmetic), and, when mt_state is
 * given, ReplayBuffer.sample's two index draws for the gradient step behind it (core/common/buffers.py:112-113, :309; numpy
 * legacy MT19937, masked rejection, bit-exact) on one otherwise idle wave: sample_idx int32 [2][batch] = { batch_inds,
 * env_indices }, drawn against the ring AS THE ADD OF THIS LAUNCH LEAVES IT (upper = rows if full else pos + 1), mt_state
 * advanced. No control word is written here: ring_ctl is only read (the row goes to position pos), the Philox offset of
 * rng_ctl stays; the next launch must be cstr_replay_gather_packed_f32(advance_ring = 1, rng_ctl, rng_advance = n_envs), or
 * the caller advances them another way. Needs the tile-major W2 copy (net->w2_swizzled), hidden widths <= 512, k0 <= 16 in
 * 16-byte-aligned rows (CSTR_E_UNSUPPORTED otherwise: use the separate launches). head 0: rng_ctl required; head 1: NULL.
 * action_out [n_envs][act_dim] or NULL: the policy output (what cstr_policy_rows_fwd_f32 would have written). */
int cstr_rollout_step_f32(const cstr_policy_mlp_t *net, const float *x, int64_t ldx, uint64_t *rng_ctl, const cstr_coef_t *coef,
                          int integrator, const cstr_ring_t *ring, const int64_t *ring_ctl, float *env_obs, int32_t *step_count,
                          int squashed, const float *act_low, const float *act_high, const float *noise, const float *reset_obs,
                          uint64_t *pcg_state, double *static_init, float *reward_out, float *done_out, float *ep_return,
                          double *ep_stats, float *action_out, uint32_t *mt_state, int64_t batch, int32_t *sample_idx,
                          cstr_stream_t stream);

/* TD3 / MADDPG target policy smoothing (core/td3/td3.py:167-173; core/maddpg/maddpg.py:131-142) in one launch:
 * noise = clamp(N(0, sigma), -clip, clip); out = clamp(action + noise, -1, 1). action [B][A] contiguous (the target
 * actor's output); `noise` [B][A] (already scaled by sigma) is read when rng_ctl is NULL, otherwise sigma * N(0,1) is drawn
 * from the Philox stream rng_ctl (see cstr_gaussian_head_fwd_f32). out rows are out_stride floats apart. */
int cstr_target_smooth_f32(const float *action, const float *noise, uint64_t *rng_ctl, float sigma, float clip, float *out,
                           int64_t out_stride, int64_t batch, int act_dim, cstr_stream_t stream);

/* A deterministic target actor's LAST Linear with that smoothing inside (core/td3/td3.py:167-173: actor_target(next_obs), then
 * noise, clamp, add, clamp): out[row][j] = clamp(act(x W^T + b)[row][j] + clamp(z, -clip, clip), -1, 1), z as in
 * cstr_target_smooth_f32 (noise [m][n] read, or sigma * N(0,1) drawn from rng_ctl with the same counters; the offset advances by
 * m). One launch instead of two; bit-identical to cstr_linear_act_fwd_f32 followed by cstr_target_smooth_f32. w [n][k], n <= 16,
 * k > 32, m <= 32768 (CSTR_E_UNSUPPORTED otherwise: use the two launches). */
int cstr_linear_smooth_fwd_f32(const float *x, int64_t ldx, const float *w, const float *bias, int act, const float *noise,
                               uint64_t *rng_ctl, float sigma, float clip, float *out, int64_t out_stride, int64_t m, int64_t n,
                               int64_t k, cstr_stream_t stream);

/* SAC entropy coefficient (core/sac/sac.py:230-243): ent_coef_out = exp(log_alpha); grad_out = d/dlog_alpha of
 * -mean(log_alpha * (logp + target_entropy)) = -mean(logp + target_entropy). loss_out (stored) and loss_sum /
 * ent_coef_sum (accumulated) are device scalars, each may be NULL: the values train() logs (:232, :236), no host sync. */
int cstr_sac_alpha_f32(const float *log_alpha, const float *logp, float target_entropy, float *grad_out, float *ent_coef_out,
                       float *loss_out, float *loss_sum, float *ent_coef_sum, int64_t batch, cstr_stream_t stream);

/* Twin-critic loss as a backward root: loss = scale * (mse(q1, t) + mse(q2, t)) (scale 0.5: core/sac/sac.py:261;
 * scale 1: core/td3/td3.py:182, core/maddpg/maddpg.py:157); gq_k = d loss / d q_k. */
int cstr_twin_q_loss_f32(const float *q1, const float *q2, const float *target, float scale, float *gq1, float *gq2,
                         float *loss_out, float *loss_sum, int64_t batch, cstr_stream_t stream);

/* TD target + twin-critic loss (+ optionally SAC's entropy-coefficient loss) in ONE launch: cstr_td_target_min_f32,
 * cstr_twin_q_loss_f32 and cstr_sac_alpha_f32 back to back, bit for bit (core/sac/sac.py:230-261; core/td3/td3.py:167-182).
 * q1_t / q2_t: the target critics on the next state; next_logp NULL for TD3; `alpha` NULL (or log_alpha NULL) = no
 * entropy-coefficient part, then the target uses ent_coef[0]; with it the target uses exp(log_alpha[0]) and ent_coef is
 * ignored. target_out [B] may be NULL. */
typedef struct cstr_alpha_part {
    const float *log_alpha; /* [1] */
    const float *logp_pi;   /* [B] log pi(a|s) of the actor pass on the sampled observations */
    float target_entropy;
    float *grad_out;        /* [1] d loss / d log_alpha */
    float *ent_coef_out;    /* [1] exp(log_alpha) */
    float *loss_out;        /* [1] or NULL: stored */
    float *loss_sum;        /* [1] or NULL: accumulated */
    float *ent_coef_sum;    /* [1] or NULL: accumulated */
} cstr_alpha_part_t;
int cstr_td_twin_q_loss_f32(const float *q1_t, const float *q2_t, const float *next_logp, const float *rew, const float *done,
                            const float *ent_coef, float gamma, const float *q1, const float *q2, float scale, float *target_out,
                            float *gq1, float *gq2, float *loss_out, float *loss_sum, const cstr_alpha_part_t *alpha,
                            int64_t batch, cstr_stream_t stream);

/* cstr_hidden_head_bwd_f32 with the loss root INSIDE: d(loss)/dq of the twin Q networks (groups = 2) is computed per row in
 * the kernel instead of being read from gq, and one extra workgroup performs the batch reductions of the separate loss
 * launch (logged loss, entropy-coefficient part, g_logp) with that launch's arithmetic and reduction order -- the critic
 * backward's / the actor backward's first launch and its loss launch become ONE.
 *   mode 1 (critic loss, core/sac/sac.py:245-261, core/td3/td3.py:174-182) = cstr_td_twin_q_loss_f32's fields;
 *   mode 2 (SAC actor loss, core/sac/sac.py:273-275)                       = cstr_sac_actor_loss_f32's fields (q1/q2 = Q(s, pi(s)));
 *   mode 3 (deterministic actors' loss -mean(Q1(s, pi(s))), core/td3/td3.py:194, core/maddpg/maddpg.py:177) = cstr_neg_mean_loss_f32's
 *          fields (q1, loss_out, loss_sum), through the FIRST Q network alone: y / dz [1][m][k], w2 [k], gb1 / gw2 / gb2 of that network.
 * Unused pointers NULL; gq1 / gq2 of the separate launches are not produced (nothing else reads them). m <= 1024 rows. */
typedef struct cstr_head_root {
    int32_t mode, batch;
    float gamma, scale;
    const float *q1_t, *q2_t, *next_logp, *rew, *done; /* mode 1 */
    const float *ent_coef;                             /* [1]: mode 1 without alpha part, mode 2 */
    const float *q1, *q2;                              /* [batch] each */
    float *target_out;                                 /* mode 1, [batch] or NULL */
    const float *logp;                                 /* mode 2 */
    float *g_logp;                                     /* mode 2 out [batch] */
    float *loss_out, *loss_sum;                        /* [1] or NULL */
    cstr_alpha_part_t alpha;                           /* mode 1: log_alpha NULL = absent */
} cstr_head_root_t;
int cstr_hidden_head_bwd_root_f32(const cstr_head_root_t *root, const float *y, int act, const float *w2, float *dz, float *gb1,
                                  float *gw2, float *gb2, int64_t m, int64_t k, cstr_stream_t stream);

/* SAC actor loss as a backward root (core/sac/sac.py:273-275): loss = mean(ent_coef * logp - min(q1, q2)). */
int cstr_sac_actor_loss_f32(const float *logp, const float *q1, const float *q2, const float *ent_coef, float *g_logp, float *gq1,
                            float *gq2, float *loss_out, float *loss_sum, int64_t batch, cstr_stream_t stream);

/* Deterministic-policy actor loss (core/td3/td3.py:194, core/maddpg/maddpg.py:174): loss = -mean(q), gq = -1/B. */
int cstr_neg_mean_loss_f32(const float *q, float *gq, float *loss_out, float *loss_sum, int64_t batch, cstr_stream_t stream);

/* cstr_linear_bwd_weight_sets_f32 with the OPTIMISER STEP inside (single-GPU training: no collective sits between a gradient and
 * its Adam step; torch.optim.Adam's arithmetic, core/sac/sac.py:266-268, :279-281): the workgroup that has reduced a 16 x 16 tile of
 * dW (and db) applies Adam to exactly those parameters -- parameter / moment quads requested at entry, one launch instead of two.
 * dw / db are still written (the gradient arena stays inspectable). `flat` (n_flat <= 4): segments WITHOUT a weight-gradient tile
 * in the same launch -- Adam over a flat range (SAC's entropy coefficient, whose gradient an earlier launch wrote) or a soft target
 * update (polyak_source set) of parameters this launch does not change.
 * Every step counter must have been advanced by an EARLIER launch (cstr_chain_root_t.adam_advance: state["step"] += 1 and the
 * running beta powers): this launch only reads adam_ctl and writes no control word (no last-workgroup ticket). m > 32 rows. */
typedef struct cstr_adam_opt {
    const int64_t *adam_ctl; /* {step, ticket, beta1^step, beta2^step (f64 bits)} AFTER this step's increment */
    const double *lr;        /* [1] */
    double beta1, beta2, eps;
    float grad_scale;
    int32_t reserved;
} cstr_adam_opt_t;
typedef struct cstr_wgrad_adam_set {
    cstr_wgrad_set_t g;      /* the gradient part: dz, x, ldx, dw, db, m, n, k */
    float *w, *w_m, *w_v;    /* the weight [n][k] and its exp_avg / exp_avg_sq */
    float *b, *b_m, *b_v;    /* the bias [n] and its moments (with g.db) */
    float *shadow;           /* tile-major copy of w kept current (cstr_policy_swizzle_f32's layout) or NULL */
    int32_t opt, reserved;   /* index into opts */
    float *w_target, *b_target; /* these parameters' OWN target (soft update with the values just computed,
                                   core/common/utils.py:478-481: target = tau * p + (1 - tau) * target) or NULL */
    float tau, reserved2;
} cstr_wgrad_adam_set_t;
int cstr_linear_bwd_weight_adam_sets_f32(const cstr_wgrad_adam_set_t *sets, int n_sets, const cstr_adam_opt_t *opts, int n_opts,
                                         const cstr_adam_seg_t *flat, int n_flat, cstr_stream_t stream);

/* ---- row-chain kernels: a gradient step's forward / backward chains with FEWER launch boundaries ------------------------------
 * A chain of Linear layers is row-local (row r of layer l+1 needs row r of layer l only), but a launch boundary is the cheapest
 * way on this chip to hand data between workgroups (an in-kernel cross-workgroup barrier costs 4-17 us, an empty dependent launch
 * 1.6 us: tools/probes/cluster_chain_probe.hip, profiles/r03_notes.md). These kernels cut boundaries WITHOUT any hand-over inside a
 * launch: a workgroup owns 16 batch rows x one column group of the chain's WIDE layer (one f32-MFMA tile pass over K), RECOMPUTES
 * the cheap layer in front of it (K = obs_dim (+ act_dim) <= 12 inputs, or an element-wise function of stored activations), and
 * leaves the NARROW layer behind it (a Q head's H2 -> 1 dot product, the actor head's H2 -> 2A, the first layer's input gradient
 * H1 -> act_dim) as per-column-group PARTIAL sums that the next launch adds up in a fixed order in its prologue.
 * SAC.train (core/sac/sac.py:215-287) = 10 launches instead of 20; everything is deterministic (no float atomics).
 * Networks: create_mlp(in, out, [H1, H2], ReLU) (core/common/torch_layers.py:110-183); H1, H2 multiples of 4, <= 512; batch a
 * multiple of 16, <= 1024; (obs_dim, act_dim) as the ring's layouts. `tiles` = 16-column tiles per workgroup (1, 2 or 4). */
#define CSTR_CHAIN_MAX_NETS 16
#define CSTR_CHAIN_MAX_WIDTH 512

/* One Q network Linear(W, H1)-ReLU-Linear(H1, H2)-ReLU-Linear(H2, 1) (core/common/policies.py:960-987) of a chain launch. */
typedef struct cstr_chain_net {
    const float *w1, *b1, *w2, *b2, *w3, *b3; /* [H1][W], [H1], [H2][H1], [H2], [H2], [1] */
    const float *x;                            /* forward: this network's input rows [batch][W] (ld = W) */
    float *h1, *h2;                            /* [batch][H1], [batch][H2] post-activation: forward = written when not NULL (kept for
                                                  the backward); backward = read */
    float *q_part;                             /* [n_colgroups][batch] partial head sums (forward: out; backward: in) */
    int32_t role, reserved;                    /* forward, see CSTR_CHAIN_ROLE_* */
} cstr_chain_net_t;
#define CSTR_CHAIN_ROLE_PLAIN 0      /* x is complete */
#define CSTR_CHAIN_ROLE_STORE_PI 1   /* x is complete; the column-group-0 workgroups ALSO finalise the actor head of the pi(obs) rows
                                        and store x_pi's action columns, params, logp_pi (a side job: nothing here reads them) */
#define CSTR_CHAIN_ROLE_NEXT 2       /* x = x_next whose ACTION columns are not written yet: every workgroup finalises the actor head
                                        of its pi(next_obs) rows for itself */
#define CSTR_CHAIN_ROLE_NEXT_STORE 3 /* ... and the column-group-0 workgroups store them (x_next action columns, logp_next) */
#define CSTR_CHAIN_ROLE_PI 4         /* x = x_pi whose ACTION columns are not written yet (a deterministic actor's loss pass): finalised
                                        by every workgroup for itself, stored by the column-group-0 workgroups */
/* rows of an actor chain pass */
#define CSTR_CHAIN_ROWS_PAIR 0 /* [obs rows | next_obs rows], 2B rows: SAC's pi(obs) and pi(next_obs) */
#define CSTR_CHAIN_ROWS_NEXT 1 /* next_obs rows, B rows: a target actor (core/td3/td3.py:171) */
#define CSTR_CHAIN_ROWS_OBS 2  /* obs rows, B rows, activations kept: the deterministic actors' loss pass (core/td3/td3.py:194) */
/* head of the actor */
#define CSTR_CHAIN_HEAD_GAUSSIAN 0      /* [mu | log_std] (H2, 2A) + squashed-Gaussian sampling (core/sac/policies.py:147-175) */
#define CSTR_CHAIN_HEAD_DETERMINISTIC 1 /* Linear(H2, A) + Tanh (core/td3/policies.py:57-83) */

/* SAC actor Linear(D, H1)-ReLU-Linear(H1, H2)-ReLU-[mu | log_std](H2, 2A) (core/sac/policies.py:84-175), heads merged. */
typedef struct cstr_sac_actor {
    int32_t obs_dim, act_dim, h1, h2;
    const float *w1, *b1, *w2, *b2, *hw, *hb; /* hw [head_n][H2], hb [head_n]: head_n = 2A (Gaussian) or A (deterministic) */
} cstr_sac_actor_t;

/* SAC.train's actor passes pi(obs) (core/sac/sac.py:222) and pi(next_obs) (:247) as ONE launch over 2B rows (rows [0, B) = obs,
 * [B, 2B) = next_obs): ReplayBuffer.sample's gather (sample_idx = the (row, env) pairs drawn by cstr_rollout_step_f32, or NULL:
 * the observation columns of x_pi / x_next are already filled) + layer 1 (recomputed per workgroup) + layer 2 (one MFMA column
 * group per workgroup) + PARTIAL head sums head_part [n_colgroups][2B][2A] (n_colgroups = ceil(H2 / (16 * tiles))). With sample_idx
 * the column-group-0 workgroups also materialise the packed batch (cstr_linear_act_fwd_gather_f32's contract: x_data, the
 * observation columns of x_pi / x_next, rewards, dones * (1 - timeouts)) and thread 0 advances the ring position.
 * a_h1 [B][H1], a_h2 [B][H2]: the pi(obs) rows' activations, kept for cstr_sac_actor_chain_bwd_f32.
 * eps_all [2B][A] (or NULL): the sampling head's noise, drawn HERE by an otherwise idle wave (Philox4x32-10 / Box-Muller, key =
 * head_rng_ctl[0], counter = head_rng_ctl[1] + head_rng_offset + row, row in [0, 2B): the stream positions of
 * cstr_gaussian_head_gemm_fwd_f32 on the 2B-row pass) -- it does not depend on the network, and the launch that finalises the head has
 * a long enough prologue without it. head_rng_offset: draws of an earlier launch on the same stream whose offset advance is still
 * pending (cstr_rollout_step_f32 leaves it to the launches behind it). head_rng_ctl is only READ in this launch; the backward chain
 * launch advances the offset by everything that is pending (cstr_chain_root_t.rng_ctl). */
int cstr_sac_actor_chain_fwd_f32(const cstr_sac_actor_t *actor, const cstr_ring_t *ring, int64_t *ring_ctl, int advance_ring,
                                 const int32_t *sample_idx, int64_t batch, float *x_data, float *x_pi, float *x_next, float *out_done,
                                 float *out_rew, float *a_h1, float *a_h2, float *head_part, const uint64_t *head_rng_ctl,
                                 uint64_t head_rng_offset, float *eps_all, int rows_mode, int head_n, int tiles, cstr_stream_t stream);

/* How a consumer launch turns the actor's head partials into actions (core/common/distributions.py:207-260, the arithmetic of
 * cstr_gaussian_head_gemm_fwd_f32): params = sum of partials + hb; u = mean + exp(clamp(log_std)) * eps; a = tanh(u); log-prob.
 * eps [2B][A]: the actor chain launch's eps_all, or teacher-forced draws. */
typedef struct cstr_sac_head_fin {
    const float *head_part; /* [n_parts][part_rows][head_n] */
    const float *hb;        /* [head_n] */
    const float *eps;       /* [part_rows][A] standard normal draws (or teacher-forced noise); deterministic head: may be NULL (no noise) */
    int32_t n_parts, act_dim, obs_dim, kind; /* kind: CSTR_CHAIN_HEAD_* */
    int32_t part_rows, next_offset;          /* rows of the actor pass; row of batch element b: pi rows b, next rows next_offset + b */
    float sigma, clip;      /* deterministic head, next rows: a' = clamp(tanh(.) + clamp(sigma * eps, -clip, clip), -1, 1)
                               (target policy smoothing, core/td3/td3.py:167-171) */
    float *x_pi, *x_next;   /* [B][D + A]: action columns written by the STORE roles */
    float *params, *logp_pi, *logp_next; /* Gaussian head: [B][2A], [B], [B] */
} cstr_sac_head_fin_t;

/* Forward of n_nets <= 16 Q networks on `batch` rows each in ONE launch (16: every agent's critic and target critic of a 4-agent MADDPG
 * step without a policy update, core/maddpg/maddpg.py:146-164): layer 1 recomputed per workgroup (K = W <= 12), layer 2 one
 * MFMA column group per workgroup, head as partial sums q_part [n_colgroups][batch]. SAC / TD3 critic step: nets 0, 1 = the critic on
 * x_data, nets 2, 3 = the target on x_next (core/sac/sac.py:250, :258; core/td3/td3.py:173, :179); actor loss: the critic on x_pi
 * (:273). `fin` (or NULL): the SAC actor's pending head (roles above). */
int cstr_q_chain_fwd_f32(const cstr_chain_net_t *nets, int n_nets, int w_in, int obs_dim, int h1, int h2, int64_t batch,
                         const cstr_sac_head_fin_t *fin, int tiles, cstr_stream_t stream);

/* Loss root + backward of the twin Q networks in ONE launch (what cstr_hidden_head_bwd_root_f32 + cstr_linear_bwd_input_f32 did in
 * two or three): per workgroup q = sum of partials + b3, d(loss)/dq of its 16 rows (cstr_head_root_t's modes and arithmetic:
 * 1 = TD critic loss, 2 = SAC actor loss, 3 = -mean(Q1)), dz2 = dq * w3 * relu'(h2) recomputed (element-wise), one MFMA column group
 * of dz1 = (dz2 W2) * relu'(h1), and -- when gact_part is given (the actor loss: the critic is frozen) -- partial sums of
 * d(loss)/d(action) = dz1 W1[:, obs_dim:], gact_part [n_nets][n_colgroups][batch][A]. ONE extra workgroup does the batch reductions
 * (logged loss, entropy-coefficient part) and advances the actor's Philox offset. */
typedef struct cstr_chain_root {
    int32_t mode, batch;
    float gamma, scale;
    const float *q_part[4]; /* nets 0, 1: q1 / q2 (the differentiated networks); 2, 3: q1_t / q2_t (mode 1) */
    const float *b3[4];
    int32_t n_parts, reserved;
    const float *next_logp, *rew, *done;  /* mode 1 */
    const float *ent_coef;                /* [1]: mode 1 without alpha part, mode 2 */
    const float *logp;                    /* mode 2: log pi(a|obs) */
    float *target_out;                    /* mode 1, [batch] or NULL */
    float *q_out;                         /* [2][batch] finalised q1 / q2 (mode 3: [1][batch]) or NULL */
    float *gq_out;                        /* [2][batch] d(loss)/dq (the head's dW3 / db3 operand) or NULL */
    float *loss_out, *loss_sum;           /* [1] or NULL */
    cstr_alpha_part_t alpha;              /* mode 1: log_alpha NULL = absent */
    uint64_t *rng_ctl;                    /* or NULL: rng_ctl[1] += rng_advance by the loss workgroup */
    uint64_t rng_advance;
    /* Adam control words the loss workgroup advances on behalf of the optimiser launch behind this one (state["step"] += 1,
     * beta^step *= beta: cstr_linear_bwd_weight_adam_sets_f32 reads them pre-advanced); NULL = none */
    int64_t *adam_advance[2];
    double adam_beta1[2], adam_beta2[2];
} cstr_chain_root_t;
int cstr_q_chain_bwd_f32(const cstr_chain_net_t *nets, int n_nets, const cstr_chain_root_t *root, int w_in, int obs_dim, int h1, int h2,
                         float *dz2, float *dz1, float *gact_part, int tiles, cstr_stream_t stream);

/* out[row * out_stride + col] = sum over p < n_parts of part[p][row][col] (p ascending): the finalisation of a chain launch's partial
 * sums for a consumer that is not a chain kernel -- MADDPG's per-layer actor backward (core/maddpg/maddpg.py:174-179) reads
 * d(loss)/d(action) from the action columns of a critic-input gradient. */
int cstr_chain_sum_parts_f32(const float *part, int n_parts, int64_t rows, int cols, float *out, int64_t out_stride, cstr_stream_t stream);

/* Backward of the SAC actor from the critic's action-gradient partials in ONE launch (cstr_gaussian_head_bwd_input_f32 +
 * cstr_linear_bwd_input_f32): d(loss)/d(action) = sum of gact_part over networks and column groups; d(loss)/d(logp) = ent_coef / B
 * (core/sac/sac.py:275); the squashed-Gaussian head's analytic backward -> g_params [B][2A]; dz2 = (g_params hw) * relu'(a_h2)
 * recomputed (2A terms per element); one MFMA column group of dz1 = (dz2 W2) * relu'(a_h1).
 * kind = CSTR_CHAIN_HEAD_DETERMINISTIC (core/td3/td3.py:194-199): g_params [B][A] = d(loss)/d(action) * (1 - a^2) (the Tanh); ent_coef,
 * params and eps are not read. */
int cstr_sac_actor_chain_bwd_f32(const cstr_sac_actor_t *actor, const float *gact_part, int n_nets, int n_parts, const float *ent_coef,
                                 const float *x_pi, const float *params, const float *eps, const float *a_h1, const float *a_h2,
                                 float *g_params, float *dz2, float *dz1, int64_t batch, int kind, int tiles, cstr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CSTR_RL_HIP_H */
