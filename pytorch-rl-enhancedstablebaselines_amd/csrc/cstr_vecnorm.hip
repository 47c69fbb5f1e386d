// cstr_vecnorm.hip -- VecNormalize on the device (reference: core/common/vec_env/vec_normalize.py:174-290,
// core/common/running_mean_std.py:34-55; applied to sampled batches by core/common/buffers.py:143-155, :312-323).
//
// The running moments are f64 like the reference's. One 1024-lane workgroup does a whole VecNormalize.step_wait for N
// envs: two-pass batch mean / variance per observation column (f64 accumulation, LDS tree), the parallel-variance merge
// into the running moments, the normalised observations, then the same for the discounted returns and the rewards.
// N x D is a few 10^4 values at the training sizes; this is latency, not bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cstr_rl_hip.h"
#include "cstr_device.h"

namespace {

constexpr int VN_TPB = 1024, VN_MAXD = 8;
// vn_state (f64): [0,8) obs mean, [8,16) obs var, 16 obs count, 17 ret mean, 18 ret var, 19 ret count
constexpr int VN_OBS_MEAN = 0, VN_OBS_VAR = 8, VN_OBS_COUNT = 16, VN_RET_MEAN = 17, VN_RET_VAR = 18, VN_RET_COUNT = 19;

// workgroup-wide sum, result broadcast to every lane
__device__ double block_sum(double v, double *scratch)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();  // scratch may still be read from the previous call
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < VN_TPB / 64; ++w) s += scratch[w];
    return s;
}

// RunningMeanStd.update_from_moments (running_mean_std.py:40-55), one statistic
__device__ __forceinline__ void merge_moments(double &mean, double &var, double &count, double b_mean, double b_var, double n)
{
    const double delta = b_mean - mean, tot = count + n;
    const double new_mean = mean + delta * n / tot;
    const double m2 = var * count + b_var * n + delta * delta * count * n / (count + n);
    mean = new_mean;
    var = m2 / (count + n);
    count = n + count;
}

__global__ __launch_bounds__(VN_TPB) void vecnorm_step_kernel(const cstr_vecnorm_cfg_t cfg, double *__restrict__ st,
                                                              double *__restrict__ returns, const float *__restrict__ obs,
                                                              const float *__restrict__ reward, const float *__restrict__ done,
                                                              float *__restrict__ norm_obs, float *__restrict__ norm_rew,
                                                              const int64_t n)
{
    __shared__ double scratch[VN_TPB / 64];
    __shared__ double s_mean[VN_MAXD], s_inv[VN_MAXD];
    const int t = threadIdx.x, D = cfg.obs_dim;
    const double dn = (double)n;

    // ---- observations: obs_rms.update(obs) then normalize_obs (vec_normalize.py:183-191) ----
    if (cfg.norm_obs) {
        if (cfg.training) {
            for (int j = 0; j < D; ++j) {
                double a = 0.0;
                for (int64_t i = t; i < n; i += VN_TPB) a += (double)obs[i * D + j];
                const double b_mean = block_sum(a, scratch) / dn;
                a = 0.0;
                for (int64_t i = t; i < n; i += VN_TPB) {
                    const double d = (double)obs[i * D + j] - b_mean;
                    a += d * d;
                }
                const double b_var = block_sum(a, scratch) / dn;
                if (t == 0) {
                    double m = st[VN_OBS_MEAN + j], v = st[VN_OBS_VAR + j], c = st[VN_OBS_COUNT];
                    merge_moments(m, v, c, b_mean, b_var, dn);
                    st[VN_OBS_MEAN + j] = m;
                    st[VN_OBS_VAR + j] = v;
                    if (j == D - 1) st[VN_OBS_COUNT] = c;  // the count is shared by all columns
                }
            }
        }
        __syncthreads();
        if (t < D) {
            s_mean[t] = st[VN_OBS_MEAN + t];
            s_inv[t] = sqrt(st[VN_OBS_VAR + t] + cfg.epsilon);
        }
        __syncthreads();
    }
    if (norm_obs) {
        for (int64_t e = t; e < n * D; e += VN_TPB) {
            const int j = (int)(e % D);
            float o = obs[e];
            if (cfg.norm_obs) {
                const double z = ((double)o - s_mean[j]) / s_inv[j];
                o = (float)fmin(fmax(z, -cfg.clip_obs), cfg.clip_obs);
            }
            norm_obs[e] = o;
        }
    }

    // ---- rewards: _update_reward, normalize_reward, returns[dones] = 0 (vec_normalize.py:193-203, :206-209) ----
    if (!reward) {  // reset(): returns = 0 (vec_normalize.py:298)
        for (int64_t i = t; i < n; i += VN_TPB) returns[i] = 0.0;
        return;
    }
    if (cfg.training) {
        double a = 0.0;
        for (int64_t i = t; i < n; i += VN_TPB) {
            const double r = returns[i] * cfg.gamma + (double)reward[i];
            returns[i] = r;
            a += r;
        }
        const double b_mean = block_sum(a, scratch) / dn;
        a = 0.0;
        for (int64_t i = t; i < n; i += VN_TPB) {
            const double d = returns[i] - b_mean;
            a += d * d;
        }
        const double b_var = block_sum(a, scratch) / dn;
        if (t == 0) {
            double m = st[VN_RET_MEAN], v = st[VN_RET_VAR], c = st[VN_RET_COUNT];
            merge_moments(m, v, c, b_mean, b_var, dn);
            st[VN_RET_MEAN] = m; st[VN_RET_VAR] = v; st[VN_RET_COUNT] = c;
        }
        __syncthreads();
    }
    const double rs = sqrt(st[VN_RET_VAR] + cfg.epsilon);
    for (int64_t i = t; i < n; i += VN_TPB) {
        if (norm_rew) {
            float r = reward[i];
            if (cfg.norm_reward) r = (float)fmin(fmax((double)r / rs, -cfg.clip_reward), cfg.clip_reward);
            norm_rew[i] = r;
        }
        if (done && done[i] != 0.0f) returns[i] = 0.0;
    }
}

// ReplayBuffer._get_samples with env=VecNormalize: normalize_obs on observations / next_observations and
// normalize_reward on rewards of a sampled batch, in place, with the CURRENT statistics (buffers.py:312-323)
__global__ void vecnorm_apply_kernel(const cstr_vecnorm_cfg_t cfg, const double *__restrict__ st, float *__restrict__ obs,
                                     float *__restrict__ next_obs, float *__restrict__ reward, const int64_t batch)
{
    const int D = cfg.obs_dim;
    const int64_t n_obs = batch * D, total = 2 * n_obs + batch;
    const double rs = sqrt(st[VN_RET_VAR] + cfg.epsilon);
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < 2 * n_obs) {
            float *base = e < n_obs ? obs : next_obs;
            if (!cfg.norm_obs || !base) continue;
            float *p = base + (e < n_obs ? e : e - n_obs);
            const int j = (int)((e < n_obs ? e : e - n_obs) % D);
            const double z = ((double)*p - st[VN_OBS_MEAN + j]) / sqrt(st[VN_OBS_VAR + j] + cfg.epsilon);
            *p = (float)fmin(fmax(z, -cfg.clip_obs), cfg.clip_obs);
        } else if (cfg.norm_reward && reward) {
            float *p = reward + (e - 2 * n_obs);
            *p = (float)fmin(fmax((double)*p / rs, -cfg.clip_reward), cfg.clip_reward);
        }
    }
}

__global__ void vecnorm_init_kernel(double *__restrict__ st)
{
    const int t = threadIdx.x;
    if (t < VN_MAXD) { st[VN_OBS_MEAN + t] = 0.0; st[VN_OBS_VAR + t] = 1.0; }
    if (t == 0) { st[VN_OBS_COUNT] = 1e-4; st[VN_RET_MEAN] = 0.0; st[VN_RET_VAR] = 1.0; st[VN_RET_COUNT] = 1e-4; }
}

bool cfg_ok(const cstr_vecnorm_cfg_t *c)
{
    return c && c->obs_dim >= 1 && c->obs_dim <= VN_MAXD && c->clip_obs > 0.0 && c->clip_reward > 0.0 && c->epsilon >= 0.0;
}

}  // namespace

extern "C" int cstr_vecnorm_init_f64(double *vn_state, cstr_stream_t stream)
{
    if (!vn_state) return CSTR_E_BADARG;
    vecnorm_init_kernel<<<1, 64, 0, (hipStream_t)stream>>>(vn_state);
    return (int)hipGetLastError();
}

extern "C" int cstr_vecnorm_step_f64(const cstr_vecnorm_cfg_t *cfg, double *vn_state, double *returns, const float *obs,
                                     const float *reward, const float *done, float *norm_obs_out, float *norm_reward_out,
                                     int64_t n_envs, cstr_stream_t stream)
{
    if (!cfg_ok(cfg)) return cfg ? CSTR_E_UNSUPPORTED : CSTR_E_BADARG;
    if (!vn_state || !returns || !obs || n_envs <= 0) return CSTR_E_BADARG;
    if (!reward && (done || norm_reward_out)) return CSTR_E_BADARG;  // reset() form carries no reward / done
    vecnorm_step_kernel<<<1, VN_TPB, 0, (hipStream_t)stream>>>(*cfg, vn_state, returns, obs, reward, done, norm_obs_out,
                                                               norm_reward_out, n_envs);
    return (int)hipGetLastError();
}

extern "C" int cstr_vecnorm_apply_f32(const cstr_vecnorm_cfg_t *cfg, const double *vn_state, float *obs, float *next_obs,
                                      float *reward, int64_t batch, cstr_stream_t stream)
{
    if (!cfg_ok(cfg)) return cfg ? CSTR_E_UNSUPPORTED : CSTR_E_BADARG;
    if (!vn_state || batch <= 0 || (!obs && !next_obs && !reward)) return CSTR_E_BADARG;
    int block, grid;
    flat_launch_shape(batch * (2 * cfg->obs_dim + 1), block, grid);
    vecnorm_apply_kernel<<<grid, block, 0, (hipStream_t)stream>>>(*cfg, vn_state, obs, next_obs, reward, batch);
    return (int)hipGetLastError();
}
