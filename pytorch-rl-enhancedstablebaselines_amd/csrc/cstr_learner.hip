// cstr_learner.hip -- element-wise learner kernels for gfx950: target-Q min, polyak soft update and Adam
// over flat fp32 parameter arenas. Pure HBM streaming: 16-byte accesses per lane, grid capped at 8
// workgroups per CU with a grid-stride loop; rounding points match torch's CPU/GPU kernels bit for bit
// (fused multiply-adds are written explicitly, the TU is built with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cstr_rl_hip.h"
#include "cstr_device.h"
#include "cstr_adam_device.h"

namespace {

// SAC: core/sac/sac.py:250-254; TD3: core/td3/td3.py:174-176
__global__ void td_target_min_kernel(const float *__restrict__ q1, const float *__restrict__ q2, const float *__restrict__ logp,
                                     const float *__restrict__ rew, const float *__restrict__ done,
                                     const float *__restrict__ ent_coef, const float gamma, float *__restrict__ out,
                                     const int64_t n)
{
    const float alpha = ent_coef ? ent_coef[0] : 0.0f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float q = fminf(q1[i], q2[i]);                    // th.min(cat(q1, q2), dim=1)
        if (logp) q = q - alpha * logp[i];                // sac.py:252
        out[i] = rew[i] + (1.0f - done[i]) * gamma * q;   // sac.py:254
    }
}

__global__ void polyak_kernel(const float *__restrict__ param, float *__restrict__ target, const float tau, const float om,
                              const int64_t n)
{
    const int64_t nv = n >> 2;
    const float4 *p4 = reinterpret_cast<const float4 *>(param);
    float4 *t4 = reinterpret_cast<float4 *>(target);
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < nv; i += stride) {
        const float4 p = p4[i];
        float4 t = t4[i];
        t.x = polyak1(p.x, t.x, tau, om); t.y = polyak1(p.y, t.y, tau, om);
        t.z = polyak1(p.z, t.z, tau, om); t.w = polyak1(p.w, t.w, tau, om);
        t4[i] = t;
    }
    for (int64_t i = (nv << 2) + tid; i < n; i += stride) target[i] = polyak1(param[i], target[i], tau, om);
}

__global__ void adam_kernel(float *__restrict__ param, const float *__restrict__ grad, float *__restrict__ exp_avg,
                            float *__restrict__ exp_avg_sq, int64_t *__restrict__ adam_ctl, const double *__restrict__ lr,
                            const double beta1, const double beta2, const double eps, const float gscale, const int64_t n)
{
    adam_body(param, grad, exp_avg, exp_avg_sq, adam_ctl, lr, beta1, beta2, eps, gscale, n);
}

// Several optimisers' steps in ONE launch (e.g. SAC's entropy coefficient -- a single parameter -- next to the critic):
// blockIdx.y selects the segment, every segment keeps its own control words, learning rate and hyper-parameters.
struct AdamSegs { cstr_adam_seg_t s[CSTR_MAX_ADAM_SEGS]; };

__global__ void adam_multi_kernel(const AdamSegs segs)
{
    const cstr_adam_seg_t &s = segs.s[blockIdx.y];
    if (s.polyak_source) {  // a soft target update riding in the same launch: param = target arena
        polyak_body(s.polyak_source, s.param, (float)s.tau, (float)(1.0 - s.tau), s.n);
        return;
    }
    AdamShadow sh = {nullptr, 0, 0, 4, 1};
    if (s.shadow)
        sh = AdamShadow{reinterpret_cast<float4 *>(s.shadow), s.shadow_begin >> 2, (s.shadow_begin + s.shadow_n * s.shadow_k) >> 2,
                        (int)s.shadow_k, (int)((s.shadow_k + 15) >> 4)};
    adam_body(s.param, s.grad, s.exp_avg, s.exp_avg_sq, s.adam_ctl, s.lr, s.beta1, s.beta2, s.eps, s.grad_scale, s.n, sh, s.own_target,
              (float)s.tau, (float)(1.0 - s.tau));
}

}  // namespace

extern "C" int cstr_td_target_min_f32(const float *q1, const float *q2, const float *logp, const float *rew, const float *done,
                                      const float *ent_coef, float gamma, float *out, int64_t n, cstr_stream_t stream)
{
    if (!q1 || !q2 || !rew || !done || !out || n <= 0 || ((logp == nullptr) != (ent_coef == nullptr))) return CSTR_E_BADARG;
    int block, grid;
    flat_launch_shape(n, block, grid);
    td_target_min_kernel<<<grid, block, 0, (hipStream_t)stream>>>(q1, q2, logp, rew, done, ent_coef, gamma, out, n);
    return (int)hipGetLastError();
}

extern "C" int cstr_polyak_f32(const float *param, float *target, double tau, int64_t n, cstr_stream_t stream)
{
    if (!param || !target || n <= 0 || !aligned16(param) || !aligned16(target)) return CSTR_E_BADARG;
    int block, grid;
    flat_launch_shape((n + 3) / 4, block, grid);
    polyak_kernel<<<grid, block, 0, (hipStream_t)stream>>>(param, target, (float)tau, (float)(1.0 - tau), n);
    return (int)hipGetLastError();
}

extern "C" int cstr_adam_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t *adam_ctl,
                             const double *lr, double beta1, double beta2, double eps, float grad_scale, int64_t n,
                             cstr_stream_t stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || !adam_ctl || !lr || n <= 0) return CSTR_E_BADARG;
    if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq)) return CSTR_E_BADARG;
    int block, grid;
    flat_launch_shape((n + 3) / 4, block, grid);
    adam_kernel<<<grid, block, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, adam_ctl, lr, beta1, beta2, eps,
                                                         grad_scale, n);
    return (int)hipGetLastError();
}

extern "C" int cstr_adam_multi_f32(const cstr_adam_seg_t *segs, int n_segs, cstr_stream_t stream)
{
    if (!segs || n_segs <= 0) return CSTR_E_BADARG;
    if (n_segs > CSTR_MAX_ADAM_SEGS) return CSTR_E_UNSUPPORTED;
    AdamSegs a;
    int grid = 1;
    for (int i = 0; i < n_segs; ++i) {
        const cstr_adam_seg_t &s = segs[i];
        if (s.polyak_source) {
            if (!s.param || s.n <= 0 || !aligned16(s.param) || !aligned16(s.polyak_source) || s.param == s.polyak_source) return CSTR_E_BADARG;
        } else {
            if (!s.param || !s.grad || !s.exp_avg || !s.exp_avg_sq || !s.adam_ctl || !s.lr || s.n <= 0) return CSTR_E_BADARG;
            if (!aligned16(s.param) || !aligned16(s.grad) || !aligned16(s.exp_avg) || !aligned16(s.exp_avg_sq)) return CSTR_E_BADARG;
            if (s.own_target && (!aligned16(s.own_target) || s.own_target == s.param)) return CSTR_E_BADARG;
            if (s.shadow && (!aligned16(s.shadow) || s.shadow_begin < 0 || (s.shadow_begin & 3) || s.shadow_n <= 0 || s.shadow_k <= 0 ||
                             (s.shadow_k & 3) || s.shadow_k > 0x7fffffff || s.shadow_begin + s.shadow_n * s.shadow_k > s.n))
                return CSTR_E_BADARG;
        }
        int block, g;
        flat_launch_shape((s.n + 3) / 4, block, g);
        grid = g > grid ? g : grid;
        a.s[i] = s;
    }
    adam_multi_kernel<<<dim3((unsigned)grid, (unsigned)n_segs), 256, 0, (hipStream_t)stream>>>(a);
    return (int)hipGetLastError();
}
