// cstr_learner.hip -- element-wise learner kernels for gfx950: target-Q min, polyak soft update and Adam
// over flat fp32 parameter arenas. Pure HBM streaming: 16-byte accesses per lane, grid capped at 8
// workgroups per CU with a grid-stride loop; rounding points match torch's CPU/GPU kernels bit for bit
// (fused multiply-adds are written explicitly, the TU is built with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cstr_rl_hip.h"
#include "cstr_device.h"

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

// polyak_update (core/common/utils.py:478-481): t.mul_(1 - tau); t = t + tau * p  (add(alpha=) is one fma)
__device__ __forceinline__ float polyak1(float p, float t, float tau, float om) { return __fmaf_rn(tau, p, t * om); }

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

// torch/optim/adam.py::_single_tensor_adam rounding points (amsgrad=False, weight_decay=0):
//   m = lerp(m, g, 1-b1) = fma(1-b1, g-m, m);  v = fma((1-b2)*g, g, v*b2)
//   p = p + (-(lr/bc1) * m) / (sqrt(v)/sqrt(bc2) + eps)
struct AdamScalars { float step_size_neg, bc2_sqrt, w1, b2, omb2, eps, gscale; };

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, const AdamScalars &a)
{
    g = g * a.gscale;
    m = __fmaf_rn(a.w1, g - m, m);
    v = __fmaf_rn(a.omb2 * g, g, v * a.b2);
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p + (a.step_size_neg * m) / denom;
}

// tile-major shadow of one weight matrix inside the arena (cstr_policy_swizzle_f32's layout), kept current by the update itself
struct AdamShadow { float4 *out; int64_t begin4, end4; int k, kc; };

__device__ __forceinline__ void adam_body(float *__restrict__ param, const float *__restrict__ grad, float *__restrict__ exp_avg,
                                          float *__restrict__ exp_avg_sq, int64_t *__restrict__ adam_ctl,
                                          const double *__restrict__ lr, const double beta1, const double beta2, const double eps,
                                          const float gscale, const int64_t n, const AdamShadow sh = AdamShadow{nullptr, 0, 0, 4, 1},
                                          float *__restrict__ own_target = nullptr, const float tau = 0.0f, const float om = 1.0f)
{
    __shared__ AdamScalars sa;
    const int64_t nv = n >> 2;
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    float4 *p4 = reinterpret_cast<float4 *>(param), *m4 = reinterpret_cast<float4 *>(exp_avg), *v4 = reinterpret_cast<float4 *>(exp_avg_sq);
    const float4 *g4 = reinterpret_cast<const float4 *>(grad);
    // arenas far beyond the caches (a microbenchmark regime, not the learners'): stream the state past L2 in BOTH directions
    // (non-temporal loads as well as stores: every byte is touched once) and keep two 16-byte quads per stream in flight per lane
    const bool stream_out = n >= (int64_t)(16 << 20) && !sh.out && !own_target;
    // The learners' regime is launch latency: every lane's FIRST quad of the four streams is requested before thread 0 turns the
    // control words into the step's scalars (a dependent load of adam_ctl / lr, an f64 division and square root, a barrier) --
    // the two latencies overlap instead of adding up.
    const bool first = !stream_out && tid < nv;
    float4 fp = make_float4(0.0f, 0.0f, 0.0f, 0.0f), fm = fp, fv = fp, fg = fp;
    if (first) { fp = p4[tid]; fm = m4[tid]; fv = v4[tid]; fg = g4[tid]; }
    if (threadIdx.x == 0) {
        // state["step"] += 1; beta^step is carried in adam_ctl as a running product (two f64 multiplies instead of two
        // f64 pow() calls in every workgroup's prologue: 5.4 -> ~2 us per launch at 136 k parameters)
        const double *pw = reinterpret_cast<const double *>(adam_ctl + 2);
        const double bc1 = 1.0 - pw[0] * beta1, bc2 = 1.0 - pw[1] * beta2;
        sa.step_size_neg = -(float)(lr[0] / bc1);
        sa.bc2_sqrt = (float)sqrt(bc2);
        sa.w1 = (float)(1.0 - beta1);
        sa.b2 = (float)beta2;
        sa.omb2 = (float)(1.0 - beta2);
        sa.eps = (float)eps;
        sa.gscale = gscale;
    }
    __syncthreads();
    const AdamScalars a = sa;
    typedef float v4f __attribute__((ext_vector_type(4)));
    auto update = [&](const int64_t i, float4 p, float4 m, float4 v, const float4 g) {
        adam1(p.x, g.x, m.x, v.x, a); adam1(p.y, g.y, m.y, v.y, a);
        adam1(p.z, g.z, m.z, v.z, a); adam1(p.w, g.w, m.w, v.w, a);
        p4[i] = p; m4[i] = m; v4[i] = v;
        if (own_target) {  // the soft update of these parameters' target with the value just computed (cstr_polyak_f32's arithmetic)
            float4 t = reinterpret_cast<float4 *>(own_target)[i];
            t.x = polyak1(p.x, t.x, tau, om); t.y = polyak1(p.y, t.y, tau, om);
            t.z = polyak1(p.z, t.z, tau, om); t.w = polyak1(p.w, t.w, tau, om);
            reinterpret_cast<float4 *>(own_target)[i] = t;
        }
        if (sh.out && i >= sh.begin4 && i < sh.end4) {  // this float4 is one lane's operand quad of the matrix
            const int64_t e = (i - sh.begin4) * 4;
            const int row = (int)(e / sh.k), col = (int)(e - (int64_t)row * sh.k);
            sh.out[((int64_t)(row >> 4) * sh.kc + (col >> 4)) * 64 + (row & 15) + 16 * ((col & 15) >> 2)] = p;
        }
    };
    if (stream_out) {
        int64_t i = tid;
        for (; i + stride < nv; i += 2 * stride) {
            const int64_t j = i + stride;
            v4f p0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p4 + i)), p1 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p4 + j));
            v4f m0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(m4 + i)), m1 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(m4 + j));
            v4f v0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(v4 + i)), v1 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(v4 + j));
            const v4f g0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(g4 + i)), g1 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(g4 + j));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pp = p0[e], mm = m0[e], vv = v0[e];
                adam1(pp, g0[e], mm, vv, a);
                p0[e] = pp; m0[e] = mm; v0[e] = vv;
                pp = p1[e]; mm = m1[e]; vv = v1[e];
                adam1(pp, g1[e], mm, vv, a);
                p1[e] = pp; m1[e] = mm; v1[e] = vv;
            }
            __builtin_nontemporal_store(p0, reinterpret_cast<v4f *>(p4 + i)); __builtin_nontemporal_store(p1, reinterpret_cast<v4f *>(p4 + j));
            __builtin_nontemporal_store(m0, reinterpret_cast<v4f *>(m4 + i)); __builtin_nontemporal_store(m1, reinterpret_cast<v4f *>(m4 + j));
            __builtin_nontemporal_store(v0, reinterpret_cast<v4f *>(v4 + i)); __builtin_nontemporal_store(v1, reinterpret_cast<v4f *>(v4 + j));
        }
        for (; i < nv; i += stride) {
            float4 p = p4[i], m = m4[i], v = v4[i];
            const float4 g = g4[i];
            adam1(p.x, g.x, m.x, v.x, a); adam1(p.y, g.y, m.y, v.y, a);
            adam1(p.z, g.z, m.z, v.z, a); adam1(p.w, g.w, m.w, v.w, a);
            p4[i] = p; m4[i] = m; v4[i] = v;
        }
    } else {
        if (first) update(tid, fp, fm, fv, fg);
        for (int64_t i = tid + stride; i < nv; i += stride) update(i, p4[i], m4[i], v4[i], g4[i]);
    }
    for (int64_t i = (nv << 2) + tid; i < n; i += stride) {
        float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
        adam1(p, grad[i], m, v, a);
        param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
        if (own_target) own_target[i] = polyak1(p, own_target[i], tau, om);
    }
    if (last_block_ticket(reinterpret_cast<unsigned long long *>(adam_ctl + 1)) && threadIdx.x == 0) {
        double *pw = reinterpret_cast<double *>(adam_ctl + 2);
        adam_ctl[0] += 1;
        pw[0] *= beta1;
        pw[1] *= beta2;
    }
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

__device__ __forceinline__ void polyak_body(const float *__restrict__ param, float *__restrict__ target, const float tau,
                                            const float om, const int64_t n)
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
