// cstr_adam_device.h -- Adam / polyak device code shared by cstr_learner.hip (flat-arena launches) and cstr_mlp.hip (the weight-
// gradient launch that applies the optimiser step to the tile it has just reduced). Internal, not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cstr_device.h"

namespace {

// polyak_update (core/common/utils.py:478-481): t.mul_(1 - tau); t = t + tau * p  (add(alpha=) is one fma)
__device__ __forceinline__ float polyak1(float p, float t, float tau, float om) { return __fmaf_rn(tau, p, t * om); }


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
                                          float *__restrict__ own_target = nullptr, const float tau = 0.0f, const float om = 1.0f,
                                          const bool pre_advanced = false, const int64_t bid = blockIdx.x, const int64_t nblk = gridDim.x)
{
    // pre_advanced: an EARLIER launch has already performed state["step"] += 1 on adam_ctl (step counter and the running beta
    // powers): this launch only reads them, and there is no last-workgroup ticket at its end (cstr_chain_root_t.adam_advance)
    __shared__ AdamScalars sa;
    const int64_t nv = n >> 2;
    const int64_t tid = bid * (int64_t)blockDim.x + threadIdx.x, stride = nblk * (int64_t)blockDim.x;
    float4 *p4 = reinterpret_cast<float4 *>(param), *m4 = reinterpret_cast<float4 *>(exp_avg), *v4 = reinterpret_cast<float4 *>(exp_avg_sq);
    const float4 *g4 = reinterpret_cast<const float4 *>(grad);
    // arenas far beyond the caches (a microbenchmark regime, not the learners'): stream the state past L2 in BOTH directions
    // (non-temporal loads as well as stores: every byte is touched once) and keep two 16-byte quads per stream in flight per lane
    const bool stream_out = n >= (int64_t)(16 << 20) && !sh.out && !own_target;
    // The learners' regime is launch latency: every lane's FIRST quad of the four streams is requested before thread 0 turns the
    // control words into the step's scalars (a dependent load of adam_ctl / lr, an f64 division and square root, a barrier) --
    // the two latencies overlap instead of adding up.
    const bool first = !stream_out && tid < nv;
    float4 fp = make_float4(0.0f, 0.0f, 0.0f, 0.0f), fm = fp, fv = fp, fg = fp, ft = fp;
    if (first) {
        fp = p4[tid]; fm = m4[tid]; fv = v4[tid]; fg = g4[tid];
        if (own_target) ft = reinterpret_cast<const float4 *>(own_target)[tid];  // (behind the stores of p / m / v it is one more round trip)
    }
    if (threadIdx.x == 0) {
        // state["step"] += 1; beta^step is carried in adam_ctl as a running product (two f64 multiplies instead of two
        // f64 pow() calls in every workgroup's prologue: 5.4 -> ~2 us per launch at 136 k parameters)
        const double *pw = reinterpret_cast<const double *>(adam_ctl + 2);
        const double bc1 = 1.0 - (pre_advanced ? pw[0] : pw[0] * beta1), bc2 = 1.0 - (pre_advanced ? pw[1] : pw[1] * beta2);
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
    auto update = [&](const int64_t i, float4 p, float4 m, float4 v, const float4 g, float4 t) {
        adam1(p.x, g.x, m.x, v.x, a); adam1(p.y, g.y, m.y, v.y, a);
        adam1(p.z, g.z, m.z, v.z, a); adam1(p.w, g.w, m.w, v.w, a);
        p4[i] = p; m4[i] = m; v4[i] = v;
        if (own_target) {  // the soft update of these parameters' target with the value just computed (cstr_polyak_f32's arithmetic)
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
        if (first) update(tid, fp, fm, fv, fg, ft);
        for (int64_t i = tid + stride; i < nv; i += stride)
            update(i, p4[i], m4[i], v4[i], g4[i], own_target ? reinterpret_cast<const float4 *>(own_target)[i] : fp);
    }
    for (int64_t i = (nv << 2) + tid; i < n; i += stride) {
        float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
        adam1(p, grad[i], m, v, a);
        param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
        if (own_target) own_target[i] = polyak1(p, own_target[i], tau, om);
    }
    if (pre_advanced) return;
    if (last_block_ticket(reinterpret_cast<unsigned long long *>(adam_ctl + 1)) && threadIdx.x == 0) {
        double *pw = reinterpret_cast<double *>(adam_ctl + 2);
        adam_ctl[0] += 1;
        pw[0] *= beta1;
        pw[1] *= beta2;
    }
}


__device__ __forceinline__ void polyak_body(const float *__restrict__ param, float *__restrict__ target, const float tau,
                                            const float om, const int64_t n, const int64_t bid = blockIdx.x, const int64_t nblk = gridDim.x)
{
    const int64_t nv = n >> 2;
    const float4 *p4 = reinterpret_cast<const float4 *>(param);
    float4 *t4 = reinterpret_cast<float4 *>(target);
    const int64_t tid = bid * (int64_t)blockDim.x + threadIdx.x, stride = nblk * (int64_t)blockDim.x;
    for (int64_t i = tid; i < nv; i += stride) {
        const float4 p = p4[i];
        float4 t = t4[i];
        t.x = polyak1(p.x, t.x, tau, om); t.y = polyak1(p.y, t.y, tau, om);
        t.z = polyak1(p.z, t.z, tau, om); t.w = polyak1(p.w, t.w, tau, om);
        t4[i] = t;
    }
    for (int64_t i = (nv << 2) + tid; i < n; i += stride) target[i] = polyak1(param[i], target[i], tau, om);
}

// the step's scalars from (pre-advanced) control words: what thread 0 of adam_body computes
__device__ __forceinline__ AdamScalars adam_scalars_advanced(const int64_t *adam_ctl, const double *lr, const double beta1, const double beta2,
                                                             const double eps, const float gscale)
{
    const double *pw = reinterpret_cast<const double *>(adam_ctl + 2);
    const double bc1 = 1.0 - pw[0], bc2 = 1.0 - pw[1];
    AdamScalars a;
    a.step_size_neg = -(float)(lr[0] / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.w1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.omb2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.gscale = gscale;
    return a;
}

}  // namespace
