// The optimizer step of the training loop as one launch: torch.nn.utils.clip_grad_value_ + torch.optim.Adam.step over EVERY parameter tensor of the
// model (train.py:430-433 builds the Adam, :541-543 clips and steps; 360 - 440 tensors in the shipped models).  torch runs that as ~50 multi-tensor
// launches behind Python loops over the tensor list: 2.3 ms of GPU time and 4 - 5 ms of host time per step at the end of a step, when nothing is
// left to overlap it with.  Here: the caller keeps a device table of (parameter, gradient, exp_avg, exp_avg_sq, count) entries and one kernel
// walks it (grid = 4 096-element chunks x parameters).
//
// Arithmetic = torch.optim.Adam (torch/optim/adam.py, _multi_tensor_adam, default flags) element for element:
//   g <- clamp(g, -clip, clip)  (written back, as clip_grad_value_ does)      [clip > 0]
//   g' = g + weight_decay p                                                    [weight_decay != 0]
//   m <- m + (g' - m) (1 - beta1)            (Tensor.lerp_)
//   v <- v beta2 + (1 - beta2) g' g'         (mul_ + addcmul_)
//   p <- p - (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)    (addcdiv_)
// mode 1: the clamp alone (clip_grad_value_ as a call of its own).  The hyper-parameters arrive as doubles (they are Python floats upstream) and
// 1 - beta, lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) are formed in double before the cast, as torch does: 1.0f - 0.999f is 1.3e-5 off.
#include "engine.h"

namespace kpd {

namespace {

constexpr int OPT_CHUNK = 4096;

struct AdamK {
    int mode;
    float clip, wd, one_minus_b1, b2, one_minus_b2, step_size, inv_sqrt_bc2, eps;
};

__device__ __forceinline__ void adam_elem(const AdamK &k, float &g, float &p, float &m, float &v) {
    if (k.clip > 0.0f) g = fminf(fmaxf(g, -k.clip), k.clip);
    if (k.mode == 1) return;
    float ge = g;
    if (k.wd != 0.0f) ge = fmaf(k.wd, p, ge);
    m = m + (ge - m) * k.one_minus_b1;
    v = v * k.b2 + k.one_minus_b2 * ge * ge;
    const float denom = sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
    p = p - k.step_size * (m / denom);
}

// block (x, y): elements [x OPT_CHUNK, (x + 1) OPT_CHUNK) of parameter y (blocks past a parameter's end leave at once); 16 bytes per lane where the
// four arrays are 16-byte aligned (torch's allocator: always), scalars otherwise and for the last n % 4 elements
__global__ __launch_bounds__(256) void k_adam(const kpd_adam_param *__restrict__ params, AdamK k) {
    const kpd_adam_param q = params[blockIdx.y];
    const long long start = (long long)blockIdx.x * OPT_CHUNK;
    if (start >= q.n) return;
    const long long end = q.n < start + OPT_CHUNK ? q.n : start + OPT_CHUNK;
    float *gp = const_cast<float *>(q.g);
    const bool vec = (((uintptr_t)q.p | (uintptr_t)q.g | (k.mode == 1 ? 0 : ((uintptr_t)q.m | (uintptr_t)q.v))) & 15) == 0;
    long long i = start;
    if (vec) {
        const long long end4 = start + ((end - start) & ~3LL);
        for (long long j = start + 4 * threadIdx.x; j < end4; j += 1024) {
            float4 g = *reinterpret_cast<const float4 *>(gp + j), p, m, v;
            if (k.mode == 0) {
                p = *reinterpret_cast<const float4 *>(q.p + j);
                m = *reinterpret_cast<const float4 *>(q.m + j);
                v = *reinterpret_cast<const float4 *>(q.v + j);
            }
            adam_elem(k, g.x, p.x, m.x, v.x);
            adam_elem(k, g.y, p.y, m.y, v.y);
            adam_elem(k, g.z, p.z, m.z, v.z);
            adam_elem(k, g.w, p.w, m.w, v.w);
            if (k.clip > 0.0f) *reinterpret_cast<float4 *>(gp + j) = g;
            if (k.mode == 0) {
                *reinterpret_cast<float4 *>(q.p + j) = p;
                *reinterpret_cast<float4 *>(q.m + j) = m;
                *reinterpret_cast<float4 *>(q.v + j) = v;
            }
        }
        i = end4;
    }
    for (i += threadIdx.x; i < end; i += 256) {
        float g = gp[i], p = 0.0f, m = 0.0f, v = 0.0f;
        if (k.mode == 0) p = q.p[i], m = q.m[i], v = q.v[i];
        adam_elem(k, g, p, m, v);
        if (k.clip > 0.0f) gp[i] = g;
        if (k.mode == 0) q.p[i] = p, q.m[i] = m, q.v[i] = v;
    }
}

}  // namespace

// include/kpd.h
extern "C" kpd_status kpd_adam_step(const kpd_adam_param *params_dev, int32_t n_params, int64_t max_numel, int32_t mode, double lr, double beta1, double beta2,
                                    double eps, double weight_decay, int64_t step, double clip_value, void *stream) {
    KPD_REQUIRE(n_params >= 0 && (n_params == 0 || params_dev) && (mode == 0 || mode == 1) && max_numel >= 0 && n_params <= 65535, KPD_ERR_INVALID,
                "kpd_adam_step: bad arguments (%d parameters)", n_params);
    KPD_REQUIRE(mode == 1 || (step >= 1 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0), KPD_ERR_INVALID,
                "kpd_adam_step: step=%lld beta=(%g, %g) eps=%g", (long long)step, beta1, beta2, eps);
    KPD_REQUIRE(mode == 0 || clip_value > 0.0, KPD_ERR_INVALID, "kpd_adam_step: the clamp alone needs a clip value");
    if (n_params == 0 || max_numel == 0) return KPD_OK;
    const double bc1 = mode == 1 ? 1.0 : 1.0 - pow(beta1, (double)step), bc2 = mode == 1 ? 1.0 : 1.0 - pow(beta2, (double)step);
    const AdamK k{mode, (float)clip_value, (float)weight_decay, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1),
                  (float)(1.0 / sqrt(bc2)), (float)eps};
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((max_numel + OPT_CHUNK - 1) / OPT_CHUNK), (unsigned)n_params), dim3(256), 0, static_cast<hipStream_t>(stream),
                       params_dev, k);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd
