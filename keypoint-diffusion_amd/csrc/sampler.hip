// Reverse-diffusion state update around the denoiser: the elementwise part of
// KeypointDiffusion.sample_p_zs_given_zt (models/ligand_diffuser.py:515-536) fused with the
// ligand-COM removal (remove_com :185-203).  One workgroup per complex; the new ligand
// coordinates are staged in LDS so the COM is a fixed-order sum (deterministic).
#include "engine.h"

namespace kpd {

// coef[b] = {alpha_t_given_s, var_terms, sigma}
__global__ __launch_bounds__(256) void k_sample_update(const int *__restrict__ lig_ptr, const int *__restrict__ kp_ptr,
                                                       int atom_nf, float *__restrict__ lig_x, float *__restrict__ lig_h,
                                                       float *__restrict__ kp_x, const float *__restrict__ eps_x,
                                                       const float *__restrict__ eps_h, const float *__restrict__ noise_x,
                                                       const float *__restrict__ noise_h, const float *__restrict__ coef) {
    extern __shared__ float sx[];          // [3 * n_lig of this complex]
    __shared__ float s_com[3];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int llo = lig_ptr[b], nl = lig_ptr[b + 1] - llo;
    const int klo = kp_ptr[b], nk = kp_ptr[b + 1] - klo;
    const float alpha = coef[3 * b], var = coef[3 * b + 1], sigma = coef[3 * b + 2];

    for (int i = tid; i < nl * 3; i += blockDim.x) {
        const size_t g = (size_t)llo * 3 + i;
        sx[i] = lig_x[g] / alpha - var * eps_x[g] + sigma * noise_x[g];
    }
    for (int i = tid; i < nl * atom_nf; i += blockDim.x) {
        const size_t g = (size_t)llo * atom_nf + i;
        lig_h[g] = lig_h[g] / alpha - var * eps_h[g] + sigma * noise_h[g];
    }
    __syncthreads();
    if (tid < 3) {
        float s = 0.0f;
        for (int i = 0; i < nl; ++i) s += sx[3 * i + tid];
        s_com[tid] = s / (float)nl;
    }
    __syncthreads();
    for (int i = tid; i < nl * 3; i += blockDim.x) lig_x[(size_t)llo * 3 + i] = sx[i] - s_com[i % 3];
    for (int i = tid; i < nk * 3; i += blockDim.x) kp_x[(size_t)klo * 3 + i] -= s_com[i % 3];
}

// Per-complex coefficients of one reverse step from the noise-schedule table (ligand_diffuser.py:505-526, 654-690):
// gamma lookup at round(t T), sigma^2_t|s = -expm1(softplus(g_s) - softplus(g_t)), alpha_t|s = exp((softplus(g_s) -
// softplus(g_t)) / 2), sigma = sqrt(sigmoid(gamma)).  Replaces ~30 elementwise launches on B-element tensors.
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }   // torch default threshold

__global__ void k_step_coef(const float *__restrict__ gamma, int n_gamma, const float *__restrict__ s,
                            const float *__restrict__ t, int B, float *__restrict__ coef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float T = (float)(n_gamma - 1);
    const int is = min(max((int)rintf(s[b] * T), 0), n_gamma - 1), it = min(max((int)rintf(t[b] * T), 0), n_gamma - 1);
    const float gs = gamma[is], gt = gamma[it];
    const float dsp = softplusf_(gs) - softplusf_(gt);
    const float sigma2_ts = -expm1f(dsp);
    const float alpha_ts = expf(0.5f * dsp);
    const float sig_s = sqrtf(1.0f / (1.0f + expf(-gs))), sig_t = sqrtf(1.0f / (1.0f + expf(-gt)));
    coef[3 * b] = alpha_ts;
    coef[3 * b + 1] = sigma2_ts / alpha_ts / sig_t;
    coef[3 * b + 2] = sqrtf(sigma2_ts) * sig_s / sig_t;
}

// ---- per-complex counter-based noise ------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) keyed by (seed, complex id), counter = (element quad within the complex, step,
// tag); Box-Muller on the four 32-bit outputs.  The value of an element depends only on (seed, complex id, step, tag,
// position inside the complex), never on which batch or rank the complex was placed in, so a sharded run reproduces the
// single-process run.  The reference draws one global torch.randn over the batch (ligand_diffuser.py:367,
// 530-531); this is the opt-in replacement, torch.randn stays the default.

__global__ void k_complex_noise(const int *__restrict__ ptr, int B, int width, const long long *__restrict__ complex_id,
                                unsigned long long seed, int step, int tag, float *__restrict__ out) {
    const int b = blockIdx.x;
    const int lo = ptr[b], n = (ptr[b + 1] - lo) * width;
    const unsigned long long cid = (unsigned long long)complex_id[b];
    const unsigned k0 = (unsigned)seed ^ (unsigned)cid, k1 = (unsigned)(seed >> 32) ^ (unsigned)(cid >> 32) ^ 0x5bd1e995u;
    float *o = out + (size_t)lo * width;
    for (int quad = threadIdx.x; 4 * quad < n; quad += blockDim.x) {
        unsigned c[4] = {(unsigned)quad, (unsigned)step, (unsigned)tag, 0u};
        philox4x32_10(c, k0, k1);
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);        // (0, 1)
            const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float r = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            z[2 * h] = r * cs;
            z[2 * h + 1] = r * sn;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * quad + i < n) o[4 * quad + i] = z[i];
    }
}

}  // namespace kpd

using namespace kpd;

extern "C" kpd_status kpd_complex_noise(int32_t B, const int32_t *node_ptr, int32_t width, const int64_t *complex_id, uint64_t seed,
                                        int32_t step, int32_t tag, float *out, void *stream) {
    KPD_REQUIRE(node_ptr && complex_id && out, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(B >= 1 && width >= 1, KPD_ERR_INVALID, "B=%d width=%d", B, width);
    hipLaunchKernelGGL(k_complex_noise, dim3(B), dim3(128), 0, static_cast<hipStream_t>(stream), node_ptr, B, width,
                       reinterpret_cast<const long long *>(complex_id), (unsigned long long)seed, step, tag, out);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

extern "C" kpd_status kpd_step_coefficients(const float *gamma, int32_t n_gamma, const float *s, const float *t, int32_t B,
                                            float *coef, void *stream) {
    KPD_REQUIRE(gamma && s && t && coef, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(n_gamma >= 2 && B >= 1, KPD_ERR_INVALID, "n_gamma=%d B=%d", n_gamma, B);
    hipLaunchKernelGGL(k_step_coef, dim3(cdiv(B, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), gamma, n_gamma, s, t, B,
                       coef);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

extern "C" kpd_status kpd_sample_update(int32_t B, const int32_t *lig_ptr, const int32_t *kp_ptr, int32_t atom_nf,
                                        float *lig_x, float *lig_h, float *kp_x, const float *eps_x, const float *eps_h,
                                        const float *noise_x, const float *noise_h, const float *coef, int32_t max_lig,
                                        void *stream) {
    KPD_REQUIRE(lig_ptr && kp_ptr && lig_x && lig_h && kp_x && eps_x && eps_h && noise_x && noise_h && coef,
                KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(B >= 1 && max_lig >= 1 && max_lig <= 4096, KPD_ERR_INVALID, "B=%d max_lig=%d", B, max_lig);
    hipLaunchKernelGGL(k_sample_update, dim3(B), dim3(256), (size_t)max_lig * 3 * sizeof(float),
                       static_cast<hipStream_t>(stream), lig_ptr, kp_ptr, atom_nf, lig_x, lig_h, kp_x, eps_x, eps_h,
                       noise_x, noise_h, coef);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}
