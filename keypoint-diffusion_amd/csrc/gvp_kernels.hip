// GVP denoiser: scalar encoders (models/dynamics_gvp.py:124-134, 161-169).  The GVP stacks themselves -- edge
// messages, node updates, per-node blocks of the first message Linear, the noise head -- are the register-chained
// kernels of gvp_chain.hip.
#include <stdlib.h>

#include <algorithm>

#include "gvp_kernels.h"
#include "mfma_core.h"

namespace kpd {

// ---- encoders (dynamics_gvp.py:124-134, 161-169): out = LN(SiLU(W [h, t] + b)) --------------
constexpr int GEMB_NODES = 4;
__global__ __launch_bounds__(256) void k_gvp_embed(const float *__restrict__ in, int n, int fin,
                                                   const float *__restrict__ W, const float *__restrict__ b,
                                                   const float *__restrict__ lw, const float *__restrict__ lb,
                                                   const float *__restrict__ t, const int *__restrict__ bidx, int S, int St,
                                                   float *__restrict__ out) {
    __shared__ float s_in[GEMB_NODES][260];
    __shared__ float s_red[GEMB_NODES][2][4];
    const int node0 = blockIdx.x * GEMB_NODES, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < GEMB_NODES * (fin + 1); i += 256) {
        const int j = i / (fin + 1), k = i - j * (fin + 1);
        const int v = node0 + j;
        s_in[j][k] = v < n ? (k < fin ? in[(size_t)v * fin + k] : t[bidx[v]]) : 0.0f;
    }
    __syncthreads();
    float y[GEMB_NODES];
    // S: row width of `out`; St <= S: the model's features (the rest is padding: zero weight rows, excluded from the statistics)
    const bool on = tid < St, row = tid < S;
#pragma unroll
    for (int j = 0; j < GEMB_NODES; ++j) y[j] = on ? b[tid] : 0.0f;
    if (on)
        for (int k = 0; k <= fin; ++k) {
            const float wv = W[(size_t)tid * (fin + 1) + k];
#pragma unroll
            for (int j = 0; j < GEMB_NODES; ++j) y[j] = fmaf(wv, s_in[j][k], y[j]);
        }
#pragma unroll
    for (int j = 0; j < GEMB_NODES; ++j) y[j] = on ? silu(y[j]) : 0.0f;
    // LayerNorm over the S features of each node: two block reductions
    float mean[GEMB_NODES], rstd[GEMB_NODES];
#pragma unroll
    for (int j = 0; j < GEMB_NODES; ++j) {
        float v = y[j];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) s_red[j][0][wave] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GEMB_NODES; ++j)
        mean[j] = (s_red[j][0][0] + s_red[j][0][1] + s_red[j][0][2] + s_red[j][0][3]) / (float)St;
#pragma unroll
    for (int j = 0; j < GEMB_NODES; ++j) {
        const float d = on ? y[j] - mean[j] : 0.0f;
        float v = d * d;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) s_red[j][1][wave] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GEMB_NODES; ++j) {
        rstd[j] = 1.0f / sqrtf((s_red[j][1][0] + s_red[j][1][1] + s_red[j][1][2] + s_red[j][1][3]) / (float)St + 1e-5f);
        const int v = node0 + j;
        if (row && v < n) out[(size_t)v * S + tid] = on ? (y[j] - mean[j]) * rstd[j] * lw[tid] + lb[tid] : 0.0f;
    }
}

kpd_status launch_gvp_embed(const float *in, int n, int fin, const float *W, const float *b, const float *ln_w,
                            const float *ln_b, const float *t, const int *bidx, int S, int S_true, float *out, hipStream_t st) {
    if (n == 0) return KPD_OK;
    KPD_REQUIRE(fin + 1 <= 260 && S <= 256 && S_true >= 1 && S_true <= S, KPD_ERR_INVALID, "gvp embed: fin=%d S=%d", fin, S);
    hipLaunchKernelGGL(k_gvp_embed, dim3(cdiv(n, GEMB_NODES)), dim3(256), 0, st, in, n, fin, W, b, ln_w, ln_b, t, bidx, S, S_true, out);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd
