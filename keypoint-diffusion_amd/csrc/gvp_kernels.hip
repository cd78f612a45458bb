// GVP denoiser kernels (models/gvp.py GVP :89-116, GVPLayerNorm :159-166, GVPMultiEdgeConv
// :459-551; models/dynamics_gvp.py NoisePredictionBlock :38-44, LigRecDynamicsGVP :149-199)
// on the fp32-MFMA row-tile core.
//
// A "GVP stage" on a 64-row tile (rows = edges or nodes) keeps the scalar inputs in the LDS A tile
// and the vector inputs in an LDS vector buffer, and runs
//   vec1  Vh = Wh^T v (per xyz),  sh = |Vh|  -> appended to the A tile behind the scalars
//   GEMM  s' = act(W [s, sh] + b [+ gathered per-row term])          (32x32x2 f32 MFMA)
//   gate  g  = Wg s' + bg                                            (16x16x4 f32 MFMA)
//   vec2  v' = sigmoid(g) * (Wu^T Vh)
// with s' written back over the A tile and v' over the vector buffer, ready for the next stage.
// The first Linear of an edge message is linear in h_src, so its 256-wide block is applied once
// per source node (k_gvp_proj) and enters the edge stage as the gathered per-row term.
#include <stdlib.h>

#include <algorithm>

#include "gvp_kernels.h"
#include "mfma_core.h"

namespace kpd {

typedef float f32x4_ __attribute__((ext_vector_type(4)));

struct GvpSmem {
    float *A;                 // [64][SA_G]
    float *V0, *V1, *V2;      // [64][VST]: current vectors, hidden vectors, residual vectors
    int *src, *dst;           // [64]
    float *rowf;              // [128] per-row scratch (LN statistics)
    int *misc;                // [16]
};

__device__ __forceinline__ GvpSmem gvp_smem(float *smem) {
    GvpSmem s;
    s.A = smem;
    s.V0 = s.A + TM * SA_G;
    s.V1 = s.V0 + TM * VST;
    s.V2 = s.V1 + TM * VST;
    s.src = reinterpret_cast<int *>(s.V2 + TM * VST);
    s.dst = s.src + TM;
    s.rowf = reinterpret_cast<float *>(s.dst + TM);
    s.misc = reinterpret_cast<int *>(s.rowf + 2 * TM);
    return s;
}

// ---- one GVP stage ------------------------------------------------------------------------
// (node-update and noise-head GVPs; the edge-message chains run in gvp_chain.hip)
__device__ __forceinline__ void gvp_stage(const GvpSmem &s, const GvpW &w, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    // 16x16x4 MFMA roles of this lane: rows (edges / nodes) 16 wave + (lane & 15) as A operand, output
    // column lane & 15, output rows 16 wave + 4 (lane >> 4) + reg.  Every wave works on its own 16 rows in the
    // vector stages and the gate GEMM, so those need no workgroup barrier between them.
    constexpr int KSMAX = (GVH + 3) / 4, NTMAX = (GVH + 15) / 16;
    const int lrow = 16 * wave + (lane & 15), kq = lane >> 4, ncol = lane & 15;
    const int orow0 = 16 * wave + 4 * kq;

    // vec1: Vh[(c, e)][h] = sum_v v[e][v][c] Wh[v][h] as three 16-row MFMA products (one per xyz component, so
    // that the three components of one Vh entry sit at the same register of three accumulators);
    // sh[h] = sqrt(max(|Vh[h]|^2, 1e-8)) goes behind the scalars in the A tile            (gvp.py:96-99)
    {
        const int ksn = (w.vin + 3) >> 2, ntn = (w.h + 15) >> 4;
        float av[3][KSMAX], bv[NTMAX][KSMAX];
        const float *vin = s.V0 + lrow * VST;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int k = 4 * ks + kq;
            if (ks < ksn) {
                const bool kin = k < w.vin;
#pragma unroll
                for (int c = 0; c < 3; ++c) av[c][ks] = kin ? vin[3 * k + c] : 0.0f;
#pragma unroll
                for (int nt = 0; nt < NTMAX; ++nt)
                    if (nt < ntn) {
                        const int h = 16 * nt + ncol;
                        bv[nt][ks] = (kin && h < w.h) ? w.Wh[k * w.h + h] : 0.0f;
                    }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTMAX; ++nt) {
            if (nt < ntn) {
                f32x4_ acc[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = f32x4_{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KSMAX; ++ks)
                    if (ks < ksn) {
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][ks], bv[nt][ks], acc[c], 0, 0, 0);
                    }
                const int h = 16 * nt + ncol;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int e = orow0 + reg;
                    if (h < w.h) {
                        float *o = s.V1 + e * VST + 3 * h;
                        o[0] = acc[0][reg]; o[1] = acc[1][reg]; o[2] = acc[2][reg];
                        const float n2 = acc[0][reg] * acc[0][reg] + acc[1][reg] * acc[1][reg] + acc[2][reg] * acc[2][reg];
                        s.A[e * SA_G + w.n_s + h] = sqrtf(fmaxf(n2, 1e-8f));
                    } else if (w.n_s + h < 8 * w.ng) {
                        s.A[e * SA_G + w.n_s + h] = 0.0f;   // K padding: the packed weight rows are zero, A must be finite
                    }
                }
            }
        }
    }
    lds_barrier();

    // GEMM + activation -> A tile columns 0..255
    {
        f32x16 acc[2][2];
        acc_zero(acc);
        gemm_rows64_rt<SA_G>(s.A, w.wp, w.ng, acc, wave, lane);
        lds_barrier();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = acc_col(nt, wave, lane);
            const float bb = w.b[col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float val = acc[mt][nt][reg] + bb;
                    s.A[acc_row(mt, reg, lane) * SA_G + col] = silu(val);
                }
        }
    }
    lds_barrier();

    // gates (gvp.py:105-107): 16 rows per wave, K = sout; the result stays in registers -- its layout
    // (column u on the lane, rows in the 4 registers) is exactly that of the vec2 product below
    f32x4_ gate;
    {
        f32x4_ c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        const float *ap = s.A + lrow * SA_G + 4 * kq;
        const f32x4_ *bp = reinterpret_cast<const f32x4_ *>(w.wg) + lane;
        const int groups = w.sout >> 4;
        f32x4_ b_next = bp[0];
#pragma unroll 4
        for (int g = 0; g < groups; ++g) {
            const f32x4_ a = *reinterpret_cast<const f32x4_ *>(ap + 16 * g);
            const f32x4_ b = b_next;
            b_next = bp[(g + 1 < groups ? g + 1 : g) * 64];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c1, 0, 0, 0);
        }
        gate = c0 + c1 + w.bg[ncol];
        if (w.vec_sigmoid) {
            gate[0] = sigmoidf_(gate[0]); gate[1] = sigmoidf_(gate[1]);
            gate[2] = sigmoidf_(gate[2]); gate[3] = sigmoidf_(gate[3]);
        }
    }

    // vec2: v'[e][u][c] = act(gate[e][u]) * sum_h Vh[e][h][c] Wu[h][u]  (gvp.py:97, 111), same MFMA shape
    {
        const int ksn = (w.h + 3) >> 2;
        float av[3][KSMAX], bv[KSMAX];
        const float *vh = s.V1 + lrow * VST;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks)
            if (ks < ksn) {
                const int k = 4 * ks + kq;
                const bool kin = k < w.h;
#pragma unroll
                for (int c = 0; c < 3; ++c) av[c][ks] = kin ? vh[3 * k + c] : 0.0f;
                bv[ks] = (kin && ncol < w.vout) ? w.Wu[k * w.vout + ncol] : 0.0f;
            }
        f32x4_ acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = f32x4_{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks)
            if (ks < ksn) {
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][ks], bv[ks], acc[c], 0, 0, 0);
            }
        if (ncol < w.vout) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float *o = s.V0 + (orow0 + reg) * VST + 3 * ncol;
                o[0] = gate[reg] * acc[0][reg]; o[1] = gate[reg] * acc[1][reg]; o[2] = gate[reg] * acc[2][reg];
            }
        }
    }
    lds_barrier();
}

// LayerNorm over the first S columns of every A-tile row (affine), in place.  (gvp.py:161)
__device__ __forceinline__ void tile_layernorm(const GvpSmem &s, int S, const float *__restrict__ lw,
                                               const float *__restrict__ lb, int tid) {
    const int row = tid >> 2, q = tid & 3;
    float *tr = s.A + row * SA_G;
    float sum = 0.0f;
    for (int c = q; c < S; c += 4) sum += tr[c];
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    const float mean = sum / (float)S;
    float var = 0.0f;
    for (int c = q; c < S; c += 4) {
        const float d = tr[c] - mean;
        var = fmaf(d, d, var);
    }
    var += __shfl_xor(var, 1);
    var += __shfl_xor(var, 2);
    const float rstd = 1.0f / sqrtf(var / (float)S + 1e-5f);
    for (int c = q; c < S; c += 4) tr[c] = (tr[c] - mean) * rstd * lw[c] + lb[c];
}

// Vector half of GVPLayerNorm (gvp.py:163-165): v / (sqrt(mean_i max(|v_i|^2, 1e-8) + eps) + eps), in place on V0.
__device__ __forceinline__ void tile_vecnorm(const GvpSmem &s, int tid) {
    const int row = tid >> 2, q = tid & 3;
    float *v = s.V0 + row * VST;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float *p = v + 3 * (q + 4 * i);
        acc += fmaxf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2], 1e-8f);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    const float vn = sqrtf(acc * (1.0f / GV) + 1e-5f) + 1e-5f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float *p = v + 3 * (q + 4 * i);
        p[0] /= vn; p[1] /= vn; p[2] /= vn;
    }
}

// ---- encoders (dynamics_gvp.py:124-134, 161-169): out = LN(SiLU(W [h, t] + b)) --------------
constexpr int GEMB_NODES = 4;
__global__ __launch_bounds__(256) void k_gvp_embed(const float *__restrict__ in, int n, int fin,
                                                   const float *__restrict__ W, const float *__restrict__ b,
                                                   const float *__restrict__ lw, const float *__restrict__ lb,
                                                   const float *__restrict__ t, const int *__restrict__ bidx, int S,
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
    const bool on = tid < S;
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
        mean[j] = (s_red[j][0][0] + s_red[j][0][1] + s_red[j][0][2] + s_red[j][0][3]) / (float)S;
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
        rstd[j] = 1.0f / sqrtf((s_red[j][1][0] + s_red[j][1][1] + s_red[j][1][2] + s_red[j][1][3]) / (float)S + 1e-5f);
        const int v = node0 + j;
        if (on && v < n) out[(size_t)v * S + tid] = (y[j] - mean[j]) * rstd[j] * lw[tid] + lb[tid];
    }
}

// ---- noise prediction block (dynamics_gvp.py:38-44) ----------------------------------------------
__global__ __launch_bounds__(256) void k_gvp_noise(GvpNoiseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GvpSmem s = gvp_smem(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int node0 = blockIdx.x * TM;
    const int S = a.S, chunks = S >> 2;
    for (int rr = 0; rr < 16; ++rr) {
        const int r = wave * 16 + rr, v = node0 + r;
        for (int c = lane; c < chunks; c += 64) {
            f32x4_ val = {0.f, 0.f, 0.f, 0.f};
            if (v < a.n) val = reinterpret_cast<const f32x4_ *>(a.s + (size_t)v * S)[c];
            *reinterpret_cast<f32x4_ *>(s.A + r * SA_G + 4 * c) = val;
        }
        if (lane < 48) s.V0[r * VST + lane] = v < a.n ? a.v[(size_t)v * 48 + lane] : 0.0f;
    }
    lds_barrier();
    for (int k = 0; k < a.n_gvps; ++k) gvp_stage(s, a.g[k], tid);
    // eps_h = W_out s (64 -> F), eps_x = the single output vector
    const int row = tid >> 2, q = tid & 3, v = node0 + row;
    if (v < a.n) {
        for (int f = q; f < a.F; f += 4) {
            float acc = a.bout[f];
            for (int k = 0; k < 64; ++k) acc = fmaf(a.Wout[f * 64 + k], s.A[row * SA_G + k], acc);
            a.eps_h[(size_t)v * a.F + f] = acc;
        }
        if (q < 3) a.eps_x[(size_t)v * 3 + q] = s.V0[row * VST + q];
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
static bool g_gvp_attr = false;

kpd_status gvp_kernels_init() {
    if (g_gvp_attr) return KPD_OK;
    KPD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gvp_noise), hipFuncAttributeMaxDynamicSharedMemorySize,
                                GVP_LDS_BYTES));
    g_gvp_attr = true;
    return KPD_OK;
}

kpd_status launch_gvp_embed(const float *in, int n, int fin, const float *W, const float *b, const float *ln_w,
                            const float *ln_b, const float *t, const int *bidx, int S, float *out, hipStream_t st) {
    if (n == 0) return KPD_OK;
    KPD_REQUIRE(fin + 1 <= 260 && S <= 256, KPD_ERR_INVALID, "gvp embed: fin=%d S=%d", fin, S);
    hipLaunchKernelGGL(k_gvp_embed, dim3(cdiv(n, GEMB_NODES)), dim3(256), 0, st, in, n, fin, W, b, ln_w, ln_b, t, bidx, S, out);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_gvp_noise(const GvpNoiseArgs &a, hipStream_t st) {
    if (a.n == 0) return KPD_OK;
    hipLaunchKernelGGL(k_gvp_noise, dim3(cdiv(a.n, TM)), dim3(256), GVP_LDS_BYTES, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd
