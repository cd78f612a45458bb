// Weight-stationary tall-skinny GEMM for the training engines:  Y[rows, 257] = epilogue(X[rows, 257] op(W) (+ b)).
//
// The per-edge products of the EGNN training path (pre2 = a1 W2^T, da1 = dpre2 W2: hundreds of thousands of rows against
// one 257 x 257 matrix) have the shape k_proj_ws was built for (egnn_chain.hip): a workgroup keeps half of the 256 x 256
// block (128 outputs x 256 inputs = 128 KB of MFMA A-fragments) resident in LDS and walks 128-row tiles with eight
// independent waves, rows in B-operand registers, no barrier and no weight refill inside the GEMM; the 257th input /
// output are a VALU rank-1 update / dot product.  The weights change every optimizer step, so they are packed per call
// (one launch: fragments of W or W^T by strides, column 256, row 256), and the elementwise work that followed the
// library GEMM is fused into the epilogue:
//   WS_BIAS_SILU   : Y = X W^T + b (kept: the pre-activation), A = SiLU(Y)
//   WS_SILU_BWD    : Y = (X W) * SiLU'(P)            (P: the saved pre-activation of the previous Linear)
//   WS_PLAIN       : Y = X op(W)
// `accumulate` adds what Y already holds before the epilogue (a partial pre-activation from another, narrow product);
// `has257 = false` is the 256-wide form used by the GVP engine (no 257th input / output, rows may be 256 floats apart).
// Narrow extras of that form (a GVP's to_feats_out reads [s | vector norms]: 256 + 16 or 17 inputs, gvp.py:100-106), has257 = false only:
//   extra inputs  (WsgExtra::X2, WS_BIAS_SILU): Y = [X | X2] [W | W2]^T + b -- a 17th resident k-block of A-fragments, X2's columns 0..15 as one
//                 more B operand, its column 16 (if any) through the rank-1 path of the 257th input;
//   extra outputs (WsgExtra::Y2, WS_PLAIN): Y2 = X W2, 16 columns as a ninth output tile of the second half's workgroups, a 17th through
//                 the row-dot path of the 257th output.
//   gates (WsgExtra::G2, with extra inputs): G2 / G2b[row][0..15] = the two halves' shares of A Wg^T, the product of the ACTIVATED output with a
//                 16 x 256 matrix (scalar_to_vector_gates, gvp.py:108-111): the epilogue's registers are already the B operand of a
//                 chained 16x16x4 MFMA, so it costs 32 MFMAs per 16 rows; the consumer adds the two halves and the bias.
// Each replaces a separate rows x 256 x 16 library product and its pass over a rows x 256 matrix.
#include "chain_core.h"
#include "engine.h"

namespace kpd {

namespace {

constexpr int WSG_TILE = 128;
constexpr int WSG_W4 = 16 * 8 * 64;                           // float4 of the resident half block
constexpr int WSG_LDS_BYTES = WSG_W4 * 16 + (128 + 128 + HS + HS) * 4;      // half block, column 256, bias, row 256, row-dot vector
constexpr int WSG_X4 = 16 * 64;                               // float4 of the extra fragments: one k-block of this half's 8 output tiles (extra inputs:
                                                              // 8 x 64, padded) or 16 k-blocks of one output tile (extra outputs)
constexpr int WSG_G4 = 8 * 64;                                // float4 of the gate fragments of one half (8 k-blocks of 16 columns)
constexpr int WSG_LDS_BYTES_X = WSG_LDS_BYTES + WSG_X4 * 16 + WSG_G4 * 16;
constexpr int WSG_PACK_EXT = 16 * 16 * 64 * 4 + 256 + HS;     // offset of the extra fragments in the pack
constexpr int WSG_PACK_GATE = WSG_PACK_EXT + 16 * 256;        // offset of the gate fragments (16 k-blocks x 64 lanes x 4)
constexpr int WSG_PACK_FLOATS = WSG_PACK_GATE + 16 * 256;     // fragments, column 256, row 256 (padded), extra fragments, gate fragments

__device__ __forceinline__ float wsg_sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }      // (1 ulp; the division sequence cost 12 % of the kernel)

// fragments of the 256 x 256 block of M (M[n][k] = src[n * sn + k * sk]) in chain-chunk order, then M[:, 256], then M[256, :]
// ext_mode 1 (extra inputs kx < ext_n <= 17, E[n][kx] = ext[n * ext_sn + kx * ext_sk]): the fragments of k-block 16 (inputs 0..15) and, in
// the slot of M[:, 256], E[:, 16];  ext_mode 2 (extra outputs nx < ext_n <= 17, E[nx][k] = ext[nx * ext_sn + k * ext_sk]): the fragments of
// output tile 16 for the 16 k-blocks and, in the slot of M[256, :], E[16, :]
// gate (optional): Wg [gate_n <= 16][256] (row stride gate_ld): fragments of output tile "gates" for the 16 k-blocks of the activated output
__global__ void k_wsg_pack(const float *__restrict__ src, int sn, int sk, int has257, float *__restrict__ dst, const float *__restrict__ ext,
                           int ext_sn, int ext_sk, int ext_mode, int ext_n, const float *__restrict__ gate, int gate_ld, int gate_n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 16 * 16 * 256) {
        const int r = idx & 3, lane = (idx >> 2) & 63, mt = (idx >> 8) & 15, ks = idx >> 12;
        const int n = 16 * mt + (lane & 15), k = 16 * ks + 4 * (lane >> 4) + r;
        dst[idx] = src[(size_t)n * sn + (size_t)k * sk];
    } else if (idx < 16 * 16 * 256 + 256) {
        const int n = idx - 16 * 16 * 256;
        dst[idx] = has257 ? src[(size_t)n * sn + (size_t)256 * sk] : (ext_mode == 1 && ext_n > 16) ? ext[(size_t)n * ext_sn + (size_t)16 * ext_sk] : 0.0f;
    } else if (idx < WSG_PACK_EXT) {
        const int k = idx - 16 * 16 * 256 - 256;
        dst[idx] = (has257 && k <= 256) ? src[(size_t)256 * sn + (size_t)k * sk]
                                        : (ext_mode == 2 && ext_n > 16 && k < 256) ? ext[(size_t)16 * ext_sn + (size_t)k * ext_sk] : 0.0f;
    } else if (idx >= WSG_PACK_GATE && idx < WSG_PACK_FLOATS) {
        const int i = idx - WSG_PACK_GATE, r = i & 3, lane = (i >> 2) & 63, kb = i >> 8;
        const int n = lane & 15, k = 16 * kb + 4 * (lane >> 4) + r;
        dst[idx] = (gate && n < gate_n) ? gate[(size_t)n * gate_ld + k] : 0.0f;
    } else if (idx < WSG_PACK_GATE) {
        const int i = idx - WSG_PACK_EXT, r = i & 3, lane = (i >> 2) & 63, t = i >> 8;       // t: output tile (mode 1) / k-block (mode 2)
        float v = 0.0f;
        if (ext_mode == 1) {
            const int n = 16 * t + (lane & 15), kx = 4 * (lane >> 4) + r;
            if (kx < ext_n) v = ext[(size_t)n * ext_sn + (size_t)kx * ext_sk];
        } else if (ext_mode == 2) {
            const int nx = lane & 15, k = 16 * t + 4 * (lane >> 4) + r;
            if (nx < ext_n) v = ext[(size_t)nx * ext_sn + (size_t)k * ext_sk];
        }
        dst[idx] = v;
    }
}

struct WsgArgs {
    const float *X;         // [rows, ldx], 257 columns used
    int rows, ldx;
    const float *pack;      // WSG_PACK_FLOATS as written by k_wsg_pack
    const float *bias;      // [257] or null
    const float *P;         // [rows, ldy] pre-activation for WS_SILU_BWD, else null
    float *Y, *A;           // [rows, ldy]; A only for WS_BIAS_SILU
    int ldy, mode, bpc, tpb, has257, accumulate;
    const float *rd_w;      // WS_SILU_BWD: optional row-dot vector (stride rd_stride) and its output [2][rows]
    int rd_stride;
    float *rd_out;
    const float *X2;        // extra inputs [rows, ldx2], x2n <= 17 columns (EXT = 1)
    float *Y2;              // extra outputs [rows, ldy2], y2n <= 17 columns (EXT = 2)
    int ldx2, x2n, ldy2, y2n;
    float *G2, *G2b;        // [rows][16] gate shares of the two halves (EXT = 1, optional)
};

template <int MODE, int EXT>
__device__ __forceinline__ void wsg_store(const WsgArgs &a, int row, int hf, int q, v4f (&acc)[8], float out256, const float *s_rd, const v4f &acc_e,
                                          const v4f *Wgf, int lane) {
    if (EXT == 1) {
        // (no accumulate, no 257th output, no row-dot in this form.)  The activation first, for every lane -- the gate product is an MFMA and
        // wants the whole wave -- then the stores of the valid rows.
        v4f sv[8], ga = zero4();
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) sv[m][r] = acc[m][r] * wsg_sigm(acc[m][r]);
        if (a.G2) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const v4f wg = Wgf[m * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) ga = mfma16(wg[r], sv[m][r], ga);
            }
        }
        if (row < 0 || row >= a.rows) return;
        float *yrow = a.Y + (size_t)row * a.ldy + 128 * hf, *arow = a.A + (size_t)row * a.ldy + 128 * hf;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            *reinterpret_cast<v4f *>(yrow + 16 * m + 4 * q) = acc[m];
            *reinterpret_cast<v4f *>(arow + 16 * m + 4 * q) = sv[m];
        }
        if (a.G2) *reinterpret_cast<v4f *>((hf ? a.G2b : a.G2) + (size_t)row * 16 + 4 * q) = ga;
        return;
    }
    if (row < 0 || row >= a.rows) return;             // (the four lanes of a row leave together: the shuffles below stay converged)
    if (EXT == 2) {
        float *y2 = a.Y2 + (size_t)row * a.ldy2;
        if (hf == 1) {
            if ((a.ldy2 & 3) == 0) *reinterpret_cast<v4f *>(y2 + 4 * q) = acc_e;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) y2[4 * q + r] = acc_e[r];
            }
        } else if (q == 0 && a.y2n > 16) y2[16] = out256;
    }
    float *yrow = a.Y + (size_t)row * a.ldy + 128 * hf;
    const bool tail = a.has257 && hf == 0 && q == 0;
    if (a.accumulate) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] += *reinterpret_cast<const v4f *>(yrow + 16 * m + 4 * q);
        if (tail) out256 += yrow[256];
    }
    if (MODE == WS_BIAS_SILU) {
        float *arow = a.A + (size_t)row * a.ldy + 128 * hf;
        float dot = 0.0f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const v4f v = acc[m];
            v4f s;
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = v[r] * wsg_sigm(v[r]);
            *reinterpret_cast<v4f *>(yrow + 16 * m + 4 * q) = v;
            *reinterpret_cast<v4f *>(arow + 16 * m + 4 * q) = s;
            if (a.rd_out) {
                const v4f w = *reinterpret_cast<const v4f *>(s_rd + 128 * hf + 16 * m + 4 * q);
                dot += s[0] * w[0] + s[1] * w[1] + s[2] * w[2] + s[3] * w[3];
            }
        }
        if (tail) {
            const float s256 = out256 * wsg_sigm(out256);
            yrow[256] = out256;
            arow[256] = s256;
            if (a.rd_out) dot += s256 * s_rd[256];
        }
        if (a.rd_out) {                                // this half's share of the activated row's product with rd_w (a head of the MLP)
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            if (q == 0) a.rd_out[(size_t)hf * a.rows + row] = dot;
        }
    } else if (MODE == WS_SILU_BWD) {
        const float *prow = a.P + (size_t)row * a.ldy + 128 * hf;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const v4f p = *reinterpret_cast<const v4f *>(prow + 16 * m + 4 * q);
            v4f v = acc[m];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sg = wsg_sigm(p[r]);
                v[r] *= sg * (1.0f + p[r] * (1.0f - sg));
            }
            *reinterpret_cast<v4f *>(yrow + 16 * m + 4 * q) = v;
            acc[m] = v;
        }
        float y256 = 0.0f;
        if (tail) {
            const float p = prow[256], sg = wsg_sigm(p);
            y256 = out256 * sg * (1.0f + p * (1.0f - sg));
            yrow[256] = y256;
        }
        if (a.rd_out) {                                // this half's share of the row's product with rd_w, from the finished registers
            float dot = tail ? y256 * s_rd[256] : 0.0f;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const v4f w = *reinterpret_cast<const v4f *>(s_rd + 128 * hf + 16 * m + 4 * q);
                dot += acc[m][0] * w[0] + acc[m][1] * w[1] + acc[m][2] * w[2] + acc[m][3] * w[3];
            }
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            if (q == 0) a.rd_out[(size_t)hf * a.rows + row] = dot;
        }
    } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) *reinterpret_cast<v4f *>(yrow + 16 * m + 4 * q) = acc[m];
        if (tail) yrow[256] = out256;
    }
}

template <int MODE, int EXT>
__global__ __launch_bounds__(512, 1) void k_ws_gemm(WsgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    v4f *W = reinterpret_cast<v4f *>(smem);
    float *s_wcol = smem + WSG_W4 * 4, *s_bias = s_wcol + 128, *s_wrow = s_bias + 128, *s_rd = s_wrow + HS;
    [[maybe_unused]] v4f *Wx = reinterpret_cast<v4f *>(s_rd + HS);      // EXT: the extra fragments
    [[maybe_unused]] v4f *Wgf = Wx + WSG_X4;                            // EXT = 1: the gate fragments of this half
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int hf = blockIdx.x / a.bpc, chunk = blockIdx.x - hf * a.bpc;
    const int tiles = (a.rows + WSG_TILE - 1) / WSG_TILE;
    const int t0 = chunk * a.tpb, t1 = min(tiles, t0 + a.tpb);
    if (t0 >= t1) return;                                     // uniform over the workgroup
    {
        const v4f *src = reinterpret_cast<const v4f *>(a.pack);
#pragma unroll
        for (int j = 0; j < WSG_W4 / 512; ++j) {
            const int i = tid + 512 * j, ks = i >> 9, rem = i & 511;
            W[i] = src[(size_t)ks * 1024 + 8 * hf * 64 + rem];
        }
        const float *wcol = a.pack + 16 * 16 * 256, *wrow = wcol + 256;
        if (tid < 128) {
            s_wcol[tid] = wcol[128 * hf + tid];
            s_bias[tid] = a.bias ? a.bias[128 * hf + tid] : 0.0f;
        }
        for (int i = tid; i < HS; i += 512) s_wrow[i] = wrow[i];
        if (EXT == 1) {
            Wx[tid] = reinterpret_cast<const v4f *>(a.pack + WSG_PACK_EXT)[8 * hf * 64 + tid];                            // 8 output tiles x 64 lanes
            Wgf[tid] = reinterpret_cast<const v4f *>(a.pack + WSG_PACK_GATE)[8 * hf * 64 + tid];                          // 8 k-blocks x 64 lanes
        }
        if (EXT == 2) {
            Wx[tid] = reinterpret_cast<const v4f *>(a.pack + WSG_PACK_EXT)[tid];                                         // 16 k-blocks x 64 lanes
            Wx[512 + tid] = reinterpret_cast<const v4f *>(a.pack + WSG_PACK_EXT)[512 + tid];
        }
        if (MODE != WS_PLAIN && a.rd_out)
            for (int i = tid; i < HS; i += 512) s_rd[i] = i < (a.has257 ? 257 : 256) ? a.rd_w[(size_t)i * a.rd_stride] : 0.0f;
    }
    __syncthreads();
    const int el = lane & 15, q = lane >> 4;
    const float bias256 = (a.bias && a.has257) ? a.bias[256] : 0.0f;
    constexpr int NX = EXT == 1 ? 17 : 16;                    // B-operand registers of a row: its 16 k-blocks (+ the extra inputs)
    auto load_x = [&](int t, v4f (&x)[NX], float &x256) {
        const int row = min(t * WSG_TILE + 16 * wave + el, a.rows - 1);
        const float *xrow = a.X + (size_t)row * a.ldx;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) x[nt] = *reinterpret_cast<const v4f *>(xrow + 16 * nt + 4 * q);
        x256 = a.has257 ? xrow[256] : 0.0f;
        if constexpr (EXT == 1) {
            const float *x2 = a.X2 + (size_t)row * a.ldx2;
            if ((a.ldx2 & 3) == 0) x[16] = *reinterpret_cast<const v4f *>(x2 + 4 * q);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) x[16][r] = x2[4 * q + r];
            }
            if (a.x2n > 16) x256 = x2[16];
        }
    };
    v4f x[NX], xn[NX], accp[8];
    [[maybe_unused]] v4f acc_ep = zero4();
    float x256, x256n, out256p = 0.0f;
    int rowp = -1;
#pragma unroll
    for (int m = 0; m < 8; ++m) accp[m] = zero4();
    load_x(t0, xn, x256n);
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        // see k_proj_ws: x is only written by these moves, so the compiler's wait counts the row loads alone and the previous
        // tile's stores (issued right after) never sit in front of it in the in-order memory counter
#pragma unroll
        for (int nt = 0; nt < NX; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(x[nt][r]) : "v"(xn[nt][r]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(x256) : "v"(x256n));
        __builtin_amdgcn_sched_barrier(0);
        wsg_store<MODE, EXT>(a, rowp, hf, q, accp, out256p, s_rd, acc_ep, Wgf, lane);
        if (t + 1 < t1) load_x(t + 1, xn, x256n);
        v4f acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m)
            acc[m] = x256 * *reinterpret_cast<const v4f *>(s_wcol + 16 * m + 4 * q) + *reinterpret_cast<const v4f *>(s_bias + 16 * m + 4 * q);
        float part = 0.0f;
        const bool row_dot = hf == 0 && (a.has257 || (EXT == 2 && a.y2n > 16));      // the 257th output / the 17th extra output
        if (row_dot) {
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) {
                const v4f wv = *reinterpret_cast<const v4f *>(s_wrow + 16 * nt + 4 * q);
                part += x[nt][0] * wv[0] + x[nt][1] * wv[1] + x[nt][2] * wv[2] + x[nt][3] * wv[3];
            }
        }
        const v4f *wp = W + lane;
        v4f w[2][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) w[0][m] = wp[m * 64];
        constexpr int NB = 2 * NX;                            // half k-blocks: 4 output tiles each
        [[maybe_unused]] v4f acc_e = zero4(), we = zero4();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 1 < NB) {
                const v4f *wn = b + 1 < 32 ? wp + 4 * (b + 1) * 64 : Wx + lane + 4 * (b + 1 - 32) * 64;
#pragma unroll
                for (int m = 0; m < 4; ++m) w[(b + 1) & 1][m] = wn[m * 64];
            }
            if (EXT == 2 && (b & 1) == 0 && hf == 1) we = Wx[(b >> 1) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
            const v4f xin = x[b >> 1];
            const int g = b & 1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[4 * g + m] = mfma16(w[b & 1][m][r], xin[r], acc[4 * g + m]);
            if (EXT == 2 && (b & 1) == 0 && hf == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_e = mfma16(we[r], xin[r], acc_e);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float out256 = 0.0f;
        if (row_dot) {
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            out256 = part + bias256 + x256 * s_wrow[256];
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) accp[m] = acc[m];
        if (EXT == 2) acc_ep = acc_e;
        rowp = t * WSG_TILE + 16 * wave + el;
        out256p = out256;
    }
    wsg_store<MODE, EXT>(a, rowp, hf, q, accp, out256p, s_rd, acc_ep, Wgf, lane);
}

}  // namespace

int ws_gemm_pack_floats() { return WSG_PACK_FLOATS; }

// Y = epilogue(X op(W) + b): W with row stride ldw; transpose_w = false: Y = X W^T (W in the torch [out, in] layout), true:
// Y = X W.  has257: 257 x 257 problem (else 256 x 256).  pack_scratch: ws_gemm_pack_floats() floats.
// ext (has257 = false only): extra inputs X2 (WS_BIAS_SILU, no accumulate) or extra outputs Y2 (WS_PLAIN); ext.W addresses element
// (n, k) of the narrow block as ext.W[n * ext.sn + k * ext.sk] with n the output and k the input index of "Y = X M^T".
kpd_status ws_gemm(int mode, const float *X, int rows, int ldx, const float *W, int ldw, bool transpose_w, const float *bias,
                   const float *P, float *Y, float *A, int ldy, float *pack_scratch, hipStream_t st, bool has257, bool accumulate,
                   const float *rowdot_w, int rowdot_stride, float *rowdot_out, const WsgExtra *ext) {
    if (rows == 0) return KPD_OK;
    const int need = has257 ? 260 : 256;
    KPD_REQUIRE(X && W && Y && pack_scratch && (ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= need && ldy >= need, KPD_ERR_INVALID,
                "ws_gemm: bad operands");
    KPD_REQUIRE((mode == WS_BIAS_SILU && A) || (mode == WS_SILU_BWD && P) || mode == WS_PLAIN, KPD_ERR_INVALID, "ws_gemm: mode %d operands",
                mode);
    const int em = !ext ? 0 : ext->X2 ? 1 : 2;
    if (ext)
        KPD_REQUIRE(!has257 && ext->W && ext->n >= 1 && ext->n <= 17 && ((em == 1 && mode == WS_BIAS_SILU && !accumulate && ext->ld >= ext->n) ||
                                                                          (em == 2 && mode == WS_PLAIN && ext->Y2 && ext->ld >= ext->n)),
                    KPD_ERR_INVALID, "ws_gemm: extra block does not fit the mode");
    static bool lds_set = false;
    if (!lds_set) {
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_ws_gemm<WS_BIAS_SILU, 0>), WSG_LDS_BYTES));
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_ws_gemm<WS_SILU_BWD, 0>), WSG_LDS_BYTES));
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_ws_gemm<WS_PLAIN, 0>), WSG_LDS_BYTES));
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_ws_gemm<WS_BIAS_SILU, 1>), WSG_LDS_BYTES_X));
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_ws_gemm<WS_PLAIN, 2>), WSG_LDS_BYTES_X));
        lds_set = true;
    }
    // M[n][k] of "Y = X M^T": transpose_w false -> M = W (sn = ldw, sk = 1); true -> M = W^T (sn = 1, sk = ldw)
    hipLaunchKernelGGL(k_wsg_pack, dim3(cdiv(WSG_PACK_FLOATS, 256)), dim3(256), 0, st, W, transpose_w ? 1 : ldw, transpose_w ? ldw : 1,
                       has257 ? 1 : 0, pack_scratch, ext ? ext->W : nullptr, ext ? ext->sn : 0, ext ? ext->sk : 0, em, ext ? ext->n : 0,
                       (ext && em == 1) ? ext->Wg : nullptr, ext ? ext->ldg : 0, ext ? ext->ng : 0);
    KPD_LAUNCH_CHECK();
    WsgArgs a;
    a.X = X; a.rows = rows; a.ldx = ldx; a.pack = pack_scratch; a.bias = bias; a.P = P; a.Y = Y; a.A = A; a.ldy = ldy; a.mode = mode;
    a.has257 = has257 ? 1 : 0; a.accumulate = accumulate ? 1 : 0;
    KPD_REQUIRE(!rowdot_out || (mode != WS_PLAIN && rowdot_w), KPD_ERR_INVALID, "ws_gemm: row-dot output without its mode / vector");
    a.rd_w = rowdot_w; a.rd_stride = rowdot_stride; a.rd_out = rowdot_out;
    a.X2 = em == 1 ? ext->X2 : nullptr; a.Y2 = em == 2 ? ext->Y2 : nullptr;
    a.ldx2 = a.ldy2 = ext ? ext->ld : 0; a.x2n = a.y2n = ext ? ext->n : 0;
    a.G2 = (em == 1 && ext->Wg) ? ext->G2 : nullptr;
    a.G2b = a.G2 ? ext->G2b : nullptr;
    KPD_REQUIRE(!(ext && ext->Wg) || (em == 1 && ext->G2 && ext->G2b && ext->ng >= 1 && ext->ng <= 16 && ext->ldg >= 256), KPD_ERR_INVALID, "ws_gemm: gate block does not fit");
    const int tiles = cdiv(rows, WSG_TILE);
    // one workgroup per CU, one round (see launch_proj_chain): 2 * bpc <= CUs of this device
    const int cus = cu_count();
    for (a.tpb = std::max(1, cdiv(2 * tiles, cus));; ++a.tpb) {
        a.bpc = cdiv(tiles, a.tpb);
        if (2 * a.bpc <= cus || a.tpb >= tiles) break;
    }
    if (em == 1) hipLaunchKernelGGL((k_ws_gemm<WS_BIAS_SILU, 1>), dim3(2 * a.bpc), dim3(512), WSG_LDS_BYTES_X, st, a);
    else if (em == 2) hipLaunchKernelGGL((k_ws_gemm<WS_PLAIN, 2>), dim3(2 * a.bpc), dim3(512), WSG_LDS_BYTES_X, st, a);
    else if (mode == WS_BIAS_SILU) hipLaunchKernelGGL((k_ws_gemm<WS_BIAS_SILU, 0>), dim3(2 * a.bpc), dim3(512), WSG_LDS_BYTES, st, a);
    else if (mode == WS_SILU_BWD) hipLaunchKernelGGL((k_ws_gemm<WS_SILU_BWD, 0>), dim3(2 * a.bpc), dim3(512), WSG_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((k_ws_gemm<WS_PLAIN, 0>), dim3(2 * a.bpc), dim3(512), WSG_LDS_BYTES, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd
