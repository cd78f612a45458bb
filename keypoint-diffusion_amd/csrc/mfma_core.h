// fp32-MFMA row-tile GEMM core shared by every dense block on the hot path.
//
// Shape: a workgroup of 256 threads (4 waves, one per SIMD) owns a tile of TM = 64 rows
// (edges or nodes).  The A tile [64][KP] lives in LDS (row stride SA floats); the weight is
// streamed from L2/HBM in a pre-packed fragment order, each wave owning 64 of the 256 output
// columns, so no weight byte is read twice inside a workgroup.  v_mfma_f32_32x32x2_f32 is
// exact fp32 (one rounding per product, k-ordered), so results agree with an fp32 CPU
// reference to accumulation-order noise.
//
// K ordering: MFMA k-step (g, j), g in [0, NG), j in [0, 4), multiplies k = 8g + 4h + j on
// lane half h = lane >> 5.  Lanes therefore read their A operands for 4 k-steps with one
// ds_read_b128 and their B operands for 4 k-steps x 2 column tiles with two dwordx4 loads.
//
// Packed weight (WP_FLOATS floats):  Wp[g][wave][lane][nt][j] = W[n = 64 wave + 32 nt + (lane & 31)]
//                                                               [k = 8 g + 4 (lane >> 5) + j]
// with zeros for k >= K or n >= N.  Output column 256 (the "+1" of hidden_nf + 1) is not
// worth a ninth 32-wide MFMA column tile; it is a 264-long dot per row done on the VALU
// (extra_col) against wx[k] = W[256][k].
#pragma once
#include "common.h"

namespace kpd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_exp_f32 + v_rcp_f32 (1 ulp each): far inside the 1e-4 parity budget, ~4x fewer VALU ops than
// the IEEE division sequence.
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ void acc_zero(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
}

// acc[mt][nt] += A[64 x KP] * W[KP x (this wave's 64 columns)].
// Output column 256 rides along on the VALU: every lane already holds A[r][k] and A[32 + r][k]
// for its k's, so ex0 / ex1 accumulate A[r][:] . wx and A[32 + r][:] . wx (8 FMAs per 16 MFMAs);
// after the final cross-half add every lane of every wave holds the dots of rows r and 32 + r.
__device__ __forceinline__ void gemm_rows64(const float *__restrict__ A, const float *__restrict__ Wp,
                                            const float *__restrict__ wx, f32x16 (&acc)[2][2], float &ex0,
                                            float &ex1, int wave, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const float *a0p = A + r * SA + 4 * h;
    const float *a1p = A + (32 + r) * SA + 4 * h;
    const f32x4 *bp = reinterpret_cast<const f32x4 *>(Wp) + (wave * 64 + lane) * 2;
    const f32x4 *wxp = reinterpret_cast<const f32x4 *>(wx) + h;
    f32x4 b0 = bp[0], b1 = bp[1], w = wxp[0];
    float e0 = 0.0f, e1 = 0.0f;
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(a0p + 8 * g);
        const f32x4 a1 = *reinterpret_cast<const f32x4 *>(a1p + 8 * g);
        const int gn = g + 1 < NG ? g + 1 : g;
        const f32x4 nb0 = bp[gn * 512], nb1 = bp[gn * 512 + 1], nw = wxp[2 * gn];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            e0 = fmaf(a0[j], w[j], e0);
            e1 = fmaf(a1[j], w[j], e1);
        }
        b0 = nb0;
        b1 = nb1;
        w = nw;
    }
    ex0 += e0 + __shfl_xor(e0, 32);
    ex1 += e1 + __shfl_xor(e1, 32);
}

// Row/column owned by accumulator register `reg` of tile (mt, nt) on this lane.
__device__ __forceinline__ int acc_row(int mt, int reg, int lane) {
    return 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int nt, int wave, int lane) { return 64 * wave + 32 * nt + (lane & 31); }

// dot over the 257 valid columns of row (tid >> 2) of a T tile with a weight vector (w in LDS
// or global).  Groups of 4 consecutive threads own a row and stride the columns by 4
// (conflict-free at SA = 268).  Returns the full dot on all 4 lanes.
__device__ __forceinline__ float row_dot257(const float *__restrict__ T, const float *__restrict__ w, int tid) {
    const int row = tid >> 2, q = tid & 3;
    const float *a = T + row * SA + q;
    float s = 0.0f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) s = fmaf(a[4 * i], w[4 * i + q], s);
    if (q == 0) s = fmaf(T[row * SA + 256], w[256], s);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    return s;
}

}  // namespace kpd
