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
// (row_dot_chunks) against wx[k] = W[256][k].
#pragma once
#include "common.h"

namespace kpd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_exp_f32 + v_rcp_f32 (1 ulp each): far inside the 1e-4 parity budget, ~4x fewer VALU ops than
// the IEEE division sequence.
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// SiLU in "pre-scaled" form: for x' = c x with c = -log2(e), x' / (1 + 2^x') = c silu(x).  The producers of x'
// (packed weights, biases) carry the factor c and the linear consumers carry 1 / c, so the activation is
// v_exp_f32 + v_add + v_rcp_f32 + v_mul -- one VALU multiply fewer per element than silu().
constexpr float SILU_C = -1.4426950408889634f;
__device__ __forceinline__ float silu_pre(float xs) { return xs * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xs)); }
// four values at a time; same operations, same bits as silu_pre.
// (Round 5 spelled the 1 + e step out as inline-asm v_pk_add_f32 to save 190 VALU issues per tile for 0.16 %.  gfx950 needs one wait state
//  between a transcendental (v_exp_f32) and a non-transcendental consumer of its result; hipcc inserts it for its own instructions but
//  does not look inside an asm block, so wherever the scheduler put the second v_exp_f32 right before the asm the add read a stale register:
//  wrong and run-to-run different outputs, caught by tests/test_cold_start_gpu.py.  No instruction-bearing inline asm next to compiler
//  code: hazards are the compiler's to track.)
typedef float silu_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ silu_f32x4 silu_pre4(silu_f32x4 v) {
    const silu_f32x4 q = {__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[0])), __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[1])),
                          __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[2])), __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[3]))};
    return v * q;
}
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
// One k-group: 16 MFMAs.
#define KPD_GEMM_STEP(A0, A1, B0, B1)                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[j], B0[j], acc[0][0], 0, 0, 0);       \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[j], B1[j], acc[0][1], 0, 0, 0);       \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[j], B0[j], acc[1][0], 0, 0, 0);       \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[j], B1[j], acc[1][1], 0, 0, 0);       \
    }

// Tail group of a K = 264 tile: the 257 real features end at k = 256 and the bias "one" sits at k = 260, i.e. both live in
// k-step j = 0 of group 32 (k = 8 g + 4 h + j, one k per lane half); steps 1..3 of that group multiply zero padding only.
template <int NG_>
struct TailSteps {
    static constexpr int J = NG_ == 33 ? 1 : 4;
};
#define KPD_GEMM_STEP_N(A0, A1, B0, B1, NJ)                                                      \
    _Pragma("unroll") for (int j = 0; j < (NJ); ++j) {                                            \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[j], B0[j], acc[0][0], 0, 0, 0);       \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[j], B1[j], acc[0][1], 0, 0, 0);       \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[j], B0[j], acc[1][0], 0, 0, 0);       \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[j], B1[j], acc[1][1], 0, 0, 0);       \
    }

#define KPD_GEMM_LOAD(A0, A1, B0, B1, G)                              \
    A0 = *reinterpret_cast<const f32x4 *>(a0p + 8 * (G));             \
    A1 = *reinterpret_cast<const f32x4 *>(a1p + 8 * (G));             \
    B0 = bp[(G) * 512];                                               \
    B1 = bp[(G) * 512 + 1];

// NG_ = k-groups of 8 (K padded to 8 NG_), SA_ = LDS row stride of the A tile in floats.
// Pointer with an explicit global address space: when address-space inference loses track of a pointer (it did for the
// in-loop weight loads of gemm_rows64_pre) hipcc falls back to flat loads, which wait on vmcnt(0) and lgkmcnt(0) together
// and serialise the prefetch.
typedef __attribute__((address_space(1))) const f32x4 gf32x4;
__device__ __forceinline__ gf32x4 *as_global(const f32x4 *p) { return (gf32x4 *)p; }

// The weight fragments of k-groups 0 and 1 of a block, issued by the caller long before the GEMM (they depend on the edge
// type only, not on the tile): the first MFMAs then wait for two LDS reads instead of an L2 round trip.
struct BPrefetch {
    f32x4 x0, x1, y0, y1;
};
__device__ __forceinline__ void gemm_b_prefetch(BPrefetch &p, const float *__restrict__ Wp, int wave, int lane) {
    gf32x4 *bp = as_global(reinterpret_cast<const f32x4 *>(Wp) + (wave * 64 + lane) * 2);
    p.x0 = bp[0];
    p.x1 = bp[1];
    p.y0 = bp[512];
    p.y1 = bp[513];
}

#ifndef KPD_GEMM_SETS
#define KPD_GEMM_SETS 2          // operand register sets of gemm_rows64_pre: fragments of k-group g + SETS are requested behind the MFMAs of group g
                                 // (3 sets -- two groups of cover for the L2 round trip -- measured in round 5: 0.8130 vs 0.8132 ms, no difference)
#endif
template <int NG_, int SA_>
__device__ __forceinline__ void gemm_rows64_pre(const float *__restrict__ A, const float *__restrict__ Wp,
                                                f32x16 (&acc)[2][2], int wave, int lane, const BPrefetch &pre) {
    static_assert(NG_ > KPD_GEMM_SETS, "prefetched form needs more k-groups than operand sets");
    constexpr int NS = KPD_GEMM_SETS;
    const int r = lane & 31, h = lane >> 5;
    const float *a0p = A + r * SA_ + 4 * h;
    const float *a1p = A + (32 + r) * SA_ + 4 * h;
    // weight fragments of k-group G: scalar base Wp + 8 KB * G (the opaque asm keeps it in scalar registers) + this lane's 32-bit offset.
    // As `bp[G * 512]` on a per-lane 64-bit pointer every group past the 4-KB immediate range cost two 64-bit vector adds (3 VALU + 2 hazard
    // nops) inside the MFMA stream: 64 of them per tile.
    const unsigned b_lane = (unsigned)(wave * 64 + lane) * 32u;
#define KPD_GEMM_LOAD_S(A0, A1, B0, B1, G)                                                              \
    A0 = *reinterpret_cast<const f32x4 *>(a0p + 8 * (G));                                               \
    A1 = *reinterpret_cast<const f32x4 *>(a1p + 8 * (G));                                               \
    {                                                                                                   \
        const char *gb_ = reinterpret_cast<const char *>(Wp) + (size_t)(G) * 8192;                      \
        asm volatile("" : "+s"(gb_));                                                                   \
        B0 = *reinterpret_cast<gf32x4 *>((__attribute__((address_space(1))) const char *)gb_ + b_lane);        \
        B1 = *reinterpret_cast<gf32x4 *>((__attribute__((address_space(1))) const char *)gb_ + b_lane + 16u);  \
    }
    // NS operand sets rotate (fully unrolled: set indices are compile-time, no register moves): the fragments of k-group g + NS are requested
    // right behind the MFMAs of group g, so a weight load from L2 has NS - 1 groups (1 024 cycles each) to land.
    f32x4 sa0[NS], sa1[NS], sb0[NS], sb1[NS];
    sb0[0] = pre.x0; sb1[0] = pre.x1;
    sb0[1] = pre.y0; sb1[1] = pre.y1;
    sa0[0] = *reinterpret_cast<const f32x4 *>(a0p);
    sa1[0] = *reinterpret_cast<const f32x4 *>(a1p);
    sa0[1] = *reinterpret_cast<const f32x4 *>(a0p + 8);
    sa1[1] = *reinterpret_cast<const f32x4 *>(a1p + 8);
#pragma unroll
    for (int i = 2; i < NS; ++i) { KPD_GEMM_LOAD_S(sa0[i], sa1[i], sb0[i], sb1[i], i) }
#pragma unroll
    for (int g = 0; g < NG_; ++g) {
        const int st = g % NS;
        __builtin_amdgcn_sched_barrier(0);
        if ((NG_ & 1) && g == NG_ - 1) {
            KPD_GEMM_STEP_N(sa0[st], sa1[st], sb0[st], sb1[st], TailSteps<NG_>::J)
        } else {
            KPD_GEMM_STEP(sa0[st], sa1[st], sb0[st], sb1[st])
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g + NS < NG_) { KPD_GEMM_LOAD_S(sa0[st], sa1[st], sb0[st], sb1[st], g + NS) }
    }
#undef KPD_GEMM_LOAD_S
}

template <int NG_, int SA_>
__device__ __forceinline__ void gemm_rows64_t(const float *__restrict__ A, const float *__restrict__ Wp,
                                              f32x16 (&acc)[2][2], int wave, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const float *a0p = A + r * SA_ + 4 * h;
    const float *a1p = A + (32 + r) * SA_ + 4 * h;
    const f32x4 *bp = reinterpret_cast<const f32x4 *>(Wp) + (wave * 64 + lane) * 2;
    // Two operand register sets X / Y alternate (no register rotation: on gfx950 the f32 MFMA and
    // ordinary VALU work do not overlap, so every VALU instruction in this loop is lost MFMA time).
    // The set that was just consumed is refilled for two k-groups ahead right after its MFMAs
    // issue, so each refill has one full group (16 MFMAs = 1024 cycles) to land.  sched_barriers
    // pin that order; without them the scheduler sinks the loads next to their use.
    f32x4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
    KPD_GEMM_LOAD(xa0, xa1, xb0, xb1, 0)
    if (NG_ > 1) {
        KPD_GEMM_LOAD(ya0, ya1, yb0, yb1, 1)
    }
    constexpr int PAIRS = NG_ / 2;          // full (X, Y) pairs
#pragma unroll 1
    for (int p = 0; p < PAIRS; ++p) {
        const int g = 2 * p;
        __builtin_amdgcn_sched_barrier(0);
        KPD_GEMM_STEP(xa0, xa1, xb0, xb1)
        __builtin_amdgcn_sched_barrier(0);
        const int g2 = g + 2 < NG_ ? g + 2 : NG_ - 1;
        KPD_GEMM_LOAD(xa0, xa1, xb0, xb1, g2)
        __builtin_amdgcn_sched_barrier(0);
        KPD_GEMM_STEP(ya0, ya1, yb0, yb1)
        __builtin_amdgcn_sched_barrier(0);
        const int g3 = g + 3 < NG_ ? g + 3 : NG_ - 1;
        KPD_GEMM_LOAD(ya0, ya1, yb0, yb1, g3)
    }
    if (NG_ & 1) {
        __builtin_amdgcn_sched_barrier(0);
        KPD_GEMM_STEP_N(xa0, xa1, xb0, xb1, TailSteps<NG_>::J)
    }
}

// ---- 32-row tile variant (node kernels: 4 workgroups per CU, finer work items) --------------------------
// acc[nt] += A[32 x 8 NG_] * W[.. x (this wave's 64 columns)]; one A read + two B loads per 8 MFMAs.
template <int NG_, int SA_>
__device__ __forceinline__ void gemm_rows32_t(const float *__restrict__ A, const float *__restrict__ Wp,
                                              f32x16 (&acc)[2], int wave, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const float *a0p = A + r * SA_ + 4 * h;
    const f32x4 *bp = reinterpret_cast<const f32x4 *>(Wp) + (wave * 64 + lane) * 2;
    // A k-group is only 8 MFMAs (512 cycles) here, less than an L2 round trip under load, so the operand
    // ring is four sets deep: the set consumed at group g is refilled for group g + 4, i.e. every refill has
    // three groups (~1500 cycles) to land.
    f32x4 a0[4], b0[4], b1[4];
#define KPD_LOAD32(S, G)                                                  \
    a0[S] = *reinterpret_cast<const f32x4 *>(a0p + 8 * (G));              \
    b0[S] = bp[(G) * 512];                                                \
    b1[S] = bp[(G) * 512 + 1];
#define KPD_STEP32(S)                                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S][j], b0[S][j], acc[0], 0, 0, 0);       \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S][j], b1[S][j], acc[1], 0, 0, 0);       \
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = i < NG_ ? i : NG_ - 1;
        KPD_LOAD32(i, g)
    }
    constexpr int QUADS = NG_ / 4;
#pragma unroll 1
    for (int q = 0; q < QUADS; ++q) {
        const int g = 4 * q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            KPD_STEP32(i)
            __builtin_amdgcn_sched_barrier(0);
            const int gn = g + 4 + i < NG_ ? g + 4 + i : NG_ - 1;
            KPD_LOAD32(i, gn)
        }
    }
    static_assert((NG_ & 3) <= 1, "tail handling assumes at most one group past the last quad");
    if (NG_ & 3) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TailSteps<NG_>::J; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0][j], b0[0][j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0][j], b1[0][j], acc[1], 0, 0, 0);
        }
    }
#undef KPD_LOAD32
#undef KPD_STEP32
}

// 8-wave form of gemm_rows32_t: one 32-column tile per wave (wave w reads column tile (w & 1) of the packed block of wave
// (w >> 1) of 4).  Twice the waves per 32-row tile halve the dependent MFMA chain of a GEMM phase, which is what bounds the
// node kernel: its grid is a single round of workgroups, so its duration is a workgroup's latency.
template <int NG_, int SA_>
__device__ __forceinline__ void gemm_rows32_t8(const float *__restrict__ A, const float *__restrict__ Wp, f32x16 &acc, int wave,
                                               int lane) {
    const int r = lane & 31, h = lane >> 5;
    const float *a0p = A + r * SA_ + 4 * h;
    const f32x4 *bp = reinterpret_cast<const f32x4 *>(Wp) + ((wave >> 1) * 64 + lane) * 2 + (wave & 1);
    f32x4 a0[4], b0[4];
#define KPD_LOAD32(S, G)                                     \
    a0[S] = *reinterpret_cast<const f32x4 *>(a0p + 8 * (G)); \
    b0[S] = bp[(G) * 512];
#define KPD_STEP32(S) \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S][j], b0[S][j], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = i < NG_ ? i : NG_ - 1;
        KPD_LOAD32(i, g)
    }
    constexpr int QUADS = NG_ / 4;
#pragma unroll 1
    for (int q = 0; q < QUADS; ++q) {
        const int g = 4 * q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            KPD_STEP32(i)
            __builtin_amdgcn_sched_barrier(0);
            const int gn = g + 4 + i < NG_ ? g + 4 + i : NG_ - 1;
            KPD_LOAD32(i, gn)
        }
    }
    static_assert((NG_ & 3) <= 1, "tail handling assumes at most one group past the last quad");
    if (NG_ & 3) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TailSteps<NG_>::J; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0][j], b0[0][j], acc, 0, 0, 0);
    }
#undef KPD_LOAD32
#undef KPD_STEP32
}

__device__ __forceinline__ int acc_row32(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---- wave-count-generic variants ------------------------------------------------------------------
// NW waves share the 256 output columns: NW = 4 -> 64 columns (2 column tiles) per wave, NW = 8 -> 32
// columns (1 column tile) per wave.  Same packed weight buffer: wave w of 8 reads column tile (w & 1) of
// the block of wave (w >> 1) of 4.  With 8 waves a workgroup puts two waves on every SIMD, which gives
// the scheduler something to issue while the other wave sits in a gather, LDS round trip or barrier.
template <int NW>
struct WaveCols {
    static constexpr int NT = 8 / NW;
};

template <int NW>
__device__ __forceinline__ void acc_zero_w(f32x16 (&acc)[2][WaveCols<NW>::NT]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < WaveCols<NW>::NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
}

template <int NW>
__device__ __forceinline__ int acc_col_w(int nt, int wave, int lane) {
    return NW == 4 ? 64 * wave + 32 * nt + (lane & 31) : 32 * wave + (lane & 31);
}

template <int NW, int NG_, int SA_>
__device__ __forceinline__ void gemm_rows64_w(const float *__restrict__ A, const float *__restrict__ Wp,
                                              f32x16 (&acc)[2][WaveCols<NW>::NT], int wave, int lane) {
    if constexpr (NW == 4) {
        gemm_rows64_t<NG_, SA_>(A, Wp, acc, wave, lane);
    } else {
        const int r = lane & 31, h = lane >> 5;
        const float *a0p = A + r * SA_ + 4 * h;
        const float *a1p = A + (32 + r) * SA_ + 4 * h;
        const f32x4 *bp = reinterpret_cast<const f32x4 *>(Wp) + ((wave >> 1) * 64 + lane) * 2 + (wave & 1);
        f32x4 xa0, xa1, xb0, ya0, ya1, yb0;
#define KPD_LOAD1(A0, A1, B0, G)                                      \
    A0 = *reinterpret_cast<const f32x4 *>(a0p + 8 * (G));             \
    A1 = *reinterpret_cast<const f32x4 *>(a1p + 8 * (G));             \
    B0 = bp[(G) * 512];
#define KPD_STEP1(A0, A1, B0)                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[j], B0[j], acc[0][0], 0, 0, 0);       \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[j], B0[j], acc[1][0], 0, 0, 0);       \
    }
        KPD_LOAD1(xa0, xa1, xb0, 0)
        if (NG_ > 1) {
            KPD_LOAD1(ya0, ya1, yb0, 1)
        }
        constexpr int PAIRS = NG_ / 2;
#pragma unroll 1
        for (int p = 0; p < PAIRS; ++p) {
            const int g = 2 * p;
            __builtin_amdgcn_sched_barrier(0);
            KPD_STEP1(xa0, xa1, xb0)
            __builtin_amdgcn_sched_barrier(0);
            const int g2 = g + 2 < NG_ ? g + 2 : NG_ - 1;
            KPD_LOAD1(xa0, xa1, xb0, g2)
            __builtin_amdgcn_sched_barrier(0);
            KPD_STEP1(ya0, ya1, yb0)
            __builtin_amdgcn_sched_barrier(0);
            const int g3 = g + 3 < NG_ ? g + 3 : NG_ - 1;
            KPD_LOAD1(ya0, ya1, yb0, g3)
        }
        if (NG_ & 1) {
            __builtin_amdgcn_sched_barrier(0);
            KPD_STEP1(xa0, xa1, xb0)
        }
#undef KPD_LOAD1
#undef KPD_STEP1
    }
}

// ---- f16x2 mode: fp32-accurate product on the f16 matrix pipe -----------------------------------------------------------
// a ~ a_hi + a_lo, w ~ w_hi + w_lo with f16 planes (11 significand bits each, 2^-21..2^-22 relative residual); a w is taken as
// a_hi w_hi + a_hi w_lo + a_lo w_hi (each product exact in fp32, accumulated in fp32 by the MFMA; the dropped a_lo w_lo term is
// 2^-22 relative).  Three v_mfma_f32_32x32x16_f16 do the work of eight v_mfma_f32_32x32x2_f32 in 96 instead of 512 pipe
// cycles, and unlike the f32 MFMA they leave the vector issue port free three quarters of the time.
// A tile in LDS: two planes [64][SAH] of f16 (hi, then lo at + PLANE_H halves); weights: pack_f16_split (pack.hip).
// Scaling: the f16 MFMA flushes subnormal inputs, and the lo plane of a value x is ~2^-11 |x|, subnormal below |x| = 0.125.  The A
// planes therefore hold 2^6 a and the weight planes 2^10 w (both split after scaling): lo planes are normal down to |a| = 2e-3,
// |w| = 1.2e-4, hi planes stay finite up to |a| = 1023, |w| = 63; what is flushed below those bounds is < 2^-11 of an already
// negligible term.  The accumulator holds 2^16 times the product; the epilogue multiplies by 2^-16.
constexpr float H_SCALE_A = 64.0f, H_SCALE_W = 1024.0f, H_UNSCALE = 1.0f / 65536.0f;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int SAH = 280;                    // halves per row of an A plane (272 used): 560-B rows, conflict-free ds_read_b128
constexpr int PLANE_H = TM * SAH;           // halves per plane

__device__ __forceinline__ h8 as_h8(const f32x4 &v) { return __builtin_bit_cast(h8, v); }

// (a, b) -> packed f16 hi pair and packed f16 lo pair (v_cvt_pkrtz_f16_f32: truncation is fine, the remainder goes to lo)
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo) {
    typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
    const hp2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const hp2 l = __builtin_amdgcn_cvt_pkrtz(a - (float)h[0], b - (float)h[1]);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

#define KPD_H_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A), as_h8(B), ACC, 0, 0, 0)

// acc[mt][nt] += A[64 x 272] W[272 x (this wave's 64 columns)], A = the two f16 planes at Ah, W = split block Wh.
// A k-step is only 12 MFMAs (384 cycles) -- a fraction of an L2 round trip -- so the weight fragments of H_B_DEPTH steps are kept in
// flight (16 VGPRs each; with two steps the loop ran at one L2 latency per step, 20 k cycles for 6.5 k of MFMA); the A fragments come
// from LDS one step ahead.  Small terms first, the hi x hi product last.
constexpr int H_B_DEPTH = 6;
// rot (wave-uniform, 0 .. KH_STEPS - 1) rotates the order of the k-steps: workgroups that start together would otherwise walk the
// same weight lines in step and queue on the same L2 channels.
__device__ __forceinline__ void gemm_rows64_h(const _Float16 *__restrict__ Ah, const void *__restrict__ Wh, f32x16 (&acc)[2][2],
                                              int wave, int lane, int rot = 0) {
    const int r = lane & 31, h = lane >> 5;
    const _Float16 *a0p = Ah + r * SAH + 8 * h;
    const _Float16 *a1p = Ah + (32 + r) * SAH + 8 * h;
    gf32x4 *bp = as_global(reinterpret_cast<const f32x4 *>(Wh) + wave * 256 + lane);           // 4 fragments of 1 KB per wave and k-step
    f32x4 b[H_B_DEPTH][4];          // [.][hi nt0, lo nt0, hi nt1, lo nt1]
    f32x4 a[3][4];                  // [.][hi mt0, hi mt1, lo mt0, lo mt1]
    auto step_of = [&](int s) {     // s-th step of this workgroup's order
        const int t = s + rot;
        return t >= KH_STEPS ? t - KH_STEPS : t;
    };
    // No load is ever issued into registers that the MFMAs issued just before it read: a fragment buffer is refilled one whole step
    // (12 MFMAs) after its last reader was issued.  (Refilled right behind its readers, results were not reproducible run to run
    // -- a returning load can overtake MFMAs that are still queued; profiles/tools/repro_check.py.)
#pragma unroll
    for (int i = 0; i < H_B_DEPTH - 1; ++i) {
        const int st = step_of(i);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[i][j] = bp[st * 1024 + j * 64];
    }
    {
        const int st = step_of(0);
        a[0][0] = *reinterpret_cast<const f32x4 *>(a0p + 16 * st);
        a[0][1] = *reinterpret_cast<const f32x4 *>(a1p + 16 * st);
        a[0][2] = *reinterpret_cast<const f32x4 *>(a0p + PLANE_H + 16 * st);
        a[0][3] = *reinterpret_cast<const f32x4 *>(a1p + PLANE_H + 16 * st);
    }
#pragma unroll
    for (int s = 0; s < KH_STEPS; ++s) {
        const int cb = s % H_B_DEPTH, ca = s % 3, can = (s + 1) % 3;
        if (s + 1 < KH_STEPS) {         // buffer last read by step s - 2
            const int st = step_of(s + 1);
            a[can][0] = *reinterpret_cast<const f32x4 *>(a0p + 16 * st);
            a[can][1] = *reinterpret_cast<const f32x4 *>(a1p + 16 * st);
            a[can][2] = *reinterpret_cast<const f32x4 *>(a0p + PLANE_H + 16 * st);
            a[can][3] = *reinterpret_cast<const f32x4 *>(a1p + PLANE_H + 16 * st);
        }
        if (s + H_B_DEPTH - 1 < KH_STEPS) {   // buffer last read by step s - 1
            const int st = step_of(s + H_B_DEPTH - 1), nb = (s + H_B_DEPTH - 1) % H_B_DEPTH;
#pragma unroll
            for (int j = 0; j < 4; ++j) b[nb][j] = bp[st * 1024 + j * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        KPD_H_MFMA(acc[0][0], a[ca][2], b[cb][0]); KPD_H_MFMA(acc[0][1], a[ca][2], b[cb][2]);
        KPD_H_MFMA(acc[1][0], a[ca][3], b[cb][0]); KPD_H_MFMA(acc[1][1], a[ca][3], b[cb][2]);
        KPD_H_MFMA(acc[0][0], a[ca][0], b[cb][1]); KPD_H_MFMA(acc[0][1], a[ca][0], b[cb][3]);
        KPD_H_MFMA(acc[1][0], a[ca][1], b[cb][1]); KPD_H_MFMA(acc[1][1], a[ca][1], b[cb][3]);
        KPD_H_MFMA(acc[0][0], a[ca][0], b[cb][0]); KPD_H_MFMA(acc[0][1], a[ca][0], b[cb][2]);
        KPD_H_MFMA(acc[1][0], a[ca][1], b[cb][0]); KPD_H_MFMA(acc[1][1], a[ca][1], b[cb][2]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same product with 8 waves per 64-row tile: wave w owns columns 32 w .. 32 w + 31 (column tile w & 1 of the block of wave
// w >> 1 of 4), two row tiles, 6 MFMAs per k-step.
__device__ __forceinline__ void gemm_rows64_h8(const _Float16 *__restrict__ Ah, const void *__restrict__ Wh, f32x16 (&acc)[2][1],
                                               int wave, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const _Float16 *a0p = Ah + r * SAH + 8 * h;
    const _Float16 *a1p = Ah + (32 + r) * SAH + 8 * h;
    gf32x4 *bp = as_global(reinterpret_cast<const f32x4 *>(Wh) + (wave >> 1) * 256 + (wave & 1) * 128 + lane);
    constexpr int DEPTH = 5;
    f32x4 b[DEPTH][2];              // [.][hi, lo]
    f32x4 a[3][4];                  // [.][hi mt0, hi mt1, lo mt0, lo mt1]
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) {
        b[i][0] = bp[i * 1024];
        b[i][1] = bp[i * 1024 + 64];
    }
    a[0][0] = *reinterpret_cast<const f32x4 *>(a0p);
    a[0][1] = *reinterpret_cast<const f32x4 *>(a1p);
    a[0][2] = *reinterpret_cast<const f32x4 *>(a0p + PLANE_H);
    a[0][3] = *reinterpret_cast<const f32x4 *>(a1p + PLANE_H);
#pragma unroll
    for (int s = 0; s < KH_STEPS; ++s) {
        const int cb = s % DEPTH, ca = s % 3, can = (s + 1) % 3;
        if (s + 1 < KH_STEPS) {         // (refill discipline of gemm_rows64_h: never right behind the readers)
            a[can][0] = *reinterpret_cast<const f32x4 *>(a0p + 16 * (s + 1));
            a[can][1] = *reinterpret_cast<const f32x4 *>(a1p + 16 * (s + 1));
            a[can][2] = *reinterpret_cast<const f32x4 *>(a0p + PLANE_H + 16 * (s + 1));
            a[can][3] = *reinterpret_cast<const f32x4 *>(a1p + PLANE_H + 16 * (s + 1));
        }
        if (s + DEPTH - 1 < KH_STEPS) {
            const int nb = (s + DEPTH - 1) % DEPTH;
            b[nb][0] = bp[(s + DEPTH - 1) * 1024];
            b[nb][1] = bp[(s + DEPTH - 1) * 1024 + 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        KPD_H_MFMA(acc[0][0], a[ca][2], b[cb][0]); KPD_H_MFMA(acc[1][0], a[ca][3], b[cb][0]);
        KPD_H_MFMA(acc[0][0], a[ca][0], b[cb][1]); KPD_H_MFMA(acc[1][0], a[ca][1], b[cb][1]);
        KPD_H_MFMA(acc[0][0], a[ca][0], b[cb][0]); KPD_H_MFMA(acc[1][0], a[ca][1], b[cb][0]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// output column 256 of the same product on the VALU: row (tid / 4) of the two A planes against w (fp32, 272 floats in LDS, already
// carrying H_SCALE_W).  Every element is rebuilt exactly (hi + lo fits fp32) and multiplied in fp32: the reference form of
// row_dot_h2 below (profiles/tools/f16x2_gemm_probe.hip compares the two).  Returns the dot on the 4 lanes of the row.
__device__ __forceinline__ float row_dot_h(const _Float16 *__restrict__ Ah, const float *__restrict__ w, int tid) {
    const int row = tid >> 2, q = tid & 3;
    const _Float16 *ahi = Ah + row * SAH, *alo = Ah + PLANE_H + row * SAH;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int c = q + 4 * i;                       // 34 chunks of 8 elements per row
        if (c < 34) {
            const h8 ah = *reinterpret_cast<const h8 *>(ahi + 8 * c), al = *reinterpret_cast<const h8 *>(alo + 8 * c);
            const f32x4 wa = *reinterpret_cast<const f32x4 *>(w + 8 * c), wb = *reinterpret_cast<const f32x4 *>(w + 8 * c + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = fmaf((float)ah[j] + (float)al[j], wa[j], s);
                s = fmaf((float)ah[4 + j] + (float)al[4 + j], wb[j], s);
            }
        }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    return s;
}

// The same dot with packed f16 dot products: w as two f16 planes of 272 (hi, then lo at + 272, both carrying H_SCALE_W), three
// v_dot2_f32_f16 per pair of elements (a_lo w_hi + a_hi w_lo + a_hi w_hi, fp32 accumulation) -- 1.5 instead of 4 VALU
// instructions per element.
template <int TPR>
__device__ __forceinline__ float row_dot_h2(const _Float16 *__restrict__ Ah, const _Float16 *__restrict__ wh, int tid) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int row = tid / TPR, q = tid % TPR;
    const _Float16 *ahi = Ah + row * SAH, *alo = Ah + PLANE_H + row * SAH;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < (34 + TPR - 1) / TPR; ++i) {
        const int c = q + TPR * i;
        if (c < 34) {
            const h8 ah = *reinterpret_cast<const h8 *>(ahi + 8 * c), al = *reinterpret_cast<const h8 *>(alo + 8 * c);
            const h8 wa = *reinterpret_cast<const h8 *>(wh + 8 * c), wl = *reinterpret_cast<const h8 *>(wh + 272 + 8 * c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h2 a_h = {ah[2 * j], ah[2 * j + 1]}, a_l = {al[2 * j], al[2 * j + 1]};
                const h2 w_h = {wa[2 * j], wa[2 * j + 1]}, w_l = {wl[2 * j], wl[2 * j + 1]};
                s = __builtin_amdgcn_fdot2(a_l, w_h, s, false);
                s = __builtin_amdgcn_fdot2(a_h, w_l, s, false);
                s = __builtin_amdgcn_fdot2(a_h, w_h, s, false);
            }
        }
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o);
    return s;
}

// 32-row tile, 8 waves, one 32-column tile per wave (the node kernel): A planes [32][SAH] halves (lo plane at + plane_h), B planes
// from the pack_f16_split block (wave w of 8 = column tile w & 1 of wave w >> 1 of 4).  17 k-steps of three MFMAs; the B fragments of
// three steps are in flight ahead of the MFMAs, the A fragments of one.
__device__ __forceinline__ void gemm_rows32_h8(const _Float16 *__restrict__ Ah, int plane_h, const void *__restrict__ wh, f32x16 &acc,
                                               int wave, int lane) {
    const _Float16 *ap = Ah + (lane & 31) * SAH + 8 * (lane >> 5);
    const gf32x4 *bp = as_global(reinterpret_cast<const f32x4 *>(wh)) + (wave >> 1) * 256 + (wave & 1) * 128 + lane;
    f32x4 bh[4], bl[4], ah[3], al[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        bh[i] = bp[i * 1024];
        bl[i] = bp[i * 1024 + 64];
    }
    ah[0] = *reinterpret_cast<const f32x4 *>(ap);
    al[0] = *reinterpret_cast<const f32x4 *>(ap + plane_h);
#pragma unroll
    for (int s = 0; s < KH_STEPS; ++s) {
        if (s + 1 < KH_STEPS) {         // (refill discipline of gemm_rows64_h: never right behind the readers)
            ah[(s + 1) % 3] = *reinterpret_cast<const f32x4 *>(ap + 16 * (s + 1));
            al[(s + 1) % 3] = *reinterpret_cast<const f32x4 *>(ap + plane_h + 16 * (s + 1));
        }
        if (s + 3 < KH_STEPS) {
            bh[(s + 3) & 3] = bp[(s + 3) * 1024];
            bl[(s + 3) & 3] = bp[(s + 3) * 1024 + 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(al[s % 3]), as_h8(bh[s & 3]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[s % 3]), as_h8(bl[s & 3]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[s % 3]), as_h8(bh[s & 3]), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// dot of row (tid / TPR) of an f16-plane tile (planes carry H_SCALE_A) with an fp32 vector w[0 .. KP): TPR threads per row, eight
// columns per chunk; returns H_SCALE_A x the dot on all TPR lanes
template <int TPR>
__device__ __forceinline__ float row_dot_planes(const _Float16 *__restrict__ Ah, int plane_h, const float *__restrict__ w, int tid) {
    const int row = tid / TPR, q = tid % TPR;
    const _Float16 *ahi = Ah + row * SAH, *alo = ahi + plane_h;
    float s = 0.0f;
#pragma unroll
    for (int c = q; c < KP / 8; c += TPR) {
        const h8 hi = *reinterpret_cast<const h8 *>(ahi + 8 * c), lo = *reinterpret_cast<const h8 *>(alo + 8 * c);
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(w + 8 * c), w1 = *reinterpret_cast<const f32x4 *>(w + 8 * c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s = fmaf((float)hi[j] + (float)lo[j], w0[j], s);
            s = fmaf((float)hi[4 + j] + (float)lo[4 + j], w1[j], s);
        }
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o);
    return s;
}

// 2^6 SiLU in the pre-scaled form of silu_pre: the factor rides in the reciprocal's argument (an fma instead of the add)
__device__ __forceinline__ float silu_pre_x64(float xs) {
    return xs * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(xs), 1.0f / H_SCALE_A, 1.0f / H_SCALE_A));
}

// dot of row (tid / TPR) of an LDS tile (stride SA, 16-B aligned) with a vector over the first `chunks`
// float4 chunks; TPR consecutive threads own one row.  Returns the full dot on all TPR lanes.
template <int TPR>
__device__ __forceinline__ float row_dot_chunks(const float *__restrict__ T, const float *__restrict__ w, int chunks, int tid) {
    const int row = tid / TPR, q = tid % TPR;
    const f32x4 *a = reinterpret_cast<const f32x4 *>(T + row * SA);
    const f32x4 *wv = reinterpret_cast<const f32x4 *>(w);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < (66 + TPR - 1) / TPR; ++i) {
        const int c = q + TPR * i;
        if (c < chunks) {
            const f32x4 av = a[c], wq = wv[c];
            s = fmaf(av[0], wq[0], s);
            s = fmaf(av[1], wq[1], s);
            s = fmaf(av[2], wq[2], s);
            s = fmaf(av[3], wq[3], s);
        }
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o);
    return s;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter, i.e. every wave would wait at each barrier for its outstanding global stores (segment
// pieces, projections) to be acknowledged; nothing in these kernels reads those back.
__device__ __forceinline__ void lds_barrier() {
#ifdef KPD_HZ_FULLBAR   // hazard hunt (profiles/tools/hz_variant.sh): every barrier also drains the vector-memory counter
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// Row/column owned by accumulator register `reg` of tile (mt, nt) on this lane.
__device__ __forceinline__ int acc_row(int mt, int reg, int lane) {
    return 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int nt, int wave, int lane) { return 64 * wave + 32 * nt + (lane & 31); }

}  // namespace kpd
