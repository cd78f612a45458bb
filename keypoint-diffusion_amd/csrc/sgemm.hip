// General fp32 GEMM of the training engines on v_mfma_f32_32x32x2_f32:  C[M,N] = alpha op(A) op(B) + beta C, row-major, any sizes and
// leading dimensions (the 2H+1 = 513-float rows of the EGNN first Linears, 16-wide GVP vector channels, B = 64-row keypoint products).
//
// Everything the training engines multiplied through the vendor BLAS runs here (train_ops.h: gemm, grad_gemm, gemv_n); the shapes the
// weight-stationary kernels were built for stay on ws_gemm.hip.  One template, three tile shapes x four operand forms:
//   * workgroup = 4 waves stacked along M; a wave owns WM x WN blocks of 32 x 32 (tile = 128 WM x 32 WN; WM = 1, WN in {1, 2, 4}), K in slabs of 16;
//   * both operands go through LDS as 16-B units in the order the direct global->LDS loads (global_load_lds_dwordx4) deliver them: an
//     operand that is contiguous along k in memory (A of NN / NT, B of NT / TT) as units (k-quad, row) -- an MFMA operand read is 32 lanes x
//     4 B at a 16-B stride; one contiguous along m / n (A of TN, B of NN / TN) row-major [k][row] -- 32 consecutive floats per half-wave;
//   * interior tiles of aligned operands: three LDS stages, the loads of slab kt + 2 issued (no registers) before the MFMAs of slab kt;
//     vmcnt counts them in order, so "one slab's worth outstanding" = slab kt + 1 has landed; one LDS-only barrier per slab;
//   * everything else (matrix edges, odd leading dimensions / offset pointers such as the 513-float rows, the K tail): the register
//     path -- clamped scalar or float4 loads issued before the slab's MFMAs, masked to zero and written to the same LDS layout after
//     them; no divergent branch, no padding contract on the caller's arrays;
//   * split-K (weight gradients, K = edge count): grid.z slices of K write partial tiles to scratch, summed in slice order by
//     k_sgemm_reduce: no atomics, bitwise reproducible.
// The arithmetic is the exact-fp32 MFMA of the inference kernels; accumulation order is fixed by the shape alone.
#include <stdlib.h>

#include <algorithm>

#include "engine.h"
#include <cstring>

#include "sgemm.h"

namespace kpd {

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int SG_BK = 16;
constexpr int SG_MAX_STAGES = 8;

struct SgemmArgs {
    const float *A, *B;
    float *C;
    int M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int k_chunk;            // K range of one grid.z slice (multiple of SG_BK); == K rounded up when not split
    long long c_slice;      // floats between the outputs of consecutive slices (split-K partials), 0 otherwise
    int vecA, vecB;         // base pointer 16-B aligned and leading dimension a multiple of 4
    int direct;             // 0: never take the global->LDS path (KPD_SGEMM_DIRECT=0, A/B runs)
    int stages;             // LDS stages of the direct path (3 .. SG_MAX_STAGES)
    float *colsum;          // op(A) = A^T only: colsum[m] += sum_k A[k][m] (the bias gradient that goes with a weight gradient), or null
    float *cs_part;         // split-K: per-slice partial column sums [slice][M] instead (summed by k_sgemm_reduce)
    const float *silu_pre;  // epilogue: C = (alpha AB + beta C) * SiLU'(silu_pre[m][n]) (same leading dimension as C), or null
    const float *bias;      // epilogue: + bias[n] (after alpha / beta), or null
    float *act_out;         // epilogue: act_out[m][n] = SiLU(C[m][n]) as a second output laid out like C, or null
    // Fringe: M, N above describe the TILED part of the output; one more output row xr (= M) and / or column xc (= N) -- the 257th feature
    // of a 257-wide product -- are computed as riders (xr / xc = -1: none), so that they do not cost a row / column of tiles of their own
    int xr, xc;
    float *x_part;          // split-K: per slice [M column-fringe values | N row-fringe values | corner | column sum of the fringe row]
};

// x or +0.0 by a bit mask: the value is consumed on both outcomes, so the load stays unconditional (a select lets the compiler sink the
// load under a divergent branch with a wait of its own -- one round trip per element)
__device__ __forceinline__ float masked(float x, bool keep) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & (keep ? 0xffffffffu : 0u));
}

// Tile operand with R rows (m or n) x SG_BK: either contiguous along k in memory (KCONT: element (r, k) at p[r * ld + k]) or along r
// (element (r, k) at p[k * ld + r]).  Its LDS form is made of 16-B units in the order the direct global->LDS loads deliver them:
//   KCONT : unit (kq, r) = the four k 4 kq .. 4 kq + 3 of row r, at float4 index kq * R + r   (an MFMA operand read is 32 lanes x 4 B at a
//           16-B stride: two lanes per bank)
//   else  : row-major [k][R]                                                                  (32 consecutive floats per half-wave)
template <int R, bool KCONT>
struct TileLoader {
    static constexpr int NV4 = R * SG_BK / 4;                    // units of the tile
    static constexpr int NV = (NV4 + 255) / 256;                 // per thread (register path) = wave instructions per wave (direct path)
    v4f v[NV];

    // float index of element (r, k) in the LDS tile
    static __device__ __forceinline__ int at(int r, int k) { return KCONT ? ((k >> 2) * R + r) * 4 + (k & 3) : k * R + r; }

    // ---- register path: any position / alignment ----
    // GUARD = false: the tile lies inside the matrix and float4 loads are aligned; true: every element is read from a clamped (always
    // valid) address here and replaced by zero in store() when it lies outside: no divergent branch in either form, and nothing consumes
    // the loaded registers before the slab's MFMAs have been issued.  Addresses are a uniform 64-bit base of the slab plus a 32-bit
    // per-thread offset that does not change from slab to slab.
    template <bool GUARD>
    __device__ __forceinline__ void load(const float *__restrict__ p, int ld, int r0, int rmax, int k0, int kmax, int tid) {
        const float *base = KCONT ? p + (size_t)r0 * ld + k0 : p + (size_t)k0 * ld + r0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            if (NV4 < 256 && idx >= NV4) break;
            const int rl = KCONT ? idx / (SG_BK / 4) : 4 * (idx % (R / 4));          // tile-local row / first of 4 rows
            const int kl = KCONT ? 4 * (idx % (SG_BK / 4)) : idx / (R / 4);          // tile-local first of 4 k / k
            if (!GUARD) {
                v[i] = *reinterpret_cast<const v4f *>(base + (unsigned)(KCONT ? rl * ld + kl : kl * ld + rl));
            } else if (KCONT) {
                const unsigned row = (unsigned)(min(rl, rmax - 1 - r0) * ld);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = base[row + (unsigned)min(kl + j, kmax - 1 - k0)];
            } else {
                const unsigned row = (unsigned)(min(kl, kmax - 1 - k0) * ld);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = base[row + (unsigned)min(rl + j, rmax - 1 - r0)];
            }
        }
    }
    // the arguments of the load this store completes
    template <bool GUARD>
    __device__ __forceinline__ void store(float *s, int r0, int rmax, int k0, int kmax, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            if (NV4 < 256 && idx >= NV4) break;
            const int rl = KCONT ? idx / (SG_BK / 4) : 4 * (idx % (R / 4));
            const int kl = KCONT ? 4 * (idx % (SG_BK / 4)) : idx / (R / 4);
            if (GUARD) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[i][j] = masked(v[i][j], KCONT ? (r0 + rl < rmax && k0 + kl + j < kmax) : (k0 + kl < kmax && r0 + rl + j < rmax));
            }
            *reinterpret_cast<v4f *>(s + at(rl, kl)) = v[i];
        }
    }

    // ---- direct path: tile inside the matrix, 16-B aligned; global -> LDS without registers ----
    // wave instruction n of the tile covers units 64 n .. 64 n + 63 (lane = unit - 64 n); every wave issues NV of them (wave + 4 j, wrapped:
    // a tile of fewer than 4 NV instructions is loaded twice into the same place, so that all waves have the same number in flight)
    // The per-thread element offsets and the per-wave LDS slots of those instructions do not change from slab to slab: plan() computes them
    // once, direct() adds them to the slab's wave-uniform base (scalar base + 32-bit lane offset loads: no vector address arithmetic in
    // the k-loop -- recomputed per slab it was ~40 VALU instructions and four v_readfirstlane between the MFMAs).
    unsigned doff[NV];
    int dslot[NV];
    __device__ __forceinline__ void plan(int ld, int wave, int lane) {
        constexpr int NI = NV4 / 64;                             // wave instructions of the tile
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int n = (wave + 4 * j) % NI, u = 64 * n + lane;
            doff[j] = 4u * (KCONT ? (unsigned)((u % R) * ld + 4 * (u / R)) : (unsigned)((u / (R / 4)) * ld + 4 * (u % (R / 4))));      // bytes
            dslot[j] = 256 * n;
        }
    }
    // base: the slab's wave-uniform origin (KCONT: p + r0 ld + k0, else p + k0 ld + r0); s: the tile's place in the LDS stage
    __device__ __forceinline__ void direct(const float *base, float *s) const {
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const char *b = reinterpret_cast<const char *>(base);
            unsigned o = doff[j];
            asm volatile("" : "+s"(b), "+v"(o));          // (scalar base, 32-bit lane offset: the addressing mode the load has; a hoisted 64-bit extension of the offset turns it back into a vector add)
            __builtin_amdgcn_global_load_lds((glb_void *)(b + o), (lds_void *)(s + dslot[j]), 16, 0, 0);
        }
    }
};

template <int WM, int WN, bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void k_sgemm(SgemmArgs a) {
    constexpr int BM = 128 * WM, BN = 32 * WN;
    constexpr int STAGE = (BM + BN) * SG_BK;                     // floats of one LDS stage: A tile, then B tile
    extern __shared__ __attribute__((aligned(16))) float smem[];  // a.stages stages
    typedef TileLoader<BM, !TA> LA;                               // op(A)[m][k]: contiguous along k unless transposed
    typedef TileLoader<BN, TB> LB;                                // op(B)[k][n]: contiguous along k only when transposed
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * a.k_chunk, kend = min(a.K, kbeg + a.k_chunk);
    float *C = a.C + (size_t)blockIdx.z * a.c_slice;

    v16f acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = (kend - kbeg + SG_BK - 1) / SG_BK, nk_full = (kend - kbeg) / SG_BK;
    // aligned float4 access where the tile lies inside the matrix (uniform over the workgroup) and the slab inside the K range
    const bool fastA = a.vecA != 0 && m0 + BM <= a.M, fastB = a.vecB != 0 && n0 + BN <= a.N;
    // (Issuing all operand reads of a slab before its first MFMA -- 8 (WM + WN) registers, one exposed LDS latency per slab -- was measured in
    // round 5 and is SLOWER: 314 vs 265 us on the 166k x 256 x 256 product; it costs an occupancy step and queues 40 LDS reads in front of the
    // first MFMA.  The reads stay next to their k-step; the second wave of the SIMD covers their latency.)
    auto compute = [&](const float *st) {
        const float *as = st, *bs = st + BM * SG_BK;
#pragma unroll
        for (int ks = 0; ks < SG_BK / 2; ++ks) {
            const int k = 2 * ks + half;
            float av[WM], bv[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) av[i] = as[LA::at(wave * 32 * WM + 32 * i + col, k)];
#pragma unroll
            for (int j = 0; j < WN; ++j) bv[j] = bs[LB::at(32 * j + col, k)];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    // column sums of the A operand ride along in the workgroups of the first column tile: thread m adds the slab's 16 values of its column
    // from the LDS tile ([k][m] rows: conflict-free), in slab order
    const bool sum_cols = TA && a.colsum != nullptr && blockIdx.y == 0 && tid < BM;
    float cs = 0.0f;
    auto add_cols = [&](const float *st) {
        if (TA && sum_cols) {
#pragma unroll
            for (int k = 0; k < SG_BK; ++k) cs += st[LA::at(tid, k)];
        }
    };
    // Fringe riders.  The slab's 16 values of op(A)[xr][k] and op(B)[k][xc] travel one slab ahead through a register of threads 0 .. 31
    // into a small double-buffered LDS array; the workgroups of the first column tile take sum_k op(A)[m][k] op(B)[k][xc] for their rows
    // from the A tile they hold, those of the first row tile sum_k op(A)[xr][k] op(B)[k][n] for their columns from the B tile, workgroup
    // (0, 0) the corner (and the fringe row's share of the column sums).
    __shared__ float xbuf[2][2][SG_BK];
    const bool fringe = a.xr >= 0 || a.xc >= 0;
    const bool ride_c = a.xc >= 0 && blockIdx.y == 0 && tid < BM, ride_r = a.xr >= 0 && blockIdx.x == 0 && tid < BN;
    const bool ride_k = fringe && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
    float fc = 0.0f, fr = 0.0f, fk = 0.0f, fcs = 0.0f, xv = 0.0f;
    auto xfetch = [&](int kt) {
        if (fringe && tid < 2 * SG_BK) {
            const int k = kbeg + kt * SG_BK + (tid & (SG_BK - 1));
            const bool isb = tid >= SG_BK;
            const int kk = min(k, kend - 1);
            const float *src = isb ? (TB ? a.B + (size_t)max(a.xc, 0) * a.ldb + kk : a.B + (size_t)kk * a.ldb + max(a.xc, 0))
                                   : (TA ? a.A + (size_t)kk * a.lda + max(a.xr, 0) : a.A + (size_t)max(a.xr, 0) * a.lda + kk);
            xv = masked(*src, k < kend && (isb ? a.xc >= 0 : a.xr >= 0));
        }
    };
    auto xstash = [&](int kt) {
        if (fringe && tid < 2 * SG_BK) xbuf[kt & 1][tid >> 4][tid & (SG_BK - 1)] = xv;
    };
    auto riders = [&](int kt, const float *st) {
        if (!fringe) return;
        const float *xa = xbuf[kt & 1][0], *xb = xbuf[kt & 1][1];
        if (ride_c) {
#pragma unroll
            for (int k = 0; k < SG_BK; ++k) fc += st[LA::at(tid, k)] * xb[k];
        }
        if (ride_r) {
#pragma unroll
            for (int k = 0; k < SG_BK; ++k) fr += st[BM * SG_BK + LB::at(tid, k)] * xa[k];
        }
        if (ride_k) {
#pragma unroll
            for (int k = 0; k < SG_BK; ++k) { fk += xa[k] * xb[k]; fcs += xa[k]; }
        }
    };
    LA la;
    LB lb;
    auto fetch = [&](int kt) {
        const int k0 = kbeg + kt * SG_BK;
        const bool inside = k0 + SG_BK <= kend;
        if (fastA && inside) la.template load<false>(a.A, a.lda, m0, a.M, k0, kend, tid);
        else la.template load<true>(a.A, a.lda, m0, a.M, k0, kend, tid);
        if (fastB && inside) lb.template load<false>(a.B, a.ldb, n0, a.N, k0, kend, tid);
        else lb.template load<true>(a.B, a.ldb, n0, a.N, k0, kend, tid);
    };
    auto stash = [&](int kt, float *st) {
        const int k0 = kbeg + kt * SG_BK;
        const bool inside = k0 + SG_BK <= kend;
        if (fastA && inside) la.template store<false>(st, m0, a.M, k0, kend, tid);
        else la.template store<true>(st, m0, a.M, k0, kend, tid);
        if (fastB && inside) lb.template store<false>(st + BM * SG_BK, n0, a.N, k0, kend, tid);
        else lb.template store<true>(st + BM * SG_BK, n0, a.N, k0, kend, tid);
    };

    int done = 0;                       // slabs consumed
    if (a.direct && fastA && fastB && nk_full >= 2) {
        // Direct path: a ring of `stages` LDS stages (3 .. SG_MAX_STAGES, chosen by the host), the loads of slab kt + stages - 1 issued
        // before the MFMAs of slab kt -- stages - 1 slabs of MFMA time for a load to arrive, no registers held meanwhile.  Loads complete
        // in order, so "at most the slabs behind kt + 1 outstanding" means slab kt + 1 has landed; the barrier then publishes it and
        // retires the stage of slab kt.
        constexpr int IN_FLIGHT = LA::NV + LB::NV;
        static_assert(IN_FLIGHT * (SG_MAX_STAGES - 2) <= 63, "vmcnt immediate");
        const int stages = a.stages;
        la.plan(a.lda, wave, lane);
        lb.plan(a.ldb, wave, lane);
        // wave-uniform origins of the next slab to issue, and the steps from slab to slab; the ring positions as counters (no `% stages`)
        const float *pa = !TA ? a.A + (size_t)m0 * a.lda + kbeg : a.A + (size_t)kbeg * a.lda + m0;
        const float *pb = TB ? a.B + (size_t)n0 * a.ldb + kbeg : a.B + (size_t)kbeg * a.ldb + n0;
        const size_t step_a = !TA ? (size_t)SG_BK : (size_t)SG_BK * a.lda, step_b = TB ? (size_t)SG_BK : (size_t)SG_BK * a.ldb;
        int si_issue = 0, si_comp = 0;
        auto issue = [&](int) {
            float *st = smem + si_issue * STAGE;
            la.direct(pa, st);
            lb.direct(pb, st + BM * SG_BK);
            pa += step_a;
            pb += step_b;
            si_issue = si_issue + 1 == stages ? 0 : si_issue + 1;
        };
        // wait until at most `slabs` slabs of loads are outstanding (the counter takes an immediate)
        auto wait_behind = [&](int slabs) {
            switch (slabs) {
            case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IN_FLIGHT) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IN_FLIGHT) : "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * IN_FLIGHT) : "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * IN_FLIGHT) : "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * IN_FLIGHT) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * IN_FLIGHT) : "memory"); break;
            }
        };
        const int ahead = min(stages - 1, nk_full);
        xfetch(0);                                     // (before the ring's loads: the in-order counter then covers it with them)
        for (int kt = 0; kt < ahead; ++kt) issue(kt);
        xstash(0);
        wait_behind(ahead - 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
        for (int kt = 0; kt < nk_full; ++kt) {
            if (kt + 1 < nk) xfetch(kt + 1);
            if (kt + stages - 1 < nk_full) issue(kt + stages - 1);
            __builtin_amdgcn_sched_barrier(0);
            const float *st_c = smem + si_comp * STAGE;
            si_comp = si_comp + 1 == stages ? 0 : si_comp + 1;
            compute(st_c);
            add_cols(st_c);
            riders(kt, st_c);
            if (kt + 1 < nk) xstash(kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            // issued so far: slabs 0 .. min(kt + stages - 1, nk_full - 1); needed next: kt + 1
            wait_behind(max(0, min(kt + stages - 1, nk_full - 1) - (kt + 1)));
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        done = nk_full;
    }
    // Register path: edge tiles, misaligned operands, short K, and the K tail of the direct path.  Two stages, one slab of prefetch.
    if (done < nk) {
        float *st0 = smem, *st1 = smem + STAGE;          // (every stage is free here: the direct path ended on a barrier)
        fetch(done);
        if (done == 0) xfetch(0);                      // (after the direct path the fringe values of slab `done` are in place already)
        stash(done, st0);
        if (done == 0) xstash(0);
        __syncthreads();
#pragma unroll 1
        for (int kt = done; kt < nk; ++kt) {
            float *cur = ((kt - done) & 1) ? st1 : st0, *nxt = ((kt - done) & 1) ? st0 : st1;
            if (kt + 1 < nk) { fetch(kt + 1); xfetch(kt + 1); }
            compute(cur);
            add_cols(cur);
            riders(kt, cur);
            if (kt + 1 < nk) { stash(kt + 1, nxt); xstash(kt + 1); }
            __syncthreads();
        }
    }
    // fringe outputs: shares to scratch when K is split, else straight into C (same alpha / beta / activation-derivative as the tiles)
    if (fringe) {
        auto finish = [&](float v, float *dst, const float *pre) {
            v *= a.alpha;
            if (a.beta != 0.0f) v += a.beta * *dst;
            if (pre) {
                const float sg = 1.0f / (1.0f + __expf(-*pre));
                v *= sg * (1.0f + *pre * (1.0f - sg));
            }
            const size_t at = dst - a.C;               // (not split here: a.C is the caller's C)
            if (a.bias) v += a.bias[at % a.ldc];
            *dst = v;
            if (a.act_out) a.act_out[at] = v / (1.0f + __expf(-v));
        };
        float *xp = a.x_part ? a.x_part + (size_t)blockIdx.z * (a.M + a.N + 2) : nullptr;
        if (ride_c && m0 + tid < a.M) {
            if (xp) xp[m0 + tid] = a.alpha * fc;
            else finish(fc, a.C + (size_t)(m0 + tid) * a.ldc + a.xc, a.silu_pre ? a.silu_pre + (size_t)(m0 + tid) * a.ldc + a.xc : nullptr);
        }
        if (ride_r && n0 + tid < a.N) {
            if (xp) xp[a.M + n0 + tid] = a.alpha * fr;
            else finish(fr, a.C + (size_t)a.xr * a.ldc + n0 + tid, a.silu_pre ? a.silu_pre + (size_t)a.xr * a.ldc + n0 + tid : nullptr);
        }
        if (ride_k) {
            if (xp) { xp[a.M + a.N] = a.alpha * fk; xp[a.M + a.N + 1] = fcs; }
            else {
                if (a.xr >= 0 && a.xc >= 0)
                    finish(fk, a.C + (size_t)a.xr * a.ldc + a.xc, a.silu_pre ? a.silu_pre + (size_t)a.xr * a.ldc + a.xc : nullptr);
                if (TA && a.colsum && a.xr >= 0) a.colsum[a.xr] += fcs;
            }
        }
    }
    if (TA && sum_cols && m0 + tid < a.M) {
        if (a.cs_part) a.cs_part[(size_t)blockIdx.z * a.M + m0 + tid] = cs;
        else a.colsum[m0 + tid] += cs;              // one workgroup per column when K is not split
    }
    // accumulator element r of lane (col, half): row 8 (r / 4) + 4 half + r % 4, column col.  beta != 0: the 16 old values of a block are
    // read together (clamped rows) before any of them is needed
    const bool accumulate = a.beta != 0.0f, silu_bwd = a.silu_pre != nullptr;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + 32 * j + col;
            if (n < a.N) {
                const int mb = m0 + wave * 32 * WM + 32 * i + 4 * half;
                const float bn = a.bias ? a.bias[n] : 0.0f;
                float old[16], pre[16];
                if (accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = C[(size_t)min(mb + 8 * (r >> 2) + (r & 3), a.M - 1) * a.ldc + n];
                }
                if (silu_bwd) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) pre[r] = a.silu_pre[(size_t)min(mb + 8 * (r >> 2) + (r & 3), a.M - 1) * a.ldc + n];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + 8 * (r >> 2) + (r & 3);
                    float v = a.alpha * acc[i][j][r];
                    if (accumulate) v += a.beta * old[r];
                    if (silu_bwd) {                      // the activation derivative that followed as a pass of its own (train_ops.h silu_grad)
                        const float sg = 1.0f / (1.0f + __expf(-pre[r]));
                        v *= sg * (1.0f + pre[r] * (1.0f - sg));
                    }
                    v += bn;
                    if (m < a.M) {
                        C[(size_t)m * a.ldc + n] = v;
                        if (a.act_out) a.act_out[(size_t)m * a.ldc + n] = v / (1.0f + __expf(-v));      // SiLU (train_ops.h silu_f)
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // one block's 16 old values at a time, not all WM x WN blocks' (registers)
        }
}

template <int WM, int WN, bool TA, bool TB>
kpd_status launch_one(dim3 grid, hipStream_t st, const SgemmArgs &a) {
    constexpr int stage_bytes = (128 * WM + 32 * WN) * SG_BK * 4;
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_sgemm<WM, WN, TA, TB>), SG_MAX_STAGES * stage_bytes));
    hipLaunchKernelGGL((k_sgemm<WM, WN, TA, TB>), grid, dim3(256), a.stages * stage_bytes, st, a);
    return KPD_OK;
}

template <int WM, int WN>
kpd_status launch_shape(bool tA, bool tB, dim3 grid, hipStream_t st, const SgemmArgs &a) {
    if (!tA && tB) return launch_one<WM, WN, false, true>(grid, st, a);
    if (!tA && !tB) return launch_one<WM, WN, false, false>(grid, st, a);
    if (tA && !tB) return launch_one<WM, WN, true, false>(grid, st, a);
    return launch_one<WM, WN, true, true>(grid, st, a);
}

// Weight gradient of a narrow Linear (the 16-wide vector channels of the GVPs): C[M,N] = alpha A[K,M]^T B[K,N] with M, N <= 32 and K = rows
// of a tall activation matrix.  No LDS staging: lane (col, half) of a wave reads A[k + half][col] and B[k + half][col] straight into the
// operands of one v_mfma_f32_32x32x2_f32 per two rows (16 rows in flight); the workgroup's four waves split its K range, their
// accumulators are added in wave order through LDS, and the workgroup writes one partial [M,N] for k_sgemm_reduce.
__global__ __launch_bounds__(256) void k_sgemm_tn_skinny(SgemmArgs a) {
    __shared__ float red[4][32 * 33];
    __shared__ float red_cs[4][32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, half = lane >> 5;
    float cs = 0.0f;                                   // column sums of A over this lane's rows (colsum requested)
    // rows of this wave: a.k_chunk rows per workgroup, a quarter (even) per wave
    const int per_wave = a.k_chunk / 4;
    const int kbeg = min(a.K, (int)blockIdx.x * a.k_chunk + wave * per_wave), kend = min(a.K, kbeg + per_wave);
    const int ca = min(col, a.M - 1), cb = min(col, a.N - 1);
    const bool va = col < a.M, vb = col < a.N;
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float *pa = a.A + ca, *pb = a.B + cb;
    int k = kbeg;
#pragma unroll 1
    for (; k + 16 <= kend; k += 16) {
        float x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = pa[(size_t)(k + 2 * u + half) * a.lda];
            y[u] = pb[(size_t)(k + 2 * u + half) * a.ldb];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float xa = masked(x[u], va);
            cs += xa;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, masked(y[u], vb), acc, 0, 0, 0);
        }
    }
    for (; k < kend; k += 2) {
        const int kk = min(k + half, a.K - 1);
        const bool in = k + half < kend;
        const float xa = masked(pa[(size_t)kk * a.lda], va && in);
        cs += xa;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, masked(pb[(size_t)kk * a.ldb], vb && in), acc, 0, 0, 0);
    }
    cs += __shfl_xor(cs, 32);                           // even rows + odd rows of this wave
    if (half == 0) red_cs[wave][col] = cs;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(8 * (r >> 2) + 4 * half + (r & 3)) * 33 + col] = acc[r];
    __syncthreads();
    float *C = a.C + (size_t)blockIdx.x * a.c_slice;
    for (int i = tid; i < a.M * a.N; i += 256) {
        const int m = i / a.N, n = i - m * a.N;
        C[i] = a.alpha * (((red[0][m * 33 + n] + red[1][m * 33 + n]) + red[2][m * 33 + n]) + red[3][m * 33 + n]);
    }
    if (a.cs_part && tid < a.M) a.cs_part[(size_t)blockIdx.x * a.M + tid] = ((red_cs[0][tid] + red_cs[1][tid]) + red_cs[2][tid]) + red_cs[3][tid];
}

// ---- weight gradient with the WHOLE 256 x 256 output in one workgroup ------------------------------------------------------------------
// C[256 (+1), 256 (+1)] = A[K, 256 (+1)]^T B[K, 256 (+1)] for K = edge count (the dW2 = dpre2^T a1 products of the EGNN trainer, the
// 256 x 256 scalar blocks of the GVP trainers).  The tiled form above cuts the output into 2 x 2 tiles of 128 x 128, so every operand
// row is fetched twice (684 MB for the 166 k kk edges against 342 MB of operands): those products ran at HBM / MALL speed, not at
// MFMA speed.  Here a workgroup of eight waves holds all 256 x 256 accumulators (wave (wr, wc): rows 64 wr .., columns 128 wc ..;
// 2 x 4 blocks of 32 x 32 = 128 VGPRs per lane) and streams its K range through LDS ONCE: a slab is 16 rows of A and 16 rows of B,
// each row one wave instruction of the direct global -> LDS load (1 KB contiguous), TN256_STAGES stages, 64 MFMAs per wave and slab.
// One workgroup per CU (2 waves per SIMD).  Fringe row / column 256, the column sums of A and the split-K shares use
// the formats of k_sgemm, so the same k_sgemm_reduce finishes the product.
constexpr int TN256_STAGES = 4;        // 128 KB of LDS: three slabs (12 k MFMA cycles) of prefetch distance
__device__ __forceinline__ void tn256_body(const SgemmArgs &a, const int slice) {
    constexpr int ROW = 256, SLAB = SG_BK * ROW, STAGE = 2 * SLAB, NSTAGE = TN256_STAGES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float xbuf[2][2][SG_BK];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, half = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int kbeg = slice * a.k_chunk, kend = min(a.K, kbeg + a.k_chunk);
    const int nk = (kend - kbeg + SG_BK - 1) / SG_BK, nk_full = (kend - kbeg) / SG_BK;
    v16f acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // rows wave, wave + 8 of either operand: one 1-KB instruction each (LDS destination = wave-uniform row base + 16 B per lane)
    auto issue = [&](int kt) {
        float *st = smem + (kt % NSTAGE) * STAGE;
        const size_t k0 = (size_t)kbeg + (size_t)kt * SG_BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wave + 8 * j;
            __builtin_amdgcn_global_load_lds((glb_void *)(a.A + (k0 + row) * a.lda + 4 * lane), (lds_void *)(st + row * ROW), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(a.B + (k0 + row) * a.ldb + 4 * lane), (lds_void *)(st + SLAB + row * ROW), 16, 0, 0);
        }
    };
    auto wait_behind = [&](int slabs) {                 // at most `slabs` slabs of this wave's loads outstanding (4 instructions each)
        if (slabs <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (slabs == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };
    auto compute = [&](const float *st) {
        const float *as = st + 64 * wr + col, *bs = st + SLAB + 128 * wc + col;
#pragma unroll
        for (int ks = 0; ks < SG_BK / 2; ++ks) {
            const int k = 2 * ks + half;
            float av[2], bv[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = as[k * ROW + 32 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = bs[k * ROW + 32 * j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    // riders (as in k_sgemm): threads 0 .. 255 own a row m of the output (column sums of A, column fringe), 256 .. 511 a column n (row
    // fringe).  The slab's 16 values of A[:, xr] / B[:, xc] travel one slab ahead through a register of threads 0 .. 31 into xbuf.
    // (Loading them with the slab as lane-0-only direct loads was tried: four more memory instructions per wave and slab, 9 % slower.)
    const bool fringe = a.xr >= 0 || a.xc >= 0;
    const bool sum_cols = a.colsum != nullptr && tid < ROW, ride_c = a.xc >= 0 && tid < ROW, ride_r = a.xr >= 0 && tid >= ROW;
    float cs = 0.0f, fc = 0.0f, fr = 0.0f, fk = 0.0f, fcs = 0.0f, xv = 0.0f;
    auto xfetch = [&](int kt) {
        if (fringe && tid < 2 * SG_BK) {
            const int k = kbeg + kt * SG_BK + (tid & (SG_BK - 1));
            const bool isb = tid >= SG_BK;
            const int kk = min(k, kend - 1);
            const float *src = isb ? a.B + (size_t)kk * a.ldb + max(a.xc, 0) : a.A + (size_t)kk * a.lda + max(a.xr, 0);
            xv = masked(*src, k < kend && (isb ? a.xc >= 0 : a.xr >= 0));
        }
    };
    auto xstash = [&](int kt) {
        if (fringe && tid < 2 * SG_BK) xbuf[kt & 1][tid >> 4][tid & (SG_BK - 1)] = xv;
    };
    auto riders = [&](int kt, const float *st) {
        const float *xa = xbuf[kt & 1][0], *xb = xbuf[kt & 1][1];
        if (sum_cols || ride_c) {
#pragma unroll
            for (int k = 0; k < SG_BK; ++k) {
                const float v = st[k * ROW + tid];
                cs += v;
                if (ride_c) fc += v * xb[k];
            }
        }
        if (ride_r) {
#pragma unroll
            for (int k = 0; k < SG_BK; ++k) fr += xa[k] * st[SLAB + k * ROW + tid - ROW];
        }
        if (fringe && tid == 0) {
#pragma unroll
            for (int k = 0; k < SG_BK; ++k) { fk += xa[k] * xb[k]; fcs += xa[k]; }
        }
    };

    const int ahead = min(NSTAGE - 1, nk_full);
    xfetch(0);
    for (int kt = 0; kt < ahead; ++kt) issue(kt);
    xstash(0);
    wait_behind(ahead - 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
    for (int kt = 0; kt < nk_full; ++kt) {
        if (kt + 1 < nk) xfetch(kt + 1);
        if (kt + NSTAGE - 1 < nk_full) issue(kt + NSTAGE - 1);
        __builtin_amdgcn_sched_barrier(0);
        const float *st = smem + (kt % NSTAGE) * STAGE;
        compute(st);
        riders(kt, st);
        if (kt + 1 < nk) xstash(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        wait_behind(min(kt + NSTAGE - 1, nk_full - 1) - (kt + 1));      // issued so far: slabs .. min(kt + NSTAGE - 1, nk_full - 1); needed next: kt + 1
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (nk > nk_full) {          // K tail of this range: rows past kend read a clamped address and are replaced by zero
        float *st = smem;        // (every stage is free: the loop ended on a barrier with nothing in flight)
        const int k0 = kbeg + nk_full * SG_BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = tid + 512 * j, op = u >> 10, row = (u & 1023) >> 6, c4 = u & 63;
            const int k = min(k0 + row, kend - 1);
            v4f v = *reinterpret_cast<const v4f *>((op ? a.B + (size_t)k * a.ldb : a.A + (size_t)k * a.lda) + 4 * c4);
            const bool in = k0 + row < kend;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = masked(v[e], in);
            *reinterpret_cast<v4f *>(st + op * SLAB + row * ROW + 4 * c4) = v;
        }
        __syncthreads();
        compute(st);
        riders(nk_full, st);
    }
    // shares of this K range (formats of k_sgemm: tile [256][256], column sums [M], fringe [M | N | corner | fringe-row column sum])
    float *C = a.C + (size_t)slice * a.c_slice;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(size_t)(64 * wr + 32 * i + 8 * (r >> 2) + 4 * half + (r & 3)) * a.ldc + 128 * wc + 32 * j + col] = a.alpha * acc[i][j][r];
    if (sum_cols && a.cs_part) a.cs_part[(size_t)slice * a.M + tid] = cs;
    if (a.x_part) {
        float *xp = a.x_part + (size_t)slice * (a.M + a.N + 2);
        if (ride_c) xp[tid] = a.alpha * fc;
        if (ride_r) xp[a.M + tid - ROW] = a.alpha * fr;
        if (tid == 0) { xp[a.M + a.N] = a.alpha * fk; xp[a.M + a.N + 1] = fcs; }
    }
}

__global__ __launch_bounds__(512, 1) void k_sgemm_tn256(SgemmArgs a) { tn256_body(a, (int)blockIdx.x); }

// Several such products in one launch: block -> (product, K slice) by a prefix table, every product with a share of the workgroups
// proportional to its K (the eight dW2 of an EGNN layer: 256 partial tiles and one pass of the CUs together instead of 256 and a pass each;
// the ligand-ligand products, too shallow for a launch of their own, ride along).
constexpr int TN256_BATCH = 8;
struct Tn256Batch {
    int n;
    int first[TN256_BATCH + 1];
    SgemmArgs p[TN256_BATCH];
};
__global__ __launch_bounds__(512, 1) void k_sgemm_tn256_batch(Tn256Batch b) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < TN256_BATCH; ++i)
        if (i < b.n && (int)blockIdx.x >= b.first[i]) pi = i;
    tn256_body(b.p[pi], (int)blockIdx.x - b.first[pi]);
}

// ---- weight gradients of a GVP message chain, several products per launch -----------------------------------------------------------------
// k_sgemm_tn256 with two narrow products riding on the operand slabs it streams anyway, as MFMA blocks instead of VALU riders:
//   C  [256, 256] += A^T B      (to_feats_out's scalar block: A = dpre of GVP j, B = the kept scalars of GVP j - 1)
//   X1 [256, nb2] += A^T B2     (its |Vh| block: B2 = the kept vector norms, nb2 <= 31; column 31 of the B2 tile is 1: the bias gradient)
//   X2 [na2, 256] += A2^T B     (the gate matrix of GVP j - 1: A2 = its dgate, na2 <= 32)
// Wave (wr, wc) adds one 32 x 32 block of X1 (rows 64 wr + 32 wc ..) and one of X2 (columns 128 wc + 32 wr ..) to its 2 x 4 blocks of C,
// re-using an A and a B fragment it has already read: + 2 LDS reads and + 2 MFMAs per 8.  The narrow operands travel one slab ahead
// through one register per thread and operand into a 2 x 2 KB LDS tile.  Before, each of the two narrow products was a launch of its own
// over the same [E, 256] arrays, bandwidth-bound at a third of HBM speed.
// Several products share one launch (block -> (product, K slice) by a prefix table), each with a share of the CUs proportional to its K:
// a conv's six edge-sized products then write 256 partial tiles together instead of 256 each, and the reductions read a sixth.
struct WgradProd {
    const float *A, *B, *B2, *A2, *B3;
    int lda, ldb, ldb2, lda2, ldb3, nb2, na2, nb3, K, k_chunk, slices, first;
    float *part;            // MAIN: main [slices][65536] | x1 [slices][256 nb2] | colsum [slices][256] | x2 [slices][na2 256] | colsum2 [slices][32]
                            // TOP:  x1 [slices][256 nb2] | colsum [slices][256] | x3 [slices][256 nb3] | x2 [slices][na2 256] | colsum2 [slices][32]
};
constexpr int WGRAD_MAX = 8;
struct WgradBatch {
    int n;
    WgradProd p[WGRAD_MAX];
};

// TOP = 1: the riders alone (no 256 x 256 block), with a second narrow block X3 [256, nb3] += A^T B3 -- the head GVP of a chain (A = its dpre,
// B2 = the rbf code, B3 = its 17 vector norms) sharing a pass with the gate matrix of the chain's last GVP (A2 = its dgate, B = its scalars).
// The narrow operands travel like the wide ones: one 4-byte LDS-DMA instruction per wave, tile and slab (thread t carries element
// (k = t >> 5, c = t & 31); columns past the operand's width read a clamped address -- they feed output columns nobody stores -- and column 31
// of the B2 tile reads a constant 1), NSTAGE - 1 slabs ahead.  (Through registers one slab ahead, the rider-only form waited an HBM round
// trip per 1.3-us slab: 479 us for a conv's four products; the slab ring hides it.)
__device__ const float kWgradOne = 1.0f;
// RID = 0 (with TOP = 0): the 256 x 256 blocks alone -- the EGNN trainer's dW2 = dpre2^T a1 products, eight per layer in one launch
template <int TOP, int RID = 1>
__global__ __launch_bounds__(512, 1) void k_wgrad_tnx(WgradBatch bt) {
    constexpr int ROW = 256, SLAB = SG_BK * ROW, STAGE = 2 * SLAB, NSTAGE = TN256_STAGES, NXT = !RID ? 0 : TOP ? 3 : 2, PER_SLAB = 4 + NXT;
    static_assert(RID || !TOP, "a product without a 256 x 256 block is its riders");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float xbuf[RID ? NSTAGE : 1][3][SG_BK][32];          // [stage][B2 | A2 | B3][k][column]
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < WGRAD_MAX; ++i)
        if (i < bt.n && (int)blockIdx.x >= bt.p[i].first) pi = i;
    const WgradProd &a = bt.p[pi];
    const int slice = (int)blockIdx.x - a.first;
    if (slice >= a.slices) return;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, col = lane & 31, half = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int kbeg = slice * a.k_chunk, kend = min(a.K, kbeg + a.k_chunk);
    const int nk = (kend - kbeg + SG_BK - 1) / SG_BK, nk_full = (kend - kbeg) / SG_BK;
    v16f acc[TOP ? 1 : 2][TOP ? 1 : 4], ax1, ax2, ax3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        ax1[r] = 0.0f; ax2[r] = 0.0f; ax3[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < (TOP ? 1 : 2); ++i)
#pragma unroll
            for (int j = 0; j < (TOP ? 1 : 4); ++j) acc[i][j][r] = 0.0f;
    }
    // this thread's element of the narrow tiles: row xk of the slab, column xc
    const int xk = tid >> 5, xc = tid & 31;
    const bool b2_live = a.B2 != nullptr && xc < a.nb2, a2_live = a.A2 != nullptr && xc < a.na2, b3_live = TOP && a.B3 != nullptr && xc < a.nb3;
    const float *b2p = xc == 31 ? &kWgradOne : a.B2 ? a.B2 + min(xc, max(a.nb2 - 1, 0)) : a.A;
    const float *a2p = a.A2 ? a.A2 + min(xc, max(a.na2 - 1, 0)) : a.A, *b3p = (TOP && a.B3) ? a.B3 + min(xc, max(a.nb3 - 1, 0)) : a.A;
    const size_t ld2b = xc == 31 ? 0 : a.B2 ? a.ldb2 : a.lda, ld2a = a.A2 ? a.lda2 : a.lda, ld3b = (TOP && a.B3) ? a.ldb3 : a.lda;
    auto issue = [&](int kt) {
        const int sg = kt % NSTAGE;
        float *st = smem + sg * STAGE;
        const size_t k0 = (size_t)kbeg + (size_t)kt * SG_BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wave + 8 * j;
            __builtin_amdgcn_global_load_lds((glb_void *)(a.A + (k0 + row) * a.lda + 4 * lane), (lds_void *)(st + row * ROW), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(a.B + (k0 + row) * a.ldb + 4 * lane), (lds_void *)(st + SLAB + row * ROW), 16, 0, 0);
        }
        if constexpr (RID) {
            // rows 2 wave, 2 wave + 1 of the narrow tiles: 64 consecutive floats of LDS per wave and tile
            __builtin_amdgcn_global_load_lds((glb_void *)(b2p + (k0 + xk) * ld2b), (lds_void *)(&xbuf[sg][0][2 * wave][0]), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(a2p + (k0 + xk) * ld2a), (lds_void *)(&xbuf[sg][1][2 * wave][0]), 4, 0, 0);
            if (TOP) __builtin_amdgcn_global_load_lds((glb_void *)(b3p + (k0 + xk) * ld3b), (lds_void *)(&xbuf[sg][2][2 * wave][0]), 4, 0, 0);
        }
    };
    auto wait_behind = [&](int slabs) {                 // at most `slabs` slabs of this wave's loads outstanding (PER_SLAB instructions each)
        if (slabs <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (slabs == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_SLAB) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_SLAB) : "memory");
    };
    float cs2 = 0.0f;
    auto compute = [&](const float *st, int sg) {
        const float *as = st + 64 * wr + col, *bs = st + SLAB + 128 * wc + col;
        const int sx = RID ? sg : 0;
        const float *xb = &xbuf[sx][0][0][col], *xa = &xbuf[sx][1][0][col], *x3 = &xbuf[sx][2][0][col];
        if constexpr (RID) cs2 += masked(xbuf[sx][1][xk][xc], a2_live);          // column sums of A2 (the gate bias gradient): this thread's row of every slab
#pragma unroll
        for (int ks = 0; ks < SG_BK / 2; ++ks) {
            const int k = 2 * ks + half;
            if constexpr (TOP) {
                const float av = as[k * ROW + 32 * wc], bv = bs[k * ROW + 32 * wr];
                ax1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xb[k * 32], ax1, 0, 0, 0);
                ax3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, x3[k * 32], ax3, 0, 0, 0);
                ax2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[k * 32], bv, ax2, 0, 0, 0);
            } else {
                // (fragment i of this wave is row block i ^ wc, fragment j column block j ^ wr: fragment 0 is then the one the riders pair
                //  with, without a select in the MFMA stream)
                float av[2], bv[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) av[i] = as[k * ROW + 32 * (i ^ wc)];
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = bs[k * ROW + 32 * (j ^ wr)];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                if constexpr (RID) {
                    ax1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], xb[k * 32], ax1, 0, 0, 0);
                    ax2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[k * 32], bv[0], ax2, 0, 0, 0);
                }
            }
        }
    };

    const int ahead = min(NSTAGE - 1, nk_full);
    for (int kt = 0; kt < ahead; ++kt) issue(kt);
    wait_behind(ahead - 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
    for (int kt = 0; kt < nk_full; ++kt) {
        if (kt + NSTAGE - 1 < nk_full) issue(kt + NSTAGE - 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(smem + (kt % NSTAGE) * STAGE, kt % NSTAGE);
        __builtin_amdgcn_sched_barrier(0);
        wait_behind(min(kt + NSTAGE - 1, nk_full - 1) - (kt + 1));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (nk > nk_full) {          // K tail of this range: rows past kend read a clamped address and are replaced by zero
        float *st = smem;        // (every stage is free: the loop ended on a barrier with nothing in flight)
        const int k0 = kbeg + nk_full * SG_BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = tid + 512 * j, op = u >> 10, row = (u & 1023) >> 6, c4 = u & 63;
            const int k = min(k0 + row, kend - 1);
            v4f v = *reinterpret_cast<const v4f *>((op ? a.B + (size_t)k * a.ldb : a.A + (size_t)k * a.lda) + 4 * c4);
            const bool in = k0 + row < kend;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = masked(v[e], in);
            *reinterpret_cast<v4f *>(st + op * SLAB + row * ROW + 4 * c4) = v;
        }
        if constexpr (RID) {
            const bool in = k0 + xk < kend;
            const size_t kk = (size_t)min(k0 + xk, kend - 1);
            xbuf[0][0][xk][xc] = xc == 31 ? (in ? 1.0f : 0.0f) : masked(b2p[kk * ld2b], in && b2_live);
            xbuf[0][1][xk][xc] = masked(a2p[kk * ld2a], in && a2_live);
            if (TOP) xbuf[0][2][xk][xc] = masked(b3p[kk * ld3b], in && b3_live);
        }
        __syncthreads();
        compute(st, 0);
    }
    // shares of this K range
    float *q = a.part;
    if constexpr (!TOP) {
        float *C = q + (size_t)slice * (ROW * ROW);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    C[(size_t)(64 * wr + 32 * (i ^ wc) + 8 * (r >> 2) + 4 * half + (r & 3)) * ROW + 128 * wc + 32 * (j ^ wr) + col] = acc[i][j][r];
        q += (size_t)a.slices * (ROW * ROW);
    }
    if constexpr (!RID) return;
    float *x1 = q + (size_t)slice * (ROW * a.nb2);
    q += (size_t)a.slices * (ROW * a.nb2);
    float *cs = q + (size_t)slice * ROW;
    q += (size_t)a.slices * ROW;
    float *x3 = q + (size_t)slice * (ROW * a.nb3);
    if (TOP) q += (size_t)a.slices * (ROW * a.nb3);
    float *x2 = q + (size_t)slice * (a.na2 * ROW);
    q += (size_t)a.slices * (a.na2 * ROW);
    float *c2 = q + (size_t)slice * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = 8 * (r >> 2) + 4 * half + (r & 3);
        const int m = 64 * wr + 32 * wc + rr;
        if (col < a.nb2) x1[m * a.nb2 + col] = ax1[r];
        if (col == 31) cs[m] = ax1[r];
        if (TOP && col < a.nb3) x3[m * a.nb3 + col] = ax3[r];
        if (rr < a.na2) x2[rr * ROW + 128 * wc + 32 * wr + col] = ax2[r];
    }
    // column sums of A2: the 16 row-in-slab shares of a column, in row order
    __syncthreads();
    xbuf[0][0][xk][xc] = cs2;
    __syncthreads();
    if (tid < 32) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < SG_BK; ++k) sum += xbuf[0][0][k][tid];
        c2[tid] = sum;
    }
}

// Sums of the split-K shares.  Up to six kinds of output share one launch, each a run of elements whose shares lie `stride` floats apart
// from slice to slice: the tiles (element (r, c) -> C[r][c]), the column sums of A, the column fringe (-> C[r][xc]), the row fringe
// (-> C[xr][c]), the corner, the fringe row's column sum.  An output element is summed by FOUR adjacent lanes, each over a quarter of
// the slices (in slice order, eight loads in flight), combined as ((q0 + q1) + q2) + q3: a fixed tree, so the result depends on the
// slice count alone; the slice loop is a quarter as long as with one thread per element.
struct RedSeg {
    const float *src;       // first slice's run
    long long stride;       // floats between slices
    int count;              // elements of the run
    float *dst;
    int cols;               // > 0: element j -> dst[(j / cols) * ld + j % cols] (tiles); 0: dst[j * ld]
    int ld;
    int accumulate;         // 1: dst += sum (column sums); 0: dst = sum + beta dst
};
struct RedArgs {
    RedSeg seg[6];
    int n_seg, slices;
    float beta;
};

__device__ __forceinline__ void reduce_body(const RedArgs &a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    int e = t >> 2;
    const int q = t & 3;
    int si = 0;
    while (si < a.n_seg && e >= a.seg[si].count) { e -= a.seg[si].count; ++si; }
    const bool live = si < a.n_seg;
    const RedSeg &g = a.seg[live ? si : 0];
    const int per = (a.slices + 3) >> 2, k0 = min(a.slices, q * per), k1 = min(a.slices, k0 + per);
    float s = 0.0f;
    if (live) {
        const float *p = g.src + e;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * g.stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < k1; ++k) s += p[(size_t)k * g.stride];
    }
    const float s1 = __shfl_down(s, 1), s2 = __shfl_down(s, 2), s3 = __shfl_down(s, 3);
    if (!live || q != 0) return;
    const float sum = ((s + s1) + s2) + s3;
    float *dst = g.cols > 0 ? g.dst + (size_t)(e / g.cols) * g.ld + e % g.cols : g.dst + (size_t)e * g.ld;
    if (g.accumulate) *dst += sum;
    else *dst = a.beta != 0.0f ? sum + a.beta * *dst : sum;
}
__global__ void k_sgemm_reduce(RedArgs a) { reduce_body(a); }
// the reductions of a batched product launch (wgrad_batch, grad257_batch) in one launch: blockIdx.y = product
constexpr int RED_BATCH = 8;
struct RedBatch {
    RedArgs r[RED_BATCH];
};
__global__ void k_sgemm_reduce_batch(RedBatch b) { reduce_body(b.r[blockIdx.y]); }

// y[m] = beta y[m] + sum_k A[m][k] x[k * incx]: one wave per row
__global__ __launch_bounds__(256) void k_sgemv_rows(const float *__restrict__ A, int lda, int M, int K, const float *__restrict__ x,
                                                    int incx, float beta, float *__restrict__ y, int incy) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float *a = A + (size_t)row * lda;
    float s = 0.0f;
    for (int k = lane; k < K; k += 64) s += a[k] * x[(size_t)k * incx];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[(size_t)row * incy] = beta != 0.0f ? s + beta * y[(size_t)row * incy] : s;
}

}  // namespace

int sgemm_split_slices(int M, int N, int K) {
    static const int pct = tool_env_int("KPD_SGEMM_SPLIT_PCT", 200);      // workgroups per 100 CUs (A/B runs)
    const int tiles = cdiv(M, 128) * cdiv(N, 128);
    int s = std::max(1, (pct * cu_count() / 100) / std::max(tiles, 1));
    // a slice is at least 256 deep -- or 64 (four slabs) when the whole product is a handful of tiles (the ligand-sized products of a
    // training step: 13 row tiles, K = 257): a workgroup then spends its time in the load latency of 16 consecutive slabs, which four
    // workgroups share better than one
    static const int small_pct = tool_env_int("KPD_SGEMM_SMALL_PCT", 25);      // A/B runs
    const int min_depth = tiles * 100 <= small_pct * cu_count() ? 64 : 256;
    s = std::min(s, std::max(1, K / min_depth));
    return std::min(s, SGEMM_MAX_SPLIT);
}

static kpd_status launch_reduce_batch(const RedBatch &b, int n, hipStream_t st) {
    long long most = 0;
    for (int p = 0; p < n; ++p) {
        long long total = 0;
        for (int i = 0; i < b.r[p].n_seg; ++i) total += b.r[p].seg[i].count;
        most = std::max(most, total);
    }
    if (n == 0 || most == 0) return KPD_OK;
    hipLaunchKernelGGL(k_sgemm_reduce_batch, dim3((unsigned)cdiv((int)std::min<long long>(4 * most, 0x7fffff00), 256), n), dim3(256), 0, st, b);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

static kpd_status launch_reduce(const RedArgs &r, hipStream_t st) {
    long long total = 0;
    for (int i = 0; i < r.n_seg; ++i) total += r.seg[i].count;
    hipLaunchKernelGGL(k_sgemm_reduce, dim3((unsigned)cdiv((int)std::min<long long>(4 * total, 0x7fffff00), 256)), dim3(256), 0, st, r);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status sgemm(bool tA, bool tB, int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb, float beta,
                 float *C, int ldc, hipStream_t st, float *part, size_t part_floats, float *colsum, const float *silu_pre, const float *bias,
                 float *act_out) {
    if (M <= 0 || N <= 0) return KPD_OK;
    KPD_REQUIRE(A && B && C && K > 0, KPD_ERR_INVALID, "sgemm: null operand or empty K (M=%d N=%d K=%d)", M, N, K);
    KPD_REQUIRE(!colsum || (tA && !tB), KPD_ERR_INVALID, "sgemm: column sums ride along with A^T B products only");
    SgemmArgs a;
    a.colsum = colsum; a.cs_part = nullptr; a.silu_pre = silu_pre; a.bias = bias; a.act_out = act_out;
    a.xr = a.xc = -1; a.x_part = nullptr;
    if (silu_pre || bias || act_out) part = nullptr;          // these epilogues live in the product kernel: no split along K
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.alpha = alpha; a.beta = beta;
    a.vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
    a.vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
    a.k_chunk = cdiv(K, SG_BK) * SG_BK;
    a.c_slice = 0;
    if (tA && !tB && M <= 32 && N <= 32 && K >= 8192 && part && part_floats >= (size_t)cu_count() * (M * N + M)) {
        // one partial per workgroup, about one workgroup per CU; rows per workgroup a multiple of 8 (even quarters)
        const int groups = std::min(cu_count(), cdiv(K, 2048));
        a.k_chunk = cdiv(cdiv(K, groups), 8) * 8;
        const int used = cdiv(K, a.k_chunk);
        a.C = part; a.ldc = N; a.beta = 0.0f; a.c_slice = (long long)M * N;
        if (colsum) a.cs_part = part + (size_t)used * M * N;
        hipLaunchKernelGGL(k_sgemm_tn_skinny, dim3(used), dim3(256), 0, st, a);
        KPD_LAUNCH_CHECK();
        RedArgs r;
        r.n_seg = 0; r.slices = used; r.beta = beta;
        r.seg[r.n_seg++] = RedSeg{part, (long long)M * N, M * N, C, N, ldc, 0};
        if (colsum) r.seg[r.n_seg++] = RedSeg{a.cs_part, (long long)M, M, colsum, 0, 1, 1};
        return launch_reduce(r, st);
    }
    // Fringe: a 129- / 257- / ...-wide side is tiled over its first M - 1 (N - 1) rows (columns); the last one rides along (k_sgemm)
    // instead of costing a row (column) of tiles of its own -- 9 tiles for the 257 x 257 weight gradients of the EGNN layers otherwise.
    static const bool use_fringe = tool_env_int("KPD_SGEMM_FRINGE", 1) != 0;          // A/B runs
    int Mt = M, Nt = N;                          // tiled part
    if (use_fringe && M >= 129 && (M - 1) % 128 == 0) { a.xr = M - 1; Mt = M - 1; }
    if (use_fringe && N >= 65 && (N - 1) % 64 == 0) { a.xc = N - 1; Nt = N - 1; }
    const size_t per_slice = (size_t)Mt * Nt + (colsum ? Mt : 0) + ((a.xr >= 0 || a.xc >= 0) ? (size_t)Mt + Nt + 2 : 0);
    // edge-sized weight gradients with a 256 x 256 tiled part: the whole output in one workgroup, every operand row fetched once
    static const bool use_tn256 = tool_env_int("KPD_SGEMM_TN256", 1) != 0;          // A/B runs
    if (use_tn256 && tA && !tB && part && Mt == 256 && Nt == 256 && a.vecA && a.vecB && K >= 65536 && part_floats >= 2 * per_slice) {
        int sl = (int)std::min<size_t>(std::min(std::min(cu_count(), K / 256), SGEMM_MAX_SPLIT), part_floats / per_slice);
        a.k_chunk = cdiv(cdiv(K, sl), SG_BK) * SG_BK;
        sl = cdiv(K, a.k_chunk);
        a.M = Mt; a.N = Nt;
        a.C = part; a.ldc = Nt; a.beta = 0.0f; a.c_slice = (long long)Mt * Nt;
        float *nxt = part + (size_t)sl * Mt * Nt;
        if (colsum) { a.cs_part = nxt; nxt += (size_t)sl * Mt; }
        if (a.xr >= 0 || a.xc >= 0) a.x_part = nxt;
        constexpr int lds = TN256_STAGES * 2 * SG_BK * 256 * 4;
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_sgemm_tn256), lds));
        hipLaunchKernelGGL(k_sgemm_tn256, dim3(sl), dim3(512), lds, st, a);
        KPD_LAUNCH_CHECK();
        RedArgs r;
        r.n_seg = 0; r.slices = sl; r.beta = beta;
        r.seg[r.n_seg++] = RedSeg{part, (long long)Mt * Nt, Mt * Nt, C, Nt, ldc, 0};
        if (colsum) r.seg[r.n_seg++] = RedSeg{a.cs_part, (long long)Mt, Mt, colsum, 0, 1, 1};
        if (a.x_part) {
            const long long xs = (long long)Mt + Nt + 2;
            if (a.xc >= 0) r.seg[r.n_seg++] = RedSeg{a.x_part, xs, Mt, C + a.xc, 0, ldc, 0};
            if (a.xr >= 0) r.seg[r.n_seg++] = RedSeg{a.x_part + Mt, xs, Nt, C + (size_t)a.xr * ldc, 0, 1, 0};
            if (a.xr >= 0 && a.xc >= 0) r.seg[r.n_seg++] = RedSeg{a.x_part + Mt + Nt, xs, 1, C + (size_t)a.xr * ldc + a.xc, 0, 1, 0};
            if (a.xr >= 0 && colsum) r.seg[r.n_seg++] = RedSeg{a.x_part + Mt + Nt + 1, xs, 1, colsum + a.xr, 0, 1, 1};
        }
        return launch_reduce(r, st);
    }
    int slices = part ? (int)std::min<size_t>(sgemm_split_slices(Mt, Nt, K), part_floats / per_slice) : 1;
    if (slices > 1) {
        a.k_chunk = cdiv(cdiv(K, slices), SG_BK) * SG_BK;
        slices = cdiv(K, a.k_chunk);
    }
    if (slices > 1) {
        a.C = part; a.ldc = Nt; a.beta = 0.0f; a.c_slice = (long long)Mt * Nt;
        float *nxt = part + (size_t)slices * Mt * Nt;
        if (colsum) { a.cs_part = nxt; nxt += (size_t)slices * Mt; }
        if (a.xr >= 0 || a.xc >= 0) a.x_part = nxt;
    } else {
        slices = 1;
        a.k_chunk = cdiv(K, SG_BK) * SG_BK;
    }
    a.M = Mt; a.N = Nt;
    // tile shape (measured on the engines' shapes, profiles/r03_sgemm_bench.txt): 128-row tiles throughout (256-row tiles lose 10-15 %
    // on every shape: half the workgroups per CU to hide the short K loops behind); 128 columns when that still gives every CU two
    // workgroups, else 64 (node-sized products: more, smaller workgroups balance the 256 CUs better); 32 for the 16-wide vector channels
    static const int force_wn = tool_env_int("KPD_SGEMM_WN", 0);          // A/B runs
    static const int direct = tool_env_int("KPD_SGEMM_DIRECT", 1);
    a.direct = direct;
    int wn = Nt > 64 ? 4 : Nt > 32 ? 2 : 1;
    if (wn == 4 && slices == 1 && (long long)cdiv(Mt, 128) * cdiv(Nt, 128) < 2ll * cu_count()) wn = 2;
    if (force_wn) wn = force_wn;
    const dim3 grid(cdiv(Mt, 128), cdiv(Nt, 32 * wn), slices);
    // ring depth of the direct path
    static const int force_stages = tool_env_int("KPD_SGEMM_STAGES", 0);          // A/B runs
    a.stages = 3;          // deeper rings measured slower on every shape (56 -> 71 us on the node-sized gradients at 8 stages): kept as an A/B switch
    if (force_stages) a.stages = std::min(std::max(force_stages, 3), SG_MAX_STAGES);
    KPD_REQUIRE(grid.y <= 65535u && grid.z <= 65535u, KPD_ERR_CAPACITY, "sgemm: N = %d too wide for one launch", N);
    if (wn == 4) KPD_TRY((launch_shape<1, 4>(tA, tB, grid, st, a)));
    else if (wn == 2) KPD_TRY((launch_shape<1, 2>(tA, tB, grid, st, a)));
    else KPD_TRY((launch_shape<1, 1>(tA, tB, grid, st, a)));
    KPD_LAUNCH_CHECK();
    if (slices > 1) {
        RedArgs r;
        r.n_seg = 0; r.slices = slices; r.beta = beta;
        r.seg[r.n_seg++] = RedSeg{part, (long long)Mt * Nt, Mt * Nt, C, Nt, ldc, 0};
        if (colsum) r.seg[r.n_seg++] = RedSeg{a.cs_part, (long long)Mt, Mt, colsum, 0, 1, 1};
        if (a.x_part) {
            const long long xs = (long long)Mt + Nt + 2;
            if (a.xc >= 0) r.seg[r.n_seg++] = RedSeg{a.x_part, xs, Mt, C + a.xc, 0, ldc, 0};
            if (a.xr >= 0) r.seg[r.n_seg++] = RedSeg{a.x_part + Mt, xs, Nt, C + (size_t)a.xr * ldc, 0, 1, 0};
            if (a.xr >= 0 && a.xc >= 0) r.seg[r.n_seg++] = RedSeg{a.x_part + Mt + Nt, xs, 1, C + (size_t)a.xr * ldc + a.xc, 0, 1, 0};
            if (a.xr >= 0 && colsum) r.seg[r.n_seg++] = RedSeg{a.x_part + Mt + Nt + 1, xs, 1, colsum + a.xr, 0, 1, 1};
        }
        return launch_reduce(r, st);
    }
    return KPD_OK;
}

kpd_status grad257_batch(const Grad257Item *items, int n, float *part, size_t part_floats, hipStream_t st) {
    if (n <= 0) return KPD_OK;
    KPD_REQUIRE(n <= TN256_BATCH && part, KPD_ERR_INVALID, "grad257_batch: %d products (at most %d) / no scratch", n, TN256_BATCH);
    Tn256Batch bt;
    memset(&bt, 0, sizeof(bt));
    long long ksum = 0;
    for (int i = 0; i < n; ++i) {
        const Grad257Item &it = items[i];
        KPD_REQUIRE(it.A && it.B && it.C && it.K >= 1 && (it.lda & 3) == 0 && (it.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(it.A) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(it.B) & 15) == 0 && it.ldc >= 257,
                    KPD_ERR_INVALID, "grad257_batch: bad product %d", i);
        ksum += it.K;
    }
    const int cus = cu_count();
    const size_t per_slice = (size_t)256 * 256 + 256 + (256 + 256 + 2);
    size_t used = 0;
    int first = 0;
    for (int i = 0; i < n; ++i) {
        const Grad257Item &it = items[i];
        SgemmArgs &a = bt.p[i];
        a.A = it.A; a.B = it.B; a.M = 256; a.N = 256; a.K = it.K; a.lda = it.lda; a.ldb = it.ldb; a.alpha = 1.0f; a.beta = 0.0f;
        a.vecA = a.vecB = 1; a.direct = 1; a.stages = 3;
        a.colsum = it.colsum; a.silu_pre = nullptr; a.bias = nullptr; a.act_out = nullptr;
        a.xr = 256; a.xc = 256;
        int sl = (int)std::max<long long>(1, ((long long)cus * it.K + ksum / 2) / ksum);
        sl = std::min(sl, std::max(1, it.K / 256));
        sl = (int)std::min<size_t>(sl, (part_floats - used) / per_slice / (size_t)(n - i));
        KPD_REQUIRE(sl >= 1, KPD_ERR_CAPACITY, "grad257_batch: split-sum scratch too small");
        a.k_chunk = cdiv(cdiv(it.K, sl), SG_BK) * SG_BK;
        sl = cdiv(it.K, a.k_chunk);
        float *q = part + used;
        a.C = q; a.ldc = 256; a.c_slice = 65536;
        a.cs_part = q + (size_t)sl * 65536;                 // (written only when colsum is wanted; the room is there either way)
        a.x_part = a.cs_part + (size_t)sl * 256;
        bt.first[i] = first;
        first += sl;
        used += (size_t)sl * per_slice;
    }
    bt.first[n] = first;
    bt.n = n;
    constexpr int lds = TN256_STAGES * 2 * SG_BK * 256 * 4;
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_sgemm_tn256_batch), lds));
    hipLaunchKernelGGL(k_sgemm_tn256_batch, dim3(first), dim3(512), lds, st, bt);
    KPD_LAUNCH_CHECK();
    RedBatch rb;
    memset(&rb, 0, sizeof(rb));
    for (int i = 0; i < n; ++i) {
        const Grad257Item &it = items[i];
        const SgemmArgs &a = bt.p[i];
        const int sl = bt.first[i + 1] - bt.first[i];
        RedArgs &r = rb.r[i];
        r.n_seg = 0; r.slices = sl; r.beta = 1.0f;
        r.seg[r.n_seg++] = RedSeg{a.C, 65536, 65536, it.C, 256, it.ldc, 0};
        if (it.colsum) r.seg[r.n_seg++] = RedSeg{a.cs_part, 256, 256, it.colsum, 0, 1, 1};
        const long long xs = 256 + 256 + 2;
        r.seg[r.n_seg++] = RedSeg{a.x_part, xs, 256, it.C + 256, 0, it.ldc, 0};
        r.seg[r.n_seg++] = RedSeg{a.x_part + 256, xs, 256, it.C + (size_t)256 * it.ldc, 0, 1, 0};
        r.seg[r.n_seg++] = RedSeg{a.x_part + 512, xs, 1, it.C + (size_t)256 * it.ldc + 256, 0, 1, 0};
        if (it.colsum) r.seg[r.n_seg++] = RedSeg{a.x_part + 513, xs, 1, it.colsum + 256, 0, 1, 1};
    }
    return launch_reduce_batch(rb, n, st);
}

kpd_status wgrad_batch(const WgradItem *items, int n, float *part, size_t part_floats, hipStream_t st) {
    if (n <= 0) return KPD_OK;
    KPD_REQUIRE(n <= WGRAD_MAX && part, KPD_ERR_INVALID, "wgrad_batch: %d products (at most %d) / no scratch", n, WGRAD_MAX);
    WgradBatch bt;
    memset(&bt, 0, sizeof(bt));
    long long ksum = 0;
    const bool top = items[0].C == nullptr;
    bool riders = top;
    for (int i = 0; i < n; ++i) {
        const WgradItem &it = items[i];
        KPD_REQUIRE(it.A && it.B && it.K >= 1 && (it.lda & 3) == 0 && (it.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(it.A) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(it.B) & 15) == 0 && it.nb2 >= 0 && it.nb2 <= 31 && it.na2 >= 0 && it.na2 <= 32 && it.nb3 >= 0 &&
                        it.nb3 <= 32 && (it.nb2 == 0 || (it.B2 && it.Cx1)) && (it.na2 == 0 || (it.A2 && it.Cx2)) && (it.nb3 == 0 || (it.B3 && it.Cx3)),
                    KPD_ERR_INVALID, "wgrad_batch: bad product %d", i);
        KPD_REQUIRE((it.C == nullptr) == top && (top || it.nb3 == 0), KPD_ERR_INVALID, "wgrad_batch: products with and without a 256 x 256 block in one batch");
        ksum += it.K;
        riders = riders || it.nb2 || it.na2 || it.colsum;
    }
    const int cus = cu_count();
    size_t used = 0;
    int first = 0;
    for (int i = 0; i < n; ++i) {
        const WgradItem &it = items[i];
        WgradProd &p = bt.p[i];
        p.A = it.A; p.B = it.B; p.B2 = it.nb2 ? it.B2 : nullptr; p.A2 = it.na2 ? it.A2 : nullptr; p.B3 = it.nb3 ? it.B3 : nullptr;
        p.lda = it.lda; p.ldb = it.ldb; p.ldb2 = it.ldb2; p.lda2 = it.lda2; p.ldb3 = it.ldb3; p.nb2 = it.nb2; p.na2 = it.na2; p.nb3 = it.nb3; p.K = it.K;
        const size_t per_slice = (top ? 0 : (size_t)256 * 256) + (riders ? 256 * it.nb2 + 256 + 256 * it.nb3 + (size_t)it.na2 * 256 + 32 : 0);
        // a share of the CUs proportional to K, a slice at least 256 rows deep
        int sl = (int)std::max<long long>(1, ((long long)cus * it.K + ksum / 2) / ksum);
        sl = std::min(sl, std::max(1, it.K / 256));
        sl = (int)std::min<size_t>(sl, (part_floats - used) / per_slice / (size_t)(n - i));
        KPD_REQUIRE(sl >= 1, KPD_ERR_CAPACITY, "wgrad_batch: split-sum scratch too small");
        p.k_chunk = cdiv(cdiv(it.K, sl), SG_BK) * SG_BK;
        p.slices = cdiv(it.K, p.k_chunk);
        p.first = first;
        first += p.slices;
        p.part = part + used;
        used += (size_t)p.slices * per_slice;
    }
    bt.n = n;
    constexpr int lds = TN256_STAGES * 2 * SG_BK * 256 * 4;
    if (top) {
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_wgrad_tnx<1>), lds));
        hipLaunchKernelGGL(k_wgrad_tnx<1>, dim3(first), dim3(512), lds, st, bt);
    } else if (riders) {
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_wgrad_tnx<0>), lds));
        hipLaunchKernelGGL(k_wgrad_tnx<0>, dim3(first), dim3(512), lds, st, bt);
    } else {
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_wgrad_tnx<0, 0>), lds));
        hipLaunchKernelGGL((k_wgrad_tnx<0, 0>), dim3(first), dim3(512), lds, st, bt);
    }
    KPD_LAUNCH_CHECK();
    RedBatch rb;
    memset(&rb, 0, sizeof(rb));
    for (int i = 0; i < n; ++i) {
        const WgradItem &it = items[i];
        const WgradProd &p = bt.p[i];
        RedArgs &r = rb.r[i];
        r.n_seg = 0; r.slices = p.slices; r.beta = 1.0f;
        float *q = p.part;
        if (!top) {
            r.seg[r.n_seg++] = RedSeg{q, 65536, 65536, it.C, 256, it.ldc, 0};
            q += (size_t)p.slices * 65536;
        }
        if (!riders) continue;
        if (it.nb2) r.seg[r.n_seg++] = RedSeg{q, (long long)256 * it.nb2, 256 * it.nb2, it.Cx1, it.nb2, it.ldx1, 0};
        q += (size_t)p.slices * 256 * it.nb2;
        if (it.colsum) r.seg[r.n_seg++] = RedSeg{q, 256, 256, it.colsum, 0, 1, 1};
        q += (size_t)p.slices * 256;
        if (it.nb3) r.seg[r.n_seg++] = RedSeg{q, (long long)256 * it.nb3, 256 * it.nb3, it.Cx3, it.nb3, it.ldx3, 0};
        q += (size_t)p.slices * 256 * it.nb3;
        if (it.na2) r.seg[r.n_seg++] = RedSeg{q, (long long)it.na2 * 256, it.na2 * 256, it.Cx2, 256, it.ldx2, 0};
        q += (size_t)p.slices * it.na2 * 256;
        if (it.na2 && it.colsum2) r.seg[r.n_seg++] = RedSeg{q, 32, it.na2, it.colsum2, 0, 1, 1};
    }
    return launch_reduce_batch(rb, n, st);
}

kpd_status sgemv_rows(int M, int K, const float *A, int lda, const float *x, int incx, float beta, float *y, int incy, hipStream_t st) {
    if (M <= 0) return KPD_OK;
    hipLaunchKernelGGL(k_sgemv_rows, dim3(cdiv(M, 4)), dim3(256), 0, st, A, lda, M, K, x, incx, beta, y, incy);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// include/kpd.h
extern "C" kpd_status kpd_sgemm(int32_t trans_a, int32_t trans_b, int32_t M, int32_t N, int32_t K, float alpha, const float *A, int32_t lda,
                                const float *B, int32_t ldb, float beta, float *C, int32_t ldc, float *colsum, float *workspace,
                                int64_t workspace_floats, void *stream) {
    KPD_REQUIRE(M >= 0 && N >= 0 && K >= 0, KPD_ERR_INVALID, "kpd_sgemm: negative size");
    KPD_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, KPD_ERR_INVALID, "kpd_sgemm: leading dimension too small");
    KPD_REQUIRE(workspace_floats >= 0 && (workspace || workspace_floats == 0), KPD_ERR_INVALID, "kpd_sgemm: workspace size without a workspace");
    if (M == 0 || N == 0) return KPD_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    KPD_REQUIRE(!colsum || (trans_a && !trans_b), KPD_ERR_INVALID, "kpd_sgemm: colsum goes with A^T B products only");
    if (K == 0) {
        KPD_REQUIRE(beta == 0.0f || beta == 1.0f, KPD_ERR_INVALID, "kpd_sgemm: K = 0 needs beta 0 or 1");
        if (beta == 0.0f) KPD_HIP(hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, M, st));
        return KPD_OK;
    }
    return sgemm(trans_a != 0, trans_b != 0, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st, workspace, (size_t)workspace_floats, colsum, nullptr, nullptr, nullptr);
}

// include/kpd.h
extern "C" kpd_status kpd_wgrad_batch(int32_t kind, int32_t n, const kpd_wgrad_item *items, float *workspace, int64_t workspace_floats, void *stream) {
    KPD_REQUIRE((kind == 0 || kind == 1) && n >= 0 && n <= 8 && (n == 0 || items) && workspace && workspace_floats > 0, KPD_ERR_INVALID,
                "kpd_wgrad_batch: kind %d, %d products", kind, n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (kind == 1) {
        Grad257Item g[8];
        for (int i = 0; i < n; ++i) g[i] = Grad257Item{items[i].A, items[i].B, items[i].lda, items[i].ldb, items[i].K, items[i].C, items[i].ldc, items[i].colsum};
        return grad257_batch(g, n, workspace, (size_t)workspace_floats, st);
    }
    WgradItem w[8];
    for (int i = 0; i < n; ++i) {
        const kpd_wgrad_item &s = items[i];
        w[i] = WgradItem{s.A, s.B, s.lda, s.ldb, s.K, s.C, s.ldc, s.B2, s.ldb2, s.nb2, s.Cx1, s.ldx1, s.colsum, s.A2, s.lda2, s.na2, s.Cx2, s.ldx2, s.colsum2,
                         s.B3, s.ldb3, s.nb3, s.Cx3, s.ldx3};
    }
    return wgrad_batch(w, n, workspace, (size_t)workspace_floats, st);
}

}  // namespace kpd
