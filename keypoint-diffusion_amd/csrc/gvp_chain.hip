// Edge-message GVP chain with the activations kept in registers (models/gvp.py:43-116, 287-345, 474-497).
//
// One workgroup = 64 edges, one wave = 16 of them, and a wave carries its 16 edges through the whole message
// chain on its own.  Every product is computed transposed, T^T[n][e] = sum_k W[n][k] X^T[k][e], on the 16x16x4
// fp32 MFMA: the weight is the A operand, the activations the B operand, and the result lands with the edge on
// the lane (e = lane & 15) and four consecutive features n = 4 (lane >> 4) + r in the four result registers.
// That is exactly the B-operand layout of the next product (k = 4 (lane >> 4) + r per 16-wide k tile), so
// scalars, hidden vectors, gates and output vectors of one GVP feed the next GVP straight from registers: no
// LDS round trip, and no workgroup barrier between the stages of a GVP.  Only the weights go through LDS: the
// to_feats_out / gate matrices are streamed as 16-row k-slabs ("chunks", pre-packed in A-fragment order by
// gvp_host.hip) through the LDS-DMA ring of chain_core.h that the four waves share, one barrier per chunk.
//
// Per 64-edge tile this leaves LDS traffic of one ds_read_b128 per four MFMAs and an LDS footprint of 79 KB
// (the ring is reused for the final segmented sum), so two workgroups fit a CU and one's epilogues and
// gathers overlap the other's MFMA stream.
#include <algorithm>

#include "chain_core.h"
#include "gvp_kernels.h"

namespace kpd {

typedef const __attribute__((address_space(4))) GvpTrainGvp cTrainGvp;        // entries of the trainers' slot tables, read as constant memory (CT, chain_core.h)
typedef const __attribute__((address_space(4))) GvpTrainSlot cTrainSlot;
typedef const __attribute__((address_space(4))) GvpBwdGvp cBwdGvp;
typedef const __attribute__((address_space(4))) GvpBwdSlot cBwdSlot;


namespace {

#ifndef KPD_CHAIN_NBUF
#define KPD_CHAIN_NBUF 3          // buffers of the edge kernel's weight ring (chain_core.h)
#endif
#ifndef KPD_CHAIN_SWAVE
#define KPD_CHAIN_SWAVE 1         // the wave index of the edge kernel as a scalar (ring hand-offs and LDS-DMA bases become scalar work)
#endif

template <int NTS>
struct ChainSmem {
    static constexpr int S = 16 * NTS;
    static constexpr int CH4 = NTS * 64;                 // float4 per chunk
    static constexpr int SO = S + 4;                     // row stride of the output staging tile
    static constexpr int REGION0 = (KPD_CHAIN_NBUF * CH4 * 4 > TM * SO) ? KPD_CHAIN_NBUF * CH4 * 4 : TM * SO;   // ring / output tile (floats)
    static constexpr int FLOATS = REGION0 + TM * 48 + TM + 16;
};

}  // namespace

// One GVP whose scalars x and vectors Vc already sit in registers (every GVP but the head of an edge-message chain, and
// all node-update GVPs): vec1, the [x | sh] GEMM over NTS + 1 chunks, SiLU, gates (one chunk), vec2.  acc enters holding
// the bias of this GVP and leaves holding `next_bias` (when given) for the following one.
// TR = 1 (training forward): tg names where this GVP's activations go (GvpTrainGvp), erow the lane's edge row, live whether it exists.
// (TG: const GvpTrainGvp in the kernel's arguments -- the node kernels -- or in a constant-memory slot table -- the edge kernel)
template <int NTS, class Ring, int TR = 0, class TG = cTrainGvp>
__device__ __forceinline__ void chain_generic_gvp(Ring &ring, const v4f *cb, const v4f *nb, const GvpW &gk, const float *next_bias,
                                                  v4f (&x)[NTS], v4f (&acc)[NTS], v4f (&Vc)[3], int lane, int q,
                                                  TG *tg = nullptr, size_t erow = 0, bool live = false, int skip = 0) {
    // cb: this GVP's chunks (NTS scalar slabs, the sh slab, the gate slab); nb: the next GVP's -- or, after the last one, cb + NTS chunks,
    // so that the two refills past the end re-read chunks that exist.  The chunk two ahead of local chunk i:
    constexpr int CH4 = NTS * 64;
    auto ahead = [&](int i) -> const v4f * { return i + 2 < NTS + 2 ? cb + (size_t)(i + 2) * CH4 : nb + (size_t)(i + 2 - (NTS + 2)) * CH4; };
    const v4f wh = reinterpret_cast<const v4f *>(gk.whp)[lane];
    v4f Vh[3], sh;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v4f t = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) t = mfma16(wh[r], Vc[c][r], t);
        Vh[c] = t;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[r] = sqrt1(fmaxf(Vh[0][r] * Vh[0][r] + Vh[1][r] * Vh[1][r] + Vh[2][r] * Vh[2][r], 1e-8f));
    if constexpr (TR) {
        if (live && !(skip & 4)) {
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<gv4f *>(G(tg->Vh) + (erow * 3 + c) * 16 + 4 * q) = Vh[c];
            *reinterpret_cast<gv4f *>(G(tg->sh) + erow * 16 + 4 * q) = sh;
        }
    }
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) {
        chunk_gemm<NTS>(ring.current(), x[nt], acc, lane, 4, [&] { ring.prefetch(ahead(nt)); });
        ring.release();
    }
    chunk_gemm<NTS>(ring.current(), sh, acc, lane, 4, [&] { ring.prefetch(ahead(NTS)); });
    ring.release();
    const v4f bgv = *reinterpret_cast<const v4f *>(gk.bg + 4 * q);
    const v4f wu = reinterpret_cast<const v4f *>(gk.wup)[lane];
    if constexpr (TR) {
        if (live && !(skip & 1)) {
            gfloat *pr = G(tg->pre) + erow * (16 * NTS) + 4 * q;
#pragma unroll
            for (int mt = 0; mt < NTS; ++mt) *reinterpret_cast<gv4f *>(pr + 16 * mt) = acc[mt];
        }
    }
#pragma unroll
    for (int mt = 0; mt < NTS; ++mt) x[mt] = silu4(acc[mt]);
    if constexpr (TR) {
        if (live && !(skip & 2)) {
            gfloat *sr = G(tg->s) + erow * (16 * NTS) + 4 * q;
#pragma unroll
            for (int mt = 0; mt < NTS; ++mt) *reinterpret_cast<gv4f *>(sr + 16 * mt) = x[mt];
        }
    }
    {   // the next GVP's bias -- or, after the last one, this GVP's again (never used): an unconditional load, because a conditional one
        // made hipcc copy all 64 accumulator registers before the branch in every GVP
        const float *nbias = next_bias ? next_bias : gk.b;
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) acc[mt] = *reinterpret_cast<const v4f *>(nbias + 16 * mt + 4 * q);
    }
    v4f gate;
    {
        const v4f *buf = ring.current() + lane;
        v4f ga[4] = {zero4(), zero4(), zero4(), zero4()};
        v4f wg[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wg[nt] = buf[nt * 64];
        __builtin_amdgcn_sched_barrier(0);
        ring.prefetch(ahead(NTS + 1));
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) {
            const v4f wv = nt < 4 ? wg[nt < 4 ? nt : 0] : buf[nt * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) ga[r] = mfma16(wv[r], x[nt][r], ga[r]);
        }
        ring.release();
        gate = (ga[0] + ga[1]) + (ga[2] + ga[3]) + bgv;
        if constexpr (TR) {
            if (live && !(skip & 4)) *reinterpret_cast<gv4f *>(G(tg->gate) + erow * 16 + 4 * q) = gate;          // before the sigmoid (k_gvp_gate_bwd applies it)
        }
        if (gk.vec_sigmoid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gate[r] = sigmoidf_(gate[r]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v4f t = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) t = mfma16(wu[r], Vh[c][r], t);
        Vc[c] = gate * t;
        if constexpr (TR) {
            if (live && !(skip & 4)) {
                *reinterpret_cast<gv4f *>(G(tg->Vu) + (erow * 3 + c) * 16 + 4 * q) = t;
                *reinterpret_cast<gv4f *>(G(tg->V) + (erow * 3 + c) * 16 + 4 * q) = Vc[c];
            }
        }
    }
}

// ---- f16x2 building blocks (unit layouts: pack.hip, k_pack_gvp_unit_h) -------------------------------------
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// eight scalars of a lane (result tiles 2 kb and 2 kb + 1) -> hi / lo B operands of v_mfma_f32_16x16x32_f16, x 2^6
__device__ __forceinline__ void split8(const v4f &a, const v4f &b, f32x4 &xh, f32x4 &xl) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split_pair(H_SCALE_A * a[0], H_SCALE_A * a[1], h0, l0);
    split_pair(H_SCALE_A * a[2], H_SCALE_A * a[3], h1, l1);
    split_pair(H_SCALE_A * b[0], H_SCALE_A * b[1], h2, l2);
    split_pair(H_SCALE_A * b[2], H_SCALE_A * b[3], h3, l3);
    xh = __builtin_bit_cast(f32x4, u32x4{h0, h1, h2, h3});
    xl = __builtin_bit_cast(f32x4, u32x4{l0, l1, l2, l3});
}

// acc (2^16 domain) += slab[16 k][256] . xin for one 16-row slab (rbf, sh) on v_mfma_f32_16x16x16_f16: unit kind 1
template <int NTS>
__device__ __forceinline__ void chunk_gemm16_h(const v4f *__restrict__ buf, v4f xin, v4f (&acc)[NTS], int lane) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned h0, h1, l0, l1;
    split_pair(H_SCALE_A * xin[0], H_SCALE_A * xin[1], h0, l0);
    split_pair(H_SCALE_A * xin[2], H_SCALE_A * xin[3], h1, l1);
    const h4 xh = __builtin_bit_cast(h4, u32x2{h0, h1}), xl = __builtin_bit_cast(h4, u32x2{l0, l1});
    const f32x2v *wp = reinterpret_cast<const f32x2v *>(buf) + lane;          // 8-B fragments
    f32x2v w[3][8];                                                           // [batch buffer][4 tiles x (hi, lo)]
#pragma unroll
    for (int i = 0; i < 8; ++i) w[0][i] = wp[i * 64];
#pragma unroll
    for (int b = 0; b < NTS / 4; ++b) {
        if (b + 1 < NTS / 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) w[(b + 1) % 3][i] = wp[(8 * (b + 1) + i) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[4 * b + m] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, w[b % 3][2 * m + 1]), xh, acc[4 * b + m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[4 * b + m] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, w[b % 3][2 * m]), xl, acc[4 * b + m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[4 * b + m] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, w[b % 3][2 * m]), xh, acc[4 * b + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// gate pre-activations (16 per edge) = Wg . x over the 256 scalars, unit kind 2; returns the sum in the plain domain
template <int NTS>
__device__ __forceinline__ v4f gates_h(const v4f *__restrict__ buf, const v4f (&x)[NTS], int lane) {
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(buf) + lane;
    v4f ga[3] = {zero4(), zero4(), zero4()};
#pragma unroll
    for (int kb = 0; kb < NTS / 2; ++kb) {
        f32x4 xh, xl;
        split8(x[2 * kb], x[2 * kb + 1], xh, xl);
        const f32x4 wh = wp[(2 * kb) * 64], wl = wp[(2 * kb + 1) * 64];
        ga[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wl), as_h8(xh), ga[0], 0, 0, 0);
        ga[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wh), as_h8(xl), ga[1], 0, 0, 0);
        ga[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wh), as_h8(xh), ga[2], 0, 0, 0);
    }
    return ((ga[0] + ga[1]) + ga[2]) * H_UNSCALE;
}

// The same GVP with the 256 x 256 part of its [x | sh] product on v_mfma_f32_16x16x32_f16 (f16x2 mode):
// units 0..15 of the re-packed chunk buffer (pack_gvp_chain_h) hold k-block kb = u >> 1 for output tiles 8 (u & 1) .. + 7 as hi / lo
// planes x 2^10; the scalars of result tiles 2 kb and 2 kb + 1 of the previous product are the eight B-operand slots of a lane,
// split into hi / lo halves x 2^6 in registers.  The sh slab (16 inputs) runs on v_mfma_f32_16x16x16_f16, the gates on the
// 16x16x32 form again.  The accumulator works in the 2^16 domain from the bias to the SiLU.  Vector channels as in chain_generic_gvp.
template <int NTS, class Ring, class Src>
__device__ __forceinline__ void chain_generic_gvp_h(Ring &ring, Src &chunk_src, const GvpW &gk, const float *next_bias,
                                                    v4f (&x)[NTS], v4f (&acc)[NTS], v4f (&Vc)[3], int lane, int q) {
    static_assert(NTS == 16, "the f16x2 form is built for 256 scalars");
    const v4f wh = reinterpret_cast<const v4f *>(gk.whp)[lane];
    v4f Vh[3], sh;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v4f t = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) t = mfma16(wh[r], Vc[c][r], t);
        Vh[c] = t;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[r] = sqrt1(fmaxf(Vh[0][r] * Vh[0][r] + Vh[1][r] * Vh[1][r] + Vh[2][r] * Vh[2][r], 1e-8f));
#pragma unroll
    for (int mt = 0; mt < NTS; ++mt) acc[mt] = acc[mt] * (1.0f / H_UNSCALE);
#pragma unroll
    for (int kb = 0; kb < NTS / 2; ++kb) {
        f32x4 xh, xl;
        split8(x[2 * kb], x[2 * kb + 1], xh, xl);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(ring.acquire(chunk_src)) + lane;
            // two batches of four output tiles; a batch's fragments are read one batch ahead into a third buffer (never refilled
            // right behind its readers: mfma_core.h, gemm_rows64_h)
            f32x4 w[3][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[0][i] = wp[i * 64];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (b + 1 < 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) w[b + 1][i] = wp[(8 * (b + 1) + i) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[8 * half + 4 * b + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(w[b][2 * m + 1]), as_h8(xh), acc[8 * half + 4 * b + m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[8 * half + 4 * b + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(w[b][2 * m]), as_h8(xl), acc[8 * half + 4 * b + m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[8 * half + 4 * b + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(w[b][2 * m]), as_h8(xh), acc[8 * half + 4 * b + m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            ring.release();
        }
    }
    {
        const v4f *buf = ring.acquire(chunk_src);
        chunk_gemm16_h<NTS>(buf, sh, acc, lane);
        ring.release();
    }
    const v4f bgv = *reinterpret_cast<const v4f *>(gk.bg + 4 * q);
    const v4f wu = reinterpret_cast<const v4f *>(gk.wup)[lane];
#pragma unroll
    for (int mt = 0; mt < NTS; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[mt][r] = silu(acc[mt][r] * H_UNSCALE);
    if (next_bias) {
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) acc[mt] = *reinterpret_cast<const v4f *>(next_bias + 16 * mt + 4 * q);
    }
    v4f gate;
    {
        const v4f *buf = ring.acquire(chunk_src);
        gate = gates_h<NTS>(buf, x, lane) + bgv;
        ring.release();
        if (gk.vec_sigmoid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gate[r] = sigmoidf_(gate[r]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v4f t = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) t = mfma16(wu[r], Vh[c][r], t);
        Vc[c] = gate * t;
    }
}

// phase-cycle sums for profiles/tools/gvp_stamps.py (a.stamps is null in production)
#define CHAIN_STAMP(idx)                                                                   \
    if (stamps && tid == 0) {                                                              \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();                     \
        atomicAdd(stamps + (idx), t_now - t_prev);                                         \
        t_prev = t_now;                                                                    \
    }

// TR = 1: the training forward (a.train, gvp_kernels.h) -- the same chain on current weights, every activation the backward pass reads stored
// on the way (8.5 KB per edge), node vectors and vector pieces in the trainers' [3][16] layout.
template <int NTS, int HM = 0, int TR = 0>
__global__ __launch_bounds__(256, 2) void k_gvp_chain(GvpEdgeArgs a) {
    using L = ChainSmem<NTS>;
    constexpr int S = L::S, CH4 = L::CH4, SO = L::SO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *O = smem;
    float *Vout = smem + L::REGION0;
    int *sdst = reinterpret_cast<int *>(Vout + TM * 48);
    int *misc = sdst + TM;

    const int tid = threadIdx.x, wave = KPD_CHAIN_SWAVE ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6, lane = tid & 63;
    unsigned long long *stamps = a.stamps;
    unsigned long long t_prev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    const int T = a.meta[8];
    const int chunk_tiles = (T + 7) >> 3;
    const int bi = blockIdx.x >> 3;
    if (bi >= chunk_tiles) return;
    const int tile = (blockIdx.x & 7) * chunk_tiles + bi;      // consecutive tiles stay on one XCD (shared weights in L2)
    if (tile >= T) return;
    int et = 0;
#pragma unroll
    for (int e = 1; e < 4; ++e)
        if (tile >= a.meta[4 + e]) et = e;
    const int tile_in_et = tile - a.meta[4 + et];
    const int e0 = tile_in_et * TM;
    const int ne = min(TM, a.meta[et] - e0);
    const int snt = (et == 1 || et == 3) ? 1 : 0, dnt = (et >= 2) ? 1 : 0;      // ll, kl, lk, kk
    const int *__restrict__ esrc = a.src[et];
    const int *__restrict__ edst = a.dst[et];
    const int n_gvps = a.n_gvps;

    // ---- weight chunk pipeline ----------------------------------------------------------------------
    const int h0 = a.g[et][0].h;
    const int n_ht = (h0 + 15) >> 4;
    const int n0 = 2 + n_ht;
    const int total = n0 + (n_gvps - 1) * (NTS + 2);
    auto chunk_src = [&](int c) -> const v4f * {
        int stage = 0, local = c;
        if (c >= n0) {
            stage = 1 + (c - n0) / (NTS + 2);
            local = (c - n0) - (stage - 1) * (NTS + 2);
        }
        return reinterpret_cast<const v4f *>(HM ? a.g[et][stage].chain_h : a.g[et][stage].chain) + (size_t)local * CH4;      // (wave-uniform)
    };
    ChunkRing<CH4, KPD_CHAIN_NBUF> ring;
    ring.init(smem, total, wave, tid);
    ring.start(chunk_src);
    auto acquire = [&]() -> const v4f * { return ring.acquire(chunk_src); };
    auto release = [&]() { ring.release(); };

    // ---- this lane's edge ---------------------------------------------------------------------------
    const int el = lane & 15, q = lane >> 4;
    const int row = 16 * wave + el;
    const int eidx = e0 + min(row, ne - 1);
    const int u = esrc[eidx], vd = edst[eidx];
    [[maybe_unused]] const bool live = row < ne;                        // (training form: rows past the end are not stored)
    [[maybe_unused]] const size_t erow = (size_t)eidx;
    [[maybe_unused]] cTrainSlot *tsl = TR ? CT(a.train + et) : nullptr;

    // run boundaries of the dst-sorted tile for the segmented sum (wave 0, one lane per row)
    if (tid < TM) {
        const int e = e0 + min(tid, ne - 1);
        const int v = edst[e];
        sdst[tid] = v;
        const int vprev = tid > 0 ? edst[e0 + min(tid - 1, ne - 1)] : (e0 > 0 ? edst[e0 - 1] : -1);
        const int vnext = tid + 1 < ne ? edst[e0 + tid + 1] : -2;
        const unsigned long long ends = __ballot(tid < ne && vnext != v);
        if (tid == 0) {
            misc[0] = (vprev == v) ? 1 : 0;
            misc[2] = (int)(ends & 0xffffffffu);
            misc[3] = (int)(ends >> 32);
        }
    }

    // ---- GVP 0: inputs ------------------------------------------------------------------------------
    v4f acc[NTS];
    {   // per-node blocks of to_feats_out (k_gvp_proj) enter as the accumulator's initial value
        const float *ps = a.Psrc[et] + (size_t)u * S + 4 * q;
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) acc[mt] = *reinterpret_cast<const v4f *>(ps + 16 * mt);
        if (a.use_dst) {
            const float *pd = a.Pdst[et] + (size_t)vd * S + 4 * q;
#pragma unroll
            for (int mt = 0; mt < NTS; ++mt) acc[mt] += *reinterpret_cast<const v4f *>(pd + 16 * mt);
        }
        if constexpr (HM) {       // f16x2 mode: the accumulator works in the 2^16 domain up to the SiLU
#pragma unroll
            for (int mt = 0; mt < NTS; ++mt) acc[mt] = acc[mt] * (1.0f / H_UNSCALE);
        }
    }
    v4f Vc[3];          // current vectors: Vc[c][r] = v[e][4 q + r][c]
    v4f x[NTS];         // current scalars: x[nt][r] = s[e][16 nt + 4 q + r]
    v4f gate;
    {
        const GvpW &g0 = a.g[et][0];
        const float *xs = a.x[snt] + (size_t)u * 3, *xd = a.x[dnt] + (size_t)vd * 3;
        const float dx = xs[0] - xd[0], dy = xs[1] - xd[1], dz = xs[2] - xd[2];
        const float dij = sqrt1(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f;
        const float inv_dij = rcp1(dij);
        const float xdv[3] = {dx * inv_dij, dy * inv_dij, dz * inv_dij};
        v4f rbf;
        {   // exp(-((d - mu_i) / sigma)^2), mu = linspace(0, D_max, 16), sigma = D_max / 16 (gvp.py:26-41): one multiply per centre
            // and exp2 of the scaled square instead of three IEEE divisions and a full-range expf per value
            const float inv_sigma = 16.0f / a.rbf_dmax, mu_step = a.rbf_dmax * (1.0f / 15.0f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float zz = (dij - mu_step * (float)(4 * q + r)) * inv_sigma;
                rbf[r] = __builtin_amdgcn_exp2f(-1.4426950408889634f * (zz * zz));
            }
        }
        // vectors of the two end points: 12 consecutive floats (4 channels x xyz) per lane
        v4f Vs[3], Vd[3];
        if constexpr (TR) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Vs[c] = *reinterpret_cast<const v4f *>(a.v[snt] + (size_t)u * 48 + 16 * c + 4 * q);
        } else {
            const v4f *vs = reinterpret_cast<const v4f *>(a.v[snt] + (size_t)u * 48 + 12 * q);
            const v4f t0 = vs[0], t1 = vs[1], t2 = vs[2];
            const float f[12] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3]};
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) Vs[c][r] = f[3 * r + c];
        }
        if (a.use_dst) {
            const v4f *vs = reinterpret_cast<const v4f *>(a.v[dnt] + (size_t)vd * 48 + 12 * q);
            const v4f t0 = vs[0], t1 = vs[1], t2 = vs[2];
            const float f[12] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3]};
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) Vd[c][r] = f[3 * r + c];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) Vd[c] = zero4();
        }

        // vec1: Vh = Wh^T [source | destination | x_diff], sh = |Vh|                     (gvp.py:96-99)
        const v4f *whp = reinterpret_cast<const v4f *>(g0.whp) + lane;
        v4f Vh[3][3], sh[3];
#pragma unroll
        for (int ht = 0; ht < 3; ++ht) {
            if (ht < n_ht) {
                const v4f ws = whp[(0 * 3 + ht) * 64], wx = whp[(2 * 3 + ht) * 64];
                v4f wd = zero4();
                if (a.use_dst) wd = whp[(1 * 3 + ht) * 64];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    v4f t = zero4();
#pragma unroll
                    for (int r = 0; r < 4; ++r) t = mfma16(ws[r], Vs[c][r], t);
                    if (a.use_dst) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) t = mfma16(wd[r], Vd[c][r], t);
                    }
                    t = mfma16(wx[0], q == 0 ? xdv[c] : 0.0f, t);
                    Vh[ht][c] = t;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sh[ht][r] = sqrt1(fmaxf(Vh[ht][0][r] * Vh[ht][0][r] + Vh[ht][1][r] * Vh[ht][1][r] + Vh[ht][2][r] * Vh[ht][2][r], 1e-8f));
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) Vh[ht][c] = zero4();
                sh[ht] = zero4();
            }
        }
        const int tail = h0 - 16 * (n_ht - 1);            // valid rows of the last hidden tile
        const int tail_reg = min(4, tail);
        if constexpr (TR) {       // geometry, message input vectors [x_diff | source], hidden vectors and their norms (17 channels)
            if (live && !(a.train_skip & 4)) {
                if (q == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        G(tsl->unit)[erow * 3 + c] = xdv[c];
                        G(tsl->vin)[(erow * 3 + c) * 17] = xdv[c];
                        G(tsl->g[0].Vh)[(erow * 3 + c) * 17 + 16] = Vh[1][c][0];
                    }
                    G(tsl->g[0].sh)[erow * 17 + 16] = sh[1][0];
                }
                *reinterpret_cast<gv4f *>(G(tsl->rbf) + erow * 16 + 4 * q) = rbf;
                // 17-float rows: four consecutive channels as ONE 16-byte store at a 4-byte-aligned address (the hardware takes it; as four
                // 4-byte stores these three arrays were 30 of the head's store instructions and wrote 0.7 GB per launch in partial sectors)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    *reinterpret_cast<gv4f_u *>(G(tsl->vin) + (erow * 3 + c) * 17 + 1 + 4 * q) = Vs[c];
                    *reinterpret_cast<gv4f_u *>(G(tsl->g[0].Vh) + (erow * 3 + c) * 17 + 4 * q) = Vh[0][c];
                }
                *reinterpret_cast<gv4f_u *>(G(tsl->g[0].sh) + erow * 17 + 4 * q) = sh[0];
            }
        }

        ring.first();
        CHAIN_STAMP(0)
        // scalar GEMM: [rbf | sh] part of to_feats_out
        {
            const v4f *buf = acquire();
            if constexpr (HM) chunk_gemm16_h<NTS>(buf, rbf, acc, lane);
            else chunk_gemm<NTS>(buf, rbf, acc, lane, 4);
            release();
        }
#pragma unroll
        for (int ht = 0; ht < 3; ++ht) {
            if (ht < n_ht) {
                const v4f *buf = acquire();
                if constexpr (HM) chunk_gemm16_h<NTS>(buf, sh[ht], acc, lane);       // rows past h carry zero weights
                else chunk_gemm<NTS>(buf, sh[ht], acc, lane, ht == n_ht - 1 ? tail_reg : 4);
                release();
            }
        }
        CHAIN_STAMP(1)
        const v4f bgv = *reinterpret_cast<const v4f *>(g0.bg + 4 * q);
        const v4f *wup = reinterpret_cast<const v4f *>(g0.wup) + lane;
        v4f wu[3];
#pragma unroll
        for (int ht = 0; ht < 3; ++ht) wu[ht] = ht < n_ht ? wup[ht * 64] : zero4();
        if constexpr (TR) {
            if (live && !(a.train_skip & 1)) {
                gfloat *pr = G(tsl->g[0].pre) + erow * S + 4 * q;
#pragma unroll
                for (int mt = 0; mt < NTS; ++mt) *reinterpret_cast<gv4f *>(pr + 16 * mt) = acc[mt];
            }
        }
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) x[mt] = silu4(HM ? acc[mt] * H_UNSCALE : acc[mt]);
        if constexpr (TR) {
            if (live && !(a.train_skip & 2)) {
                gfloat *sr = G(tsl->g[0].s) + erow * S + 4 * q;
#pragma unroll
                for (int mt = 0; mt < NTS; ++mt) *reinterpret_cast<gv4f *>(sr + 16 * mt) = x[mt];
            }
        }
        {   // (unconditional: see chain_generic_gvp)
            const float *bn = a.g[et][n_gvps > 1 ? 1 : 0].b + 4 * q;
#pragma unroll
            for (int mt = 0; mt < NTS; ++mt) acc[mt] = *reinterpret_cast<const v4f *>(bn + 16 * mt);
        }
        CHAIN_STAMP(2)
        // gates                                                                       (gvp.py:105-107)
        {
            if constexpr (HM) {
                const v4f *buf = acquire();
                gate = gates_h<NTS>(buf, x, lane) + bgv;
                release();
            } else {
                const v4f *buf = acquire() + lane;
                v4f ga[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
                for (int nt = 0; nt < NTS; ++nt) {
                    const v4f wg = buf[nt * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ga[r] = mfma16(wg[r], x[nt][r], ga[r]);
                }
                release();
                gate = (ga[0] + ga[1]) + (ga[2] + ga[3]) + bgv;
            }
            if constexpr (TR) {
                if (live && !(a.train_skip & 4)) *reinterpret_cast<gv4f *>(G(tsl->g[0].gate) + erow * 16 + 4 * q) = gate;
            }
            if (g0.vec_sigmoid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gate[r] = sigmoidf_(gate[r]);
            }
        }
        CHAIN_STAMP(3)
        // vec2: v' = gate * Wu^T Vh                                                   (gvp.py:97, 111)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v4f t = zero4();
#pragma unroll
            for (int ht = 0; ht < 3; ++ht) {
                if (ht < n_ht) {
                    const int nr = ht == n_ht - 1 ? tail_reg : 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < nr) t = mfma16(wu[ht][r], Vh[ht][c][r], t);
                }
            }
            Vc[c] = gate * t;
            if constexpr (TR) {
                if (live && !(a.train_skip & 4)) {
                    *reinterpret_cast<gv4f *>(G(tsl->g[0].Vu) + (erow * 3 + c) * 16 + 4 * q) = t;
                    *reinterpret_cast<gv4f *>(G(tsl->g[0].V) + (erow * 3 + c) * 16 + 4 * q) = Vc[c];
                }
            }
        }
        CHAIN_STAMP(4)
    }

    // ---- GVP 1 .. n-1: scalars and vectors come from the previous GVP's registers ------------------------
#pragma unroll 1
    for (int k = 1; k < n_gvps; ++k) {
        if constexpr (HM) chain_generic_gvp_h<NTS>(ring, chunk_src, a.g[et][k], k + 1 < n_gvps ? a.g[et][k + 1].b : nullptr, x, acc, Vc, lane, q);
        else {
            const v4f *cb = reinterpret_cast<const v4f *>(a.g[et][k].chain);
            const v4f *nb = k + 1 < n_gvps ? reinterpret_cast<const v4f *>(a.g[et][k + 1].chain) : cb + (size_t)NTS * CH4;
            if constexpr (TR) chain_generic_gvp<NTS, decltype(ring), 1>(ring, cb, nb, a.g[et][k], k + 1 < n_gvps ? a.g[et][k + 1].b : nullptr, x, acc, Vc, lane, q,
                                                                         &tsl->g[k], erow, live, a.train_skip);
            else chain_generic_gvp<NTS>(ring, cb, nb, a.g[et][k], k + 1 < n_gvps ? a.g[et][k + 1].b : nullptr, x, acc, Vc, lane, q);
        }
        CHAIN_STAMP(6)
    }

    // ---- messages -> LDS: the ring is reused, so drain the (redundant) tail fetches first -----------------
    ring.drain();
    {
        float *orow = O + row * SO + 4 * q;
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) *reinterpret_cast<v4f *>(orow + 16 * mt) = x[mt];
        if constexpr (TR) {
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<v4f *>(Vout + row * 48 + 16 * c + 4 * q) = Vc[c];
        } else {
            float f[12];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) f[3 * r + c] = Vc[c][r];
            v4f *vo = reinterpret_cast<v4f *>(Vout + row * 48 + 12 * q);
            vo[0] = v4f{f[0], f[1], f[2], f[3]};
            vo[1] = v4f{f[4], f[5], f[6], f[7]};
            vo[2] = v4f{f[8], f[9], f[10], f[11]};
        }
    }
    lds_barrier();
    CHAIN_STAMP(10)

    // ---- segmented sums over dst: scalars (thread = column) and the 48 vector floats in ONE pass over the rows --------
    // (the vector columns ride on the first 48 threads -- S = 256 -- or on the last wave's spare lanes; a second pass for them
    // kept wave 0, and with it the workgroup's slot, 5 k cycles longer).  The run structure is wave-uniform: as scalars.
    const int first_is_cont = __builtin_amdgcn_readfirstlane(misc[0]);
    const unsigned long long endmask = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(misc[3]) << 32) |
                                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(misc[2]);
    const int vt = S < 256 ? tid - 192 : tid;
    const bool do_s = tid < S, do_v = vt >= 0 && vt < 48;
    if (do_s || do_v) {
        float *smain = a.ms_main[et], *scont = a.ms_cont[et] + (size_t)tile_in_et * S;
        float *vmain = a.mv_main[et], *vcont = a.mv_cont[et] + (size_t)tile_in_et * 48;
        const int vcol = do_v ? vt : 0, scol = do_s ? tid : 0;
        float run = 0.0f, runv = 0.0f;
        int piece = 0;
#pragma unroll 1
        for (int r0 = 0; r0 < TM; r0 += 16) {
            if (r0 >= ne) break;
            float v[16], u[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = O[(r0 + i) * SO + scol];
            if (__builtin_amdgcn_readfirstlane(wave == (S < 256 ? 3 : 0))) {
#pragma unroll
                for (int i = 0; i < 16; ++i) u[i] = Vout[(r0 + i) * 48 + vcol];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) u[i] = 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // (rows past ne carry a repeated edge; they follow the last run end -- endmask has no bit there -- so what they add to the
                // running sums is never stored: no per-row select needed)
                run += v[i];
                runv += u[i];
                if ((endmask >> (r0 + i)) & 1ull) {
                    const bool cont = piece == 0 && first_is_cont;
                    const int dv = sdst[r0 + i];
                    if (do_s) (cont ? scont : smain + (size_t)dv * S)[tid] = run;
                    if (do_v) (cont ? vcont : vmain + (size_t)dv * 48)[vt] = runv;
                    run = 0.0f;
                    runv = 0.0f;
                    ++piece;
                }
            }
        }
    }
    CHAIN_STAMP(11)
}

// ---- backward of the message chains (training; autograd of gvp.py:89-116 under :540-551), register-chained --------------------------
// The same tiling as k_gvp_chain run in reverse: a wave carries the gradients of its 16 edges' messages from the aggregated-message
// gradient of the destination node down to the head GVP, through registers.  Per GVP: gate backward (kept gate / Vu), the scalar gradient
// through the gates (Wg^T, one chunk), SiLU' at the kept pre-activation, ds_in = W[:, :256]^T dpre over 16 chunks (the forward's GEMM with
// the transposed k-slabs), d|Vh| = W[:, 256:]^T dpre (one chunk of one-tile slabs), and the 16-channel vector half (Wu^T, the norm term,
// Wh^T) as 16x16 MFMAs.  What the weight-gradient products need -- dpre, dgate, dVu, d|Vh| per GVP, drbf at the head -- is stored on the
// way (1.3 KB per edge and GVP); the products themselves (K = edges) stay with sgemm.hip and the vector-weight kernels.  Replaces, per
// GVP and edge type, a gate kernel, a K = 16 GEMM with the SiLU' epilogue and a weight-stationary 256 x 256 GEMM, each a pass over
// [E, 256] arrays.
__device__ __forceinline__ v4f silu_grad4(v4f p) {          // d SiLU(p) / dp = s (1 + p (1 - s)), s = sigmoid(p)
    v4f s;
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = sigmoidf_(p[r]);
    return s * (1.0f + p * (1.0f - s));
}

// gate backward of one GVP from the kept gate pre-activation and Vu: dgate (stored), dVu (stored, returned in dV)
template <class TF, class TO>
__device__ __forceinline__ v4f gate_bwd_lane(TF *f, TO *o, v4f (&dV)[3], size_t erow, bool live, int q) {
    const v4f gp = *reinterpret_cast<const gv4f *>(G(f->gate) + erow * 16 + 4 * q);
    v4f Vu[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) Vu[c] = *reinterpret_cast<const gv4f *>(G(f->Vu) + (erow * 3 + c) * 16 + 4 * q);
    v4f sg;
#pragma unroll
    for (int r = 0; r < 4; ++r) sg[r] = sigmoidf_(gp[r]);
    v4f dgate = (dV[0] * Vu[0] + dV[1] * Vu[1] + dV[2] * Vu[2]) * (sg * (1.0f - sg));
#pragma unroll
    for (int c = 0; c < 3; ++c) dV[c] = dV[c] * sg;
    if (live) {
        *reinterpret_cast<gv4f *>(G(o->dgate) + erow * 16 + 4 * q) = dgate;
#pragma unroll
        for (int c = 0; c < 3; ++c) *reinterpret_cast<gv4f *>(G(o->dVu) + (erow * 3 + c) * 16 + 4 * q) = dV[c];
    }
    return dgate;
}

// one 16-output product over the 16 NTS registers of x against a chunk of one-tile slabs (the gate form of chain_generic_gvp)
template <int NTS, class H>
__device__ __forceinline__ v4f chunk_tile_product(const v4f *__restrict__ buf0, const v4f (&x)[NTS], int lane, H &&after_first_reads) {
    const v4f *buf = buf0 + lane;
    v4f ga[4] = {zero4(), zero4(), zero4(), zero4()};
    v4f wg[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wg[nt] = buf[nt * 64];
    __builtin_amdgcn_sched_barrier(0);
    after_first_reads();
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) {
        const v4f wv = nt < 4 ? wg[nt < 4 ? nt : 0] : buf[nt * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[r] = mfma16(wv[r], x[nt][r], ga[r]);
    }
    return (ga[0] + ga[1]) + (ga[2] + ga[3]);
}

// backward of one generic GVP: acc holds dL/ds of its outputs on entry and dL/ds of its inputs on exit, dV likewise for the vectors; x is scratch
template <int NTS, class Ring, class TF, class TO>
__device__ __forceinline__ void chain_generic_gvp_bwd(Ring &ring, const v4f *cb, const v4f *nb, const GvpBwdW &w, TF *f, TO *o,
                                                      v4f (&x)[NTS], v4f (&acc)[NTS], v4f (&dV)[3], size_t erow, bool live, int lane, int q) {
    constexpr int S = 16 * NTS, CH4 = NTS * 64, NG = NTS + 2;
    auto ahead = [&](int i) -> const v4f * { return i + 2 < NG ? cb + (size_t)(i + 2) * CH4 : nb + (size_t)(i + 2 - NG) * CH4; };
    const v4f dgate = gate_bwd_lane(f, o, dV, erow, live, q);
    // kept pre-activation: requested before the gate chunk, used behind it
    {
        const gfloat *pr = G(f->pre) + erow * S + 4 * q;
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) x[mt] = *reinterpret_cast<const gv4f *>(pr + 16 * mt);
    }
    chunk_gemm<NTS>(ring.current(), dgate, acc, lane, 4, [&] { ring.prefetch(ahead(0)); });
    ring.release();
#pragma unroll
    for (int mt = 0; mt < NTS; ++mt) x[mt] = acc[mt] * silu_grad4(x[mt]);
    if (live) {
        gfloat *dp = G(o->dpre) + erow * S + 4 * q;
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) *reinterpret_cast<gv4f *>(dp + 16 * mt) = x[mt];
    }
#pragma unroll
    for (int mt = 0; mt < NTS; ++mt) acc[mt] = zero4();
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) {
        chunk_gemm<NTS>(ring.current(), x[nt], acc, lane, 4, [&] { ring.prefetch(ahead(1 + nt)); });
        ring.release();
    }
    const v4f dsh = chunk_tile_product<NTS>(ring.current(), x, lane, [&] { ring.prefetch(ahead(NTS + 1)); });
    ring.release();
    // vector half: dVh = Wu dVu + dsh Vh / |Vh| (where the clamp of the norm is inactive), dv_in = Wh dVh
    v4f Vh[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) Vh[c] = *reinterpret_cast<const gv4f *>(G(f->Vh) + (erow * 3 + c) * 16 + 4 * q);
    const v4f sh = *reinterpret_cast<const gv4f *>(G(f->sh) + erow * 16 + 4 * q);
    if (live) *reinterpret_cast<gv4f *>(G(o->dsh) + erow * 16 + 4 * q) = dsh;
    const v4f wut = reinterpret_cast<const v4f *>(w.wut)[lane], wht = reinterpret_cast<const v4f *>(w.wht)[lane];
    v4f nrm;
#pragma unroll
    for (int r = 0; r < 4; ++r) nrm[r] = sh[r] * sh[r] > 1e-8f ? dsh[r] * rcp1(sh[r]) : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v4f t = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) t = mfma16(wut[r], dV[c][r], t);
        t += nrm * Vh[c];
        v4f u = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) u = mfma16(wht[r], t[r], u);
        dV[c] = u;
    }
}

template <int NTS>
__global__ __launch_bounds__(256, 2) void k_gvp_chain_bwd(GvpEdgeBwdArgs a) {
    constexpr int S = 16 * NTS, CH4 = NTS * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int T = a.meta[8];
    const int chunk_tiles = (T + 7) >> 3;
    const int bi = blockIdx.x >> 3;
    if (bi >= chunk_tiles) return;
    const int tile = (blockIdx.x & 7) * chunk_tiles + bi;
    if (tile >= T) return;
    int et = 0;
#pragma unroll
    for (int e = 1; e < 4; ++e)
        if (tile >= a.meta[4 + e]) et = e;
    const int e0 = (tile - a.meta[4 + et]) * TM;
    const int ne = min(TM, a.meta[et] - e0);
    const int dnt = (et >= 2) ? 1 : 0;
    const int n_gvps = a.n_gvps;
    cTrainSlot *fs = CT(a.fwd + et);
    cBwdSlot *os = CT(a.out + et);

    // chunk sequence: GVP n - 1 .. 1 (NTS + 2 chunks each), then the head's four
    constexpr int NG = NTS + 2;
    const int total = (n_gvps - 1) * NG + 4;
    const v4f *first = reinterpret_cast<const v4f *>(a.g[et][n_gvps - 1].chain);
    auto chunk_src = [&](int c) -> const v4f * { return first + (size_t)c * CH4; };       // (chunks 0 and 1 only: every GVP has at least four)
    ChunkRing<CH4, KPD_CHAIN_NBUF> ring;
    ring.init(smem, total, wave, tid);
    ring.start(chunk_src);

    const int el = lane & 15, q = lane >> 4;
    const int row = 16 * wave + el;
    const bool live = row < ne;
    const size_t erow = (size_t)(e0 + min(row, ne - 1));
    const int vd = a.dst[et][erow];
    float sc;
    if (a.mode == 1) sc = 1.0f / (float)(a.rowptr[et][vd + 1] - a.rowptr[et][vd]);
    else sc = 1.0f / (a.mode == 2 ? a.z[dnt][a.bidx[dnt][vd]] : a.norm);

    // gradient of this edge's message = scale x gradient of the destination's aggregated messages
    v4f acc[NTS], x[NTS], dV[3];
    {
        const float *gsp = a.gs[dnt] + (size_t)vd * S + 4 * q;
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) acc[mt] = *reinterpret_cast<const v4f *>(gsp + 16 * mt) * sc;
#pragma unroll
        for (int c = 0; c < 3; ++c) dV[c] = *reinterpret_cast<const v4f *>(a.gv[dnt] + (size_t)vd * 48 + 16 * c + 4 * q) * sc;
    }
    ring.first();

#pragma unroll 1
    for (int k = n_gvps - 1; k >= 1; --k) {
        const v4f *cb = reinterpret_cast<const v4f *>(a.g[et][k].chain), *nb = reinterpret_cast<const v4f *>(a.g[et][k - 1].chain);
        chain_generic_gvp_bwd<NTS>(ring, cb, nb, a.g[et][k], &fs->g[k], &os->g[k], x, acc, dV, erow, live, lane, q);
    }

    // head GVP: gates, SiLU', the rbf and |Vh| blocks; its source-scalar block and its 17-channel vector half are the caller's
    {
        cTrainGvp *f = &fs->g[0];
        cBwdGvp *o = &os->g[0];
        const v4f *hb = reinterpret_cast<const v4f *>(a.g[et][0].chain);
        auto ahead = [&](int i) -> const v4f * { return hb + (size_t)(i + 2 < 4 ? i + 2 : 3) * CH4; };
        const v4f dgate = gate_bwd_lane(f, o, dV, erow, live, q);
        {
            const gfloat *pr = G(f->pre) + erow * S + 4 * q;
#pragma unroll
            for (int mt = 0; mt < NTS; ++mt) x[mt] = *reinterpret_cast<const gv4f *>(pr + 16 * mt);
        }
        chunk_gemm<NTS>(ring.current(), dgate, acc, lane, 4, [&] { ring.prefetch(ahead(0)); });
        ring.release();
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) x[mt] = acc[mt] * silu_grad4(x[mt]);
        if (live) {
            gfloat *dp = G(o->dpre) + erow * S + 4 * q;
#pragma unroll
            for (int mt = 0; mt < NTS; ++mt) *reinterpret_cast<gv4f *>(dp + 16 * mt) = x[mt];
        }
        const v4f drbf = chunk_tile_product<NTS>(ring.current(), x, lane, [&] { ring.prefetch(ahead(1)); });
        ring.release();
        const v4f dsh0 = chunk_tile_product<NTS>(ring.current(), x, lane, [&] { ring.prefetch(ahead(2)); });
        ring.release();
        const v4f dsh1 = chunk_tile_product<NTS>(ring.current(), x, lane, [&] { ring.prefetch(ahead(3)); });
        ring.release();
        if (live) {
            *reinterpret_cast<gv4f *>(G(os->drbf) + erow * 16 + 4 * q) = drbf;
#pragma unroll
            for (int r = 0; r < 4; ++r) G(o->dsh)[erow * 17 + 4 * q + r] = dsh0[r];
            if (q == 0) G(o->dsh)[erow * 17 + 16] = dsh1[0];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's tail fetches must not outlive the workgroup's LDS
}

// the same for a node-update chain: rows = nodes of one type, every GVP generic, gradients in and out through node-sized arrays
template <int NTS>
__global__ __launch_bounds__(256, 2) void k_gvp_node_chain_bwd(GvpNodeBwdArgs a) {
    constexpr int S = 16 * NTS, CH4 = NTS * 64, NG = NTS + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int n_gvps = a.n_gvps;
    const v4f *first = reinterpret_cast<const v4f *>(a.g[n_gvps - 1].chain);
    auto chunk_src = [&](int c) -> const v4f * { return first + (size_t)c * CH4; };
    ChunkRing<CH4, KPD_CHAIN_NBUF> ring;
    ring.init(smem, n_gvps * NG, wave, tid);
    ring.start(chunk_src);
    const int el = lane & 15, q = lane >> 4;
    const int vr = (int)blockIdx.x * TM + 16 * wave + el;
    const bool live = vr < a.n;
    const size_t erow = (size_t)min(vr, a.n - 1);
    v4f acc[NTS], x[NTS], dV[3];
    {
        const float *gsp = a.ds + erow * S + 4 * q;
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) acc[mt] = *reinterpret_cast<const v4f *>(gsp + 16 * mt);
#pragma unroll
        for (int c = 0; c < 3; ++c) dV[c] = *reinterpret_cast<const v4f *>(a.dV + erow * 48 + 16 * c + 4 * q);
    }
    ring.first();
#pragma unroll 1
    for (int k = n_gvps - 1; k >= 0; --k) {
        const v4f *cb = reinterpret_cast<const v4f *>(a.g[k].chain);
        const v4f *nb = k > 0 ? reinterpret_cast<const v4f *>(a.g[k - 1].chain) : cb;          // (past the end: chunks that exist, never consumed)
        chain_generic_gvp_bwd<NTS>(ring, cb, nb, a.g[k], &a.f[k], &a.o[k], x, acc, dV, erow, live, lane, q);
    }
    if (live) {
        float *so = a.ds_in + erow * S + 4 * q;
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) *reinterpret_cast<v4f *>(so + 16 * mt) = acc[mt];
#pragma unroll
        for (int c = 0; c < 3; ++c) *reinterpret_cast<v4f *>(a.dv_in + erow * 48 + 16 * c + 4 * q) = dV[c];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's tail fetches must not outlive the workgroup's LDS
}

// ---- node update (gvp.py:499-536), register-chained ---------------------------------------------------------
// One wave = 16 nodes: aggregate the message pieces (per-etype sum or mean, cross-etype sum), s + msg / norm,
// message GVPLayerNorm, the update GVP chain (chain_generic_gvp), residual, update GVPLayerNorm.  A node's S scalars
// live on its four lanes (64 registers each), so both layer norms are in-lane sums plus two cross-lane adds; the
// residual scalars wait in the s_tmp scratch rows (the registers are needed for the GEMM operands), the 16 residual
// vectors stay in registers.
// GVPDropout scale (0 or 1 / (1 - rate)) of the four consecutive elements i0 .. i0 + 3 (i0 a multiple of 4) of one Philox stream: they
// share one counter (gvp_train_core.h, dropout_scale -- the same bits)
__device__ __forceinline__ v4f dropout_scale4(unsigned long long seed, unsigned stream, long long i0, float rate) {
    unsigned c[4] = {(unsigned)(i0 >> 2), (unsigned)((unsigned long long)i0 >> 34), stream, 0x6b70646fu};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    const unsigned thr = (unsigned)fminf(rate * 4294967296.0f, 4294967040.0f);
    const float keep = 1.0f / (1.0f - rate);
    return v4f{c[0] >= thr ? keep : 0.0f, c[1] >= thr ? keep : 0.0f, c[2] >= thr ? keep : 0.0f, c[3] >= thr ? keep : 0.0f};
}
__device__ __forceinline__ float dropout_scale1(unsigned long long seed, unsigned stream, long long i, float rate) {
    unsigned c[4] = {(unsigned)(i >> 2), (unsigned)((unsigned long long)i >> 34), stream, 0x6b70646fu};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    const unsigned thr = (unsigned)fminf(rate * 4294967296.0f, 4294967040.0f);
    return c[i & 3] >= thr ? 1.0f / (1.0f - rate) : 0.0f;
}
// scalars x (row v of an [n][16 NTS] array) and vectors V ([n][live] masks shared by the three components; channels past `live` are padding)
template <int NTS>
__device__ __forceinline__ void lanes_dropout(v4f (&x)[NTS], v4f (&V)[3], const GvpNodeTrain &t, int pos, int v, int q) {
    if (t.rate <= 0.0f) return;
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) x[nt] = x[nt] * dropout_scale4(t.seed, t.stream[2 * pos], (long long)v * (16 * NTS) + 16 * nt + 4 * q, t.rate);
    v4f m;
    if (t.live_v == 16) m = dropout_scale4(t.seed, t.stream[2 * pos + 1], (long long)v * 16 + 4 * q, t.rate);
    else {
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = 4 * q + r < t.live_v ? dropout_scale1(t.seed, t.stream[2 * pos + 1], (long long)v * t.live_v + 4 * q + r, t.rate) : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) V[c] = V[c] * m;
}

// TR = 1: the training form (GvpNodeTrain, gvp_kernels.h)
template <int NTS, int HM = 0, int TR = 0>
__global__ __launch_bounds__(256, 2) void k_gvp_node_chain(GvpNodePair p) {
    constexpr int S = 16 * NTS, CH4 = NTS * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = HM ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6, lane = tid & 63;      // (scalar in the f16x2 form: 2 spilled dwords otherwise; the exact form is left as measured)
    const int which = (int)blockIdx.x >= p.tiles0 ? 1 : 0;
    const GvpNodeArgs &a = p.nt[which];
    const int node0 = ((int)blockIdx.x - (which ? p.tiles0 : 0)) * TM;
    const int n_gvps = a.n_gvps;

    auto chunk_src = [&](int c) -> const v4f * {
        const int stage = c / (NTS + 2), local = c - stage * (NTS + 2);
        return reinterpret_cast<const v4f *>(HM ? a.g[stage].chain_h : a.g[stage].chain) + (size_t)local * CH4;      // (wave-uniform)
    };
    ChunkRing<CH4> ring;
    ring.init(smem, n_gvps * (NTS + 2), wave, tid);
    ring.start(chunk_src);

    const int el = lane & 15, q = lane >> 4;
    const int vr = node0 + 16 * wave + el;
    const bool valid = vr < a.n;
    const int v = valid ? vr : a.n - 1;            // rows past the end repeat the last node and are not stored
    float inv_norm = 1.0f / a.norm_const;
    if (a.z) inv_norm = 1.0f / a.z[a.bidx[v]];

    v4f x[NTS], acc[NTS], Vc[3], Vm[3];
    if constexpr (TR) {   // s + dropout(msg / norm), v + dropout(msg_v / norm); the sums are kept
        const GvpNodeTrain &t = a.tr;
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) x[nt] = zero4();
#pragma unroll
        for (int c = 0; c < 3; ++c) Vc[c] = zero4();
        for (int i = 0; i < a.n_in; ++i) {
            const int lo = a.rowptr[i][v], hi = a.rowptr[i][v + 1];
            if (hi > lo) {
                const float w = (a.mean ? 1.0f / (float)(hi - lo) : 1.0f) * inv_norm;
                const float *mp = a.ms_main[i] + (size_t)v * S + 4 * q;
                v4f m[NTS], mv[3];
#pragma unroll
                for (int nt = 0; nt < NTS; ++nt) m[nt] = *reinterpret_cast<const v4f *>(mp + 16 * nt);
#pragma unroll
                for (int c = 0; c < 3; ++c) mv[c] = *reinterpret_cast<const v4f *>(a.mv_main[i] + (size_t)v * 48 + 16 * c + 4 * q);
                for (int tl = lo / TM + 1; tl <= (hi - 1) / TM; ++tl) {       // pieces continued into later tiles
                    const float *cp = a.ms_cont[i] + (size_t)tl * S + 4 * q;
#pragma unroll
                    for (int nt = 0; nt < NTS; ++nt) m[nt] += *reinterpret_cast<const v4f *>(cp + 16 * nt);
#pragma unroll
                    for (int c = 0; c < 3; ++c) mv[c] += *reinterpret_cast<const v4f *>(a.mv_cont[i] + (size_t)tl * 48 + 16 * c + 4 * q);
                }
#pragma unroll
                for (int nt = 0; nt < NTS; ++nt) x[nt] += m[nt] * w;
#pragma unroll
                for (int c = 0; c < 3; ++c) Vc[c] += mv[c] * w;
            }
        }
        lanes_dropout<NTS>(x, Vc, t, 0, v, q);
        const float *sp = t.s_in + (size_t)v * S + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) x[nt] += *reinterpret_cast<const v4f *>(sp + 16 * nt);
#pragma unroll
        for (int c = 0; c < 3; ++c) Vc[c] += *reinterpret_cast<const v4f *>(t.v_in + (size_t)v * 48 + 16 * c + 4 * q);
        if (valid) {
            float *so = t.sa + (size_t)v * S + 4 * q;
#pragma unroll
            for (int nt = 0; nt < NTS; ++nt) *reinterpret_cast<v4f *>(so + 16 * nt) = x[nt];
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<v4f *>(t.va + (size_t)v * 48 + 16 * c + 4 * q) = Vc[c];
        }
    } else {   // s + msg / norm, v + msg_v / norm
        const float *sp = a.s + (size_t)v * S + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) x[nt] = *reinterpret_cast<const v4f *>(sp + 16 * nt);
        load_vec12(a.v + (size_t)v * 48 + 12 * q, Vc);
        for (int i = 0; i < a.n_in; ++i) {
            const int lo = a.rowptr[i][v], hi = a.rowptr[i][v + 1];
            if (hi > lo) {
                const float w = (a.mean ? 1.0f / (float)(hi - lo) : 1.0f) * inv_norm;
                const float *mp = a.ms_main[i] + (size_t)v * S + 4 * q;
                v4f m[NTS], mv[3];
#pragma unroll
                for (int nt = 0; nt < NTS; ++nt) m[nt] = *reinterpret_cast<const v4f *>(mp + 16 * nt);
                load_vec12(a.mv_main[i] + (size_t)v * 48 + 12 * q, mv);
                for (int t = lo / TM + 1; t <= (hi - 1) / TM; ++t) {       // pieces continued into later tiles
                    const float *cp = a.ms_cont[i] + (size_t)t * S + 4 * q;
#pragma unroll
                    for (int nt = 0; nt < NTS; ++nt) m[nt] += *reinterpret_cast<const v4f *>(cp + 16 * nt);
                    v4f cv[3];
                    load_vec12(a.mv_cont[i] + (size_t)t * 48 + 12 * q, cv);
#pragma unroll
                    for (int c = 0; c < 3; ++c) mv[c] += cv[c];
                }
#pragma unroll
                for (int nt = 0; nt < NTS; ++nt) x[nt] += m[nt] * w;
#pragma unroll
                for (int c = 0; c < 3; ++c) Vc[c] += mv[c] * w;
            }
        }
    }
    // message layer norm (gvp.py:519-521); its output is also the residual of the update block
    lanes_layernorm<NTS>(x, a.ln1_w, a.ln1_b, q, a.ln_inv_n, a.ln_pad);
    lanes_vecnorm(Vc, a.vn_inv_n, a.vn_pad);
    float *tmp = (TR ? a.tr.s1 : a.s_tmp) + (size_t)v * S + 4 * q;
    if (valid) {
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) *reinterpret_cast<v4f *>(tmp + 16 * nt) = x[nt];
        if constexpr (TR) {
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<v4f *>(a.tr.v1 + (size_t)v * 48 + 16 * c + 4 * q) = Vc[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) Vm[c] = Vc[c];
    {
        const float *b0 = a.g[0].b + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) acc[nt] = *reinterpret_cast<const v4f *>(b0 + 16 * nt);
    }
    ring.first();
#pragma unroll 1
    for (int k = 0; k < n_gvps; ++k) {
        if constexpr (HM) chain_generic_gvp_h<NTS>(ring, chunk_src, a.g[k], k + 1 < n_gvps ? a.g[k + 1].b : nullptr, x, acc, Vc, lane, q);
        else {
            const v4f *cb = reinterpret_cast<const v4f *>(a.g[k].chain);
            const v4f *nb = k + 1 < n_gvps ? reinterpret_cast<const v4f *>(a.g[k + 1].chain) : cb + (size_t)NTS * CH4;
            if constexpr (TR) chain_generic_gvp<NTS, decltype(ring), 1>(ring, cb, nb, a.g[k], k + 1 < n_gvps ? a.g[k + 1].b : nullptr, x, acc, Vc, lane, q,
                                                                         &a.tr.g[k], (size_t)v, valid);
            else chain_generic_gvp<NTS>(ring, cb, nb, a.g[k], k + 1 < n_gvps ? a.g[k + 1].b : nullptr, x, acc, Vc, lane, q);
        }
    }
    // residual + update layer norm (gvp.py:524-532)
    if constexpr (TR) lanes_dropout<NTS>(x, Vc, a.tr, 1, v, q);
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) x[nt] += *reinterpret_cast<const v4f *>(tmp + 16 * nt);
#pragma unroll
    for (int c = 0; c < 3; ++c) Vc[c] += Vm[c];
    if constexpr (TR) {
        if (valid) {
            float *so = a.tr.sb + (size_t)v * S + 4 * q;
#pragma unroll
            for (int nt = 0; nt < NTS; ++nt) *reinterpret_cast<v4f *>(so + 16 * nt) = x[nt];
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<v4f *>(a.tr.vb + (size_t)v * 48 + 16 * c + 4 * q) = Vc[c];
        }
    }
    lanes_layernorm<NTS>(x, a.ln2_w, a.ln2_b, q, a.ln_inv_n, a.ln_pad);
    lanes_vecnorm(Vc, a.vn_inv_n, a.vn_pad);
    if constexpr (TR) {
        if (valid) {
            float *so = a.tr.s_out + (size_t)v * S + 4 * q;
#pragma unroll
            for (int nt = 0; nt < NTS; ++nt) *reinterpret_cast<v4f *>(so + 16 * nt) = x[nt];
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<v4f *>(a.tr.v_out + (size_t)v * 48 + 16 * c + 4 * q) = Vc[c];
        }
    } else if (valid) {
        float *so = a.s + (size_t)v * S + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) *reinterpret_cast<v4f *>(so + 16 * nt) = x[nt];
        float f[12];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) f[3 * r + c] = Vc[c][r];
        v4f *vo = reinterpret_cast<v4f *>(a.v + (size_t)v * 48 + 12 * q);
        vo[0] = v4f{f[0], f[1], f[2], f[3]};
        vo[1] = v4f{f[4], f[5], f[6], f[7]};
        vo[2] = v4f{f[8], f[9], f[10], f[11]};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's tail fetches must not outlive the workgroup's LDS
}

// ---- per-node blocks of the first message Linear (h_src / h_dst part of to_feats_out), register-chained -------------
// P[slot][node][:] = W_block s[node] (+ b): one 64-node tile of one slot per workgroup, the S x S block streamed as
// two-slab chunks (gvp_host.hip packs wproj / wproj_dst in chunk order).
template <int NTS, int HM = 0>
__global__ __launch_bounds__(256, 2) void k_gvp_proj_chain(GvpProjArgs a) {
    constexpr int S = 16 * NTS, CH4 = NTS * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int sl = 0;
#pragma unroll
    for (int e = 1; e < GVP_PROJ_SLOTS; ++e)
        if (e < a.n_slots && (int)blockIdx.x >= a.tiles_first[e]) sl = e;
    const int node0 = ((int)blockIdx.x - a.tiles_first[sl]) * TM;
    const int n = a.n[sl];

    const v4f *stream = reinterpret_cast<const v4f *>(HM ? a.wp_h[sl] : a.wp[sl]) + tid;
    auto chunk_src = [&](int c) -> const v4f * { return stream + (size_t)c * (2 * CH4); };
    ChunkRing2<2 * CH4> ring;
    ring.init(smem, NTS / 2, wave);
    ring.start(chunk_src);

    const int el = lane & 15, q = lane >> 4;
    const int vr = node0 + 16 * wave + el;
    const int v = min(vr, n - 1);
    const float *sp = a.s[sl] + (size_t)v * S + 4 * q;
    v4f x[NTS], acc[NTS];
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) x[nt] = *reinterpret_cast<const v4f *>(sp + 16 * nt);
    const float *bias = a.b[sl];
#pragma unroll
    for (int mt = 0; mt < NTS; ++mt) acc[mt] = bias ? *reinterpret_cast<const v4f *>(bias + 16 * mt + 4 * q) : zero4();
    ring.first();
    if constexpr (HM) {       // f16x2 mode: a 32-KB chunk = one 32-wide k-block for all 16 output tiles (units 2 kb, 2 kb + 1)
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) acc[mt] = acc[mt] * (1.0f / H_UNSCALE);
#pragma unroll
        for (int kb = 0; kb < NTS / 2; ++kb) {
            f32x4 xh, xl;
            split8(x[2 * kb], x[2 * kb + 1], xh, xl);
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(ring.acquire(chunk_src)) + lane;
            f32x4 w[3][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[0][i] = wp[i * 64];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) w[(b + 1) % 3][i] = wp[(8 * (b + 1) + i) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[4 * b + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(w[b % 3][2 * m + 1]), as_h8(xh), acc[4 * b + m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[4 * b + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(w[b % 3][2 * m]), as_h8(xl), acc[4 * b + m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[4 * b + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(w[b % 3][2 * m]), as_h8(xh), acc[4 * b + m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            ring.release();
        }
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) acc[mt] = acc[mt] * H_UNSCALE;
    } else {
#pragma unroll
        for (int nt = 0; nt < NTS; nt += 2) {
            const v4f *buf = ring.acquire(chunk_src);
            chunk_gemm2<NTS>(buf, x[nt], x[nt + 1], acc, lane);
            ring.release();
        }
    }
    if (vr < n) {
        float *out = a.P[sl] + (size_t)v * S + 4 * q;
#pragma unroll
        for (int mt = 0; mt < NTS; ++mt) *reinterpret_cast<v4f *>(out + 16 * mt) = acc[mt];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's tail fetch must not outlive the workgroup's LDS
}

// ---- noise prediction block (dynamics_gvp.py:10-44), register-chained --------------------------------------------
// n_gvps - 1 GVPs of the generic kind, then the head GVP (S scalars, 16 vectors) -> (64 scalars, 1 vector, identity vector
// activation), eps_h = Linear(64, F) of its scalars, eps_x = its vector.  16 ligand atoms per wave; the head GVP's weights
// (NTS + 1 k-slabs of 4 output tiles, 4 gate tiles) are read straight from global memory: ligand atoms are few.
template <int NTS, int HM = 0>
__global__ __launch_bounds__(256, 2) void k_gvp_noise_chain(GvpNoiseArgs a) {
    constexpr int S = 16 * NTS, CH4 = NTS * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int node0 = blockIdx.x * TM;
    const int n_gen = a.n_gvps - 1;

    auto chunk_src = [&](int c) -> const v4f * {
        const int stage = c / (NTS + 2), local = c - stage * (NTS + 2);
        return reinterpret_cast<const v4f *>(HM ? a.g[stage].chain_h : a.g[stage].chain) + (size_t)local * CH4;      // (wave-uniform)
    };
    ChunkRing<CH4> ring;
    ring.init(smem, std::max(n_gen, 1) * (NTS + 2), wave, tid);
    if (n_gen > 0) ring.start(chunk_src);

    const int el = lane & 15, q = lane >> 4;
    const int vr = node0 + 16 * wave + el;
    const bool valid = vr < a.n;
    const int v = valid ? vr : a.n - 1;
    v4f x[NTS], acc[NTS], Vc[3];
    const float *sp = a.s + (size_t)v * S + 4 * q;
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) x[nt] = *reinterpret_cast<const v4f *>(sp + 16 * nt);
    load_vec12(a.v + (size_t)v * 48 + 12 * q, Vc);
    if (n_gen > 0) {
        const float *b0 = a.g[0].b + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) acc[nt] = *reinterpret_cast<const v4f *>(b0 + 16 * nt);
        ring.first();
#pragma unroll 1
        for (int k = 0; k < n_gen; ++k) {
            if constexpr (HM) chain_generic_gvp_h<NTS>(ring, chunk_src, a.g[k], k + 1 < n_gen ? a.g[k + 1].b : nullptr, x, acc, Vc, lane, q);
            else {
                const v4f *cb = reinterpret_cast<const v4f *>(a.g[k].chain);
                const v4f *nb = k + 1 < n_gen ? reinterpret_cast<const v4f *>(a.g[k + 1].chain) : cb + (size_t)NTS * CH4;
                chain_generic_gvp<NTS>(ring, cb, nb, a.g[k], k + 1 < n_gen ? a.g[k + 1].b : nullptr, x, acc, Vc, lane, q);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's tail fetches must not outlive the workgroup's LDS
    }

    // head GVP: vec1, [x | sh] -> 64 scalars (4 output tiles), gate and output vector (row 0 of their tiles)
    const GvpW &gl = a.g[n_gen];
    const v4f wh = reinterpret_cast<const v4f *>(gl.whp)[lane];
    v4f Vh[3], sh;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v4f t = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) t = mfma16(wh[r], Vc[c][r], t);
        Vh[c] = t;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[r] = sqrt1(fmaxf(Vh[0][r] * Vh[0][r] + Vh[1][r] * Vh[1][r] + Vh[2][r] * Vh[2][r], 1e-8f));
    const v4f *wl = reinterpret_cast<const v4f *>(gl.chain) + lane;            // [slab][4 tiles][64 lanes]
    v4f so[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) so[mt] = *reinterpret_cast<const v4f *>(gl.b + 16 * mt + 4 * q);
#pragma unroll 4
    for (int slab = 0; slab <= NTS; ++slab) {
        v4f xin = sh;
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt)
            if (slab == nt) xin = x[nt];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const v4f w = wl[(slab * 4 + mt) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) so[mt] = mfma16(w[r], xin[r], so[mt]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) so[mt][r] = silu(so[mt][r]);
    const v4f *wgl = wl + (size_t)(NTS + 1) * 4 * 64;                            // gate slab: 4 k-tiles of one output tile
    v4f ga = zero4();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const v4f w = wgl[nt * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) ga = mfma16(w[r], so[nt][r], ga);
    }
    const v4f wu = reinterpret_cast<const v4f *>(gl.wup)[lane];
    v4f vu[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v4f t = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) t = mfma16(wu[r], Vh[c][r], t);
        vu[c] = t;
    }
    float gate = ga[0] + gl.bg[0];                                            // output vector 0 = row 0: lanes q == 0, r == 0
    if (gl.vec_sigmoid) gate = sigmoidf_(gate);
    if (valid && q == 0) {
        a.eps_x[(size_t)v * 3] = gate * vu[0][0];
        a.eps_x[(size_t)v * 3 + 1] = gate * vu[1][0];
        a.eps_x[(size_t)v * 3 + 2] = gate * vu[2][0];
    }
    // eps_h = W_out s + b_out: this lane holds s[16 mt + 4 q + r]
    for (int f = 0; f < a.F; ++f) {
        const float *wo = a.Wout + (size_t)f * 64 + 4 * q;
        float part = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const v4f w = *reinterpret_cast<const v4f *>(wo + 16 * mt);
            part += so[mt][0] * w[0] + so[mt][1] * w[1] + so[mt][2] * w[2] + so[mt][3] * w[3];
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (valid && q == 0) a.eps_h[(size_t)v * a.F + f] = part + a.bout[f];
    }
}

kpd_status launch_gvp_edge(const GvpEdgeArgs &a, int tile_cap, hipStream_t st) {
    if (tile_cap == 0) return KPD_OK;
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    for (int et = 0; et < 4; ++et)
        if (a.src[et])
            for (int k = 0; k < a.n_gvps; ++k)
                KPD_REQUIRE(a.g[et][k].chain && a.g[et][k].whp && a.g[et][k].wup, KPD_ERR_STATE,
                            "message GVP %d of edge type %d was not prepared for the chained edge kernel", k, et);
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_chain<16>), ChainSmem<16>::FLOATS * 4));
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_chain<8>), ChainSmem<8>::FLOATS * 4));
    KPD_REQUIRE(a.S == 256 || a.S == 128, KPD_ERR_INVALID, "gvp chain kernel: S=%d (supported 128, 256)", a.S);
    KPD_REQUIRE(!a.train || (a.S == 256 && a.gemm_mode == 0), KPD_ERR_INVALID, "gvp chain kernel: the training form runs at S = 256 in the exact fp32 mode");
    const dim3 grid(8 * cdiv(tile_cap, 8));
    if (a.S == 256 && a.gemm_mode == 1) {
        for (int et = 0; et < 4; ++et)
            if (a.src[et])
                for (int k = 0; k < a.n_gvps; ++k)
                    KPD_REQUIRE(a.g[et][k].chain_h, KPD_ERR_STATE, "message GVP %d of edge type %d has no f16x2 chunks", k, et);
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_chain<16, 1>), ChainSmem<16>::FLOATS * 4));
        hipLaunchKernelGGL((k_gvp_chain<16, 1>), grid, dim3(256), ChainSmem<16>::FLOATS * 4, st, a);
    } else if (a.S == 256 && a.train) {
        KPD_REQUIRE(!a.use_dst, KPD_ERR_INVALID, "gvp chain kernel: the training form has no destination-feature inputs");
        for (int et = 0; et < 4; ++et)
            KPD_REQUIRE(!a.src[et] || (a.g[et][0].h == 17 && a.n_gvps <= 4), KPD_ERR_INVALID, "gvp chain kernel: the training form wants a 17-channel head GVP");
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_chain<16, 0, 1>), ChainSmem<16>::FLOATS * 4));
        GvpEdgeArgs b = a;
        b.train_skip = tool_env_int("KPD_TR_SKIP", 0);          // (TOOLS build: timing experiments; a compile-time 0 in the product)
        hipLaunchKernelGGL((k_gvp_chain<16, 0, 1>), grid, dim3(256), ChainSmem<16>::FLOATS * 4, st, b);
    } else if (a.S == 256)
        hipLaunchKernelGGL(k_gvp_chain<16>, grid, dim3(256), ChainSmem<16>::FLOATS * 4, st, a);
    else
        hipLaunchKernelGGL(k_gvp_chain<8>, grid, dim3(256), ChainSmem<8>::FLOATS * 4, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_gvp_edge_bwd(const GvpEdgeBwdArgs &a, int tile_cap, hipStream_t st) {
    if (tile_cap == 0) return KPD_OK;
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    KPD_REQUIRE(a.n_gvps >= 1 && a.n_gvps <= GVP_MAX_CHAIN && a.fwd && a.out && a.meta, KPD_ERR_INVALID, "gvp chain backward: bad arguments");
    for (int et = 0; et < 4; ++et)
        if (a.dst[et])
            for (int k = 0; k < a.n_gvps; ++k)
                KPD_REQUIRE(a.g[et][k].chain && (k == 0 || (a.g[et][k].wut && a.g[et][k].wht)), KPD_ERR_STATE,
                            "message GVP %d of edge type %d was not packed for the chained backward kernel", k, et);
    const int lds = KPD_CHAIN_NBUF * 16 * 64 * 16;
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_chain_bwd<16>), lds));
    hipLaunchKernelGGL(k_gvp_chain_bwd<16>, dim3(8 * cdiv(tile_cap, 8)), dim3(256), lds, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_gvp_node_bwd(const GvpNodeBwdArgs &a, hipStream_t st) {
    if (a.n == 0) return KPD_OK;
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    KPD_REQUIRE(a.n_gvps >= 1 && a.n_gvps <= GVP_MAX_CHAIN && a.ds && a.dV && a.ds_in && a.dv_in, KPD_ERR_INVALID, "gvp node chain backward: bad arguments");
    for (int k = 0; k < a.n_gvps; ++k)
        KPD_REQUIRE(a.g[k].chain && a.g[k].wut && a.g[k].wht && a.f[k].pre && a.o[k].dpre, KPD_ERR_STATE, "update GVP %d was not prepared for the chained backward kernel", k);
    const int lds = KPD_CHAIN_NBUF * 16 * 64 * 16;
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_node_chain_bwd<16>), lds));
    hipLaunchKernelGGL(k_gvp_node_chain_bwd<16>, dim3(cdiv(a.n, TM)), dim3(256), lds, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_gvp_proj(const GvpProjArgs &a, hipStream_t st) {
    if (a.n_slots == 0 || a.tiles_first[a.n_slots] == 0) return KPD_OK;
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    KPD_REQUIRE(a.S == 256 || a.S == 128, KPD_ERR_INVALID, "gvp projection kernel: S=%d (supported 128, 256)", a.S);
    if (!(a.S == 256 && a.gemm_mode == 1)) {
        long rows = 0;
        for (int e = 0; e < a.n_slots; ++e) rows += a.n[e];
        if (rows <= coop_rows_max(a.coop_rows)) return launch_gvp_proj_coop(a, st);
    }
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_proj_chain<16>), 4 * 16 * 64 * 16));
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_proj_chain<8>), 4 * 8 * 64 * 16));
    const dim3 grid(a.tiles_first[a.n_slots]);
    if (a.S == 256 && a.gemm_mode == 1) {
        for (int e = 0; e < a.n_slots; ++e) KPD_REQUIRE(a.wp_h[e], KPD_ERR_STATE, "projection slot %d has no f16x2 block", e);
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_proj_chain<16, 1>), 4 * 16 * 64 * 16));
        hipLaunchKernelGGL((k_gvp_proj_chain<16, 1>), grid, dim3(256), 4 * 16 * 64 * 16, st, a);
    } else if (a.S == 256) hipLaunchKernelGGL(k_gvp_proj_chain<16>, grid, dim3(256), 4 * 16 * 64 * 16, st, a);
    else hipLaunchKernelGGL(k_gvp_proj_chain<8>, grid, dim3(256), 4 * 8 * 64 * 16, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_gvp_noise(const GvpNoiseArgs &a, hipStream_t st) {
    if (a.n == 0) return KPD_OK;
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    KPD_REQUIRE(a.S == 256 || a.S == 128, KPD_ERR_INVALID, "gvp noise head: S=%d (supported 128, 256)", a.S);
    KPD_REQUIRE(a.n_gvps >= 1 && a.F >= 1, KPD_ERR_INVALID, "gvp noise head: n_gvps=%d F=%d", a.n_gvps, a.F);
    const GvpW &gl = a.g[a.n_gvps - 1];
    KPD_REQUIRE(gl.sout == 64 && gl.vout == 1 && gl.vin == GV, KPD_ERR_STATE, "noise head GVP must map (S, 16) -> (64, 1)");
    for (int k = 0; k < a.n_gvps; ++k)
        KPD_REQUIRE(a.g[k].chain && a.g[k].whp && a.g[k].wup, KPD_ERR_STATE, "noise GVP %d was not prepared for the chained kernel", k);
    if (!(a.S == 256 && a.gemm_mode == 1 && a.n_gvps > 1) && a.n <= coop_rows_max(a.coop_rows)) return launch_gvp_noise_coop(a, st);
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_noise_chain<16>), 3 * 16 * 64 * 16));
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_noise_chain<8>), 3 * 8 * 64 * 16));
    if (a.S == 256 && a.gemm_mode == 1 && a.n_gvps > 1) {
        for (int k = 0; k + 1 < a.n_gvps; ++k) KPD_REQUIRE(a.g[k].chain_h, KPD_ERR_STATE, "noise GVP %d has no f16x2 chunks", k);
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_noise_chain<16, 1>), 3 * 16 * 64 * 16));
        hipLaunchKernelGGL((k_gvp_noise_chain<16, 1>), dim3(cdiv(a.n, TM)), dim3(256), 3 * 16 * 64 * 16, st, a);
    } else if (a.S == 256) hipLaunchKernelGGL(k_gvp_noise_chain<16>, dim3(cdiv(a.n, TM)), dim3(256), 3 * 16 * 64 * 16, st, a);
    else hipLaunchKernelGGL(k_gvp_noise_chain<8>, dim3(cdiv(a.n, TM)), dim3(256), 3 * 8 * 64 * 16, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_gvp_node(const GvpNodePair &pin, hipStream_t st) {
    // a caller that does not state the model's widths (the receptor encoder: always the kernels' own) gets the full-width norms
    GvpNodePair p = pin;
    for (int nt = 0; nt < 2; ++nt) {
        GvpNodeArgs &a = p.nt[nt];
        if (a.ln_inv_n == 0.0f) { a.ln_inv_n = 1.0f / (float)std::max(a.S, 1); a.ln_pad = 0.0f; }
        if (a.vn_inv_n == 0.0f) { a.vn_inv_n = 1.0f / (float)GV; a.vn_pad = 0.0f; }
        KPD_REQUIRE(a.ln_pad >= 0.0f && a.vn_pad >= 0.0f && a.ln_inv_n > 0.0f && a.vn_inv_n > 0.0f, KPD_ERR_INVALID, "gvp node kernel: bad norm widths");
    }
    const int tiles = p.tiles0 + cdiv(p.nt[1].n, TM);
    if (tiles == 0) return KPD_OK;
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    const int S = p.nt[0].n ? p.nt[0].S : p.nt[1].S;
    KPD_REQUIRE(S == 256 || S == 128, KPD_ERR_INVALID, "gvp node kernel: S=%d (supported 128, 256)", S);
    for (int nt = 0; nt < 2; ++nt)
        if (p.nt[nt].n)
            for (int k = 0; k < p.nt[nt].n_gvps; ++k)
                KPD_REQUIRE(p.nt[nt].g[k].chain && p.nt[nt].g[k].whp && p.nt[nt].g[k].wup, KPD_ERR_STATE,
                            "update GVP %d was not prepared for the chained node kernel", k);
    if (p.nt[0].train || p.nt[1].train) {
        KPD_REQUIRE(S == 256 && p.gemm_mode == 0 && (p.nt[0].n == 0 || p.nt[0].train) && (p.nt[1].n == 0 || p.nt[1].train), KPD_ERR_INVALID,
                    "gvp node kernel: the training form runs at S = 256 in the exact fp32 mode, for both node types of a launch");
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_node_chain<16, 0, 1>), 3 * 16 * 64 * 16));
        hipLaunchKernelGGL((k_gvp_node_chain<16, 0, 1>), dim3(tiles), dim3(256), 3 * 16 * 64 * 16, st, p);
        KPD_LAUNCH_CHECK();
        return KPD_OK;
    }
    if (!(S == 256 && p.gemm_mode == 1) && p.nt[0].n + p.nt[1].n <= coop_rows_max(p.coop_rows)) return launch_gvp_node_coop(p, st);
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_node_chain<16>), 3 * 16 * 64 * 16));
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_node_chain<8>), 3 * 8 * 64 * 16));
    if (S == 256 && p.gemm_mode == 1) {
        for (int nt = 0; nt < 2; ++nt)
            if (p.nt[nt].n)
                for (int k = 0; k < p.nt[nt].n_gvps; ++k)
                    KPD_REQUIRE(p.nt[nt].g[k].chain_h, KPD_ERR_STATE, "update GVP %d has no f16x2 chunks", k);
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_node_chain<16, 1>), 3 * 16 * 64 * 16));
        hipLaunchKernelGGL((k_gvp_node_chain<16, 1>), dim3(tiles), dim3(256), 3 * 16 * 64 * 16, st, p);
    } else if (S == 256) hipLaunchKernelGGL(k_gvp_node_chain<16>, dim3(tiles), dim3(256), 3 * 16 * 64 * 16, st, p);
    else hipLaunchKernelGGL(k_gvp_node_chain<8>, dim3(tiles), dim3(256), 3 * 8 * 64 * 16, st, p);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd
