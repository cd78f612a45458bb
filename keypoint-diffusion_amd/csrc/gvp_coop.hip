// Column-split ("cooperative") forms of the GVP node-side kernels: node update, noise head, per-node projections
// (models/gvp.py:499-536, models/dynamics_gvp.py:10-44, the h_src / h_dst block of the first message Linear of gvp.py:545-549).
//
// The register-chained kernels of gvp_chain.hip give one wave 16 rows and ALL S output columns of every product: 1 100 dependent
// MFMAs per GVP on one SIMD.  That is the right shape for the edge kernel (thousands of tiles, two waves per SIMD), but a node-side
// launch of a keypoint model has 65 tiles (gvp_40kp: 4 160 nodes) for 256 CUs: a quarter of the SIMDs run one latency-bound wave each
// for 62 us and the rest idle.  Here a workgroup owns 16 rows and its four waves split the OUTPUT COLUMNS of every [x | sh] product:
// wave w computes tiles w S / 64 .. of T^T = W X^T from the full activations (each wave holds all of X^T in B-operand registers), so
// a GVP is 272 + 40 MFMAs deep instead of 1 176, and there are four times as many workgroups.  Because no two waves of a workgroup
// read the same weight fragment, the weights do not go through LDS at all: every wave streams its own A fragments from global
// memory (L2) into registers, four k-slabs ahead.  LDS carries only what the waves owe each other:
//   * after the SiLU every wave publishes its S / 4 new scalars (16 rows x S floats per workgroup) and reads back all of them:
//     the B operand of the next product -- one barrier per GVP;
//   * the gates need all S new scalars as their K dimension: every wave multiplies its own S / 4 (registers, before the exchange)
//     and publishes a 16 x 16 partial; the four partials are added in the fixed order (w0 + w1) + (w2 + w3) behind the same barrier.
// Everything that is not the big product (vector channels, norms, layer norms, aggregation of the message pieces) is computed by
// every wave redundantly: it is small, and it keeps the activations replicated without further exchanges.
//
// Summation order differs from the chained kernels in the gate product only (four partials instead of one accumulator chain), so the
// two forms agree to rounding, not bit for bit; which one a launch takes is a function of the row count alone (gvp_kernels.h,
// coop_rows_max), so every result stays bitwise repeatable.  Exact fp32 mode only (the f16x2 mode keeps the chained kernels).
#include <algorithm>

#include "chain_core.h"
#include "gvp_kernels.h"

namespace kpd {

namespace {

constexpr int CR = 16;        // rows (nodes) per cooperative workgroup
constexpr int CD = 4;         // k-slabs of weight fragments in flight per wave

template <int NTS>
struct CoopGeom {
    static constexpr int S = 16 * NTS, TPW = NTS / 4, SX = S + 4;
    static constexpr int X_FLOATS = CR * SX;                       // one exchange buffer of scalars
    static constexpr int G_FLOATS = 4 * 64 * 4;                    // one exchange buffer of gate partials (4 waves x 64 lanes x 4)
    static constexpr int FLOATS = 2 * (X_FLOATS + G_FLOATS);       // both double-buffered: one barrier per exchange suffices
};

template <int NTS>
struct Coop {
    using G = CoopGeom<NTS>;
    float *xs, *gs;
    int xc;                   // exchanges so far (selects the buffer)
    int w, lane, el, q;

    __device__ __forceinline__ void init(float *smem, int wave, int lane_) {
        xs = smem;
        gs = smem + 2 * G::X_FLOATS;
        xc = 0;
        w = wave;
        lane = lane_;
        el = lane_ & 15;
        q = lane_ >> 4;
    }
    // publish this wave's TPW tiles (and, optionally, its gate partial), barrier, read all NTS tiles (and the summed gate partials).
    // Buffer reuse: exchange i + 2 writes the buffer exchange i read; its writer has passed barrier i + 1, which every wave reaches
    // only after its reads of exchange i.
    __device__ __forceinline__ void exchange(const v4f (&own)[G::TPW], v4f (&x)[NTS], const v4f *gpart = nullptr, v4f *gsum = nullptr) {
        float *xb = xs + (xc & 1) * G::X_FLOATS + el * G::SX + 4 * q;
        float *gb = gs + (xc & 1) * G::G_FLOATS;
#pragma unroll
        for (int j = 0; j < G::TPW; ++j) *reinterpret_cast<v4f *>(xb + 16 * (w * G::TPW + j)) = own[j];
        if (gpart) *reinterpret_cast<v4f *>(gb + (w * 64 + lane) * 4) = *gpart;
        lds_barrier();
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) x[nt] = *reinterpret_cast<const v4f *>(xb + 16 * nt);
        if (gsum) {
            const v4f g0 = *reinterpret_cast<const v4f *>(gb + (0 * 64 + lane) * 4), g1 = *reinterpret_cast<const v4f *>(gb + (1 * 64 + lane) * 4);
            const v4f g2 = *reinterpret_cast<const v4f *>(gb + (2 * 64 + lane) * 4), g3 = *reinterpret_cast<const v4f *>(gb + (3 * 64 + lane) * 4);
            *gsum = (g0 + g1) + (g2 + g3);
        }
        ++xc;
    }
};

// acc[j] += sum over k-slabs s < NSLAB of W[tile w TPW + j][slab s] . xin(s).  wbase = fragment (tile w TPW, slab 0) of this lane in a
// [slab][tile][lane] stream (pack_chain_frag order); the fragments of slab s + CD are requested as soon as those of slab s are consumed.
template <int NTS, int NSLAB, class XF>
__device__ __forceinline__ void coop_slab_gemm(const v4f *__restrict__ wbase, XF &&xin_of, v4f (&acc)[NTS / 4]) {
    constexpr int TPW = NTS / 4;
    v4f wq[CD][TPW];
#pragma unroll
    for (int d = 0; d < CD; ++d)
        if (d < NSLAB) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) wq[d][j] = wbase[(d * NTS + j) * 64];
        }
#pragma unroll
    for (int s = 0; s < NSLAB; ++s) {
        const v4f xin = xin_of(s);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = mfma16(wq[s % CD][j][r], xin[r], acc[j]);
        if (s + CD < NSLAB) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) wq[s % CD][j] = wbase[((s + CD) * NTS + j) * 64];
        }
    }
}

// One generic GVP (scalars x and vectors Vc replicated in every wave's registers), cooperative form of chain_generic_gvp:
// acc enters holding this wave's tiles of the GVP's bias and leaves holding those of `next_bias` (when given); x leaves holding ALL
// new scalars (+ `res`, this wave's tiles of a residual, with add_res: the node update's second residual rides in the exchange).
template <int NTS>
__device__ __forceinline__ void coop_generic_gvp(Coop<NTS> &C, const GvpW &gk, const float *next_bias, bool add_res, const v4f (&res)[NTS / 4],
                                                 v4f (&x)[NTS], v4f (&acc)[NTS / 4], v4f (&Vc)[3]) {
    constexpr int TPW = NTS / 4;
    const int lane = C.lane, q = C.q, w = C.w;
    const v4f *chain = reinterpret_cast<const v4f *>(gk.chain) + lane;
    const v4f *wbase = chain + (size_t)(w * TPW) * 64;
    // gate fragments of this wave's k-tiles (chunk NTS + 1 of the chain: NTS k-tiles of the one 16-gate output tile)
    v4f wg[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) wg[j] = chain[((size_t)(NTS + 1) * NTS + w * TPW + j) * 64];
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
    coop_slab_gemm<NTS, NTS + 1>(wbase, [&](int s) -> v4f { return s < NTS ? x[s < NTS ? s : 0] : sh; }, acc);
    const v4f bgv = *reinterpret_cast<const v4f *>(gk.bg + 4 * q);
    const v4f wu = reinterpret_cast<const v4f *>(gk.wup)[lane];
    v4f own[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) own[j] = silu4(acc[j]);
    if (next_bias) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = *reinterpret_cast<const v4f *>(next_bias + 16 * (w * TPW + j) + 4 * q);
    }
    // this wave's share of the gate product (its own S / 4 inputs)
    v4f ga[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[r] = mfma16(wg[j][r], own[j][r], ga[r]);
    const v4f gpart = (ga[0] + ga[1]) + (ga[2] + ga[3]);
    if (add_res) {                    // (a flag beside the array: a conditional POINTER to it would put the array into scratch)
#pragma unroll
        for (int j = 0; j < TPW; ++j) own[j] += res[j];
    }
    v4f gate;
    C.exchange(own, x, &gpart, &gate);
    gate += bgv;
    if (gk.vec_sigmoid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) gate[r] = sigmoidf_(gate[r]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v4f t = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) t = mfma16(wu[r], Vh[c][r], t);
        Vc[c] = gate * t;
    }
}

// ---- node update (gvp.py:499-536): aggregate, + residual, message GVPLayerNorm, update GVPs, + residual, update GVPLayerNorm ----------
template <int NTS>
__global__ __launch_bounds__(256, 2) void k_gvp_node_coop(GvpNodePair p) {
    using G = CoopGeom<NTS>;
    constexpr int S = G::S, TPW = G::TPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int which = (int)blockIdx.x >= p.tiles0 ? 1 : 0;            // (tiles0 counts 16-row workgroups in this launch form)
    const GvpNodeArgs &a = p.nt[which];
    const int node0 = ((int)blockIdx.x - (which ? p.tiles0 : 0)) * CR;
    const int n_gvps = a.n_gvps;
    Coop<NTS> C;
    C.init(smem, wave, lane);
    const int el = C.el, q = C.q;
    const int vr = node0 + el;
    const bool valid = vr < a.n;
    const int v = valid ? vr : a.n - 1;            // rows past the end repeat the last node and are not stored
    float inv_norm = 1.0f / a.norm_const;
    if (a.z) inv_norm = 1.0f / a.z[a.bidx[v]];

    v4f x[NTS], own[TPW], Vc[3], Vm[3];
    {   // s + msg / norm for this wave's tiles; v + msg_v / norm in every wave
        const float *sp = a.s + (size_t)v * S + 16 * (wave * TPW) + 4 * q;
#pragma unroll
        for (int j = 0; j < TPW; ++j) own[j] = *reinterpret_cast<const v4f *>(sp + 16 * j);
        load_vec12(a.v + (size_t)v * 48 + 12 * q, Vc);
        for (int i = 0; i < a.n_in; ++i) {
            const int lo = a.rowptr[i][v], hi = a.rowptr[i][v + 1];
            if (hi > lo) {
                const float wgt = (a.mean ? 1.0f / (float)(hi - lo) : 1.0f) * inv_norm;
                const float *mp = a.ms_main[i] + (size_t)v * S + 16 * (wave * TPW) + 4 * q;
                v4f m[TPW], mv[3];
#pragma unroll
                for (int j = 0; j < TPW; ++j) m[j] = *reinterpret_cast<const v4f *>(mp + 16 * j);
                load_vec12(a.mv_main[i] + (size_t)v * 48 + 12 * q, mv);
                for (int t = lo / TM + 1; t <= (hi - 1) / TM; ++t) {       // pieces continued into later tiles
                    const float *cp = a.ms_cont[i] + (size_t)t * S + 16 * (wave * TPW) + 4 * q;
#pragma unroll
                    for (int j = 0; j < TPW; ++j) m[j] += *reinterpret_cast<const v4f *>(cp + 16 * j);
                    v4f cv[3];
                    load_vec12(a.mv_cont[i] + (size_t)t * 48 + 12 * q, cv);
#pragma unroll
                    for (int c = 0; c < 3; ++c) mv[c] += cv[c];
                }
#pragma unroll
                for (int j = 0; j < TPW; ++j) own[j] += m[j] * wgt;
#pragma unroll
                for (int c = 0; c < 3; ++c) Vc[c] += mv[c] * wgt;
            }
        }
    }
    C.exchange(own, x);
    // message layer norm (gvp.py:519-521), every wave on the full row; its output is also the residual of the update block, of which a
    // wave keeps its own tiles (normalised from the registers it published: no indexed access into x)
    v4f res[TPW], acc[TPW];
    {
        float mean, rstd;
        lanes_ln_stats<NTS>(x, a.ln_inv_n, a.ln_pad, mean, rstd);
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) x[nt] = lanes_ln_tile(x[nt], a.ln1_w, a.ln1_b, nt, q, mean, rstd);
#pragma unroll
        for (int j = 0; j < TPW; ++j) res[j] = lanes_ln_tile(own[j], a.ln1_w, a.ln1_b, wave * TPW + j, q, mean, rstd);
    }
    lanes_vecnorm(Vc, a.vn_inv_n, a.vn_pad);
#pragma unroll
    for (int c = 0; c < 3; ++c) Vm[c] = Vc[c];
    {
        const float *b0 = a.g[0].b + 16 * (wave * TPW) + 4 * q;
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = *reinterpret_cast<const v4f *>(b0 + 16 * j);
    }
#pragma unroll 1
    for (int k = 0; k < n_gvps; ++k)
        coop_generic_gvp<NTS>(C, a.g[k], k + 1 < n_gvps ? a.g[k + 1].b : nullptr, k + 1 == n_gvps, res, x, acc, Vc);
    // (the scalar residual came in with the last exchange) update layer norm (gvp.py:524-532)
#pragma unroll
    for (int c = 0; c < 3; ++c) Vc[c] += Vm[c];
    lanes_layernorm<NTS>(x, a.ln2_w, a.ln2_b, q, a.ln_inv_n, a.ln_pad);
    lanes_vecnorm(Vc, a.vn_inv_n, a.vn_pad);
    if (valid && wave == 0) {                      // every wave holds the full result: wave 0 stores the scalars, wave 1 the vectors
        float *so = a.s + (size_t)v * S + 4 * q;
#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) *reinterpret_cast<v4f *>(so + 16 * nt) = x[nt];
    }
    if (valid && wave == 1) {
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
}

// ---- per-node blocks of the first message Linear: P[slot][node][:] = W_block s[node] (+ b) -----------------------------------------
template <int NTS>
__global__ __launch_bounds__(256, 2) void k_gvp_proj_coop(GvpProjArgs a) {
    constexpr int S = 16 * NTS, TPW = NTS / 4;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int sl = 0;
#pragma unroll
    for (int e = 1; e < GVP_PROJ_SLOTS; ++e)
        if (e < a.n_slots && (int)blockIdx.x >= a.tiles_first[e]) sl = e;
    const int node0 = ((int)blockIdx.x - a.tiles_first[sl]) * CR;
    const int n = a.n[sl];
    const int el = lane & 15, q = lane >> 4;
    const int vr = node0 + el;
    const int v = min(vr, n - 1);
    const float *sp = a.s[sl] + (size_t)v * S + 4 * q;
    v4f x[NTS], acc[TPW];
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) x[nt] = *reinterpret_cast<const v4f *>(sp + 16 * nt);
    const float *bias = a.b[sl];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = bias ? *reinterpret_cast<const v4f *>(bias + 16 * (wave * TPW + j) + 4 * q) : zero4();
    const v4f *wbase = reinterpret_cast<const v4f *>(a.wp[sl]) + lane + (size_t)(wave * TPW) * 64;
    coop_slab_gemm<NTS, NTS>(wbase, [&](int s) -> v4f { return x[s]; }, acc);
    if (vr < n) {
        float *out = a.P[sl] + (size_t)v * S + 16 * (wave * TPW) + 4 * q;
#pragma unroll
        for (int j = 0; j < TPW; ++j) *reinterpret_cast<v4f *>(out + 16 * j) = acc[j];
    }
}

// ---- noise prediction block (dynamics_gvp.py:10-44) ---------------------------------------------------------------------------------
// n_gvps - 1 generic GVPs, then the head GVP (S scalars, 16 vectors) -> (64 scalars, 1 vector): its four output tiles go one to a
// wave, the 64 scalars are exchanged, and gate / output vector / eps_h = Linear(64, F) are finished by wave 0.
template <int NTS>
__global__ __launch_bounds__(256, 2) void k_gvp_noise_coop(GvpNoiseArgs a) {
    using G = CoopGeom<NTS>;
    constexpr int S = G::S, TPW = G::TPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int node0 = blockIdx.x * CR;
    const int n_gen = a.n_gvps - 1;
    Coop<NTS> C;
    C.init(smem, wave, lane);
    const int el = C.el, q = C.q;
    const int vr = node0 + el;
    const bool valid = vr < a.n;
    const int v = valid ? vr : a.n - 1;
    v4f x[NTS], acc[TPW], Vc[3];
    const v4f no_res[TPW] = {};
    const float *sp = a.s + (size_t)v * S + 4 * q;
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) x[nt] = *reinterpret_cast<const v4f *>(sp + 16 * nt);
    load_vec12(a.v + (size_t)v * 48 + 12 * q, Vc);
    if (n_gen > 0) {
        const float *b0 = a.g[0].b + 16 * (wave * TPW) + 4 * q;
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = *reinterpret_cast<const v4f *>(b0 + 16 * j);
#pragma unroll 1
        for (int k = 0; k < n_gen; ++k) coop_generic_gvp<NTS>(C, a.g[k], k + 1 < n_gen ? a.g[k + 1].b : nullptr, false, no_res, x, acc, Vc);
    }
    // head GVP: vec1 in every wave; output tile `wave` of [x | sh] -> 64 scalars
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
    const v4f *wl = reinterpret_cast<const v4f *>(gl.chain) + lane;            // [slab][4 tiles][64 lanes], then the gate slab
    v4f so = *reinterpret_cast<const v4f *>(gl.b + 16 * wave + 4 * q);
    {
        const v4f *wt = wl + (size_t)wave * 64;
        v4f wq[CD];
#pragma unroll
        for (int d = 0; d < CD; ++d) wq[d] = wt[(size_t)d * 4 * 64];
#pragma unroll
        for (int slab = 0; slab <= NTS; ++slab) {
            const v4f xin = slab < NTS ? x[slab < NTS ? slab : 0] : sh;
#pragma unroll
            for (int r = 0; r < 4; ++r) so = mfma16(wq[slab % CD][r], xin[r], so);
            if (slab + CD <= NTS) wq[slab % CD] = wt[(size_t)(slab + CD) * 4 * 64];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) so[r] = silu(so[r]);
    // the 64 head scalars of a row: tile `wave` from every wave (the exchange buffers are S + 4 wide; 64 columns of one are used)
    v4f sall[4];
    {
        float *xb = C.xs + (C.xc & 1) * G::X_FLOATS + el * G::SX + 4 * q;
        *reinterpret_cast<v4f *>(xb + 16 * wave) = so;
        lds_barrier();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) sall[mt] = *reinterpret_cast<const v4f *>(xb + 16 * mt);
    }
    if (wave != 0) return;
    const v4f *wgl = wl + (size_t)(NTS + 1) * 4 * 64;                            // gate slab: 4 k-tiles of one output tile
    v4f ga = zero4();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const v4f wv = wgl[nt * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) ga = mfma16(wv[r], sall[nt][r], ga);
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
            const v4f wv = *reinterpret_cast<const v4f *>(wo + 16 * mt);
            part += sall[mt][0] * wv[0] + sall[mt][1] * wv[1] + sall[mt][2] * wv[2] + sall[mt][3] * wv[3];
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (valid && q == 0) a.eps_h[(size_t)v * a.F + f] = part + a.bout[f];
    }
}

}  // namespace

// Rows up to which a node-side launch takes the cooperative form (gvp_kernels.h).  Measured on gvp_40kp / gvp_all_atom shapes
// (profiles/r05_gvp_coop_ab.txt); KPD_COOP_ROWS (TOOLS build only) overrides it for the sweep.
int coop_rows_max(int requested) {
    static const int rows = tool_env_int("KPD_COOP_ROWS", COOP_ROWS_DEFAULT);
    return requested != 0 ? requested : rows;
}

kpd_status launch_gvp_node_coop(const GvpNodePair &pin, hipStream_t st) {
    GvpNodePair p = pin;
    const int S = p.nt[0].n ? p.nt[0].S : p.nt[1].S;
    p.tiles0 = cdiv(p.nt[0].n, CR);
    const int wgs = p.tiles0 + cdiv(p.nt[1].n, CR);
    if (S == 256) {
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_node_coop<16>), CoopGeom<16>::FLOATS * 4));
        hipLaunchKernelGGL(k_gvp_node_coop<16>, dim3(wgs), dim3(256), CoopGeom<16>::FLOATS * 4, st, p);
    } else {
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_node_coop<8>), CoopGeom<8>::FLOATS * 4));
        hipLaunchKernelGGL(k_gvp_node_coop<8>, dim3(wgs), dim3(256), CoopGeom<8>::FLOATS * 4, st, p);
    }
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_gvp_proj_coop(const GvpProjArgs &ain, hipStream_t st) {
    GvpProjArgs a = ain;
    int run = 0;
    for (int e = 0; e < a.n_slots; ++e) {
        a.tiles_first[e] = run;
        run += cdiv(a.n[e], CR);
    }
    a.tiles_first[a.n_slots] = run;
    if (a.S == 256) hipLaunchKernelGGL(k_gvp_proj_coop<16>, dim3(run), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_gvp_proj_coop<8>, dim3(run), dim3(256), 0, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_gvp_noise_coop(const GvpNoiseArgs &a, hipStream_t st) {
    if (a.S == 256) {
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_noise_coop<16>), CoopGeom<16>::FLOATS * 4));
        hipLaunchKernelGGL(k_gvp_noise_coop<16>, dim3(cdiv(a.n, CR)), dim3(256), CoopGeom<16>::FLOATS * 4, st, a);
    } else {
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_gvp_noise_coop<8>), CoopGeom<8>::FLOATS * 4));
        hipLaunchKernelGGL(k_gvp_noise_coop<8>, dim3(cdiv(a.n, CR)), dim3(256), CoopGeom<8>::FLOATS * 4, st, a);
    }
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd
