// EGNN edge kernel, register-chained formulation (LigRecConv messages, models/dynamics.py:111-120, 160-185).
//
// Same contract as the LDS-staged k_egnn_edge (egnn_kernels.hip): per tile of 64 dst-sorted same-type edges,
//   f1   = SiLU(P_src + P_dst + d w_r)              first Linear of edge_mlp / coord_mlp, applied per node (k_node_layer)
//   m    = SiLU(W2 f1 + b2)                         second Linear, 257 x 257
//   msg_h = m sigmoid(att(m)),  msg_x = tanh(w3 . m_c) range x_diff
// and the segmented sums of msg_h / msg_x over the destination nodes.  Here a wave owns 16 edges and computes
// m^T[n][e] = sum_k W2[n][k] f1^T[k][e] on the 16x16x4 fp32 MFMA (chain_core.h): f1 is assembled from the gathered
// P rows directly in B-operand registers, the result stays in registers for the attention / coordinate heads, and
// only W2 moves through LDS (16 chunks of 16 KB per branch, LDS-DMA ring shared by the four waves).  The 257th
// feature does not fit the 16-wide tiles: its input column and output row are rank-1 updates on the VALU.
// The coordinate branch runs first so that the feature messages can be staged for the segmented sum in the LDS the
// ring occupied.
#include <algorithm>
#include <cstdlib>

#include "chain_core.h"
#include "egnn_kernels.h"

namespace kpd {

namespace {

constexpr int ENT = 16;                       // 16-wide feature tiles of the 256 MFMA features
constexpr int ECH4 = ENT * 64;                // float4 per weight chunk
constexpr int ESO = 260;                      // row stride of the message staging tile
constexpr int EREGION0 = TM * ESO;            // >= 3 chunks (12288 floats)
constexpr int EVEC = 10;                      // per-edge-type vectors kept in LDS, HS floats each
enum { V_WR_C = 0, V_WR_E, V_B_C, V_B_E, V_WCOL_C, V_WCOL_E, V_WROW_C, V_WROW_E, V_W3, V_WATT };
constexpr int ECHAIN_FLOATS = EREGION0 + EVEC * HS + 3 * TM + TM + 16;

__device__ __forceinline__ float reduce_q(float v) {      // sum over the four lanes (lane >> 4) that share an edge
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

}  // namespace

#define ECHAIN_STAMP(idx)                                                                  \
    if (a.stamps && tid == 0) {                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
        atomicAdd(&a.stamps[idx], (unsigned long long)(now_ - t_prev_));                   \
        t_prev_ = now_;                                                                    \
    }

// HM = 1: the 256 x 256 part of both second Linears on v_mfma_f32_16x16x32_f16 (f16x2 mode): the ring carries
// hi / lo units (pack_egnn_chain_h), the eight k-slots of a lane per 32-wide k-block are result features of tiles 2 kb, 2 kb + 1.
template <int HM>
__global__ __launch_bounds__(256, 2) void k_egnn_chain(EdgeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *O = smem;                                  // [64][ESO] message staging (aliases the weight ring)
    float *vec = smem + EREGION0;                     // [EVEC][HS]
    float *s_mx = vec + EVEC * HS;                    // [64][3]
    int *s_dst = reinterpret_cast<int *>(s_mx + 3 * TM);
    int *misc = s_dst + TM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long t_prev_ = a.stamps ? __builtin_amdgcn_s_memtime() : 0ull;

    const int T = a.meta[8];
    const int chunk_tiles = (T + 7) >> 3;
    const int bi = blockIdx.x >> 3;
    if (bi >= chunk_tiles) return;
    const int tile = (blockIdx.x & 7) * chunk_tiles + bi;      // consecutive tiles share P rows and weights: same XCD
    if (tile >= T) return;
    int et = 0;
#pragma unroll
    for (int e = 1; e < 4; ++e)
        if (tile >= a.meta[4 + e]) et = e;
    const int tile_in_et = tile - a.meta[4 + et];
    const int e0 = tile_in_et * TM;
    const int ne = min(TM, a.meta[et] - e0);
    const int snt = a.src_nt[et], dnt = a.dst_nt[et];
    const int *__restrict__ esrc = a.src[et];
    const int *__restrict__ edst = a.dst[et];

    const v4f *stream = reinterpret_cast<const v4f *>(HM ? a.chain_h[et] : a.chain[et]) + tid;
    auto chunk_src = [&](int c) -> const v4f * { return stream + (size_t)c * ECH4; };
    ChunkRing<ECH4> ring;
    ring.init(smem, 2 * ENT, wave);
    ring.start(chunk_src);

    // ---- this lane's edge: endpoints and geometry (dynamics.py:160-169, 209-217) ---------------------------
    const int el = lane & 15, q = lane >> 4;
    const int row = 16 * wave + el;
    const int eidx = e0 + min(row, ne - 1);
    const int u = esrc[eidx], vd = edst[eidx];
    float d, xdv[3];
    {
        const float *xs = a.x[snt] + (size_t)u * 3, *xd = a.x[dnt] + (size_t)vd * 3;
        const float dx = xs[0] - xd[0], dy = xs[1] - xd[1], dz = xs[2] - xd[2];
        d = sqrtf(dx * dx + dy * dy + dz * dz);
        const float inv = 1.0f / (d + 1.0f);
        xdv[0] = dx * inv; xdv[1] = dy * inv; xdv[2] = dz * inv;
    }
    if (tid < TM) {     // run structure of the dst-sorted tile as two 64-bit masks (wave 0 == rows 0..63)
        const int e = e0 + min(tid, ne - 1);
        const int v = edst[e];
        s_dst[tid] = v;
        const int vprev = tid > 0 ? edst[e0 + min(tid - 1, ne - 1)] : (e0 > 0 ? edst[e0 - 1] : -1);
        const int vnext = tid + 1 < ne ? edst[e0 + tid + 1] : -2;
        const unsigned long long heads = __ballot(tid < ne && (tid == 0 || vprev != v));
        const unsigned long long ends = __ballot(tid < ne && vnext != v);
        if (tid == 0) {
            misc[0] = (vprev == v) ? 1 : 0;
            misc[2] = (int)(ends & 0xffffffffu);
            misc[3] = (int)(ends >> 32);
            misc[4] = (int)(heads & 0xffffffffu);
            misc[5] = (int)(heads >> 32);
        }
    }
    {   // per-edge-type vectors -> LDS (the two biases pre-scaled like the activations, mfma_core.h silu_pre)
        const float *srcs[EVEC] = {a.wr_c[et], a.wr_e[et], a.b_c[et], a.b_e[et], a.wcol_c[et], a.wcol_e[et],
                                   a.wx_c[et], a.wx_e[et], a.w3[et],  a.watt[et]};
#pragma unroll
        for (int i = 0; i < EVEC; ++i) {
            if (tid < HS / 4) {
                v4f t = reinterpret_cast<const v4f *>(srcs[i])[tid];
                if (i == V_B_C || i == V_B_E) t *= SILU_C;
                reinterpret_cast<v4f *>(vec + i * HS)[tid] = t;
            }
        }
    }
    ring.first();       // barrier: vectors, run masks and chunk 0 are in LDS
    ECHAIN_STAMP(0)

    const size_t prow = (size_t)NSLOT * HS;
    v4f acc[ENT];
    float x256 = 0.0f, t256 = 0.0f;

#pragma unroll 1
    for (int br = 1; br >= 0; --br) {       // 1: coordinate branch, 0: feature branch
        const float *Ps = a.P[snt] + (size_t)u * prow + (size_t)(a.src_slot[et] + br) * HS;
        const float *Pd = a.P[dnt] + (size_t)vd * prow + (size_t)(a.dst_slot[et] + br) * HS;
        const float *wr = vec + (br ? V_WR_C : V_WR_E) * HS;
        const float *cb = vec + (br ? V_B_C : V_B_E) * HS;
        const float *wcol = vec + (br ? V_WCOL_C : V_WCOL_E) * HS;
        const float *wrow = vec + (br ? V_WROW_C : V_WROW_E) * HS;
        const float *wh = vec + (br ? V_W3 : V_WATT) * HS;

        // f1 = SiLU(P_src + P_dst + d w_r) in B-operand registers (P and w_r carry the SiLU pre-scale)
        // The scheduler fences keep hipcc from hoisting the later LDS vector reads over the gathers: all 32 gathers of
        // a lane (128 registers) are in flight at once and everything else is read just in time.
        v4f x[ENT];
        {
            v4f gs[ENT], gd[ENT];
#pragma unroll
            for (int nt = 0; nt < ENT; ++nt) {
                gs[nt] = *reinterpret_cast<const v4f *>(Ps + 16 * nt + 4 * q);
                gd[nt] = *reinterpret_cast<const v4f *>(Pd + 16 * nt + 4 * q);
            }
            x256 = Ps[256] + Pd[256];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < ENT / 4; ++g) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int nt = 4 * g + m;
                    v4f f = gs[nt] + gd[nt] + d * *reinterpret_cast<const v4f *>(wr + 16 * nt + 4 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[r] = silu_pre(f[r]);
                    x[nt] = f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            x256 = silu_pre(x256 + d * wr[256]);
        }
        // accumulators start from the bias and the rank-1 term of input feature 256; the output row 256 is a dot
        // product whose partial sums live on the four lanes of the edge
        float part = 0.0f;
#pragma unroll
        for (int g = 0; g < ENT / 4; ++g) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int mt = 4 * g + m;
                acc[mt] = *reinterpret_cast<const v4f *>(cb + 16 * mt + 4 * q) + x256 * *reinterpret_cast<const v4f *>(wcol + 16 * mt + 4 * q);
                const v4f wv = *reinterpret_cast<const v4f *>(wrow + 16 * mt + 4 * q);
                part += x[mt][0] * wv[0] + x[mt][1] * wv[1] + x[mt][2] * wv[2] + x[mt][3] * wv[3];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        ECHAIN_STAMP(1 + 4 * br)
        if constexpr (HM) {
#pragma unroll
            for (int mt = 0; mt < ENT; ++mt) acc[mt] = acc[mt] * (1.0f / H_UNSCALE);
#pragma unroll
            for (int kb = 0; kb < ENT / 2; ++kb) {
                f32x4 xh, xl;
                {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                    split_pair(H_SCALE_A * x[2 * kb][0], H_SCALE_A * x[2 * kb][1], h0, l0);
                    split_pair(H_SCALE_A * x[2 * kb][2], H_SCALE_A * x[2 * kb][3], h1, l1);
                    split_pair(H_SCALE_A * x[2 * kb + 1][0], H_SCALE_A * x[2 * kb + 1][1], h2, l2);
                    split_pair(H_SCALE_A * x[2 * kb + 1][2], H_SCALE_A * x[2 * kb + 1][3], h3, l3);
                    xh = __builtin_bit_cast(f32x4, u32x4{h0, h1, h2, h3});
                    xl = __builtin_bit_cast(f32x4, u32x4{l0, l1, l2, l3});
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const f32x4 *wp = reinterpret_cast<const f32x4 *>(ring.acquire(chunk_src)) + lane;
                    f32x4 w[2][8];          // two batches of four output tiles (hi, lo); batch 1 is read while batch 0 multiplies
#pragma unroll
                    for (int i = 0; i < 8; ++i) w[0][i] = wp[i * 64];
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        if (b == 0) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) w[1][i] = wp[(8 + i) * 64];
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
#pragma unroll
            for (int mt = 0; mt < ENT; ++mt) acc[mt] = acc[mt] * H_UNSCALE;
        } else {
#pragma unroll
            for (int nt = 0; nt < ENT; ++nt) {
                const v4f *buf = ring.acquire(chunk_src);
                chunk_gemm<ENT>(buf, x[nt], acc, lane, 4);
                ring.release();
            }
        }
        ECHAIN_STAMP(2 + 4 * br)
        // m = SiLU(.) (pre-scaled: registers hold c m), head dot product over all 257 features
        t256 = silu_pre(reduce_q(part) + wrow[BIAS_K] + x256 * wrow[256]);
        float hp = 0.0f;
#pragma unroll
        for (int mt = 0; mt < ENT; ++mt) {
            const v4f wv = *reinterpret_cast<const v4f *>(wh + 16 * mt + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[mt][r] = silu_pre(acc[mt][r]);
                hp = fmaf(acc[mt][r], wv[r], hp);
            }
        }
        const float dot = reduce_q(hp) + t256 * wh[256];
        if (br) {
            // msg_x = tanh(coord_mlp(f)) * x_diff * range                           (dynamics.py:113-120)
            float c = a.use_tanh ? tanhf(dot) * a.coords_range : dot;
            if (row >= ne) c = 0.0f;
            if (q == 0) {
                s_mx[3 * row] = c * xdv[0];
                s_mx[3 * row + 1] = c * xdv[1];
                s_mx[3 * row + 2] = c * xdv[2];
            }
        } else {
            // msg_h = m * sigmoid(att(m)): registers hold c m, w_att carries 1 / c, the weight returns 1 / c  (:111-112)
            const float att = row < ne ? sigmoidf_(dot + wh[ATT_BIAS_AT]) * (1.0f / SILU_C) : 0.0f;
#pragma unroll
            for (int mt = 0; mt < ENT; ++mt) acc[mt] *= att;
            t256 *= att;
        }
        ECHAIN_STAMP(3 + 4 * br)
    }

    // ---- messages -> LDS (the ring memory is reused: drain the tail fetches first) --------------------------
    ring.drain();
    {
        float *orow = O + row * ESO + 4 * q;
#pragma unroll
        for (int mt = 0; mt < ENT; ++mt) *reinterpret_cast<v4f *>(orow + 16 * mt) = acc[mt];
        if (q == 0) O[row * ESO + 256] = t256;
    }
    lds_barrier();
    ECHAIN_STAMP(9)

    // ---- segmented sums over dst (dynamics.py:182-185): thread = column, rows in order; run boundaries are
    // wave-uniform (endmask), LDS reads are issued 16 rows at a time --------------------------------------------
    const int first_is_cont = misc[0];
    const unsigned long long endmask = ((unsigned long long)(unsigned)misc[3] << 32) | (unsigned long long)(unsigned)misc[2];
    const unsigned long long heads = ((unsigned long long)(unsigned)misc[5] << 32) | (unsigned long long)(unsigned)misc[4];
    float *hmain = a.hn_main[et], *hcont = a.hn_cont[et] + (size_t)tile_in_et * HS;
    {
        float run = 0.0f;
        int piece = 0;
#pragma unroll 1
        for (int r0 = 0; r0 < TM; r0 += 16) {
            if (r0 >= ne) break;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = O[(r0 + i) * ESO + tid];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                run += v[i];
                if ((endmask >> (r0 + i)) & 1ull) {
                    float *out = (piece == 0 && first_is_cont) ? hcont : hmain + (size_t)s_dst[r0 + i] * HS;
                    out[tid] = run;
                    run = 0.0f;
                    ++piece;
                }
            }
        }
    }
    // column 256 (last wave) and the coordinate messages (wave 0): lane = row, segmented inclusive scan across lanes
    if (wave == 3 || wave == 0) {
        const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
        const int start = 63 - __clzll((long long)((heads & upto) | 1ull));
        const bool is_end = (endmask >> lane) & 1ull;
        const int pc = __popcll(endmask & ((1ull << lane) - 1ull));
        const bool to_cont = pc == 0 && first_is_cont;
        if (wave == 3) {
            float v = O[lane * ESO + 256];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float t = __shfl_up(v, off);
                if (lane - off >= start) v += t;
            }
            if (is_end) (to_cont ? hcont : hmain + (size_t)s_dst[lane] * HS)[256] = v;
        } else {
            float vx = s_mx[3 * lane], vy = s_mx[3 * lane + 1], vz = s_mx[3 * lane + 2];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float tx = __shfl_up(vx, off), ty = __shfl_up(vy, off), tz = __shfl_up(vz, off);
                if (lane - off >= start) {
                    vx += tx;
                    vy += ty;
                    vz += tz;
                }
            }
            if (is_end) {
                float *out = to_cont ? a.xn_cont[et] + (size_t)tile_in_et * 4 : a.xn_main[et] + (size_t)s_dst[lane] * 4;
                out[0] = vx;
                out[1] = vy;
                out[2] = vz;
            }
        }
    }
    ECHAIN_STAMP(10)
}

kpd_status launch_egnn_chain(const EdgeArgs &a, int tile_cap, hipStream_t st) {
    if (tile_cap == 0) return KPD_OK;
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    if (a.gemm_mode == 1) {
        for (int et = 0; et < 4; ++et) KPD_REQUIRE(!a.chain[et] || a.chain_h[et], KPD_ERR_STATE, "edge type %d has no f16x2 chain units", et);
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_egnn_chain<1>), ECHAIN_FLOATS * 4));
        hipLaunchKernelGGL(k_egnn_chain<1>, dim3(8 * cdiv(tile_cap, 8)), dim3(256), ECHAIN_FLOATS * 4, st, a);
        KPD_LAUNCH_CHECK();
        return KPD_OK;
    }
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_egnn_chain<0>), ECHAIN_FLOATS * 4));
    hipLaunchKernelGGL(k_egnn_chain<0>, dim3(8 * cdiv(tile_cap, 8)), dim3(256), ECHAIN_FLOATS * 4, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd

// ---- node projections, register-chained -------------------------------------------------------------------
// P[node][slot][:] = c (W1_block h[node] (+ b1)) for one 64-node tile and slots_per_block slots per workgroup (same
// contract as the projection half of k_node_layer, egnn_kernels.hip).  The nodes' features are read straight into
// B-operand registers, the // 256 x 256 block of the slot's weight streams through the LDS ring, feature 256 on either side is a rank-1 / dot
// product update on the VALU.  Many small workgroups (tiles x slots) keep the hardware dispatcher balanced.
namespace kpd {

__global__ __launch_bounds__(256, 2) void k_proj_chain(ProjPair p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int which = blockIdx.x >= p.tiles0 ? 1 : 0;
    const ProjArgs &a = p.nt[which];
    const int s0 = blockIdx.y * p.slots_per_block;
    if (s0 >= p.n_slots[which]) return;
    const int ns = min(p.slots_per_block, p.n_slots[which] - s0);
    const int node0 = (blockIdx.x - (which ? p.tiles0 : 0)) * TM;

    // chunks of two k-slabs (32 KB): slab pairs are contiguous in the packed stream
    auto chunk_src = [&](int c) -> const v4f * {
        return reinterpret_cast<const v4f *>(a.chain[s0 + (c >> 3)]) + (size_t)(c & 7) * (2 * ECH4) + tid;
    };
    ChunkRing2<2 * ECH4> ring;
    ring.init(smem, ns * (ENT / 2), wave);
    ring.start(chunk_src);

    const int el = lane & 15, q = lane >> 4;
    const int row = 16 * wave + el;
    // h and P are padded to whole tiles (rows past n: zero / never read back), so no row predicate
    const float *hrow = a.h + (size_t)(node0 + row) * HS;
    v4f x[ENT];
#pragma unroll
    for (int nt = 0; nt < ENT; ++nt) x[nt] = *reinterpret_cast<const v4f *>(hrow + 16 * nt + 4 * q);
    const float h256 = hrow[256];
#pragma unroll 1
    for (int si = 0; si < ns; ++si) {
        const int s = s0 + si;
        const float *bias = a.bias[s], *wcol = a.wcol[s], *wrow = a.wx[s];
        v4f acc[ENT];
        float part = 0.0f;
#pragma unroll
        for (int mt = 0; mt < ENT; ++mt) {
            acc[mt] = h256 * *reinterpret_cast<const v4f *>(wcol + 16 * mt + 4 * q);
            if (bias) acc[mt] += *reinterpret_cast<const v4f *>(bias + 16 * mt + 4 * q);
            const v4f wv = *reinterpret_cast<const v4f *>(wrow + 16 * mt + 4 * q);
            part += x[mt][0] * wv[0] + x[mt][1] * wv[1] + x[mt][2] * wv[2] + x[mt][3] * wv[3];
        }
        const float out256 = reduce_q(part) + (bias ? bias[256] : 0.0f) + h256 * wrow[256];
        if (si == 0) ring.first();
#pragma unroll
        for (int nt = 0; nt < ENT; nt += 2) {
            const v4f *buf = ring.acquire(chunk_src);
            chunk_gemm2<ENT>(buf, x[nt], x[nt + 1], acc, lane);
            ring.release();
        }
        float *orow = a.P + ((size_t)(node0 + row) * NSLOT + a.slot[s]) * HS;
#pragma unroll
        for (int mt = 0; mt < ENT; ++mt) *reinterpret_cast<v4f *>(orow + 16 * mt + 4 * q) = acc[mt];
        if (q == 0) orow[256] = out256;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's tail fetches must not outlive the workgroup's LDS
}

// ---- weight-stationary form -------------------------------------------------------------------------------------------
// k_proj_chain streams a slot's 256 x 256 block through LDS for every 64-node tile (2 600 workgroups x 262 KB from L2, a
// barrier per chunk).  Here a workgroup keeps HALF a slot (128 output features x 256 inputs = 128 KB of A-fragments)
// resident in LDS for its whole life and walks many 128-node tiles: 8 waves x 16 nodes, features in B-operand registers,
// no barrier and no weight refill inside the GEMM, next tile's rows prefetched during the current tile's 512 MFMAs.
// One workgroup per CU (LDS), two waves per SIMD.
struct ProjWs {
    ProjPair p;
    int blocks0;        // workgroups of node type 0
    int bpc[2];         // workgroups per (slot, half) of each node type
    int tpb;            // 128-node tiles per workgroup
};

constexpr int WS_TILE = 128;                                  // nodes per tile: 8 waves x 16
constexpr int WS_W4 = 16 * 8 * 64;                            // float4 of the resident half block
constexpr int WS_LDS_BYTES = WS_W4 * 16 + (128 + 128 + HS) * 4;

__global__ __launch_bounds__(512, 1) void k_proj_ws(ProjWs qa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    v4f *W = reinterpret_cast<v4f *>(smem);
    float *s_wcol = smem + WS_W4 * 4, *s_bias = s_wcol + 128, *s_wrow = s_bias + 128;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int which = (int)blockIdx.x >= qa.blocks0 ? 1 : 0;
    const ProjArgs &a = qa.p.nt[which];
    const int local = blockIdx.x - (which ? qa.blocks0 : 0);
    const int combo = local / qa.bpc[which], chunk = local - combo * qa.bpc[which];
    const int s = combo >> 1, hf = combo & 1;
    const int tiles = (a.n + WS_TILE - 1) / WS_TILE;
    const int t0 = chunk * qa.tpb, t1 = min(tiles, t0 + qa.tpb);
    if (t0 >= t1) return;                                     // uniform over the workgroup
    {
        const v4f *src = reinterpret_cast<const v4f *>(a.chain[s]);
#pragma unroll
        for (int j = 0; j < WS_W4 / 512; ++j) {
            const int i = tid + 512 * j, ks = i >> 9, rem = i & 511;
            W[i] = src[(size_t)ks * ECH4 + 8 * hf * 64 + rem];
        }
        const float *bias = a.bias[s];
        if (tid < 128) {
            s_wcol[tid] = a.wcol[s][128 * hf + tid];
            s_bias[tid] = bias ? bias[128 * hf + tid] : 0.0f;
        }
        for (int i = tid; i < HS; i += 512) s_wrow[i] = i <= 256 ? a.wx[s][i] : 0.0f;
    }
    __syncthreads();
    const int el = lane & 15, q = lane >> 4;
    const float bias256 = a.bias[s] ? a.bias[s][256] : 0.0f;
    auto load_x = [&](int t, v4f (&x)[ENT], float &x256) {
        const int row = min(t * WS_TILE + 16 * wave + el, a.n - 1);
        const float *hrow = a.h + (size_t)row * HS;
#pragma unroll
        for (int nt = 0; nt < ENT; ++nt) x[nt] = *reinterpret_cast<const v4f *>(hrow + 16 * nt + 4 * q);
        x256 = hrow[256];
    };
    auto store_tile = [&](const v4f (&acc)[8], int row, float out256) {
        if (row >= 0 && row < a.n) {
            float *orow = a.P + ((size_t)row * NSLOT + a.slot[s]) * HS + 128 * hf;
#pragma unroll
            for (int m = 0; m < 8; ++m) *reinterpret_cast<v4f *>(orow + 16 * m + 4 * q) = acc[m];
            if (hf == 0 && q == 0) orow[256] = out256;
        }
    };
    v4f x[ENT], xn[ENT], accp[8];
    float h256, h256n, out256p = 0.0f;
    int rowp = -1;
#pragma unroll
    for (int m = 0; m < 8; ++m) accp[m] = zero4();
    load_x(t0, xn, h256n);
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        // The rows of this tile were loaded into xn a whole tile ago; x is only ever written by these moves, so the wait the
        // compiler places here counts the loads alone (the previous tile's stores are younger and stay in flight) and the
        // MFMAs below never wait on memory.  (A plain copy is coalesced away and the wait sinks behind the stores.)
#pragma unroll
        for (int nt = 0; nt < ENT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(x[nt][r]) : "v"(xn[nt][r]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(h256) : "v"(h256n));
        __builtin_amdgcn_sched_barrier(0);
        // the previous tile's results leave only now, after the wait above: stores share the in-order memory counter with
        // the loads, and a store issued before that wait would be waited for as well (its full round trip, every tile)
        store_tile(accp, rowp, out256p);
        if (t + 1 < t1) load_x(t + 1, xn, h256n);             // in flight during this tile's MFMAs
        v4f acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m)
            acc[m] = h256 * *reinterpret_cast<const v4f *>(s_wcol + 16 * m + 4 * q) + *reinterpret_cast<const v4f *>(s_bias + 16 * m + 4 * q);
        float part = 0.0f;
        if (hf == 0) {
#pragma unroll
            for (int nt = 0; nt < ENT; ++nt) {
                const v4f wv = *reinterpret_cast<const v4f *>(s_wrow + 16 * nt + 4 * q);
                part += x[nt][0] * wv[0] + x[nt][1] * wv[1] + x[nt][2] * wv[2] + x[nt][3] * wv[3];
            }
        }
        // 32 batches of 4 output tiles (16 k-slabs x 2), LDS reads of batch b + 1 pinned ahead of the 16 MFMAs of batch b
        const v4f *wp = W + lane;
        v4f w[2][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) w[0][m] = wp[m * 64];
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            if (b + 1 < 32) {
#pragma unroll
                for (int m = 0; m < 4; ++m) w[(b + 1) & 1][m] = wp[(4 * (b + 1) + m) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            const v4f xin = x[b >> 1];
            const int g = b & 1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[4 * g + m] = mfma16(w[b & 1][m][r], xin[r], acc[4 * g + m]);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int row = t * WS_TILE + 16 * wave + el;
        const float out256 = hf == 0 ? reduce_q(part) + bias256 + h256 * s_wrow[256] : 0.0f;
#pragma unroll
        for (int m = 0; m < 8; ++m) accp[m] = acc[m];
        rowp = row;
        out256p = out256;
    }
    store_tile(accp, rowp, out256p);
}

// ---- weight-stationary form, f16x2 split (opt-in gemm mode 1; fact 10 of DESIGN.md) -------------------------------------
// Same walk as k_proj_ws with every fp32 product replaced by three f16 products of hi / lo planes on v_mfma_f32_16x16x32_f16
// (A = weight fragment 16 features x 32 k from LDS, B = the nodes' features 32 k x 16 nodes from registers, fp32 accumulate).
// The half slot lives in LDS as planes scaled by 2^10 (pack_proj_f16_split): [kb 8][m 8][plane 2][lane 64] fragments of 16 B,
// 128 KB as before.  A lane owns node el and, per k-block kb, the eight features 32 kb + 8 q + j: two dwordx4 loads of the fp32
// row, scaled by 2^6 and split on the VALU while the previous tile's stores and the next tile's loads are in flight.  The
// accumulators start at 2^16 (h256 wcol + bias) and leave multiplied by 2^-16.
__global__ __launch_bounds__(512, 1) void k_proj_ws_h(ProjWs qa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4 *W = reinterpret_cast<f32x4 *>(smem);                 // 16-B fragments (8 halves)
    float *s_wcol = smem + WS_W4 * 4, *s_bias = s_wcol + 128, *s_wrow = s_bias + 128;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int which = (int)blockIdx.x >= qa.blocks0 ? 1 : 0;
    const ProjArgs &a = qa.p.nt[which];
    const int local = blockIdx.x - (which ? qa.blocks0 : 0);
    const int combo = local / qa.bpc[which], chunk = local - combo * qa.bpc[which];
    const int s = combo >> 1, hf = combo & 1;
    const int tiles = (a.n + WS_TILE - 1) / WS_TILE;
    const int t0 = chunk * qa.tpb, t1 = min(tiles, t0 + qa.tpb);
    if (t0 >= t1) return;                                     // uniform over the workgroup
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.chain_h[s]);
#pragma unroll
        for (int j = 0; j < WS_W4 / 512; ++j) {
            const int i = tid + 512 * j, kb = i >> 10, rem = i & 1023;          // rem = (m, plane, lane)
            W[i] = src[(size_t)(kb * 16 + 8 * hf) * 128 + rem];
        }
        const float *bias = a.bias[s];
        if (tid < 128) {
            s_wcol[tid] = a.wcol[s][128 * hf + tid];
            s_bias[tid] = bias ? bias[128 * hf + tid] : 0.0f;
        }
        for (int i = tid; i < HS; i += 512) s_wrow[i] = i <= 256 ? a.wx[s][i] : 0.0f;
    }
    __syncthreads();
    const int el = lane & 15, q = lane >> 4;
    const float bias256 = a.bias[s] ? a.bias[s][256] : 0.0f;
    auto load_x = [&](int t, v4f (&x)[16], float &x256) {
        const int row = min(t * WS_TILE + 16 * wave + el, a.n - 1);
        const float *hrow = a.h + (size_t)row * HS + 8 * q;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            x[2 * kb] = *reinterpret_cast<const v4f *>(hrow + 32 * kb);
            x[2 * kb + 1] = *reinterpret_cast<const v4f *>(hrow + 32 * kb + 4);
        }
        x256 = hrow[256 - 8 * q];
    };
    auto store_tile = [&](const v4f (&acc)[8], int row) {
        if (row >= 0 && row < a.n) {
            float *orow = a.P + ((size_t)row * NSLOT + a.slot[s]) * HS + 128 * hf;
#pragma unroll
            for (int m = 0; m < 8; ++m) *reinterpret_cast<v4f *>(orow + 16 * m + 4 * q) = acc[m];
        }
    };
    v4f xn[16], accp[8];
    f32x4 xh[8], xl[8];                                       // the tile's features as f16 planes, one fragment per k-block
    float h256n;
    int rowp = -1;
#pragma unroll
    for (int m = 0; m < 8; ++m) accp[m] = zero4();
    load_x(t0, xn, h256n);
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        // split this tile's rows (loaded a whole tile ago); the extra output feature 256 is a VALU dot on the fp32 values
        const float h256 = h256n;
        float part = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const v4f u = xn[2 * kb], v = xn[2 * kb + 1];
            if (hf == 0) {
                const v4f w0 = *reinterpret_cast<const v4f *>(s_wrow + 32 * kb + 8 * q), w1 = *reinterpret_cast<const v4f *>(s_wrow + 32 * kb + 8 * q + 4);
                part += u[0] * w0[0] + u[1] * w0[1] + u[2] * w0[2] + u[3] * w0[3] + v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
            }
            unsigned h0, h1, h2, h3, l0, l1, l2, l3;
            split_pair(H_SCALE_A * u[0], H_SCALE_A * u[1], h0, l0);
            split_pair(H_SCALE_A * u[2], H_SCALE_A * u[3], h1, l1);
            split_pair(H_SCALE_A * v[0], H_SCALE_A * v[1], h2, l2);
            split_pair(H_SCALE_A * v[2], H_SCALE_A * v[3], h3, l3);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            xh[kb] = __builtin_bit_cast(f32x4, u32x4{h0, h1, h2, h3});
            xl[kb] = __builtin_bit_cast(f32x4, u32x4{l0, l1, l2, l3});
        }
        __builtin_amdgcn_sched_barrier(0);
        // the previous tile's results leave only now, after the wait on this tile's rows (see k_proj_ws)
        store_tile(accp, rowp);
        if (t + 1 < t1) load_x(t + 1, xn, h256n);             // in flight during this tile's MFMAs
        const int row = t * WS_TILE + 16 * wave + el;
        if (hf == 0) {
            const float out256 = reduce_q(part) + bias256 + h256 * s_wrow[256];
            if (q == 0 && row < a.n) a.P[((size_t)row * NSLOT + a.slot[s]) * HS + 256] = out256;
        }
        v4f acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m)
            acc[m] = (1.0f / H_UNSCALE) * (h256 * *reinterpret_cast<const v4f *>(s_wcol + 16 * m + 4 * q) + *reinterpret_cast<const v4f *>(s_bias + 16 * m + 4 * q));
        // 32 batches of two output tiles x three products; LDS reads of batch b + 1 pinned ahead of the 6 MFMAs of batch b
        const f32x4 *wp = W + lane;
        f32x4 w[3][4];                                        // [buffer][m0 hi, m0 lo, m1 hi, m1 lo]: refilled one batch after its
                                                              // readers were issued, never right behind them (mfma_core.h, gemm_rows64_h)
#pragma unroll
        for (int i = 0; i < 4; ++i) w[0][i] = wp[i * 64];
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            if (b + 1 < 32) {
#pragma unroll
                for (int i = 0; i < 4; ++i) w[(b + 1) % 3][i] = wp[(4 * (b + 1) + i) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            const int kb = b >> 2, m0 = 2 * (b & 3);
            const f32x4 *wb = w[b % 3];
            acc[m0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wb[1]), as_h8(xh[kb]), acc[m0], 0, 0, 0);
            acc[m0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wb[3]), as_h8(xh[kb]), acc[m0 + 1], 0, 0, 0);
            acc[m0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wb[0]), as_h8(xl[kb]), acc[m0], 0, 0, 0);
            acc[m0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wb[2]), as_h8(xl[kb]), acc[m0 + 1], 0, 0, 0);
            acc[m0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wb[0]), as_h8(xh[kb]), acc[m0], 0, 0, 0);
            acc[m0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wb[2]), as_h8(xh[kb]), acc[m0 + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) accp[m] = H_UNSCALE * acc[m];
        rowp = row;
    }
    store_tile(accp, rowp);
}

kpd_status launch_proj_chain(const ProjPair &p, hipStream_t st) {
    const int tiles = p.tiles0 + cdiv(p.nt[1].n, TM);
    const int slots = std::max(p.n_slots[0], p.n_slots[1]);
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    if (tiles == 0 || slots == 0) return KPD_OK;
    for (int nt = 0; nt < 2; ++nt)
        for (int s = 0; s < p.n_slots[nt]; ++s)
            KPD_REQUIRE(p.nt[nt].chain[s] && p.nt[nt].wcol[s], KPD_ERR_STATE, "projection slot %d not packed for k_proj_chain", s);
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_proj_chain), 4 * ECH4 * 16));
    static const int ws = getenv("KPD_PROJ_WS") ? atoi(getenv("KPD_PROJ_WS")) : 1;
    if (ws) {
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_proj_ws), WS_LDS_BYTES));
        static const int target_env = getenv("KPD_PROJ_WS_BLOCKS") ? std::max(1, atoi(getenv("KPD_PROJ_WS_BLOCKS"))) : 0;
        const int target = target_env ? target_env : cu_count();
        ProjWs q;
        q.p = p;
        int units = 0, tl[2];
        for (int nt = 0; nt < 2; ++nt) {
            tl[nt] = p.n_slots[nt] ? cdiv(p.nt[nt].n, WS_TILE) : 0;
            units += 2 * p.n_slots[nt] * tl[nt];
        }
        if (units == 0) return KPD_OK;
        // one workgroup per CU and ONE round: the smallest tiles-per-workgroup whose grid fits the target (a grid of 272 on
        // 256 CUs runs its last 16 workgroups alone and doubles the kernel: B = 66 measured 353 vs 190 us)
        int blocks[2];
        for (q.tpb = std::max(1, cdiv(units, target));; ++q.tpb) {
            for (int nt = 0; nt < 2; ++nt) {
                q.bpc[nt] = std::max(1, cdiv(tl[nt], q.tpb));
                blocks[nt] = tl[nt] ? 2 * p.n_slots[nt] * q.bpc[nt] : 0;
            }
            if (blocks[0] + blocks[1] <= target || q.tpb >= std::max(tl[0], tl[1])) break;
        }
        q.blocks0 = blocks[0];
        if (p.gemm_mode == 1) {
            for (int nt = 0; nt < 2; ++nt)
                for (int s = 0; s < p.n_slots[nt]; ++s)
                    KPD_REQUIRE(p.nt[nt].chain_h[s], KPD_ERR_STATE, "projection slot %d has no f16 planes (f16x2 mode)", s);
            KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_proj_ws_h), WS_LDS_BYTES));
            hipLaunchKernelGGL(k_proj_ws_h, dim3(blocks[0] + blocks[1]), dim3(512), WS_LDS_BYTES, st, q);
            KPD_LAUNCH_CHECK();
            return KPD_OK;
        }
        hipLaunchKernelGGL(k_proj_ws, dim3(blocks[0] + blocks[1]), dim3(512), WS_LDS_BYTES, st, q);
        KPD_LAUNCH_CHECK();
        return KPD_OK;
    }
    static const int spb = getenv("KPD_PROJ_SPB") ? std::max(1, atoi(getenv("KPD_PROJ_SPB"))) : 1;
    ProjPair q = p;
    q.slots_per_block = spb;
    hipLaunchKernelGGL(k_proj_chain, dim3(tiles, cdiv(slots, spb)), dim3(256), 4 * ECH4 * 16, st, q);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd
