// Node projections of the EGNN denoiser: P[node][slot][:] = W1_block h[node] (+ b1 on destination slots), the per-node halves of the
// first Linear(515, 257) of edge_mlp / coord_mlp (models/dynamics.py:37-79, 103-105), for every slot the layer's edge types need.
// (The register-chained edge kernel and the streaming projection kernel that used to live in this file were A/B losers of rounds
// 1 - 3 and were removed in round 4; the weight-stationary form below is the one path.)
#include <algorithm>
#include <cstdlib>

#include "chain_core.h"
#include "egnn_kernels.h"

namespace kpd {

namespace {

constexpr int ENT = 16;                       // 16-wide feature tiles of the 256 MFMA features
constexpr int ECH4 = ENT * 64;                // float4 per weight chunk

__device__ __forceinline__ float reduce_q(float v) {      // sum over the four lanes (lane >> 4) that share a node
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

}  // namespace

// ---- weight-stationary form -------------------------------------------------------------------------------------------
// A workgroup keeps HALF a slot (128 output features x 256 inputs = 128 KB of A-fragments)
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
    const int slots = std::max(p.n_slots[0], p.n_slots[1]);
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    if (slots == 0) return KPD_OK;
    for (int nt = 0; nt < 2; ++nt)
        for (int s = 0; s < p.n_slots[nt]; ++s)
            KPD_REQUIRE(p.nt[nt].chain[s] && p.nt[nt].wcol[s], KPD_ERR_STATE, "projection slot %d not packed", s);
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_proj_ws), WS_LDS_BYTES));
    static const int target_env = std::max(0, tool_env_int("KPD_PROJ_WS_BLOCKS", 0));      // (TOOLS build only)
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

}  // namespace kpd
