// Building blocks of the register-chained MFMA kernels (gvp_chain.hip, egnn_chain.hip).
//
// Products are computed transposed on the 16x16x4 fp32 MFMA, T^T[n][e] = sum_k W[n][k] X^T[k][e]: the weight is
// the A operand, the activations the B operand.  A wave owns 16 rows (edges / nodes) e = lane & 15; a result tile
// holds the four features n = 16 mt + 4 (lane >> 4) + r in its four registers, which is also the B-operand layout
// of the next product (k = 16 nt + 4 (lane >> 4) + r).  Activations therefore chain through registers; only the
// weights move through LDS, as 16-row k-slabs ("chunks") pre-packed in A-fragment order by pack_chain_frag
// (pack.hip): chunk[(mt * 64 + lane) * 4 + r] = W[16 mt + (lane & 15)][k0 + 4 (lane >> 4) + r].
#pragma once
#include "mfma_core.h"

namespace kpd {

typedef float v4f __attribute__((ext_vector_type(4)));

// Pointers read from a device-resident table (the trainers' activation slots) are generic to the compiler: every access through them becomes
// a FLAT instruction, which counts in vmcnt AND lgkmcnt -- each wait for an LDS read in the MFMA loops then also waits for the stores in flight.
// G() states the address space; the accesses come out as global_load / global_store.
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) v4f gv4f;
typedef float v4f_u4 __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(1))) v4f_u4 gv4f_u;          // a 16-byte access at a 4-byte aligned address (rows of 17 floats)
__device__ __forceinline__ gfloat *G(float *p) { return (gfloat *)p; }
__device__ __forceinline__ const gfloat *G(const float *p) { return (const gfloat *)p; }
// The tables themselves (device arrays of pointers, written by the host before the launch and never by a kernel): read through a plain global
// pointer, a wave-uniform entry is still a VECTOR load -- the compiler may not use the scalar unit once the kernel has stored anything -- followed
// by s_waitcnt vmcnt(0), which in the training kernels drains the tens of activation stores in flight: a store round trip per pointer, 28 times
// per tile in k_gvp_chain<16, 0, 1>.  CT() states that the table is constant memory: the entries arrive by s_load (lgkmcnt), next to the stores.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T *CT(const T *p) {
    return (const __attribute__((address_space(4))) T *)p;
}

__device__ __forceinline__ v4f mfma16(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// One-instruction square root / reciprocal (v_sqrt_f32, v_rcp_f32: 1 ulp) for the per-row geometry and the vector norms of the chained
// kernels.  The IEEE-exact sqrtf / division of hipcc are ~8 / ~10 VALU instructions each, and on gfx950 a VALU instruction is time the
// fp32 MFMA pipe of the same SIMD does not get: 33 of them per edge-kernel tile were 9 % of its VALU work.
__device__ __forceinline__ float sqrt1(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rcp1(float x) { return __builtin_amdgcn_rcpf(x); }
// SiLU of the four values of a result tile with the three non-transcendental steps as float4 operations (hipcc turns them into
// v_pk_mul_f32 / v_pk_add_f32: 6 VALU + 8 transcendental instructions per four values instead of 12 + 8).  Same operations, same bits as silu().
__device__ __forceinline__ v4f silu4(v4f x) {
    const v4f t = x * -1.4426950408889634f;
    v4f e;
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(t[r]);
    const v4f d = e + 1.0f;          // (spelling this as v_pk_add_f32 by inline asm, as silu_pre4 of mfma_core.h does, costs k_gvp_chain its last registers: 8 B scratch)
    v4f q;
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] = __builtin_amdgcn_rcpf(d[r]);
    return x * q;
}
__device__ __forceinline__ v4f zero4() { return v4f{0.f, 0.f, 0.f, 0.f}; }

// acc[mt] += chunk[16 mt .. +15][16 k] * xin  (xin[r] = X^T[4 (lane >> 4) + r][e]); nreg < 4 limits the k-steps of a
// partially filled slab.  Four output tiles per LDS batch so that consecutive MFMAs never hit the same accumulator;
// the reads of batch g + 1 are pinned ahead of the 16 MFMAs of batch g (hipcc otherwise sinks them to two MFMAs before
// their first use and every batch eats the LDS latency).
// `after_first_reads` runs once, right behind the first batch of LDS reads: the ring's refill (address arithmetic + LDS-DMA issue) goes
// there, so that a chunk starts with its own operand reads instead of waiting behind the hand-off of a chunk two ahead.
template <int NTS, class H>
__device__ __forceinline__ void chunk_gemm(const v4f *__restrict__ buf, v4f xin, v4f (&acc)[NTS], int lane, int nreg, H &&after_first_reads) {
    const v4f *wp = buf + lane;
    v4f w[2][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) w[0][m] = wp[m * 64];
    __builtin_amdgcn_sched_barrier(0);
    after_first_reads();
#pragma unroll
    for (int g = 0; g < NTS / 4; ++g) {
        if (g + 1 < NTS / 4) {
#pragma unroll
            for (int m = 0; m < 4; ++m) w[(g + 1) & 1][m] = wp[(4 * (g + 1) + m) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < nreg) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[4 * g + m] = mfma16(w[g & 1][m][r], xin[r], acc[4 * g + m]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int NTS>
__device__ __forceinline__ void chunk_gemm(const v4f *__restrict__ buf, v4f xin, v4f (&acc)[NTS], int lane, int nreg) {
    chunk_gemm<NTS>(buf, xin, acc, lane, nreg, [] {});
}

// Two consecutive k-slabs (buf, buf + NTS * 64) against xa, xb with one read pipeline across both.
template <int NTS>
__device__ __forceinline__ void chunk_gemm2(const v4f *__restrict__ buf, v4f xa, v4f xb, v4f (&acc)[NTS], int lane) {
    const v4f *wp = buf + lane;
    constexpr int NB = NTS / 4;            // batches per slab
    v4f w[2][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) w[0][m] = wp[m * 64];
#pragma unroll
    for (int b = 0; b < 2 * NB; ++b) {
        if (b + 1 < 2 * NB) {
#pragma unroll
            for (int m = 0; m < 4; ++m) w[(b + 1) & 1][m] = wp[(4 * (b + 1) + m) * 64];     // slab 1 follows slab 0 in LDS
        }
        __builtin_amdgcn_sched_barrier(0);
        const v4f xin = b < NB ? xa : xb;
        const int g = b < NB ? b : b - NB;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[4 * g + m] = mfma16(w[b & 1][m][r], xin[r], acc[4 * g + m]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Weight-chunk ring of a 256-thread workgroup: chunks travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no
// staging registers, no ds_write), NBUF - 1 chunks ahead of the one being consumed, through NBUF buffers (below: NBUF = 3).
//   src(c)     this thread's source address of chunk c (CH4 float4 per chunk, thread t owns t + 256 j)
//   start()    begin chunks 0 and 1;   first()  chunk 0 is in LDS for every wave
//   acquire()  every wave has passed the barrier that ended chunk cur - 1, so buffer (cur + 2) % 3 is free: start chunk
//              cur + 2 into it, return the buffer of chunk cur
//   release()  wait for this thread's pieces of chunk cur + 1 (those of chunk cur + 2 stay in flight), then barrier:
//              everybody's pieces of chunk cur + 1 are in LDS and chunk cur is retired
//   drain()    before the ring memory is reused for something else
// MFMA streams run at priority 0 and everything else at 2: epilogues, gathers and ring hand-offs are short and
// latency bound, and a wave stuck behind another workgroup's full-rate MFMA stream stalls its own workgroup.
template <int CH4, int NBUF = 3>
struct ChunkRing {
    static constexpr int PT = CH4 / 256;
    static constexpr int AHEAD = NBUF - 1;          // chunks in flight beyond the one being consumed
    static_assert(PT == 2 || PT == 4, "chunk size");
    static_assert(NBUF == 3 || NBUF == 4, "ring depth");
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    v4f *ring;
    int cur, total, wave;
    int bi;                 // buffer of chunk cur (cur mod NBUF as a counter: no division per chunk)
    unsigned toff;          // this thread's offset inside a 256-thread piece (float4 units)

    // wave_: the (wave-uniform) wave index; tid: the thread index.  src(c) returns the wave-UNIFORM base of chunk c: the thread offset is
    // added here as a 32-bit lane offset, so that every piece is a scalar-base + lane-offset load (the pieces are 4 KB apart, past the
    // 12-bit immediate: with a per-thread 64-bit pointer each of them cost a 64-bit vector add)
    __device__ __forceinline__ void init(float *smem, int total_chunks, int wave_, int tid = -1) {
        ring = reinterpret_cast<v4f *>(smem);
        cur = 0;
        bi = 0;
        total = total_chunks;
        wave = wave_;
        toff = tid < 0 ? 0u : (unsigned)tid;
        __builtin_amdgcn_s_setprio(2);
    }
    __device__ __forceinline__ int ahead_buffer() const {
        const int b2 = bi + AHEAD;
        return b2 >= NBUF ? b2 - NBUF : b2;
    }
    // the chunk at the wave-uniform address g into buffer b
    __device__ __forceinline__ void fetch_ptr(const v4f *g, int b) {
        v4f *dst = ring + b * CH4 + 64 * wave;
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const v4f *gj = g + 256 * j;
            asm volatile("" : "+s"(gj));          // the piece's base stays a scalar pair (else hipcc folds it into a per-thread 64-bit pointer)
            __builtin_amdgcn_global_load_lds((glb_void *)(gj + toff), (lds_void *)(dst + 256 * j), 16, 0, 0);
        }
    }
    template <class F>
    __device__ __forceinline__ void fetch(F &&src, int c, int b) { fetch_ptr(src(min(c, total - 1)), b); }
    template <class F>
    __device__ __forceinline__ void start(F &&src) {
#pragma unroll
        for (int c = 0; c < AHEAD; ++c) fetch(src, c, c);
    }
    // this thread's pieces of everything but the AHEAD - 1 most recent chunks have landed
    __device__ __forceinline__ void wait_landed() {
        constexpr int n = (AHEAD - 1) * PT;
        if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __device__ __forceinline__ void first() {
        wait_landed();
        lds_barrier();
    }
    template <class F>
    __device__ __forceinline__ const v4f *acquire(F &&src) {
        fetch(src, cur + AHEAD, ahead_buffer());
        __builtin_amdgcn_sched_barrier(0);      // keep the fetch at the head of the chunk
        __builtin_amdgcn_s_setprio(0);
        return ring + bi * CH4;
    }
    // Explicit-source form: the caller knows where the chunk AHEAD of the current one lives (a compile-time offset from a per-GVP base
    // pointer in the chained GVP kernels), so the hand-off needs no chunk -> (stage, local) arithmetic and no pointer fetch from the
    // argument block; and the refill can be issued BEHIND the chunk's first operand reads (chunk_gemm's hook):
    //   buf = current();  chunk_gemm(buf, ..., [&] { prefetch(address of chunk cur + AHEAD); });  release();
    __device__ __forceinline__ const v4f *current() const { return ring + bi * CH4; }
    __device__ __forceinline__ void prefetch(const v4f *g_ahead) {
        fetch_ptr(g_ahead, ahead_buffer());
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
    }
    __device__ __forceinline__ void release() {
        __builtin_amdgcn_s_setprio(2);
        wait_landed();
        lds_barrier();
        ++cur;
        bi = bi + 1 == NBUF ? 0 : bi + 1;
    }
    __device__ __forceinline__ void drain() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
};

// Two-buffer variant for chunks of several k-slabs (fewer hand-offs per MFMA): chunk cur + 1 is fetched while chunk
// cur is consumed, so a chunk must last longer than an L2 round trip (two 16-KB slabs = 128 MFMAs per wave do).
//   acquire()  every wave has passed the barrier that ended chunk cur - 1: the other buffer is free, start chunk cur + 1
//   release()  wait for this thread's pieces of chunk cur + 1, barrier
template <int CH4>
struct ChunkRing2 {
    static constexpr int PT = CH4 / 256;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    v4f *ring;
    int cur, total, wave;

    __device__ __forceinline__ void init(float *smem, int total_chunks, int wave_) {
        ring = reinterpret_cast<v4f *>(smem);
        cur = 0;
        total = total_chunks;
        wave = wave_;
        __builtin_amdgcn_s_setprio(2);
    }
    template <class F>
    __device__ __forceinline__ void fetch(F &&src, int c, int b) {
        const v4f *g = src(min(c, total - 1));
        v4f *dst = ring + b * CH4 + 64 * wave;
#pragma unroll
        for (int j = 0; j < PT; ++j)
            __builtin_amdgcn_global_load_lds((glb_void *)(g + 256 * j), (lds_void *)(dst + 256 * j), 16, 0, 0);
    }
    template <class F>
    __device__ __forceinline__ void start(F &&src) { fetch(src, 0, 0); }
    __device__ __forceinline__ void first() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
    template <class F>
    __device__ __forceinline__ const v4f *acquire(F &&src) {
        fetch(src, cur + 1, (cur + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        return ring + (cur & 1) * CH4;
    }
    __device__ __forceinline__ void release() {
        __builtin_amdgcn_s_setprio(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        ++cur;
    }
    __device__ __forceinline__ void drain() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
};

// ---- GVPLayerNorm halves and the 12-float vector rows of the register-chained node kernels (gvp_chain.hip, gvp_coop.hip) ---------
// A node's S scalars live on its four lanes (x[nt][r] = s[node][16 nt + 4 (lane >> 4) + r]), so a layer norm is in-lane sums plus
// two cross-lane adds.
template <int NTS>
__device__ __forceinline__ void lanes_ln_stats(const v4f (&x)[NTS], float inv_n, float pad, float &mean, float &rstd) {
    // inv_n = 1 / n_hidden_scalars; `pad` trailing registers hold 0 (narrower models on these kernels): their (0 - mean)^2 is taken out
    // of the variance and their weight / bias are 0.  pad = 0 is bit-identical to the fixed-width form.
    float sum = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) sum += (x[nt][0] + x[nt][1]) + (x[nt][2] + x[nt][3]);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    mean = sum * inv_n;
    float var = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = x[nt][r] - mean;
            var = fmaf(d, d, var);
        }
    var += __shfl_xor(var, 16);
    var += __shfl_xor(var, 32);
    rstd = 1.0f / sqrtf((var - pad * mean * mean) * inv_n + 1e-5f);
}

// one 16-column tile of the normalised row (columns 16 nt + 4 q ..): what lanes_layernorm leaves in x[nt]
__device__ __forceinline__ v4f lanes_ln_tile(v4f xt, const float *__restrict__ lw, const float *__restrict__ lb, int nt, int q, float mean, float rstd) {
    const v4f w = *reinterpret_cast<const v4f *>(lw + 16 * nt + 4 * q), b = *reinterpret_cast<const v4f *>(lb + 16 * nt + 4 * q);
    return (xt - mean) * rstd * w + b;
}

template <int NTS>
__device__ __forceinline__ void lanes_layernorm(v4f (&x)[NTS], const float *__restrict__ lw, const float *__restrict__ lb, int q, float inv_n,
                                                float pad) {
    float mean, rstd;
    lanes_ln_stats<NTS>(x, inv_n, pad, mean, rstd);
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) x[nt] = lanes_ln_tile(x[nt], lw, lb, nt, q, mean, rstd);
}

// vector half of GVPLayerNorm (gvp.py:163-165): v / (sqrt(mean_i max(|v_i|^2, 1e-8) + eps) + eps)
// inv_n = 1 / vector_size; each of the `pad` zero padding channels (narrower models) adds the clamp value 1e-8 to the sum: taken out.
// pad = 0 is bit-identical to the fixed 16-channel form.
__device__ __forceinline__ void lanes_vecnorm(v4f (&V)[3], float inv_n, float pad) {
    float a = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) a += fmaxf(V[0][r] * V[0][r] + V[1][r] * V[1][r] + V[2][r] * V[2][r], 1e-8f);
    a += __shfl_xor(a, 16);
    a += __shfl_xor(a, 32);
    const float vn = sqrtf((a - pad * 1e-8f) * inv_n + 1e-5f) + 1e-5f;
#pragma unroll
    for (int c = 0; c < 3; ++c) V[c] = V[c] / vn;
}

__device__ __forceinline__ void load_vec12(const float *p, v4f (&V)[3]) {       // 4 channels x xyz -> V[c][r]
    const v4f *vp = reinterpret_cast<const v4f *>(p);
    const v4f t0 = vp[0], t1 = vp[1], t2 = vp[2];
    const float f[12] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3]};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) V[c][r] = f[3 * r + c];
}

}  // namespace kpd
