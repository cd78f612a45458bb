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

__device__ __forceinline__ v4f mfma16(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4f zero4() { return v4f{0.f, 0.f, 0.f, 0.f}; }

// acc[mt] += chunk[16 mt .. +15][16 k] * xin  (xin[r] = X^T[4 (lane >> 4) + r][e]); nreg < 4 limits the k-steps of a
// partially filled slab.  Four output tiles per LDS batch so that consecutive MFMAs never hit the same accumulator;
// the reads of batch g + 1 are pinned ahead of the 16 MFMAs of batch g (hipcc otherwise sinks them to two MFMAs before
// their first use and every batch eats the LDS latency).
template <int NTS>
__device__ __forceinline__ void chunk_gemm(const v4f *__restrict__ buf, v4f xin, v4f (&acc)[NTS], int lane, int nreg) {
    const v4f *wp = buf + lane;
    v4f w[2][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) w[0][m] = wp[m * 64];
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
// staging registers, no ds_write), two chunks ahead of the one being consumed, through three buffers.
//   src(c)     this thread's source address of chunk c (CH4 float4 per chunk, thread t owns t + 256 j)
//   start()    begin chunks 0 and 1;   first()  chunk 0 is in LDS for every wave
//   acquire()  every wave has passed the barrier that ended chunk cur - 1, so buffer (cur + 2) % 3 is free: start chunk
//              cur + 2 into it, return the buffer of chunk cur
//   release()  wait for this thread's pieces of chunk cur + 1 (those of chunk cur + 2 stay in flight), then barrier:
//              everybody's pieces of chunk cur + 1 are in LDS and chunk cur is retired
//   drain()    before the ring memory is reused for something else
// MFMA streams run at priority 0 and everything else at 2: epilogues, gathers and ring hand-offs are short and
// latency bound, and a wave stuck behind another workgroup's full-rate MFMA stream stalls its own workgroup.
template <int CH4>
struct ChunkRing {
    static constexpr int PT = CH4 / 256;
    static_assert(PT == 2 || PT == 4, "chunk size");
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
    __device__ __forceinline__ void start(F &&src) {
        fetch(src, 0, 0);
        fetch(src, 1, 1);
    }
    __device__ __forceinline__ void wait_landed() {
        if (PT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __device__ __forceinline__ void first() {
        wait_landed();
        lds_barrier();
    }
    template <class F>
    __device__ __forceinline__ const v4f *acquire(F &&src) {
        int b2 = cur + 2;
        b2 -= 3 * (b2 / 3);
        fetch(src, cur + 2, b2);
        __builtin_amdgcn_sched_barrier(0);      // keep the fetch at the head of the chunk
        __builtin_amdgcn_s_setprio(0);
        return ring + (cur - 3 * (cur / 3)) * CH4;
    }
    __device__ __forceinline__ void release() {
        __builtin_amdgcn_s_setprio(2);
        wait_landed();
        lds_barrier();
        ++cur;
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

}  // namespace kpd
