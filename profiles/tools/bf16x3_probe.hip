// Probe: fp32-accurate tile GEMM on the bf16 MFMA by three-way operand splitting ("bf16x3").
//   x = hi + mid + lo, each a bf16 (8 mantissa bits, obtained by truncation so the split is exact to 24 bits);
//   a w ~= hi.hi + (hi.mid + mid.hi) + (hi.lo + lo.hi + mid.mid)     6 products, error ~2^-23 of |a||w|
//                                                                      3 products (first three), error ~2^-16
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32 per unit of K, so six products are 2.7x
// and three products 5.3x faster than the exact-fp32 MFMA, and (unlike the fp32 MFMA) leave 3/4 of the vector issue
// slots free for the splitting and the activations.  The tile is the edge kernel's: 64 rows x K = 272 from an fp32 LDS
// image, 256 output columns, 4 waves (2 x 2 tiles of 32 x 32 each), weights pre-split and pre-packed in B-fragment order
// and read from L2.  A fragments are split on the fly (every wave splits the whole A tile: 4x redundant, still hidden).
// Reports cycles per tile-GEMM and the error against an fp64 reference.
// Build: hipcc -O3 --offload-arch=gfx950 bf16x3_probe.hip -o bf16x3_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64, K = 272, KS = K / 16, N = 256, SA = 276;

__device__ __forceinline__ unsigned hi16(float x) { return __float_as_uint(x) & 0xffff0000u; }

// 8 fp32 -> three bf16x8 fragments (truncation split)
__device__ __forceinline__ void split8(const f32x4 &a, const f32x4 &b, bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    unsigned ph[4], pm[4], pl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned hh[2], mm[2], ll[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float v = x[2 * i + j];
            hh[j] = hi16(v);
            const float r1 = v - __uint_as_float(hh[j]);
            mm[j] = hi16(r1);
            const float r2 = r1 - __uint_as_float(mm[j]);
            ll[j] = hi16(r2);
        }
        ph[i] = (hh[0] >> 16) | hh[1];
        pm[i] = (mm[0] >> 16) | mm[1];
        pl[i] = (ll[0] >> 16) | ll[1];
    }
    u32x4 vh = {ph[0], ph[1], ph[2], ph[3]}, vm = {pm[0], pm[1], pm[2], pm[3]}, vl = {pl[0], pl[1], pl[2], pl[3]};
    h = __builtin_bit_cast(bf16x8, vh);
    m = __builtin_bit_cast(bf16x8, vm);
    l = __builtin_bit_cast(bf16x8, vl);
}

// Wp[plane][ks][wave][nt][lane] (16 B each): W[n = 64 wave + 32 nt + (lane & 31)][k = 16 ks + 8 (lane >> 5) + 0..7]
template <int NPROD, bool SPLIT = true>
__global__ __launch_bounds__(256, 2) void probe(const float *__restrict__ A, const u32x4 *__restrict__ Wp, float *__restrict__ C,
                                                int reps) {
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < TM * K; i += 256) sA[(i / K) * SA + (i % K)] = A[i];
    __syncthreads();
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const size_t plane = (size_t)KS * 4 * 2 * 64;
    for (int rep = 0; rep < reps; ++rep) {
        if (rep == reps - 1)        // the last repetition is the one that is checked
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
        // operands of k-step ks + 1 are fetched (B: L2, A: LDS) before the MFMAs of k-step ks are issued
        u32x4 bq[2][3], bn[2][3];
        f32x4 aq[2][2], an[2][2];
        auto fetch = [&](int ks, u32x4 (&bb)[2][3], f32x4 (&aa)[2][2]) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const u32x4 *bp = Wp + ((size_t)(ks * 4 + wave) * 2 + nt) * 64 + lane;
                bb[nt][0] = bp[0];
                bb[nt][1] = bp[plane];
                if (NPROD == 6) bb[nt][2] = bp[2 * plane];
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float *ap = sA + (32 * mt + r) * SA + 16 * ks + 8 * h;
                aa[mt][0] = *reinterpret_cast<const f32x4 *>(ap);
                aa[mt][1] = *reinterpret_cast<const f32x4 *>(ap + 4);
            }
        };
        // three-deep software pipeline: raw operands of k-step ks + 2 are fetched, those of ks + 1 are split on the VALU,
        // and the MFMAs of ks are issued -- interleaved one MFMA : five VALU instructions by sched_group_barrier, because
        // a bf16 MFMA occupies the vector issue port for 8 of its 32 cycles only
        fetch(0, bq, aq);
        bf16x8 ah[2], am[2], al[2], nh[2], nm[2], nl[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) split8(aq[mt][0], aq[mt][1], ah[mt], am[mt], al[mt]);
        u32x4 bc[2][3];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bc[nt][pl] = bq[nt][pl];
        fetch(1, bq, aq);
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
            fetch(ks + 2 < KS ? ks + 2 : KS - 1, bn, an);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                if (SPLIT) split8(aq[mt][0], aq[mt][1], nh[mt], nm[mt], nl[mt]);
                else { nh[mt] = ah[mt]; nm[mt] = am[mt]; nl[mt] = al[mt]; }     // calibration: MFMA + operand traffic only
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, bc[nt][0]), bm = __builtin_bit_cast(bf16x8, bc[nt][1]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, bc[nt][2]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x16 c = acc[mt][nt];
                    if (NPROD == 6) {       // smallest terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mt], bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl, c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mt], bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh, c, 0, 0, 0);
                    acc[mt][nt] = c;
                }
            }
#pragma unroll
            for (int i = 0; i < 4 * NPROD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);     // six VALU
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                ah[mt] = nh[mt]; am[mt] = nm[mt]; al[mt] = nl[mt];
                aq[mt][0] = an[mt][0];
                aq[mt][1] = an[mt][1];
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    bc[nt][pl] = bq[nt][pl];
                    bq[nt][pl] = bn[nt][pl];
                }
        }
    }
    if (blockIdx.x == 0)
        for (int mt = 0; mt < 2; ++mt)
            for (int nt = 0; nt < 2; ++nt)
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * h, col = 64 * wave + 32 * nt + r;
                    C[row * N + col] = acc[mt][nt][reg];
                }
}

static unsigned short trunc_bf16(float x, float *rest) {
    unsigned u;
    memcpy(&u, &x, 4);
    u &= 0xffff0000u;
    float hf;
    memcpy(&hf, &u, 4);
    *rest = x - hf;
    return (unsigned short)(u >> 16);
}

template <int NPROD, bool SPLIT = true>
static void run(const float *dA, const u32x4 *dW, float *dC, const std::vector<double> &ref, double refmax) {
    const int reps = 200, blocks = 512;
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<NPROD, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, TM * SA * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NPROD, SPLIT><<<blocks, 256, TM * SA * 4>>>(dA, dW, dC, reps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NPROD, SPLIT><<<blocks, 256, TM * SA * 4>>>(dA, dW, dC, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> C(TM * N);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double err = 0;
    for (size_t i = 0; i < C.size(); ++i) err = fmax(err, fabs((double)C[i] - ref[i]));
    // 2 workgroups per CU share a SIMD per wave: cycles per tile-GEMM per SIMD = time / (reps * 2 tiles)
    printf("%d products%s: %.3f ms, %.0f cycles per 64x272x256 tile-GEMM per SIMD at 2.4 GHz (fp32 MFMA floor 34816), max err / max |ref| = %.2e\n",
           NPROD, SPLIT ? "" : " (no splitting: MFMA + operand traffic only)", ms, ms * 1e-3 * 2.4e9 / (reps * 2.0), err / refmax);
}

int main() {
    std::vector<float> A(TM * K), W((size_t)N * K);
    srand(1);
    for (auto &v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto &v : W) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
    std::vector<double> ref((size_t)TM * N);
    double refmax = 0;
    for (int m = 0; m < TM; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * (double)W[(size_t)n * K + k];
            ref[(size_t)m * N + n] = s;
            refmax = fmax(refmax, fabs(s));
        }
    const size_t plane = (size_t)KS * 4 * 2 * 64;
    std::vector<unsigned short> Wp(3 * plane * 8);
    for (int ks = 0; ks < KS; ++ks)
        for (int wave = 0; wave < 4; ++wave)
            for (int nt = 0; nt < 2; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int n = 64 * wave + 32 * nt + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + i;
                        float r1, r2, r3;
                        const unsigned short hh = trunc_bf16(W[(size_t)n * K + k], &r1), mm = trunc_bf16(r1, &r2), ll = trunc_bf16(r2, &r3);
                        const size_t at = ((((size_t)ks * 4 + wave) * 2 + nt) * 64 + lane) * 8 + i;
                        Wp[at] = hh;
                        Wp[plane * 8 + at] = mm;
                        Wp[2 * plane * 8 + at] = ll;
                    }
    float *dA, *dC;
    u32x4 *dW;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dC, TM * N * 4); hipMalloc(&dW, Wp.size() * 2);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice);
    run<6>(dA, dW, dC, ref, refmax);
    run<3>(dA, dW, dC, ref, refmax);
    run<6, false>(dA, dW, dC, ref, refmax);
    return 0;
}
