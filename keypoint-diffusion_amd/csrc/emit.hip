// Output side of sampling: element decode and XYZ text of a batch of sampled ligands, on the device.
// Replaces, for the tensor -> text part, sample.py:66-90 (argmax over the feature columns, index -> element symbol) and
// utils.py:11-21 (write_xyz_file: "<n>\n\n" then one "<el> <x:.3f> <y:.3f> <z:.3f>\n" line per atom).  The text is
// byte-identical to what Python's float formatting prints: "%.3f" of the exact value of the fp32 coordinate, rounded
// half-to-even on the exact binary value (integer arithmetic below, no floating-point rounding involved).
// Byte work, HBM-bound and tiny: three launches per batch (lines, ligand offsets, compaction).
#include "common.h"

namespace kpd {

constexpr int LINE_SLOT = 80;       // 2 symbol bytes + 3 x (space, sign, <= 16 integer digits, '.', 3 digits) + '\n' <= 72

// "%.3f" of an fp32 value into buf; returns the length, or -1 if |v| >= 2^53 (not printable with 64-bit integers here)
__device__ __forceinline__ int format_f3(float v, char *buf) {
    const unsigned bits = __float_as_uint(v);
    const bool neg = bits >> 31;
    const int e8 = (bits >> 23) & 0xff;
    const unsigned frac = bits & 0x7fffffu;
    int n = 0;
    if (e8 == 255) {                // Python: 'nan' without sign, 'inf' / '-inf'
        if (frac) {
            buf[0] = 'n'; buf[1] = 'a'; buf[2] = 'n';
            return 3;
        }
        if (neg) buf[n++] = '-';
        buf[n++] = 'i'; buf[n++] = 'n'; buf[n++] = 'f';
        return n;
    }
    const unsigned long long m = e8 ? (frac | 0x800000u) : frac;
    const int e = (e8 ? e8 : 1) - 150;              // value = m * 2^e
    const unsigned long long M = m * 1000ull;       // < 2^34
    unsigned long long N;                           // round_half_even(|v| * 1000)
    if (e >= 0) {
        if (e > 29) return -1;
        N = M << e;
    } else {
        const int s = -e;
        if (s >= 64) {
            N = 0;
        } else {
            const unsigned long long q = M >> s, r = M & ((1ull << s) - 1ull), half = 1ull << (s - 1);
            N = q + ((r > half || (r == half && (q & 1ull))) ? 1ull : 0ull);
        }
    }
    if (neg) buf[n++] = '-';                        // the sign survives rounding to zero ('-0.000'), as in Python
    unsigned long long ip = N / 1000ull;
    const unsigned fp = (unsigned)(N % 1000ull);
    char tmp[20];
    int nd = 0;
    do {
        tmp[nd++] = (char)('0' + (int)(ip % 10ull));
        ip /= 10ull;
    } while (ip);
    while (nd) buf[n++] = tmp[--nd];
    buf[n++] = '.';
    buf[n++] = (char)('0' + fp / 100);
    buf[n++] = (char)('0' + (fp / 10) % 10);
    buf[n++] = (char)('0' + fp % 10);
    return n;
}

// one thread per atom: argmax of the feature row (first maximum, torch.argmax on CPU), the atom's line into its slot
__global__ void k_emit_lines(const float *__restrict__ pos, const float *__restrict__ feat, int N, int F,
                             const unsigned *__restrict__ symbols, int *__restrict__ elem, char *__restrict__ slots,
                             int *__restrict__ len, int *__restrict__ status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *f = feat + (size_t)i * F;
    int best = 0;
    float bv = f[0];
    bool has_nan = bv != bv;
    for (int k = 1; k < F && !has_nan; ++k) {       // torch.argmax: first maximum; a NaN is a maximum
        const float x = f[k];
        if (x != x) {
            best = k;
            has_nan = true;
        } else if (x > bv) {
            bv = x;
            best = k;
        }
    }
    elem[i] = best;
    char line[LINE_SLOT];
    int n = 0;
    const unsigned sym = symbols[best];             // up to 4 bytes, NUL-padded, little-endian
    for (int b = 0; b < 4; ++b) {
        const char c = (char)((sym >> (8 * b)) & 0xff);
        if (!c) break;
        line[n++] = c;
    }
    for (int c = 0; c < 3; ++c) {
        line[n++] = ' ';
        const int w = format_f3(pos[(size_t)i * 3 + c], line + n);
        if (w < 0) {
            atomicOr(status, 1);
            line[n++] = '?';
        } else {
            n += w;
        }
    }
    line[n++] = '\n';
    len[i] = n;
    char *dst = slots + (size_t)i * LINE_SLOT;
    for (int b = 0; b < n; ++b) dst[b] = line[b];
}

__device__ __forceinline__ int header_len(int n) {
    int d = 1;
    for (int v = n; v >= 10; v /= 10) ++d;
    return d + 2;
}

// one wave per ligand: bytes of its block = header + lines
__global__ void k_emit_ligand_len(const int *__restrict__ lig_ptr, int B, const int *__restrict__ len, long long *__restrict__ lig_len) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int a0 = lig_ptr[b], a1 = lig_ptr[b + 1];
    long long s = 0;
    for (int i = a0 + lane; i < a1; i += 64) s += len[i];
    for (int off = 32; off; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) lig_len[b] = s + header_len(a1 - a0);
}

// single workgroup: exclusive scan of the ligand lengths -> text_ptr [B + 1]
__global__ void k_emit_scan(const long long *__restrict__ lig_len, int B, long long *__restrict__ text_ptr) {
    __shared__ long long part[256];
    __shared__ long long carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 256) {
        const int i = base + tid;
        const long long v = i < B ? lig_len[i] : 0;
        part[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const long long t = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += t;
            __syncthreads();
        }
        if (i < B) text_ptr[i] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0) text_ptr[B] = carry;
}

// one workgroup per ligand: header, then every line at its offset (in-block scan of the line lengths)
__global__ void k_emit_compact(const int *__restrict__ lig_ptr, const int *__restrict__ len, const char *__restrict__ slots,
                               const long long *__restrict__ text_ptr, long long capacity, char *__restrict__ text,
                               int *__restrict__ status) {
    __shared__ int part[256];
    __shared__ int carry;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int a0 = lig_ptr[b], a1 = lig_ptr[b + 1], n = a1 - a0;
    const long long t0 = text_ptr[b];
    if (text_ptr[b + 1] > capacity) {
        if (tid == 0) atomicOr(status, 2);
        return;
    }
    const int hl = header_len(n);
    if (tid == 0) {
        int v = n;
        for (int d = hl - 3; d >= 0; --d) {
            text[t0 + d] = (char)('0' + v % 10);
            v /= 10;
        }
        text[t0 + hl - 2] = '\n';
        text[t0 + hl - 1] = '\n';
        carry = hl;
    }
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        const int v = i < n ? len[a0 + i] : 0;
        part[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int t = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += t;
            __syncthreads();
        }
        if (i < n) {
            char *dst = text + t0 + carry + part[tid] - v;
            const char *src = slots + (size_t)(a0 + i) * LINE_SLOT;
            for (int k = 0; k < v; ++k) dst[k] = src[k];
        }
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
}

}  // namespace kpd

using namespace kpd;

static size_t emit_len_bytes(int n_atoms) { return ((size_t)n_atoms * 4 + 255) & ~(size_t)255; }

extern "C" int64_t kpd_xyz_scratch_bytes(int32_t n_atoms, int32_t B) {
    if (n_atoms < 0 || B < 0) return -1;
    return (int64_t)((size_t)n_atoms * LINE_SLOT + emit_len_bytes(n_atoms) + (size_t)B * 8 + 8);
}

extern "C" kpd_status kpd_xyz_emit(const float *pos, const float *feat, const int32_t *lig_ptr, int32_t n_atoms, int32_t B,
                                   int32_t F, const uint32_t *symbols, int32_t *elem, uint8_t *text, int64_t capacity,
                                   int64_t *text_ptr, int32_t *status, void *scratch, void *stream) {
    KPD_REQUIRE(n_atoms >= 0 && B >= 0 && F >= 1 && capacity >= 0, KPD_ERR_INVALID, "n_atoms=%d B=%d F=%d capacity=%lld", n_atoms, B,
                F, (long long)capacity);
    KPD_REQUIRE(lig_ptr && text_ptr && status && scratch, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(!n_atoms || (pos && feat && symbols && elem), KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(!capacity || text, KPD_ERR_INVALID, "null text buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    KPD_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    char *slots = static_cast<char *>(scratch);
    int *len = reinterpret_cast<int *>(slots + (size_t)n_atoms * LINE_SLOT);
    long long *lig_len = reinterpret_cast<long long *>(reinterpret_cast<char *>(len) + emit_len_bytes(n_atoms));
    if (n_atoms) {
        hipLaunchKernelGGL(k_emit_lines, dim3(cdiv(n_atoms, 256)), dim3(256), 0, st, pos, feat, n_atoms, F, symbols, elem, slots, len,
                           status);
        KPD_LAUNCH_CHECK();
    }
    if (B) {
        hipLaunchKernelGGL(k_emit_ligand_len, dim3(cdiv(B, 4)), dim3(256), 0, st, lig_ptr, B, len, lig_len);
        KPD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_emit_scan, dim3(1), dim3(256), 0, st, lig_len, B, reinterpret_cast<long long *>(text_ptr));
    KPD_LAUNCH_CHECK();
    if (B) {
        hipLaunchKernelGGL(k_emit_compact, dim3(B), dim3(256), 0, st, lig_ptr, len, slots,
                           reinterpret_cast<const long long *>(text_ptr), (long long)capacity, reinterpret_cast<char *>(text), status);
        KPD_LAUNCH_CHECK();
    }
    return KPD_OK;
}
