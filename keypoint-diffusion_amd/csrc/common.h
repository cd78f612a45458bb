// Shared host/device helpers for libkpd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/kpd.h"

namespace kpd {

void set_error(const char *fmt, ...);

#define KPD_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            kpd::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return KPD_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define KPD_REQUIRE(cond, code, ...)        \
    do {                                    \
        if (!(cond)) {                      \
            kpd::set_error(__VA_ARGS__);    \
            return (code);                  \
        }                                   \
    } while (0)

#define KPD_LAUNCH_CHECK() KPD_HIP(hipGetLastError())
#define KPD_TRY(expr)                  \
    do {                               \
        kpd_status s_ = (expr);        \
        if (s_ != KPD_OK) return s_;   \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// A/B, ablation and debugging switches are read from the environment only in the TOOLS build (`make tools`, -DKPD_TOOLS:
// profiles/tools/*.sh load it through KPD_LIB); in the product library every one of them is its default, a compile-time
// constant, and the variable names are not even in the binary (tests/test_abi.py checks that).  kpd_build_flags() tells the two apart.
#ifdef KPD_TOOLS
#include <stdlib.h>
static inline int tool_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define KPD_TOOL_SWITCH(expr, dflt) (expr)
#else
static constexpr int tool_env_int(const char *, int dflt) { return dflt; }
#define KPD_TOOL_SWITCH(expr, dflt) (dflt)
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for kernels that need more than 64 KB of dynamic LDS: applied once per
// (kernel, device), thread-safe; cheap enough to call before every launch.  pack.hip.
kpd_status ensure_dynamic_lds(const void *kernel, int bytes);
// Compute units of the current device (cached per device; 256 if the query fails).
int cu_count();

// Philox4x32-10 counter-based generator (per-complex sampler noise, dropout masks of the GVP training path)
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}


// ---- geometry of the fp32-MFMA row-tile kernels ------------------------------------------
constexpr int TM = 64;        // rows (edges or nodes) per workgroup tile
constexpr int HID = 256;      // hidden_nf
constexpr int HW = 257;       // hidden_nf + 1 (timestep column), the reference's layer width
constexpr int HS = 264;       // row stride of width-257 arrays in HBM (floats, multiple of 8)
constexpr int KP = 264;       // K of every GEMM, zero padded
constexpr int NG = KP / 8;    // k-groups of 8 (4 MFMA k-steps of 2)
constexpr int SA = 268;       // LDS row stride of the A / T tile (floats); 16-B aligned rows
constexpr int WP_FLOATS = NG * 4 * 64 * 8;   // packed 256-column weight block
constexpr int KL_KMAX = 16;   // largest supported kl_k
constexpr int KH_STEPS = 17;  // f16x2 mode: K = 264 as 17 k-steps of 16 (v_mfma_f32_32x32x16_f16), zero padded to 272
constexpr int WH_HALVES = KH_STEPS * 4 * 64 * 2 * 2 * 8;   // halves of one split 256-column block (hi + lo planes)
constexpr int CHH_HALVES = 8 * 16 * 2 * 64 * 8;            // halves of one split 256 x 256 projection block (k_proj_ws_h)

}  // namespace kpd
