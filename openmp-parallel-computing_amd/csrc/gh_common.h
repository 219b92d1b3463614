// gh_common.h -- shared host/device helpers of libgrayhip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "grayhip.h"

// ---- error plumbing (thread-local message, see shim.hip) ---------------------------------------
int gh_set_error(int status, const char *fmt, ...);

#define GH_HIP_TRY(expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return gh_set_error(GH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                __FILE__, __LINE__);                                           \
    } while (0)

#define GH_LAUNCH_CHECK()                                                                      \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess)                                                                  \
            return gh_set_error(GH_ERR_HIP, "kernel launch failed: %s (%s:%d)",                \
                                hipGetErrorString(e_), __FILE__, __LINE__);                    \
    } while (0)

#define GH_REQUIRE(cond, msg)                                                                  \
    do {                                                                                       \
        if (!(cond)) return gh_set_error(GH_ERR_ARG, "%s: %s", __func__, msg);                 \
    } while (0)

int gh_ensure_init();   // lazy gh_init(GRAYHIP_DEVICE or 0)

// which device kernel form the last launch of this thread took (gh_last_kernel(): tests and bench.py assert that the headline shapes run
// the kernels DESIGN.md says they run, instead of silently falling to a slower form)
void gh_note_kernel(const char *name);

static inline bool gh_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// MI355X: 256 CUs in 8 XCDs
constexpr int GH_NUM_CU = 256;
constexpr int GH_NUM_XCD = 8;

// ---- device-side arithmetic shared by every kernel ---------------------------------------------
#if defined(__HIPCC__)

// Luma of one pixel.  Reference: monolithic/src/parallel_to_grayscale.c:13
//   (unsigned char)(0.299f*r + 0.587f*g + 0.114f*b)
// IEEE mode: each product and each sum rounded to float32 separately (no contraction) -- hipcc
// contracts a*b+c by default, so contraction is switched off for this function.
// FMA mode: mul, fma, fma in the order gcc 11.4 emits under -ffast-math -march=native.
// The truncating float->uint conversion is v_cvt_u32_f32 (round toward zero), values <= 255.
template <int MODE>
__device__ __forceinline__ uint32_t gh_luma(uint32_t r, uint32_t g, uint32_t b)
{
#pragma clang fp contract(off)
    const float fr = (float)r, fg = (float)g, fb = (float)b;
    float y;
    if (MODE == GH_MODE_FMA) {
        float t = 0.587f * fg;
        t = __builtin_fmaf(0.299f, fr, t);
        y = __builtin_fmaf(0.114f, fb, t);
    } else {
        const float pr = 0.299f * fr;
        const float pg = 0.587f * fg;
        const float pb = 0.114f * fb;
        float s = pr + pg;
        y = s + pb;
    }
    return (uint32_t)y;
}

// byte k (0..3) of a dword, as an unsigned int (lowers to v_cvt_f32_ubyteK / v_bfe_u32)
__device__ __forceinline__ uint32_t gh_byte(uint32_t w, int k) { return (w >> (8 * k)) & 0xffu; }

// Sobel magnitude from the squared gradient norm m = gx*gx + gy*gy (0 <= m <= 2*1020^2).
// Reference: monolithic/src/sobel.c:20-22   mag=(int)sqrtf((float)m); if (mag>255) mag=255;
// which equals min(255, isqrt(m)).  v_sqrt_f32 is accurate to 1 ulp, not correctly rounded, so a
// perfect square could come out one ulp low and truncate to k-1.  sqrt(m + 0.5) is never within
// 9e-4 of an integer for results <= 256 (|sqrt(k*k +- 0.5) - k| >= 0.25/k), far more than 1 ulp
// (3e-5 at 256), and m + 0.5 is exact in float32 (m < 2^21), so truncation is safe; larger
// results are clamped anyway.  gh_selftest_sobel_sqrt() checks every reachable m on the device.
__device__ __forceinline__ uint32_t gh_sobel_mag(int m)
{
    const float f = __builtin_amdgcn_sqrtf((float)m + 0.5f);
    const uint32_t r = (uint32_t)f;
    return r > 255u ? 255u : r;
}


// Tuning knobs (environment variables): read once per process; with GH_TUNE_DYNAMIC set they are re-read on every
// launch so that an A/B tool (tools/ab_fused.py) can interleave variants inside one process.
#include <cstdlib>
inline int gh_env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
inline bool gh_tune_dynamic()
{
    static const bool d = getenv("GH_TUNE_DYNAMIC") != nullptr;
    return d;
}
#define GH_TUNE_INT(name, dflt) ([] { static const int v_ = gh_env_int(name, dflt); return gh_tune_dynamic() ? gh_env_int(name, dflt) : v_; }())

#endif  // __HIPCC__
