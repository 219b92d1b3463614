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
int gh_sobel_init_tables(int device);   // sobel.hip: the device's lum(v, v, v) table of the luma-on-load kernels (called by gh_init)

// which device kernel form the last launch of this thread took (gh_last_kernel(): tests and bench.py assert that the headline shapes run
// the kernels DESIGN.md says they run, instead of silently falling to a slower form)
void gh_note_kernel(const char *name);

// ---- halo-flag form of the sharded steps (comm.hip) ---------------------------------------------------------------------------
// The one-row exchange of a shard runs on the communicator's side stream WHILE one launch covers the whole shard on the caller's
// stream, and the two sides meet through two dwords in device memory instead of stream events (a cross-queue event hand-over costs
// ~20 us on this runtime; a polled dword ~2):
//   started  the shard kernel's first workgroup stores the step's epoch there when it begins -- by stream order every earlier piece of
//            work on the caller's stream has completed then (the rows to send are in place, the previous step has read its halo rows);
//            the side stream polls it in front of the exchange;
//   flag     set to the exchange's number behind the ncclRecv; only the workgroups of the first / last band, whose rows depend on a halo row, poll
//            it before they issue their first load; every other workgroup starts at once.
// A wait that outlasts `ticks` of the 100 MHz constant clock gives up, counts itself in *timeouts (host-mapped memory: the next sharded
// call sees it without a synchronisation and fails) and lets the wave finish -- every wave reaches its exit whatever happens to the
// exchange.  flag == nullptr: no wait (every other caller of the kernels).
struct GhHaloWait {
    const uint32_t *flag;
    uint32_t *started;
    uint32_t *timeouts;           // host-mapped dwords: [0] waits that gave up; [2] / [4]: copies of the standstill counters (flag + 40 / + 41)
    uint32_t epoch;               // what the kernel's first workgroup stores into `started`
    uint32_t wait_epoch;          // what `flag` must have reached before an edge band may request its halo row
    uint32_t pad;                 // host side: 1 = "would this frame run a kernel form that carries the wait?" (GH_OK / GH_NO_FLAG_FORM, nothing launched)
    uint32_t pad2;
    unsigned long long ticks;
};
constexpr int GH_NO_FLAG_FORM = -100;      // internal: this frame shape runs a kernel form without the flag wait (narrow frames, per-pixel kernels)
int gh_fused_rows_halo_wait(const uint8_t *rgb, uint8_t *edge, int width, int rows, int channels, int mode, int top_is_border,
                            int bottom_is_border, gh_stream_t stream, const GhHaloWait *hw);
// luma: 0 = the plane as it is; 1 / 2 = every byte through lum(e, e, e) on load, IEEE / FMA arithmetic (pass k >= 2 of the driver loop)
int gh_sobel_rows_halo_wait(const uint8_t *src, uint8_t *dst, int width, int rows, int top_is_border, int bottom_is_border, gh_stream_t stream,
                            const GhHaloWait *hw, int luma);

static inline bool gh_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Topology the launch rules work with: the CU and XCD counts of the initialised device (gh_init reads them; shim.hip).  The rules were
// measured on an MI355X in SPX mode -- 256 CUs in 8 XCDs, the values before initialisation -- and are written in multiples of the CU count,
// so a partitioned device (CPX: 32 CUs, one XCD) gets plans sized for what it has.  GH_NUM_CU_OVERRIDE (environment, probe builds): pretend
// another CU count (launch-rule experiments).  The workgroup -> band mappings that interleave over XCDs (band_of_block) assume 8 dispatch queues;
// with another XCD count they are still permutations -- every band is processed exactly once -- only the locality they aim at is lost.
constexpr int GH_REF_NUM_CU = 256, GH_REF_NUM_XCD = 8;
int gh_num_cu();
int gh_num_xcd();

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

// the wait itself (wave-uniform: every lane polls the same dword; acquire at system scope, so that the halo rows -- written by the
// RCCL kernel of this device or, over xGMI, by the neighbour's -- are re-fetched behind it instead of served from this XCD's L2)
// GH_HALO_TRACE builds (make probes): 100 MHz timestamps of the handshake in the 8 u64 behind the flag's cache line (flag + 16 dwords):
// [0] shard kernel's first workgroup began, [1] side stream saw `started`, [2] flag set, [3] / [4] an edge band began / ended its wait
#ifdef GH_HALO_TRACE
#define GH_HALO_TRACE_AT(flagp, k) \
    do { if ((threadIdx.x & 63) == 0) reinterpret_cast<unsigned long long *>(const_cast<uint32_t *>(flagp) + 16)[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GH_HALO_TRACE_AT(flagp, k) do { } while (0)
#endif
__device__ __forceinline__ void gh_halo_started(const GhHaloWait &hw)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(hw.started, hw.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        GH_HALO_TRACE_AT(hw.flag, 0);
    }
}
__device__ __forceinline__ void gh_halo_wait(const GhHaloWait &hw)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    GH_HALO_TRACE_AT(hw.flag, 3);
    bool stood_still = false;
    for (;;) {
        const uint32_t v = __hip_atomic_load(hw.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(v - hw.wait_epoch) >= 0) {
            GH_HALO_TRACE_AT(hw.flag, 4);
            // A band that had to stand still says for how long (wave 0 of its workgroup): two counters in DEVICE memory behind the flag --
            // dword 40 the 100 MHz ticks (modulo 2^32: the host looks at differences), dword 41 the number of such waits -- which the
            // flag kernel of the next exchange copies into host-mapped words (atomics straight into host memory cost the step 15 us).
            // A few microseconds are the normal case of a small shard; tens of them mean the exchange is not running beside the launch --
            // the communicator then goes back to the serial form (comm.hip, halo_flag_review).
            if (stood_still && threadIdx.x == 0) {
                uint32_t *stats = const_cast<uint32_t *>(hw.flag) + 40;
                const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
                // (one wait counts for at most 100 us: the first exchange of a communicator takes milliseconds -- RCCL sets its channels up --
                //  and must not look like fifty slow steps)
                __hip_atomic_fetch_add(stats, dt > 10000ull ? 10000u : (uint32_t)dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(stats + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > hw.ticks) {
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(hw.timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        stood_still = true;
        __builtin_amdgcn_s_sleep(8);
    }
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


// Environment knobs, two kinds.
//   GH_KNOB_INT  the few run-time settings of the RELEASE library (INTEGRATION.md lists them): read once per process.
//   GH_LAB_INT   tuning and probe knobs of the measurement lab: they exist only in builds with -DGH_ENABLE_PROBES (`make probes` ->
//                build_probes/libgrayhip_probes.so, what tools/ and the variant-sweeping tests load); in the release library the macro IS its
//                default, a compile-time constant, and the kernel variants those knobs select are not compiled in.  With GH_TUNE_DYNAMIC
//                set a probe build re-reads every knob on each launch, so that an A/B tool can interleave variants inside one process.
#include <cstdlib>
inline int gh_env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
#ifdef GH_ENABLE_PROBES
#define GH_LAB 1
inline bool gh_tune_dynamic()
{
    static const bool d = getenv("GH_TUNE_DYNAMIC") != nullptr;
    return d;
}
#define GH_LAB_INT(name, dflt) ([] { static const int v_ = gh_env_int(name, dflt); return gh_tune_dynamic() ? gh_env_int(name, dflt) : v_; }())
#define GH_KNOB_INT(name, dflt) GH_LAB_INT(name, dflt)
#else
#define GH_LAB 0
#define GH_LAB_INT(name, dflt) (dflt)
#define GH_KNOB_INT(name, dflt) ([] { static const int v_ = gh_env_int(name, dflt); return v_; }())
#endif

#endif  // __HIPCC__
