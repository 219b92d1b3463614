// stream_core.h -- the "strip streaming" Sobel cores for gfx950 (MI355X), shared by sobel.hip and fused.hip.
//
// Reference arithmetic: monolithic/src/sobel.c:14-22
//   gx = -tl - 2l - bl + tr + 2r + br ;  gy = tl + 2t + tr - bl - 2b - br ;  dst = min(255, (int)sqrtf(gx*gx + gy*gy))
//
// Geometry: one wavefront owns a strip 1024 px wide (64 lanes x 16 px = one dwordx4 per lane per row) and
// walks down R rows; the 4 waves of a workgroup sit side by side (4 KiB of every row).  The 3x3 reuse
// lives in registers as a rolling window, the one-pixel horizontal halo comes from the neighbour lanes
// (DPP wave_shr/wave_shl) and, for lanes 0 and 63, from a 1-byte global load.  No LDS, no barrier:
// measured on MI355X the LDS-tile form (sobel_tile.h) and this form are both bound by the same
// (row-strided) HBM access pattern, and this one needs fewer instructions per pixel (kernels).
//
// Separable form (identical integers to the reference):
//   per input row:   h1[x] = p[x+1] - p[x-1]               h2[x] = p[x-1] + 2 p[x] + p[x+1]
//   per output row:  gx = h1[y-1] + 2 h1[y] + h1[y+1]      gy = h2[y-1] - h2[y+1]
//   rolling state:   c[y] = h1[y-1] + h1[y]  ->  gx = c[y] + c[y+1]   (h1p, cp) ; h2 ring of two rows
//
// Two cores:
//   row16  packed 2 x i16 (v_pk_* ops), float tail under MODE.fp_round = toward-zero: 32 state VGPRs.
//          Used by the plane -> plane kernel, which contains no rounding-sensitive float op.
//   rowf   all-f32 (integers < 2^24 are exact); rounding-mode independent tail.  Used by the fused kernel,
//          whose luma arithmetic needs round-to-nearest.
#pragma once

#include "gh_common.h"

namespace stream_core {

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }

constexpr int STRIP_PX = 1024;          // pixels per wave per row
constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;

// a*2 + c on packed u16 as ONE v_pk_mad_u16 (hipcc would emit a shift and an add)
__device__ __forceinline__ s16x2 pk_mad2(s16x2 a, s16x2 c)
{
    uint32_t d;
    const uint32_t two = 0x00020002u;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(__builtin_bit_cast(uint32_t, a)), "s"(two), "v"(__builtin_bit_cast(uint32_t, c)));
    return as_s16x2(d);
}

// sqrt(gx^2 + gy^2 + 0.5) with the inner fma forced to one v_fma_f32 (hipcc would emit v_mov 0.5 + v_fmac).
// Exact inputs: |gx|,|gy| <= 1020, so gx^2 + gy^2 + 0.5 < 2^22 is exact in float32.
__device__ __forceinline__ float root_half(float fx, float fy, float half)
{
    float t;
    asm("v_fma_f32 %0, %1, %1, %2" : "=v"(t) : "v"(fy), "v"(half));
    return __builtin_amdgcn_sqrtf(__builtin_fmaf(fx, fx, t));
}

// Magnitude byte, variant RTZ: requires MODE.fp_round(f32) = toward zero.  v_sqrt_f32 is accurate to 1 ulp;
// sqrt(m + 0.5) is never within 9e-4 of an integer for results <= 256 (|sqrt(k^2 +- 0.5) - k| >= 0.25/k), so
// truncation gives floor(sqrt(m)) and v_cvt_pk_u8_f32 (which truncates under RTZ and saturates at 255)
// gives min(255, isqrt(m)).  gh_selftest_sobel_mag(0) checks every reachable (gx, gy) on the device.
__device__ __forceinline__ uint32_t mag_pack_rtz(float fx, float fy, float half, int byte, uint32_t acc)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(root_half(fx, fy, half), byte, acc);
}

// Variant RNE (default rounding): round-to-nearest of sqrt(m + 0.5) - 0.5 is floor(sqrt(m)) by the same
// margin argument (the value is at least 9e-4 away from k +- 0.5).  gh_selftest_sobel_mag(1) checks it.
__device__ __forceinline__ uint32_t mag_pack_rne(float fx, float fy, float half, int byte, uint32_t acc)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(root_half(fx, fy, half) - half, byte, acc);
}

// ---- packed-i16 core -----------------------------------------------------------------------------
struct State16 {
    s16x2 h1p[8], cp[8], h2r[2][8];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) { h1p[k] = as_s16x2(0u); cp[k] = as_s16x2(0u); h2r[0][k] = as_s16x2(0u); h2r[1][k] = as_s16x2(0u); }
    }
};

// One input row for this lane's 16 pixels: qv = the 16 pixels, ql holds pixel x-1 in its top byte, qr holds
// pixel x+16 in its low byte.  Returns the output row that this input row completes (window rows i-2, i-1, i);
// meaningful from the third row on.  PAR = row index parity (selects the h2 ring slot; must be static).
template <int PAR>
__device__ __forceinline__ uint4 row16(State16 &st, const uint4 &qv, uint32_t ql, uint32_t qr, float half)
{
    const uint32_t q[6] = {ql, qv.x, qv.y, qv.z, qv.w, qr};
    s16x2 o[9], e[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t cur = q[j + 1];
        e[2 * j] = as_s16x2(__builtin_amdgcn_perm(0u, cur, 0x0c010c00u));        // (p[4j],   p[4j+1])
        e[2 * j + 1] = as_s16x2(__builtin_amdgcn_perm(0u, cur, 0x0c030c02u));    // (p[4j+2], p[4j+3])
        o[2 * j] = as_s16x2(__builtin_amdgcn_perm(cur, q[j], 0x0c040c03u));      // (p[4j-1], p[4j])
        o[2 * j + 1] = as_s16x2(__builtin_amdgcn_perm(0u, cur, 0x0c020c01u));    // (p[4j+1], p[4j+2])
    }
    o[8] = as_s16x2(__builtin_amdgcn_perm(q[5], q[4], 0x0c040c03u));             // (p[15], p[16])
    uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const s16x2 h1 = o[k + 1] - o[k];
        const s16x2 h2 = pk_mad2(e[k], o[k] + o[k + 1]);
        const s16x2 c = st.h1p[k] + h1;
        const s16x2 gx = st.cp[k] + c;
        const s16x2 gy = st.h2r[PAR][k] - h2;
        st.h1p[k] = h1; st.cp[k] = c; st.h2r[PAR][k] = h2;
#pragma unroll
        for (int el = 0; el < 2; ++el) {
            const int p = 2 * k + el;
            const float fx = (float)(int)(el ? gx.y : gx.x), fy = (float)(int)(el ? gy.y : gy.x);
            r[p >> 2] = mag_pack_rtz(fx, fy, half, p & 3, r[p >> 2]);
        }
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

// ---- f32 core ------------------------------------------------------------------------------------
struct StateF {
    float h1p[16], cp[16], h2r[2][16];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) { h1p[k] = 0.f; cp[k] = 0.f; h2r[0][k] = 0.f; h2r[1][k] = 0.f; }
    }
};

// p[0..17] = luma at x-1 .. x+16 as integer-valued floats
template <int PAR>
__device__ __forceinline__ uint4 rowf(StateF &st, const float (&p)[18], float half)
{
    float s[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) s[k] = p[k] + p[k + 1];
    uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float h1 = p[k + 2] - p[k];
        const float h2 = s[k] + s[k + 1];
        const float c = st.h1p[k] + h1;
        const float gx = st.cp[k] + c;
        const float gy = st.h2r[PAR][k] - h2;
        st.h1p[k] = h1; st.cp[k] = c; st.h2r[PAR][k] = h2;
        r[k >> 2] = mag_pack_rne(gx, gy, half, k & 3, r[k >> 2]);
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}


// ---- packed-f32 core ("pair layout") --------------------------------------------------------------
// Measured on MI355X every wave64 VALU instruction costs ~4 cycles in these kernels, so instructions per
// pixel is what counts; v_pk_{add,mul,fma}_f32 do two pixels per instruction when both operands sit in aligned
// 64-bit register pairs.  Pairing pixel m with pixel m+8 of the lane's 16 makes every stencil neighbour pair
// aligned too:  G[m] = (p[m-1], p[m+7]) for m = 0..9  (p[-1] and p[16] are the strip halo), so
//   h1 pair k = G[k+2] - G[k],   h2 pair k = (G[k]+G[k+1]) + (G[k+1]+G[k+2])        (pixels k and k+8, k = 0..7)
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct StateF2 {
    f32x2 h1p[8], cp[8], h2r[2][8];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) { h1p[k] = f32x2{0.f, 0.f}; cp[k] = f32x2{0.f, 0.f}; h2r[0][k] = f32x2{0.f, 0.f}; h2r[1][k] = f32x2{0.f, 0.f}; }
    }
};

// RTZ = 1: the kernel runs with MODE.fp_round = toward zero (allowed only when it contains no rounding-sensitive
// float op), so v_cvt_pk_u8_f32 floors by itself and the "- 0.5" is not needed.
template <int PAR, int RTZ = 0, int PROBE_NOSQRT = 0>
__device__ __forceinline__ uint4 rowf2(StateF2 &st, const f32x2 (&G)[10], f32x2 half2)
{
    f32x2 s[9];
#pragma unroll
    for (int m = 0; m < 9; ++m) s[m] = G[m] + G[m + 1];
    uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x2 h1 = G[k + 2] - G[k];
        const f32x2 h2 = s[k] + s[k + 1];
        const f32x2 c = st.h1p[k] + h1;
        const f32x2 gx = st.cp[k] + c;
        const f32x2 gy = st.h2r[PAR][k] - h2;
        st.h1p[k] = h1; st.cp[k] = c; st.h2r[PAR][k] = h2;
        const f32x2 m = __builtin_elementwise_fma(gx, gx, __builtin_elementwise_fma(gy, gy, half2));   // exact: < 2^22
        f32x2 f;
        if (PROBE_NOSQRT) {              // timing probe only (wrong results): what do the 16 v_sqrt_f32 per row cost?
            f = m * half2;
        } else {
            f.x = __builtin_amdgcn_sqrtf(m.x);
            f.y = __builtin_amdgcn_sqrtf(m.y);
        }
        if (!RTZ) f = f - half2;                                            // see mag_pack_rne / mag_pack_rtz
        r[k >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(f.x, k & 3, r[k >> 2]);              // pixel k
        r[2 + (k >> 2)] = __builtin_amdgcn_cvt_pk_u8_f32(f.y, k & 3, r[2 + (k >> 2)]);  // pixel k + 8
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}


// ---- packed-i16 state, packed-f32 magnitude ("hybrid") --------------------------------------------------------
// row16's linear part (h1, h2, the rolling sums: 2 pixels per v_pk_*_i16, 32 state VGPRs per strip instead of the 64 of StateF2) with
// the magnitude of rowf2 (v_pk_fma_f32 on integer-valued floats, v_sqrt_f32, v_cvt_pk_u8_f32 under RTZ).  Per 16 pixels: 17 v_perm +
// 48 packed-i16 operations + 32 conversions + 16 v_pk_fma + 16 v_sqrt + 16 packs = 9.1 VALU instructions per pixel against 8.4 for the
// all-f32 pair core and 10 for row16 -- the price of fitting FOUR strips' state into a wave's registers (k_sobel_dma4).
template <int PAR>
__device__ __forceinline__ uint4 row16p(State16 &st, const uint4 &qv, uint32_t ql, uint32_t qr, f32x2 half2)
{
    const uint32_t q[6] = {ql, qv.x, qv.y, qv.z, qv.w, qr};
    s16x2 o[9], e[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t cur = q[j + 1];
        e[2 * j] = as_s16x2(__builtin_amdgcn_perm(0u, cur, 0x0c010c00u));        // (p[4j],   p[4j+1])
        e[2 * j + 1] = as_s16x2(__builtin_amdgcn_perm(0u, cur, 0x0c030c02u));    // (p[4j+2], p[4j+3])
        o[2 * j] = as_s16x2(__builtin_amdgcn_perm(cur, q[j], 0x0c040c03u));      // (p[4j-1], p[4j])
        o[2 * j + 1] = as_s16x2(__builtin_amdgcn_perm(0u, cur, 0x0c020c01u));    // (p[4j+1], p[4j+2])
    }
    o[8] = as_s16x2(__builtin_amdgcn_perm(q[5], q[4], 0x0c040c03u));             // (p[15], p[16])
    uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const s16x2 h1 = o[k + 1] - o[k];
        const s16x2 h2 = pk_mad2(e[k], o[k] + o[k + 1]);
        const s16x2 c = st.h1p[k] + h1;
        const s16x2 gx = st.cp[k] + c;
        const s16x2 gy = st.h2r[PAR][k] - h2;
        st.h1p[k] = h1; st.cp[k] = c; st.h2r[PAR][k] = h2;
        const f32x2 fx = {(float)(int)gx.x, (float)(int)gx.y}, fy = {(float)(int)gy.x, (float)(int)gy.y};    // pixels 2k, 2k + 1
        const f32x2 m = __builtin_elementwise_fma(fx, fx, __builtin_elementwise_fma(fy, fy, half2));          // exact: < 2^22
        r[k >> 1] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_sqrtf(m.x), (2 * k) & 3, r[k >> 1]);
        r[k >> 1] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_sqrtf(m.y), (2 * k + 1) & 3, r[k >> 1]);
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

// `if (rare wave-uniform condition) out = 0` as a scalar branch around four v_mov instead of four v_cndmask on every row: the empty asm
// keeps the optimiser from converting the branch into selects
__device__ __forceinline__ void zero_if(bool uniform_cond, uint4 &out)
{
    if (__builtin_expect(uniform_cond, 0)) {
        asm volatile("" ::: "memory");
        out = make_uint4(0, 0, 0, 0);
    }
}

// opaque(): stops the optimiser from proving a float is a small integer and rewriting the f32 adds as integer/SDWA ops
__device__ __forceinline__ float opaque(float v) { asm("" : "+v"(v)); return v; }

__device__ __forceinline__ f32x2 half2_reg()
{
    f32x2 h = {0.5f, 0.5f};
    asm volatile("" : "+v"(h));
    return h;
}

// ---- luma in packed f32 (fused kernel: RGB -> luma; plane Sobel of a later pass: grey -> luma) ---------------------------------------
// luma of two pixels as integer-valued floats (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32): trunc(y) == (float)(unsigned char)y for
// 0 <= y <= 255; same arithmetic as gh_luma (gh_common.h; monolithic/src/parallel_to_grayscale.c:13), minus the float -> int -> float
// round trip.  IEEE mode must not be contracted; the FMA mode uses explicit fused multiply-adds.  Needs MODE.fp_round = nearest-even.
template <int MODE>
__device__ __forceinline__ f32x2 luma_f2(f32x2 r, f32x2 g, f32x2 b)
{
#pragma clang fp contract(off)
    const f32x2 cr = {0.299f, 0.299f}, cg = {0.587f, 0.587f}, cb = {0.114f, 0.114f};
    f32x2 y;
    if (MODE == GH_MODE_FMA) {
        f32x2 t = cg * g;
        t = __builtin_elementwise_fma(cr, r, t);
        y = __builtin_elementwise_fma(cb, b, t);
    } else {
        const f32x2 s = cr * r + cg * g;
        y = s + cb * b;
    }
    y.x = opaque(__builtin_truncf(y.x));
    y.y = opaque(__builtin_truncf(y.y));
    return y;
}

// One strip row of a PLANE for the packed-f32 core: this lane's 16 pixels (q), the dword holding pixel x-1 in its top byte (ql) and the one
// holding pixel x+16 in its low byte (qr) -> the pair layout G[m] = (p[m-1], p[m+7]), m = 0..9, that rowf2 consumes.
// LUMA = 0: the bytes as they are (sobel_edge on a luma plane, monolithic/src/sobel.c:6-25).
// LUMA = 1 ("luma on load"): every byte e is replaced by lum(e, e, e) first -- what pass k + 1 of the reference's driver loop does to pass
// k's replicated edge image before it filters it (monolithic/src/main_with_sobel.c:54-62: convert_to_grayscale of (e, e, e), plane
// extract, sobel_edge).  lum(e, e, e) is not e: 13 grey levels come out one lower (IEEE: 37, 61, 74, ...; the FMA build moves another
// 13; tests hold both lists), so the step cannot be skipped -- but it can ride on the load instead of costing a 2 B/px pass of its own.
// It is a function of ONE byte, so it is a 256-entry table: `lut` points at 256 bytes in LDS which the workgroup copied from the device
// table that gh_luma itself filled (sobel.hip, k_fill_lum_lut), and a pixel costs one byte extract, one ds_read_u8 and one conversion --
// 9 VALU instructions per pixel for the whole row step against 12.4 with the five packed-f32 operations and the truncation of the
// arithmetic spelt out (measured first: 0.1194 ms at 16384^2 in IEEE mode, 0.1135 in FMA mode), no rounding-sensitive float operation
// in the kernel (MODE.fp_round stays toward-zero, rowf2's cheaper tail) and one code path for both arithmetic modes.
template <int LUMA>
__device__ __forceinline__ void plane_row_pairs(const uint4 &q, uint32_t ql, uint32_t qr, f32x2 (&G)[10], const uint8_t *lut)
{
    const uint32_t lo[2] = {q.x, q.y}, hi[2] = {q.z, q.w};          // own pixels 0..7 live in q.x, q.y and 8..15 in q.z, q.w
    if (LUMA) {
        uint32_t a[8], c[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) { a[m] = lut[gh_byte(lo[m >> 2], m & 3)]; c[m] = lut[gh_byte(hi[m >> 2], m & 3)]; }     // all 18 reads in flight together
        const uint32_t el = lut[ql >> 24], er = lut[qr & 0xffu];
#pragma unroll
        for (int m = 0; m < 8; ++m) G[m + 1] = f32x2{opaque((float)a[m]), opaque((float)c[m])};
        G[0] = f32x2{opaque((float)el), G[8].x};
        G[9] = f32x2{G[1].y, opaque((float)er)};
        return;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
        G[m + 1] = f32x2{opaque((float)gh_byte(lo[m >> 2], m & 3)), opaque((float)gh_byte(hi[m >> 2], m & 3))};
    G[0] = f32x2{opaque((float)(ql >> 24)), G[8].x};
    G[9] = f32x2{G[1].y, opaque((float)(qr & 0xffu))};
}

// the workgroup's copy of the 256-byte luma table (every thread of the 256 brings one byte; call before any wave leaves the kernel)
__device__ __forceinline__ void load_lum_lut(uint8_t *lut_lds, const uint8_t *lut_global)
{
    lut_lds[threadIdx.x & 255] = lut_global[threadIdx.x & 255];
    __syncthreads();
}

// streaming (read-once / write-once) 16-byte accesses: `nt` keeps these lines from displacing reusable data in L2
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_stream(const uint4 *p)
{
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
// Streaming stores.  POLICY 2 = nontemporal (what every kernel uses), 1 = sc1, 0 = plain.
// A trap worth a comment: measured in a loop that rewrites ONE 268 MB output plane, sc1 / plain stores look much faster for the
// kernels that write a single plane (RGB -> plane 88.6 % vs 78.9 % nt, extract 89 % vs 78 %, plane Sobel 74.7 % vs 66.7 %) --
// the 256 MB Infinity Cache absorbs the writes, which are overwritten before they ever reach HBM.  With the output rotating
// over six planes (1.6 GB) the order flips (RGB -> plane: sc1 77.4 %, nt 79.1 %; Sobel: sc1 62.6 %, nt 66.8 %) and the
// next kernel of a pipeline pays for the deferred write-backs (gray -> Sobel back to back: 0.88 instead of 0.99 Tpixel/s).
// Nontemporal stores perform the same in both settings: they are the honest, steady-state choice.
#ifndef GH_STREAM_STORE_POLICY
#define GH_STREAM_STORE_POLICY 2
#endif
template <int POLICY>
__device__ __forceinline__ void store_stream_p(uint4 *p, const uint4 v)
{
    u32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    if (POLICY == 2) __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(p));
    else if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
    else *reinterpret_cast<u32x4 *>(p) = t;
}
__device__ __forceinline__ void store_stream(uint4 *p, const uint4 v) { store_stream_p<GH_STREAM_STORE_POLICY>(p, v); }

// ---- wave-level RGB8 de-interleave through a private 3 KiB LDS slab ---------------------------------------
// A wavefront owns 1024 consecutive pixels = 3072 bytes = 192 dwordx4.  Loading 48 B per lane directly touches every
// 128-B line from three instructions (5.7 TB/s, and nontemporal makes it worse); instead the wave moves its 3 KiB with
// three FULLY COALESCED dwordx4 accesses (lane i: vectors i, 64+i, 128+i) and re-slices them through LDS so that lane i
// owns bytes 48 i .. 48 i + 47 (ds_read/ds_write_b128 at a 48-byte stride are conflict-free).  One wave, DS ops in
// order: the wave_barrier()s are scheduling fences only.
__device__ __forceinline__ void wave_load_rgb48(uint8_t *slab, const uint4 *p, int lane, uint32_t (&w)[12])
{
    uint4 *sw = reinterpret_cast<uint4 *>(slab);
    const uint4 v0 = load_stream(p + lane), v1 = load_stream(p + 64 + lane), v2 = load_stream(p + 128 + lane);
    sw[lane] = v0; sw[64 + lane] = v1; sw[128 + lane] = v2;
    __builtin_amdgcn_wave_barrier();
    const uint4 *sl = reinterpret_cast<const uint4 *>(slab + 48 * lane);
    const uint4 a = sl[0], b = sl[1], c = sl[2];
    __builtin_amdgcn_wave_barrier();
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
}

__device__ __forceinline__ void wave_store_rgb48(uint8_t *slab, uint4 *p, int lane, const uint32_t (&o)[12])
{
    uint4 *sl = reinterpret_cast<uint4 *>(slab + 48 * lane);
    sl[0] = make_uint4(o[0], o[1], o[2], o[3]);
    sl[1] = make_uint4(o[4], o[5], o[6], o[7]);
    sl[2] = make_uint4(o[8], o[9], o[10], o[11]);
    __builtin_amdgcn_wave_barrier();
    const uint4 *sw = reinterpret_cast<const uint4 *>(slab);
    const uint4 r0 = sw[lane], r1 = sw[64 + lane], r2 = sw[128 + lane];
    __builtin_amdgcn_wave_barrier();
    store_stream(p + lane, r0);
    store_stream(p + 64 + lane, r1);
    store_stream(p + 128 + lane, r2);
}

// Output stores of the strip kernels.  STORE_POLICY: 0 = default, 1 = sc1 buffer store, 2 = nontemporal (default).
// In the no-compute 3R:1W strip-copy probe (tools/ubench pattern31) default / sc1 stores beat nt stores by 8 %, but in the
// real kernels the order flips: fused 0.2006 ms (nt) vs 0.2107 (default) vs 0.2129 (sc1); conv 0.325 / 0.342 / 0.344;
// Sobel 0.105 / 0.101 / 0.103 -- so nt stays (build with -DGH_STRIP_STORE_POLICY=n to re-measure).
#ifndef GH_STRIP_STORE_POLICY
#define GH_STRIP_STORE_POLICY 2
#endif
template <int POLICY>
__device__ __forceinline__ void store_strip_p(uint8_t *wave_row /* wave-uniform: first byte of this wave's 1 KiB piece */, int lane, const uint4 v)
{
    if (POLICY == 1) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wave_row, 0, STRIP_PX, 0x00020000);
        u32x4 t;
        t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_amdgcn_raw_buffer_store_b128(t, rs, lane * 16, 0, 16 /* sc1 */);
    } else if (POLICY == 2) {
        store_stream(reinterpret_cast<uint4 *>(wave_row) + lane, v);
    } else {
        reinterpret_cast<uint4 *>(wave_row)[lane] = v;
    }
}
__device__ __forceinline__ void store_strip(uint8_t *wave_row, int lane, const uint4 v) { store_strip_p<GH_STRIP_STORE_POLICY>(wave_row, lane, v); }

// 0.5f kept in a VGPR (opaque to the optimiser) so the fma above can take it as a register operand
__device__ __forceinline__ float half_reg()
{
    float h = 0.5f;
    asm volatile("" : "+v"(h));
    return h;
}

// Row index to LOAD for band-relative row yy: rows outside the readable range are clamped onto a readable row
// instead of being zero-filled -- every output row that would see the substitute is a frame-border row and is
// written as 0 anyway, and unconditional loads need no per-row zero fill / predication.
__device__ __forceinline__ int clamp_row(int yy, int rows, int zero_top, int zero_bot)
{
    const int lo = zero_top ? 0 : -1, hi = zero_bot ? rows - 1 : rows;
    return yy < lo ? lo : (yy > hi ? hi : yy);
}

// band-relative row yy readable?  rows -1 and `rows` are the halo rows of a shard
__device__ __forceinline__ bool row_readable(int yy, int rows, int zero_top, int zero_bot)
{
    return (yy >= 0 && yy < rows) || (yy == -1 && !zero_top) || (yy == rows && !zero_bot);
}

// ---- LDS-DMA (global_load_lds_*) helpers of the ring kernels (fused.hip k_fused_dma, sobel.hip k_sobel_dma) -------------
// hipcc does not count asm memory operations: the kernels place their own s_waitcnt vmcnt(N) (wait_vm).  M0 carries the
// wave-uniform LDS destination; each lane's data lands at M0 + instruction offset + lane * size.
// LP: cache policy of the DMA loads (0 default, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt); nt measured best, as for register loads.
// `slot` (a constant once the row loop is unrolled) is the ring slot the row goes to.  It is printed into the assembly as a
// comment -- "; gh_dma slot=S" on every DMA instruction, "; gh_wait slot=S" on the wait that precedes the reads of slot S --
// so that tests/test_isa_vmcnt.py can check ON THE GENERATED CODE that on every path at least N vector-memory instructions
// are issued between the last DMA into slot S and a `vmcnt(N)` wait for it (vmcnt retires in order: with fewer than N younger
// operations the wait would return before the row has landed -- silent, timing-dependent corruption).
// One ring row = ONE asm statement: the wave's 1 KiB pieces (all 64 lanes) followed by the halo dword (EXEC narrowed to
// lanes 0 and 1 inside the statement, destination right behind the pieces).  A statement is never split or reordered
// against other volatile asm, so a row is always exactly `pieces + 1` vector-memory instructions in program order -- the
// quantity the waits are counted in.  (Every lane is active where rows are issued: the strip kernels branch wave-uniformly.)
#define GH_GLDS_HEAD  "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
#define GH_GLDS_HALO_OP(OPC) "s_mov_b64 %1, exec\n\ts_mov_b32 m0, %5\n\ts_mov_b64 exec, 3\n\t"  \
                      OPC " %6, %3 ; gh_dma slot=%7\n\t"                                           \
                      "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
#define GH_GLDS_HALO  GH_GLDS_HALO_OP("global_load_lds_dword")
// rows of frames whose pitch or base is not a multiple of 16 (see k_sobel_dma, UNAL): the piece starts at a 16-byte aligned
// address at or before the strip, and the two lanes of the last instruction fetch the one or two 16-byte blocks behind it
#define GH_GLDS_TAIL  GH_GLDS_HALO_OP("global_load_lds_dwordx4")
#define GH_GLDS_OPS   : "=&s"(keep), "=&s"(keepx)                                                    \
                      : "v"(voff), "s"(rowb), "s"(lds_slot), "s"(lds_halo), "v"(hoff), "i"(slot)    \
                      : "memory"
#define GH_PIECE(OFF, POL) "global_load_lds_dwordx4 %2, %3" OFF POL " ; gh_dma slot=%7\n\t"
// 3 KiB row (RGB8 strip): LP as above
template <int LP = 1>
__device__ __forceinline__ void glds_row3k_halo(uint32_t lds_slot, uint32_t lds_halo, const uint8_t *rowb, uint32_t voff, uint32_t hoff, int slot)
{
    uint32_t keep;
    uint64_t keepx;
#define GH_ROW3(POL) asm volatile(GH_GLDS_HEAD GH_PIECE("", POL) GH_PIECE(" offset:1024", POL) GH_PIECE(" offset:2048", POL) GH_GLDS_HALO GH_GLDS_OPS)
    if (LP == 0) GH_ROW3("");
    else if (LP == 2) GH_ROW3(" sc1");
    else if (LP == 3) GH_ROW3(" sc0 sc1");
    else if (LP == 4) GH_ROW3(" sc1 nt");
    else GH_ROW3(" nt");
#undef GH_ROW3
}
// 4 KiB row (RGBA8 strip)
__device__ __forceinline__ void glds_row4k_halo(uint32_t lds_slot, uint32_t lds_halo, const uint8_t *rowb, uint32_t voff, uint32_t hoff, int slot)
{
    uint32_t keep;
    uint64_t keepx;
    asm volatile(GH_GLDS_HEAD GH_PIECE("", " nt") GH_PIECE(" offset:1024", " nt") GH_PIECE(" offset:2048", " nt") GH_PIECE(" offset:3072", " nt")
                 GH_GLDS_HALO GH_GLDS_OPS);
}
// 1 KiB row (u8 plane strip)
template <int LP = 1>
__device__ __forceinline__ void glds_row1k_halo(uint32_t lds_slot, uint32_t lds_halo, const uint8_t *rowb, uint32_t voff, uint32_t hoff, int slot)
{
    uint32_t keep;
    uint64_t keepx;
    if (LP == 0) asm volatile(GH_GLDS_HEAD GH_PIECE("", "") GH_GLDS_HALO GH_GLDS_OPS);     // default policy: probe only
    else asm volatile(GH_GLDS_HEAD GH_PIECE("", " nt") GH_GLDS_HALO GH_GLDS_OPS);
}
// probe only (k_sobel_dma PROBE = 5): the 1 KiB piece WITHOUT the halo dword pair -- one vector-memory instruction per row
__device__ __forceinline__ void glds_row1k_nohalo(uint32_t lds_slot, uint32_t lds_halo, const uint8_t *rowb, uint32_t voff, uint32_t hoff, int slot)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt ; gh_dma slot=%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rowb), "s"(lds_slot), "i"(slot) : "memory");
    (void)lds_halo; (void)hoff;
}
// 2 KiB row (two neighbouring u8 plane strips owned by one wave: k_sobel_dma2)
__device__ __forceinline__ void glds_row2k_halo(uint32_t lds_slot, uint32_t lds_halo, const uint8_t *rowb, uint32_t voff, uint32_t hoff, int slot)
{
    uint32_t keep;
    uint64_t keepx;
    asm volatile(GH_GLDS_HEAD GH_PIECE("", " nt") GH_PIECE(" offset:1024", " nt") GH_GLDS_HALO GH_GLDS_OPS);
}
// the same rows with a 32-byte tail instead of the halo dword pair (operand names as above: lds_halo = where the tail lands,
// hoff = the two lanes' offsets of the tail blocks)
__device__ __forceinline__ void glds_row1k_tail(uint32_t lds_slot, uint32_t lds_halo, const uint8_t *rowb, uint32_t voff, uint32_t hoff, int slot)
{
    uint32_t keep;
    uint64_t keepx;
    asm volatile(GH_GLDS_HEAD GH_PIECE("", " nt") GH_GLDS_TAIL GH_GLDS_OPS);
}
__device__ __forceinline__ void glds_row2k_tail(uint32_t lds_slot, uint32_t lds_halo, const uint8_t *rowb, uint32_t voff, uint32_t hoff, int slot)
{
    uint32_t keep;
    uint64_t keepx;
    asm volatile(GH_GLDS_HEAD GH_PIECE("", " nt") GH_PIECE(" offset:1024", " nt") GH_GLDS_TAIL GH_GLDS_OPS);
}
__device__ __forceinline__ void glds_row3k_tail(uint32_t lds_slot, uint32_t lds_halo, const uint8_t *rowb, uint32_t voff, uint32_t hoff, int slot)
{
    uint32_t keep;
    uint64_t keepx;
    asm volatile(GH_GLDS_HEAD GH_PIECE("", " nt") GH_PIECE(" offset:1024", " nt") GH_PIECE(" offset:2048", " nt") GH_GLDS_TAIL GH_GLDS_OPS);
}
__device__ __forceinline__ void glds_row4k_tail(uint32_t lds_slot, uint32_t lds_halo, const uint8_t *rowb, uint32_t voff, uint32_t hoff, int slot)
{
    uint32_t keep;
    uint64_t keepx;
    asm volatile(GH_GLDS_HEAD GH_PIECE("", " nt") GH_PIECE(" offset:1024", " nt") GH_PIECE(" offset:2048", " nt") GH_PIECE(" offset:3072", " nt")
                 GH_GLDS_TAIL GH_GLDS_OPS);
}
#undef GH_PIECE
#undef GH_GLDS_OPS
#undef GH_GLDS_TAIL
#undef GH_GLDS_HALO
#undef GH_GLDS_HALO_OP
#undef GH_GLDS_HEAD

// Band plan of a strip kernel launch: level l has n[l] bands of R[l] rows starting at row y[l]; the last level runs to the
// end of the frame.  Workgroups are dispatched in blockIdx order, so the short bands of the last levels are the ones still
// running when the grid drains.
struct BandPlan {
    int n[2];
    int R[3];
    int y[3];
    int bands;      // total
    int order;      // workgroup -> (band, column group) mapping, see band_of_block(); 0 = band-major (default)
};
inline BandPlan make_band_plan(int rows, int R0, int n0, int R1, int n1, int R2)
{
    BandPlan p{};
    const int full0 = (rows + R0 - 1) / R0;
    if (n0 < 0 || n0 >= full0 || R1 <= 0) {        // single level
        p.n[0] = full0; p.n[1] = 0; p.R[0] = p.R[1] = p.R[2] = R0; p.y[0] = 0; p.y[1] = p.y[2] = full0 * R0; p.bands = full0;
        return p;
    }
    p.R[0] = R0; p.n[0] = n0; p.y[0] = 0; p.y[1] = n0 * R0; p.R[1] = R1;
    const int left1 = rows - p.y[1], full1 = (left1 + R1 - 1) / R1;
    if (n1 < 0 || n1 >= full1 || R2 <= 0) { p.n[1] = full1; p.R[2] = R1; p.y[2] = p.y[1] + full1 * R1; p.bands = n0 + full1; return p; }
    p.n[1] = n1; p.y[2] = p.y[1] + n1 * R1; p.R[2] = R2;
    const int left2 = rows - p.y[2];
    p.bands = n0 + n1 + (left2 + R2 - 1) / R2;
    return p;
}

// blockIdx -> (column group gx, band gy).  `order` packs the tuning parameters of the mapping (GH_FUSED_DMA_ORDER /
// GH_SOBEL_ORDER, decimal digits "ASCM"):
//   M (units)     0: column group fastest (a band's workgroups are dispatched together); 2: band fastest; 4: XCD-major --
//                 workgroup b belongs to XCD b % 8, and every XCD walks its own contiguous run of (band, column) pairs;
//                 6: band-per-XCD -- XCD x owns the bands k = x (mod 8) with all their column groups; 8: the same with
//                 pairs of neighbouring bands
//   C (tens)      column group rotated by C * band
//   S (hundreds)  bands visited with stride {1, 37, 3, 5, 7, 11, 13, 17, 19, 23}[S] (a permutation when co-prime to the count)
//   A (thousands) 1: band heights alternate R-1, R+1 (band starts alternate between even and odd rows); 2: odd bands walk
//                 upward (band_walks_up)
// Which mapping is fastest is decided by how the rows that are in flight at the same time spread over the memory
// channels; measured on MI355X: see DESIGN.md.
__device__ __forceinline__ void band_of_block(const BandPlan &plan, unsigned blk, int groups_x, int &gx, int &gy)
{
    const unsigned bands = (unsigned)plan.bands, total = bands * (unsigned)groups_x;
    const int M = plan.order % 10, C = (plan.order / 10) % 10, S = (plan.order / 100) % 10;
    if (M == 4) {                       // XCD-major: the workgroups of XCD x, in dispatch order, walk one contiguous run of the default order
        const unsigned per = (total + 7) / 8, x = blk % 8, i = blk / 8;
        blk = x * per + i;
        if (blk >= total) { gx = 0; gy = -1; return; }      // only when total is not a multiple of 8: surplus slots idle
    }
    if (M == 6) {                       // band-per-XCD: XCD x (= blk % 8) owns the bands k = x (mod 8), all column groups of each
        const unsigned x = blk % 8, i = blk / 8;
        gy = (int)(8 * (i / (unsigned)groups_x) + x);
        gx = (int)(i % (unsigned)groups_x);
        if ((unsigned)gy >= bands) gy = -1;
    } else if (M == 8) {                // pairs-per-XCD: XCD x owns the band pairs p = x (mod 8), i.e. bands 2p and 2p + 1
        const unsigned x = blk % 8, i = blk / 8, bi = i / (unsigned)groups_x;
        gy = (int)(2 * (8 * (bi >> 1) + x) + (bi & 1));
        gx = (int)(i % (unsigned)groups_x);
        if ((unsigned)gy >= bands) gy = -1;
    } else if (M == 2) { gy = (int)(blk % bands); gx = (int)(blk / bands); }
    else { gx = (int)(blk % (unsigned)groups_x); gy = (int)(blk / (unsigned)groups_x); }
    if (C) gx = (gx + C * gy) % groups_x;
    if (S) {
        const unsigned strides[10] = {1, 37, 3, 5, 7, 11, 13, 17, 19, 23};
        gy = (int)(((unsigned long long)gy * strides[S]) % bands);
    }
}

// Walk direction of a band (order thousands digit A = 2): odd bands walk UPWARD, so that the two bands of a pair meet at
// their common boundary at the end of their lives and the shared halo rows are read by both at about the same time.
// The rolling Sobel window is symmetric under the reversal (gy changes sign, and is squared).
__device__ __forceinline__ bool band_walks_up(const BandPlan &plan, int gy) { return (plan.order / 1000) % 10 == 2 && (gy & 1); }

// ---- band height and channel aliasing (host side) ----------------------------------------------------------------------------------
// The bands of a strip kernel start R * pitch bytes apart, and the ~256 row streams in flight advance together.  Measured on MI355X
// (tools/sweep.py r2_bg: plane Sobel, R = 64 against R = 16): where the band starts take only a FEW distinct values modulo 1 MiB the
// streams pile onto the same channels -- 12288-wide planes (64 rows = 768 KiB: 4 values) 0.0872 vs 0.0753 ms, 24576 wide (1.5 MiB: 2) 0.0910
// vs 0.0755, 20480 wide (1.25 MiB: 4) 0.0785 vs 0.0633, 10240 wide (640 KiB: 8) 0.0864 vs 0.0681, 12288^2 0.0805 vs 0.0570 -- while ONE value
// (16384- and 32768-wide planes: whole MiB, the address hash spreads the streams by their high bits) and MANY values (16368 / 16370 wide)
// are both fine, and there the taller band wins on halo rows.  The same arithmetic explains why 16-row bands lose on 16384-wide planes
// (256 KiB: 4 values).  band_start_phases() counts the values; prefer_short_bands() is the rule drawn from it.
inline int band_start_phases(long long stride_bytes)
{
    const long long W = 1ll << 20;
    long long a = stride_bytes % W, b = W;
    if (a == 0) return 1;
    while (b) { const long long t = a % b; a = b; b = t; }
    return (int)(W / a);
}
// true: the tall band height aliases (2 .. 15 distinct band-start values) and the short one does better (one value, or more values)
inline bool prefer_short_bands(long long pitch_bytes, int tall, int shrt)
{
    const int pt = band_start_phases((long long)tall * pitch_bytes), ps = band_start_phases((long long)shrt * pitch_bytes);
    return pt > 1 && pt < 16 && (ps == 1 || ps > pt);
}

// workgroups / (CUs x rounds needed): 1.0 = every CU gets the same number of workgroups (matters below ~2 workgroups per CU)
inline double wg_balance(long long nb, int num_cu)
{
    const long long rounds = (nb + num_cu - 1) / num_cu;
    return rounds ? (double)nb / (double)(rounds * num_cu) : 0.0;
}

// Regular-band kernels shift a last band that would be ragged UP, i.e. they launch one band row more than the frame has whole bands.  Every
// CU sustains only its share of the HBM rate (a 64-row band of two strips keeps one CU busy for the whole kernel), so the extra
// workgroups are free only while they do not add a LAYER of workgroups to some CUs: 16384 x 8200 with two strips per wave, 258 workgroups
// on 256 CUs, 0.0817 ms against 0.0532; 4096 x 4100, 257 workgroups, 0.0141 against 0.0131.  With many layers one more hardly shows.
inline bool shifted_band_overflows(long long groups_x, int rows, int R, int /*wg_per_cu*/, int num_cu)
{
    if (rows % R == 0) return false;
    const long long n_full = groups_x * (rows / R), n_all = n_full + groups_x;
    const long long layers_full = (n_full + num_cu - 1) / num_cu, layers_all = (n_all + num_cu - 1) / num_cu;
    return layers_all > layers_full && layers_full < 8;
}

// ---- band height of the ring kernels: pure functions of the geometry and the device's CU count (fused.hip / sobel.hip call them with
// gh_num_cu(); tests/native/launch_rules_dump.hip prints them for 256 and 32 CUs) --------------------------------------------------------
// Fused kernel: tall bands re-read fewer halo rows, but between one and two workgroups per CU they only pay when the workgroups spread
// evenly over the CUs (16384 x 4096: 256 workgroups of 64-row bands, one per CU, 0.0495 vs 0.0555 ms with 16-row bands; 12288 x 8192:
// 384 = 1.5 per CU, 0.0887 vs 0.0813 with 32-row and 0.0771 with 16-row bands; 10240^2: 480, 0.0887 vs 0.0840 with 32) -- "balance" =
// workgroups / (CUs x rounds needed); band starts that alias onto a few channel phases are demoted (prefer_short_bands: 10240 x 16384
// 0.1377 ms with 64-row bands, 0.1298 with 32; 12288 x 8192 0.0813 with 32, 0.0771 with 16).
// beside_exchange: the launch is the shard kernel of a sharded step in its halo-flag form (an RCCL exchange runs beside it, its edge
// bands wait for their halo row).  There 32-row bands that give every CU exactly one workgroup beat the 16-row bands the kernel alone
// prefers: 16384 x 2048 (the shard of eight GPUs) 32.6-32.9 us per step against 39.2 (tools/sweep.py shard_step_heights) -- an edge band of
// 32 rows reaches its halo row when the kernel is nearly over, and 256 fat workgroups leave the exchange's kernel more room than 512
// thin ones -- while the kernel alone measures the same either way (29.3 vs 29.8 us).
inline int fused_band_rows(int groups_x, int rows, long long pitch_bytes, int num_cu, bool alias_rule, bool beside_exchange = false)
{
    const long long nb64 = (long long)groups_x * ((rows + 63) / 64), nb32 = (long long)groups_x * ((rows + 31) / 32);
    if (beside_exchange && nb64 < num_cu && nb32 >= num_cu && nb32 < 2ll * num_cu && wg_balance(nb32, num_cu) >= 0.97) return 32;
    int rd = (nb64 >= 2ll * num_cu || (nb64 >= num_cu && wg_balance(nb64, num_cu) >= 0.97)) ? 64
             : ((nb32 >= 2ll * num_cu && wg_balance(nb32, num_cu) >= 0.9) ? 32 : 16);
    if (alias_rule) {
        if (rd == 64 && prefer_short_bands(pitch_bytes, 64, 32) && nb32 >= num_cu) rd = 32;
        if (rd == 64 && prefer_short_bands(pitch_bytes, 64, 16)) rd = 16;
        if (rd == 32 && prefer_short_bands(pitch_bytes, 32, 16)) rd = 16;
    }
    return rd;
}
// Plane Sobel, one strip per wave: 64-row bands from two workgroups per CU up (16384-wide planes: 66-67 % of 8 TB/s from 8192 rows up
// against 59-61 % for 32 / 16), else 32, else 16; an aliasing 64 is demoted to 16; where the branch-free regular form of 16-row bands
// applies it beats the 32-row choice (12288 x 8192 0.0380 vs 0.0460 ms, 16384 x 4096 0.0301 vs 0.0309).
inline int sobel_band_rows(int groups_x, int rows, long long pitch_bytes, int num_cu, bool ring_kernel, bool alias_rule, bool regular_form)
{
    const long long nb64 = (long long)groups_x * ((rows + 63) / 64), nb32 = (long long)groups_x * ((rows + 31) / 32);
    int R = nb64 >= 2ll * num_cu ? 64 : (nb32 >= 2ll * num_cu ? 32 : 16);
    if (R == 64 && ring_kernel && alias_rule && prefer_short_bands(pitch_bytes, 64, 16)) R = 16;
    if (R == 32 && ring_kernel && rows >= 16 && regular_form && !shifted_band_overflows(groups_x, rows, 16, 3, num_cu)) R = 16;
    return R;
}

// host side: workgroups to launch for a plan (the band-per-XCD order pads the band count to a multiple of 8: surplus
// workgroups find gy < 0 and exit)
inline long long plan_grid(const BandPlan &plan, int groups_x)
{
    const int M = plan.order % 10;
    const long long bands = M == 6 ? ((plan.bands + 7) / 8) * 8ll : (M == 8 ? ((plan.bands + 15) / 16) * 16ll : plan.bands);
    return bands * groups_x;
}

// host side: drop the parts of a mapping request that would not be a permutation of the (band, column) pairs -- and the XCD-interleaved
// mappings (M = 4, 6, 8: "workgroup b runs on XCD b % 8") on a device that does not have the eight XCDs they are written for (a CPX
// partition has one: the interleave would only scatter its bands and pad its grids to multiples of 8 / 16)
inline int sanitize_order(int order, int bands, int groups_x, int num_xcd = GH_REF_NUM_XCD)
{
    if (order <= 0) return 0;
    int M = order % 10, C = (order / 10) % 10, S = (order / 100) % 10, A = (order / 1000) % 10;
    const unsigned strides[10] = {1, 37, 3, 5, 7, 11, 13, 17, 19, 23};
    if (M != 0 && M != 2 && M != 4 && M != 6 && M != 8) M = 0;
    if (num_xcd != GH_REF_NUM_XCD && (M == 4 || M == 6 || M == 8)) M = 0;
    if (M == 4 && ((long long)bands * groups_x) % 8 != 0) M = 0;
    if (S && (unsigned)bands % strides[S] == 0) S = 0;
    if (bands < 2 || A > 2) A = 0;          // A = 1: alternating heights; A = 2: odd bands walk upward
    return A * 1000 + S * 100 + C * 10 + M;
}

// wait until at most N vector-memory operations of this wave are outstanding; `slot` = the ring slot whose row the
// following LDS reads consume (see the gh_dma / gh_wait comment markers above)
template <int N>
__device__ __forceinline__ void wait_vm(int slot)
{
    asm volatile("s_waitcnt vmcnt(%0) ; gh_wait slot=%1" ::"n"(N), "i"(slot) : "memory");
}

// wave-uniform 64-bit pointer / 32-bit value into SGPRs (inline asm "s" operands inside divergent code need it spelled out)
__device__ __forceinline__ const uint8_t *uniform_ptr(const uint8_t *p)
{
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return reinterpret_cast<const uint8_t *>(
        (uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)a)));
}
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

}  // namespace stream_core
