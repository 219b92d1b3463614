// sobel_tile.h -- the LDS-tile Sobel core shared by sobel.hip (plane -> plane) and fused.hip
// (RGB -> luma -> Sobel).  gfx950 only.
//
// Reference arithmetic: monolithic/src/sobel.c:14-22
//   gx = -tl - 2l - bl + tr + 2r + br ;  gy = tl + 2t + tr - bl - 2b - br
//   dst = min(255, (int)sqrtf(gx*gx + gy*gy))
// Separable form used here (identical integers):
//   per input row:   h1[x] = p[x+1] - p[x-1]            h2[x] = p[x-1] + 2 p[x] + p[x+1]
//   per output row:  gx = h1[up] + 2 h1[mid] + h1[dn]   gy = h2[up] - h2[dn]
// All of it in packed 2 x i16 (v_pk_* VALU ops): |h1| <= 255, h2 <= 1020, |gx|,|gy| <= 1020.
//
// Tile: TW = 1024 px (64 lanes x 16 px, one dwordx4 per lane per row) by TH = 32 rows per 256-thread
// workgroup, staged in LDS with a one-pixel halo: LDS rows are [16 B left chunk | 1024 B | 16 B right
// chunk] so every global load and every LDS access stays 16-byte aligned; only byte 15 of the left
// chunk and byte 0 of the right chunk are consumed.  34 rows x 1056 B = 35,904 B per workgroup ->
// 4 workgroups (16 waves) per CU.  Each wave then walks 8 output rows with a 3-row rolling window of
// (h1,h2) in registers, so an input row is unpacked once per wave and each output pixel costs one
// LDS dwordx4 read amortised 10/8.
#pragma once

#include "gh_common.h"

namespace sobel_tile {

constexpr int BLOCK = 256;
constexpr int TW = 1024;               // output pixels per tile row
constexpr int TH = 32;                 // output rows per tile
constexpr int ROWS_PER_WAVE = 8;
constexpr int LDS_ROW = TW + 32;       // bytes per staged row
constexpr int LDS_ROWS = TH + 2;
constexpr int CHUNKS = LDS_ROW / 16;   // 66 16-byte chunks per staged row
constexpr int LDS_BYTES = LDS_ROWS * LDS_ROW;
constexpr int STAGE_ITERS = (LDS_ROWS * CHUNKS + BLOCK - 1) / BLOCK;   // 9

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }

// Tile order: plain row-major blockIdx.  An XCD-aware remap (each XCD gets one contiguous run of tiles, so that
// vertically adjacent tiles share an L2) was measured on MI355X and REMOVED: it made every row-strided kernel
// slower (strip copy 0.115 ms vs 0.095 ms on a 16384^2 plane) because it concentrates each XCD on a distant
// region instead of spreading concurrent requests over the HBM channels; halo re-reads are only 0.2 % of the
// traffic (profiles/r1_first/pmc_hbm_traffic.json), so there was nothing to gain.

// Horizontal stage of one staged row for this lane's 16 pixels.
// q[0..3] = the lane's 16 pixels, ql = dword holding pixel x-1 in its top byte, qr = dword holding
// pixel x+16 in its low byte.
__device__ __forceinline__ void hstage(const uint4 &qv, uint32_t ql, uint32_t qr, s16x2 (&h1)[8], s16x2 (&h2)[8])
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
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        h1[k] = o[k + 1] - o[k];
        h2[k] = o[k] + o[k + 1] + e[k] * (short)2;
    }
}

// Vertical stage + magnitude for 16 pixels -> 16 packed bytes.
__device__ __forceinline__ uint4 vstage(const s16x2 (&h1u)[8], const s16x2 (&h1m)[8], const s16x2 (&h1d)[8],
                                        const s16x2 (&h2u)[8], const s16x2 (&h2d)[8])
{
    uint32_t mag[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const s16x2 gx = h1u[k] + h1m[k] * (short)2 + h1d[k];
        const s16x2 gy = h2u[k] - h2d[k];
        const int gx0 = gx.x, gx1 = gx.y, gy0 = gy.x, gy1 = gy.y;
        mag[2 * k] = gh_sobel_mag(gx0 * gx0 + gy0 * gy0);
        mag[2 * k + 1] = gh_sobel_mag(gx1 * gx1 + gy1 * gy1);
    }
    uint4 r;
    r.x = mag[0] | (mag[1] << 8) | (mag[2] << 16) | (mag[3] << 24);
    r.y = mag[4] | (mag[5] << 8) | (mag[6] << 16) | (mag[7] << 24);
    r.z = mag[8] | (mag[9] << 8) | (mag[10] << 16) | (mag[11] << 24);
    r.w = mag[12] | (mag[13] << 8) | (mag[14] << 16) | (mag[15] << 24);
    return r;
}

// Compute phase: the tile (luma bytes, halo included) is in LDS; every wave produces its 8 rows.
//   x0,y0  : frame coordinates of the tile's first output pixel (y relative to the band)
//   w,rows : band geometry;  zero_top/zero_bot: band edge is the frame border (row written as 0)
__device__ __forceinline__ void compute_tile(const uint8_t *tile, uint8_t *__restrict__ dst, int w, int rows, int x0,
                                             int y0, int zero_top, int zero_bot)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x = x0 + 16 * lane;
    const int r0 = wave * ROWS_PER_WAVE;
    if (y0 + r0 >= rows) return;            // wave-uniform: nothing to produce (no barrier follows)
    const bool in_frame = x < w;
    const uint8_t *lp = tile + r0 * LDS_ROW + 16 + 16 * lane;

    s16x2 h1[3][8], h2[3][8];
#pragma unroll
    for (int i = 0; i < ROWS_PER_WAVE + 2; ++i) {
        const uint4 qv = *reinterpret_cast<const uint4 *>(lp + i * LDS_ROW);
        const uint32_t ql = *reinterpret_cast<const uint32_t *>(lp + i * LDS_ROW - 4);
        const uint32_t qr = *reinterpret_cast<const uint32_t *>(lp + i * LDS_ROW + 16);
        hstage(qv, ql, qr, h1[i % 3], h2[i % 3]);
        if (i >= 2) {
            const int y = y0 + r0 + i - 2;   // output row (window rows i-2, i-1, i)
            uint4 out = vstage(h1[(i - 2) % 3], h1[(i - 1) % 3], h1[i % 3], h2[(i - 2) % 3], h2[i % 3]);
            if (x == 0) out.x &= 0xffffff00u;            // frame column 0
            if (x + 15 == w - 1) out.w &= 0x00ffffffu;   // frame column w-1
            if ((y == 0 && zero_top) || (y == rows - 1 && zero_bot)) out = make_uint4(0, 0, 0, 0);
            if (in_frame && y < rows) *reinterpret_cast<uint4 *>(dst + (size_t)y * w + x) = out;
        }
    }
}

}  // namespace sobel_tile
