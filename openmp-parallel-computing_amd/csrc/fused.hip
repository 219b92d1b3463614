// fused.hip -- RGB8/RGBA8 -> luma -> 3x3 Sobel in one pass for gfx950 (MI355X).
//
// Replaces one pass of the reference driver's pipeline, monolithic/src/main_with_sobel.c:51-62
// (convert_to_grayscale, plane extract, sobel_edge), without ever writing the luma plane to HBM.
// HBM traffic: 3 R + 1 W bytes per pixel instead of 6 + 4 + 2 for the reference's three loops.  Halo
// luma (rows above/below a band, pixels left/right of a strip) is recomputed from RGB, which is also the
// zero-communication alternative to exchanging luma halo rows between shards.
//
//   k_fused_stream      default: strip streaming (stream_core.h).  Per row a wave loads its 3 KiB (RGB) of the row
//                       with fully coalesced dwordx4 loads, de-interleaves them through a private LDS slab so
//                       that every lane owns 16 consecutive pixels, converts to luma in f32 and feeds the f32
//                       Sobel row step directly (no byte round trip).
//   k_gray_sobel_fused  the LDS-tile form (sobel_tile.h); GH_FUSED_IMPL=tile
#include "sobel_tile.h"
#include "stream_core.h"

#include <cstdlib>
#include <cstring>

namespace {

using namespace sobel_tile;
using stream_core::row_readable;

// luma of one pixel as an integer-valued float: trunc(y) == (float)(unsigned char)y for 0 <= y <= 255
// (same arithmetic as gh_luma, minus the float->int->float round trip)
template <int MODE>
__device__ __forceinline__ float luma_f(uint32_t r, uint32_t g, uint32_t b)
{
#pragma clang fp contract(off)
    const float fr = (float)r, fg = (float)g, fb = (float)b;
    float y;
    if (MODE == GH_MODE_FMA) {
        float t = 0.587f * fg;
        t = __builtin_fmaf(0.299f, fr, t);
        y = __builtin_fmaf(0.114f, fb, t);
    } else {
        const float s = 0.299f * fr + 0.587f * fg;
        y = s + 0.114f * fb;
    }
    y = __builtin_truncf(y);
    asm("" : "+v"(y));   // opaque: keeps the optimiser from turning the f32 adds below into (half-rate) integer ops
    return y;
}

// R output rows per strip, C rows per load chunk
template <int MODE, int CH, int R, int C>
__global__ __launch_bounds__(stream_core::BLOCK) void k_fused_stream(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ edge, int w,
                                                                    int rows, int groups_x, int zero_top, int zero_bot)
{
    using namespace stream_core;
    static_assert(C % 2 == 0, "chunk must be even: the h2 ring slot is the row parity");
    constexpr int SLAB = STRIP_PX * CH;                          // bytes of one strip row
    constexpr int NV = SLAB / 1024;                              // dwordx4 loads per lane per row (3 or 4)
    __shared__ __attribute__((aligned(16))) uint8_t slab[WAVES_PER_BLOCK][SLAB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx_i = (int)(blockIdx.x % (unsigned)groups_x), gy_i = (int)(blockIdx.x / (unsigned)groups_x);
    const int xw = (gx_i * WAVES_PER_BLOCK + wave) * STRIP_PX;
    if (xw >= w) return;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    const float half = half_reg();
    const bool in_frame = x < w;
    const int nvec = (xw + STRIP_PX <= w) ? NV * 64 : (w - xw) * CH / 16;   // 16-byte vectors of this strip row inside the frame
    const int hx = lane == 0 ? xw - 1 : xw + STRIP_PX;
    const bool hvalid = (lane == 0 || lane == 63) && hx >= 0 && hx < w;

    StateF st;
    st.clear();
    uint4 *sw = reinterpret_cast<uint4 *>(slab[wave]);
    const uint4 *sr = reinterpret_cast<const uint4 *>(slab[wave] + 16 * CH * lane);

    constexpr int NROWS = R + 2;
    constexpr int NCH = (NROWS + C - 1) / C;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        uint4 v[C][NV];
        uint32_t hp[C][3];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = ch * C + j;
            const int yy = y0 - 1 + i;
#pragma unroll
            for (int k = 0; k < NV; ++k) v[j][k] = make_uint4(0, 0, 0, 0);
            hp[j][0] = hp[j][1] = hp[j][2] = 0;
            if (i < NROWS && row_readable(yy, rows, zero_top, zero_bot)) {
                const uint8_t *rowb = rgb + ((ptrdiff_t)yy * w) * CH;
                const uint4 *rowp = reinterpret_cast<const uint4 *>(rowb + (size_t)xw * CH);
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    if (64 * k + lane < nvec) v[j][k] = load_stream(rowp + 64 * k + lane);
                if (hvalid) { const uint8_t *h = rowb + (size_t)hx * CH; hp[j][0] = h[0]; hp[j][1] = h[1]; hp[j][2] = h[2]; }
            }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = ch * C + j;
            // de-interleave: the wave's row (lane-linear vectors) -> LDS -> 16 consecutive pixels per lane
#pragma unroll
            for (int k = 0; k < NV; ++k) sw[64 * k + lane] = v[j][k];
            __builtin_amdgcn_wave_barrier();     // same wave, DS ops execute in order: only a scheduling fence
            uint32_t wv[4 * NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) { const uint4 u = sr[k]; wv[4 * k] = u.x; wv[4 * k + 1] = u.y; wv[4 * k + 2] = u.z; wv[4 * k + 3] = u.w; }
            __builtin_amdgcn_wave_barrier();
            float p[18];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (CH == 3) {
                    const int o = 3 * k;
                    p[k + 1] = luma_f<MODE>(gh_byte(wv[o >> 2], o & 3), gh_byte(wv[(o + 1) >> 2], (o + 1) & 3), gh_byte(wv[(o + 2) >> 2], (o + 2) & 3));
                } else {
                    p[k + 1] = luma_f<MODE>(gh_byte(wv[k], 0), gh_byte(wv[k], 1), gh_byte(wv[k], 2));
                }
            }
            const float hl = luma_f<MODE>(hp[j][0], hp[j][1], hp[j][2]);
            float pl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(uint32_t, p[16]), 0x138, 0xf, 0xf, false));
            float pr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(uint32_t, p[1]), 0x130, 0xf, 0xf, false));
            if (lane == 0) pl = hl;
            if (lane == 63) pr = hl;
            p[0] = pl; p[17] = pr;
            uint4 out = (j & 1) ? rowf<1>(st, p, half) : rowf<0>(st, p, half);
            if (i >= 2 && i < NROWS) {
                const int y = y0 + i - 2;
                if (x == 0) out.x &= 0xffffff00u;
                if (x + 15 == w - 1) out.w &= 0x00ffffffu;
                if ((y == 0 && zero_top) || (y == rows - 1 && zero_bot)) out = make_uint4(0, 0, 0, 0);
                if (in_frame && y < rows) store_stream(reinterpret_cast<uint4 *>(edge + (size_t)y * w + x), out);
            }
        }
    }
}

template <int MODE, int CH>
__device__ __forceinline__ uint4 luma_chunk(const uint8_t *__restrict__ p)
{
    uint32_t lum[16];
    if (CH == 3) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
        const uint4 a = q[0], b = q[1], c = q[2];
        const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = 3 * i;
            lum[i] = gh_luma<MODE>(gh_byte(w[o >> 2], o & 3), gh_byte(w[(o + 1) >> 2], (o + 1) & 3),
                                   gh_byte(w[(o + 2) >> 2], (o + 2) & 3));
        }
    } else {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = q[j];
            const uint32_t px[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) lum[4 * j + k] = gh_luma<MODE>(gh_byte(px[k], 0), gh_byte(px[k], 1), gh_byte(px[k], 2));
        }
    }
    uint4 o;
    o.x = lum[0] | (lum[1] << 8) | (lum[2] << 16) | (lum[3] << 24);
    o.y = lum[4] | (lum[5] << 8) | (lum[6] << 16) | (lum[7] << 24);
    o.z = lum[8] | (lum[9] << 8) | (lum[10] << 16) | (lum[11] << 24);
    o.w = lum[12] | (lum[13] << 8) | (lum[14] << 16) | (lum[15] << 24);
    return o;
}

template <int MODE, int CH>
__global__ __launch_bounds__(BLOCK) void k_gray_sobel_fused(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ edge, int w,
                                                            int rows, int tiles_x, int zero_top, int zero_bot)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[LDS_BYTES];

    const unsigned t = blockIdx.x;
    const int x0 = (int)(t % (unsigned)tiles_x) * TW;
    const int y0 = (int)(t / (unsigned)tiles_x) * TH;

#pragma unroll 3
    for (int it = 0; it < STAGE_ITERS; ++it) {
        const int idx = threadIdx.x + it * BLOCK;
        const int r = idx / CHUNKS, c = idx - r * CHUNKS;
        const int yy = y0 - 1 + r;
        const int gx = x0 - 16 + 16 * c;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (idx < LDS_ROWS * CHUNKS && gx >= 0 && gx < w && row_readable(yy, rows, zero_top, zero_bot))
            v = luma_chunk<MODE, CH>(rgb + ((ptrdiff_t)yy * w + gx) * CH);
        if (idx < LDS_ROWS * CHUNKS) *reinterpret_cast<uint4 *>(tile + idx * 16) = v;
    }
    __syncthreads();

    compute_tile(tile, edge, w, rows, x0, y0, zero_top, zero_bot);
}

}  // namespace

extern "C" int gh_gray_sobel_fused_rows_dev(const uint8_t *rgb, uint8_t *edge, int width, int rows, int channels, int mode,
                                            int top_is_border, int bottom_is_border, gh_stream_t stream)
{
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    GH_REQUIRE(channels == 3 || channels == 4, "fused kernel takes RGB8 or RGBA8");
    GH_REQUIRE(mode == GH_MODE_IEEE || mode == GH_MODE_FMA, "bad mode");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(rgb && edge, "null pointer");
    GH_REQUIRE(width % 16 == 0 && gh_aligned16(rgb) && gh_aligned16(edge),
               "fused kernel needs width % 16 == 0 and 16-byte aligned buffers (use gh_gray_rgb8_to_plane_dev + gh_sobel_u8_rows_dev otherwise)");
    if (int rc = gh_ensure_init()) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int zt = top_is_border ? 1 : 0, zb = bottom_is_border ? 1 : 0;
    static const bool use_tile = [] { const char *e = getenv("GH_FUSED_IMPL"); return e && strcmp(e, "tile") == 0; }();
    if (!use_tile) {
        constexpr int R = 32, C = 2;
        const int span = stream_core::STRIP_PX * stream_core::WAVES_PER_BLOCK;
        const int groups_x = (width + span - 1) / span;
        const long long nb = (long long)groups_x * ((rows + R - 1) / R);
        GH_REQUIRE(nb < (1ll << 31), "frame too large for one launch");
        const unsigned g = (unsigned)nb;
        const unsigned b = stream_core::BLOCK;
        if (channels == 3) {
            if (mode == GH_MODE_FMA) k_fused_stream<GH_MODE_FMA, 3, R, C><<<g, b, 0, s>>>(rgb, edge, width, rows, groups_x, zt, zb);
            else k_fused_stream<GH_MODE_IEEE, 3, R, C><<<g, b, 0, s>>>(rgb, edge, width, rows, groups_x, zt, zb);
        } else {
            if (mode == GH_MODE_FMA) k_fused_stream<GH_MODE_FMA, 4, R, C><<<g, b, 0, s>>>(rgb, edge, width, rows, groups_x, zt, zb);
            else k_fused_stream<GH_MODE_IEEE, 4, R, C><<<g, b, 0, s>>>(rgb, edge, width, rows, groups_x, zt, zb);
        }
        GH_LAUNCH_CHECK();
        return GH_OK;
    }
    const int tiles_x = (width + TW - 1) / TW;
    const int tiles_y = (rows + TH - 1) / TH;
    const long long nt = (long long)tiles_x * tiles_y;
    GH_REQUIRE(nt < (1ll << 31), "frame too large for one launch");
    const unsigned grid = (unsigned)nt;
    if (channels == 3) {
        if (mode == GH_MODE_FMA) k_gray_sobel_fused<GH_MODE_FMA, 3><<<grid, BLOCK, 0, s>>>(rgb, edge, width, rows, tiles_x, zt, zb);
        else k_gray_sobel_fused<GH_MODE_IEEE, 3><<<grid, BLOCK, 0, s>>>(rgb, edge, width, rows, tiles_x, zt, zb);
    } else {
        if (mode == GH_MODE_FMA) k_gray_sobel_fused<GH_MODE_FMA, 4><<<grid, BLOCK, 0, s>>>(rgb, edge, width, rows, tiles_x, zt, zb);
        else k_gray_sobel_fused<GH_MODE_IEEE, 4><<<grid, BLOCK, 0, s>>>(rgb, edge, width, rows, tiles_x, zt, zb);
    }
    GH_LAUNCH_CHECK();
    return GH_OK;
}
