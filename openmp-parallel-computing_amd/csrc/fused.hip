// fused.hip -- RGB8/RGBA8 -> luma -> 3x3 Sobel in one pass for gfx950 (MI355X).
//
// Replaces one pass of the reference driver's pipeline, monolithic/src/main_with_sobel.c:51-62
// (convert_to_grayscale, plane extract, sobel_edge), without ever writing the luma plane to HBM:
// the staging loop converts each 16-pixel RGB group to luma on its way into the LDS tile
// (sobel_tile.h), halo included, and the Sobel core runs from LDS.  HBM traffic: 3 R + 1 W bytes
// per pixel instead of 6 + 4 + 2 for the reference's three loops.  The 2/32 halo rows and 2/66 halo
// chunks are recomputed per tile (L2 hits), which is the zero-communication alternative to
// exchanging luma halo rows between shards.
#include "sobel_tile.h"

namespace {

using namespace sobel_tile;

__device__ __forceinline__ bool row_readable(int yy, int rows, int zero_top, int zero_bot)
{
    return (yy >= 0 && yy < rows) || (yy == -1 && !zero_top) || (yy == rows && !zero_bot);
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

    const unsigned t = remap_tile(blockIdx.x, gridDim.x);
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
