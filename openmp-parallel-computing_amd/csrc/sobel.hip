// sobel.hip -- 3x3 Sobel magnitude, u8 plane -> u8 plane, for gfx950 (MI355X).
//
// Replaces monolithic/src/sobel.c:6-25 (sobel_edge).  Bound: HBM, 1 R + 1 W byte per pixel; the
// 3x3 reuse is served from LDS (tile + one-pixel halo, see sobel_tile.h), no MFMA.
//
//   k_sobel_tiled : fast path (width % 16 == 0, 16-byte aligned planes)
//   k_sobel_px    : one pixel per lane, any geometry / alignment (odd widths, tiny frames)
#include "sobel_tile.h"

namespace {

using namespace sobel_tile;

// Is band-relative row yy backed by memory?  Rows -1 and `rows` are the halo rows of a shard.
__device__ __forceinline__ bool row_readable(int yy, int rows, int zero_top, int zero_bot)
{
    return (yy >= 0 && yy < rows) || (yy == -1 && !zero_top) || (yy == rows && !zero_bot);
}

__global__ __launch_bounds__(BLOCK) void k_sobel_tiled(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w,
                                                       int rows, int tiles_x, int zero_top, int zero_bot)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[LDS_BYTES];

    const unsigned t = remap_tile(blockIdx.x, gridDim.x);
    const int x0 = (int)(t % (unsigned)tiles_x) * TW;
    const int y0 = (int)(t / (unsigned)tiles_x) * TH;

    // ---- stage (TH+2) x (TW+32) bytes: issue every global load first, then the LDS writes ----
    uint4 v[STAGE_ITERS];
#pragma unroll
    for (int it = 0; it < STAGE_ITERS; ++it) {
        const int idx = threadIdx.x + it * BLOCK;
        const int r = idx / CHUNKS, c = idx - r * CHUNKS;
        const int yy = y0 - 1 + r;
        const int gx = x0 - 16 + 16 * c;
        v[it] = make_uint4(0, 0, 0, 0);
        if (idx < LDS_ROWS * CHUNKS && gx >= 0 && gx < w && row_readable(yy, rows, zero_top, zero_bot))
            v[it] = *reinterpret_cast<const uint4 *>(src + (ptrdiff_t)yy * w + gx);
    }
#pragma unroll
    for (int it = 0; it < STAGE_ITERS; ++it) {
        const int idx = threadIdx.x + it * BLOCK;
        if (idx < LDS_ROWS * CHUNKS) *reinterpret_cast<uint4 *>(tile + idx * 16) = v[it];
    }
    __syncthreads();

    compute_tile(tile, dst, w, rows, x0, y0, zero_top, zero_bot);
}

// One pixel per lane; rows above/below the band are read only when they exist.
__global__ __launch_bounds__(BLOCK) void k_sobel_px(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows,
                                                    int zero_top, int zero_bot)
{
    const size_t n = (size_t)w * rows;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / w), x = (int)(i - (size_t)y * w);
        uint32_t out = 0;
        const bool border = x == 0 || x == w - 1 || (y == 0 && zero_top) || (y == rows - 1 && zero_bot);
        if (!border) {
            const uint8_t *m = src + (ptrdiff_t)y * w + x;
            const uint8_t *u = m - w, *d = m + w;
            const int gx = -(int)u[-1] - 2 * (int)m[-1] - (int)d[-1] + (int)u[1] + 2 * (int)m[1] + (int)d[1];
            const int gy = (int)u[-1] + 2 * (int)u[0] + (int)u[1] - (int)d[-1] - 2 * (int)d[0] - (int)d[1];
            out = gh_sobel_mag(gx * gx + gy * gy);
        }
        dst[i] = (uint8_t)out;
    }
}

__global__ void k_selftest_sqrt(uint32_t m_max, unsigned long long *mismatches)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m <= m_max; m += stride) {
        // exact integer square root by bisection on r*r <= m
        uint32_t lo = 0, hi = 65535;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if ((unsigned long long)mid * mid <= m) lo = mid; else hi = mid - 1;
        }
        const uint32_t want = lo > 255 ? 255 : lo;
        if (gh_sobel_mag((int)m) != want) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace

extern "C" {

int gh_sobel_u8_rows_dev(const uint8_t *src, uint8_t *dst, int width, int rows, int top_is_border, int bottom_is_border,
                         gh_stream_t stream)
{
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(src && dst, "null pointer");
    GH_REQUIRE(src != dst, "src and dst must differ");
    if (int rc = gh_ensure_init()) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int zt = top_is_border ? 1 : 0, zb = bottom_is_border ? 1 : 0;
    if (width % 16 == 0 && gh_aligned16(src) && gh_aligned16(dst)) {
        const int tiles_x = (width + TW - 1) / TW;
        const int tiles_y = (rows + TH - 1) / TH;
        const long long nt = (long long)tiles_x * tiles_y;
        GH_REQUIRE(nt < (1ll << 31), "frame too large for one launch");
        k_sobel_tiled<<<(unsigned)nt, BLOCK, 0, s>>>(src, dst, width, rows, tiles_x, zt, zb);
    } else {
        size_t blocks = ((size_t)width * rows + BLOCK - 1) / BLOCK;
        if (blocks > (1u << 20)) blocks = 1u << 20;
        k_sobel_px<<<(unsigned)blocks, BLOCK, 0, s>>>(src, dst, width, rows, zt, zb);
    }
    GH_LAUNCH_CHECK();
    return GH_OK;
}

int gh_sobel_u8_dev(const uint8_t *src, uint8_t *dst, int width, int height, gh_stream_t stream)
{
    return gh_sobel_u8_rows_dev(src, dst, width, height, 1, 1, stream);
}

int gh_selftest_sobel_sqrt(uint32_t m_max, uint64_t *mismatches)
{
    GH_REQUIRE(mismatches, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    unsigned long long *d = nullptr;
    GH_HIP_TRY(hipMalloc(&d, sizeof(*d)));
    GH_HIP_TRY(hipMemset(d, 0, sizeof(*d)));
    k_selftest_sqrt<<<1024, 256>>>(m_max, d);
    GH_LAUNCH_CHECK();
    unsigned long long h = 0;
    GH_HIP_TRY(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    GH_HIP_TRY(hipFree(d));
    *mismatches = h;
    return GH_OK;
}

}  // extern "C"
