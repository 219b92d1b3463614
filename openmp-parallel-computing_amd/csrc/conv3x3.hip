// conv3x3.hip -- generic 3x3 integer convolution on interleaved RGB8 for gfx950 (MI355X).
//
// Replaces old/parallel_convolution.c:8-24 restated on u8 frames: per channel
//   out(r,c,p) = clamp( trunc( sum_{i,j} in(r+i-1, c+j-1, p) * k[i][j] / norm ), 0, 255 )
// with a zero border supplied by the kernel (the reference pre-pads its array) and C integer
// division (truncation toward zero).  The frame is treated as a byte plane of width 3*w: output
// byte xb depends on input bytes xb-3, xb, xb+3 of three rows.
#include "gh_common.h"

namespace {

constexpr int BLOCK = 256;

struct Taps { int k[9]; int norm; };

__global__ __launch_bounds__(BLOCK) void k_conv3x3_rgb(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int h, Taps t)
{
    const size_t rowb = (size_t)w * 3;
    const size_t n = rowb * h;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / rowb);
        const int xb = (int)(i - (size_t)y * rowb);
        int acc = 0;
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int yy = y + di - 1;
            if (yy < 0 || yy >= h) continue;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int xx = xb + 3 * (dj - 1);
                if (xx < 0 || xx >= (int)rowb) continue;
                acc += (int)src[(size_t)yy * rowb + xx] * t.k[di * 3 + dj];
            }
        }
        acc /= t.norm;
        acc = acc < 0 ? 0 : (acc > 255 ? 255 : acc);
        dst[i] = (uint8_t)acc;
    }
}

}  // namespace

extern "C" int gh_conv3x3_rgb8_dev(const uint8_t *src, uint8_t *dst, int width, int height, const int32_t kernel[9], int32_t norm,
                                   gh_stream_t stream)
{
    GH_REQUIRE(width >= 0 && height >= 0, "negative size");
    GH_REQUIRE(kernel, "null kernel");
    GH_REQUIRE(norm != 0, "norm must be non-zero");
    if (width == 0 || height == 0) return GH_OK;
    GH_REQUIRE(src && dst && src != dst, "null or aliased pointers");
    long long ksum = 0;
    for (int i = 0; i < 9; ++i) ksum += kernel[i] < 0 ? -(long long)kernel[i] : kernel[i];
    GH_REQUIRE(ksum * 255 < (1ll << 31), "kernel taps too large for 32-bit accumulation");
    if (int rc = gh_ensure_init()) return rc;
    Taps t;
    for (int i = 0; i < 9; ++i) t.k[i] = kernel[i];
    t.norm = norm;
    size_t blocks = ((size_t)width * height * 3 + BLOCK - 1) / BLOCK;
    if (blocks > (1u << 20)) blocks = 1u << 20;
    k_conv3x3_rgb<<<(unsigned)blocks, BLOCK, 0, (hipStream_t)stream>>>(src, dst, width, height, t);
    GH_LAUNCH_CHECK();
    return GH_OK;
}
