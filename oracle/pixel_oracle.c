/*
 * pixel_oracle.c -- CPU restatement of the reference's per-pixel hot path (see pixel_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; never by the product path.
 *
 * Must be compiled with -ffp-contract=off: the IEEE luma model needs every float op rounded
 * separately (gcc would otherwise fuse a*b+c when the target has FMA).  oracle/Makefile does so.
 */
#include "pixel_oracle.h"

#include <math.h>
#include <string.h>

/* luma weights exactly as written in monolithic/src/parallel_to_grayscale.c:13 */
#define W_R 0.299f
#define W_G 0.587f
#define W_B 0.114f

/* IEEE: ((0.299f*r + 0.587f*g) + 0.114f*b), each op rounded to float32, then C truncation.
 * This is what `-O3 -fopenmp` (microservices/grayscale/c/Makefile:2) produces and what the
 * reference's golden images/test_gray.png contains. */
static inline uint8_t luma_ieee(uint8_t r, uint8_t g, uint8_t b)
{
    float pr = W_R * (float)r;
    float pg = W_G * (float)g;
    float pb = W_B * (float)b;
    float s = pr + pg;
    s = s + pb;
    return (uint8_t)s;
}

/* FMA: what gcc 11.4 emits for the same expression under monolithic/Makefile:2
 * (-O3 -march=native -ffast-math): mul for the g term, then two fused multiply-adds. */
static inline uint8_t luma_fma(uint8_t r, uint8_t g, uint8_t b)
{
    float t = W_G * (float)g;
    t = fmaf(W_R, (float)r, t);
    t = fmaf(W_B, (float)b, t);
    return (uint8_t)t;
}

uint8_t orc_luma(uint8_t r, uint8_t g, uint8_t b, int mode)
{
    return mode == ORC_MODE_FMA ? luma_fma(r, g, b) : luma_ieee(r, g, b);
}

void orc_gray_inplace(uint8_t *data, int width, int height, int channels, int mode)
{
    if (channels < 3) return;   /* reference behaviour is undefined there; this build: no-op */
    const long long npix = (long long)width * height;
    if (mode == ORC_MODE_FMA) {
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < npix; ++i) {
            uint8_t *p = data + (size_t)i * channels;
            uint8_t y = luma_fma(p[0], p[1], p[2]);
            p[0] = y; p[1] = y; p[2] = y;
        }
    } else {
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < npix; ++i) {
            uint8_t *p = data + (size_t)i * channels;
            uint8_t y = luma_ieee(p[0], p[1], p[2]);
            p[0] = y; p[1] = y; p[2] = y;
        }
    }
}

void orc_gray_to_plane(const uint8_t *rgb, uint8_t *gray, int width, int height, int channels, int mode)
{
    const long long npix = (long long)width * height;
    if (channels < 3) {   /* grey / grey+alpha input: plane = channel 0 */
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < npix; ++i) gray[i] = rgb[(size_t)i * channels];
        return;
    }
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < npix; ++i) {
        const uint8_t *p = rgb + (size_t)i * channels;
        gray[i] = mode == ORC_MODE_FMA ? luma_fma(p[0], p[1], p[2]) : luma_ieee(p[0], p[1], p[2]);
    }
}

static inline uint8_t sobel_at(const uint8_t *up, const uint8_t *mid, const uint8_t *dn, int x)
{
    /* monolithic/src/sobel.c:14-22 */
    int gx = -(int)up[x - 1] - 2 * (int)mid[x - 1] - (int)dn[x - 1]
             + (int)up[x + 1] + 2 * (int)mid[x + 1] + (int)dn[x + 1];
    int gy =  (int)up[x - 1] + 2 * (int)up[x] + (int)up[x + 1]
             - (int)dn[x - 1] - 2 * (int)dn[x] - (int)dn[x + 1];
    int mag = (int)sqrtf((float)(gx * gx + gy * gy));
    return (uint8_t)(mag > 255 ? 255 : mag);
}

void orc_sobel_interior(const uint8_t *src, uint8_t *dst, int width, int height)
{
    if (width < 3 || height < 3) return;
#pragma omp parallel for schedule(static)
    for (int y = 1; y < height - 1; ++y) {
        const uint8_t *mid = src + (size_t)y * width;
        const uint8_t *up = mid - width, *dn = mid + width;
        uint8_t *out = dst + (size_t)y * width;
        for (int x = 1; x < width - 1; ++x) out[x] = sobel_at(up, mid, dn, x);
    }
}

void orc_sobel_zero_border(const uint8_t *src, uint8_t *dst, int width, int height)
{
    if (width <= 0 || height <= 0) return;
    memset(dst, 0, (size_t)width * height);
    orc_sobel_interior(src, dst, width, height);
}

void orc_extract_plane(const uint8_t *img, uint8_t *plane, size_t npix, int channels)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)npix; ++i) plane[i] = img[(size_t)i * channels];
}

void orc_replicate_plane(const uint8_t *plane, uint8_t *img, size_t npix, int channels)
{
    if (channels < 3) {
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)npix; ++i) img[(size_t)i * channels] = plane[i];
        return;
    }
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)npix; ++i) {
        uint8_t *p = img + (size_t)i * channels;
        p[0] = p[1] = p[2] = plane[i];
    }
}

void orc_pipeline(uint8_t *img, uint8_t *gray, uint8_t *edge, int width, int height, int channels,
                  int passes, int mode)
{
    const size_t npix = (size_t)width * height;
    if (passes < 1) passes = 1;
    memset(edge, 0, npix);
    for (int p = 0; p < passes; ++p) {
        orc_gray_inplace(img, width, height, channels, mode);
        orc_extract_plane(img, gray, npix, channels);
        orc_sobel_interior(gray, edge, width, height);
        orc_replicate_plane(edge, img, npix, channels);
    }
}

void orc_gray_passes(uint8_t *img, int width, int height, int channels, int passes, int mode)
{
    if (passes < 1) passes = 1;
    for (int p = 0; p < passes; ++p) orc_gray_inplace(img, width, height, channels, mode);
}

void orc_conv3x3_rgb8(const uint8_t *src, uint8_t *dst, int width, int height, const int32_t k[9], int32_t norm)
{
    /* old/parallel_convolution.c:12-23 with the image at offset (1,1) of a zero padded array:
     * out(r,c,p) = trunc( sum_{i,j} in(r+i-1, c+j-1, p) * k[i][j] / norm ), C integer division. */
#pragma omp parallel for schedule(static)
    for (int r = 0; r < height; ++r) {
        for (int c = 0; c < width; ++c) {
            for (int p = 0; p < 3; ++p) {
                int64_t acc = 0;
                for (int i = 0; i < 3; ++i) {
                    int rr = r + i - 1;
                    if (rr < 0 || rr >= height) continue;
                    for (int j = 0; j < 3; ++j) {
                        int cc = c + j - 1;
                        if (cc < 0 || cc >= width) continue;
                        acc += (int64_t)src[((size_t)rr * width + cc) * 3 + p] * k[i * 3 + j];
                    }
                }
                acc /= norm;
                if (acc < 0) acc = 0;
                if (acc > 255) acc = 255;
                dst[((size_t)r * width + c) * 3 + p] = (uint8_t)acc;
            }
        }
    }
}

void orc_sum_rgb8(const uint8_t *rgb, size_t npix, uint64_t sums[3])
{
    uint64_t s0 = 0, s1 = 0, s2 = 0;
#pragma omp parallel for schedule(static) reduction(+ : s0, s1, s2)
    for (long long i = 0; i < (long long)npix; ++i) {
        s0 += rgb[3 * (size_t)i + 0];
        s1 += rgb[3 * (size_t)i + 1];
        s2 += rgb[3 * (size_t)i + 2];
    }
    sums[0] = s0; sums[1] = s1; sums[2] = s2;
}

void orc_avg_from_sums(const uint64_t sums[3], size_t npix, int64_t avgs_inout[3])
{
    /* old/parallel_avg_pixel.c:17-31,38-41: each channel is added three times (inner `pixel`
     * loop), accumulated INTO avgs, then divided by ROW*COL*RGB. */
    const int64_t count = (int64_t)npix * 3;
    for (int c = 0; c < 3; ++c) {
        avgs_inout[c] += 3 * (int64_t)sums[c];
        if (count) avgs_inout[c] /= count;
    }
}

void orc_mean_gray_minmax(const uint8_t *rgb, uint8_t *out_rgb, size_t npix, int32_t minmax[2])
{
    int lo = 256, hi = -1;   /* old/parallel_to_grayscale.c:9-10 */
#pragma omp parallel for schedule(static) reduction(min : lo) reduction(max : hi)
    for (long long i = 0; i < (long long)npix; ++i) {
        const uint8_t *p = rgb + 3 * (size_t)i;
        int g = ((int)p[0] + p[1] + p[2]) / 3;
        uint8_t *o = out_rgb + 3 * (size_t)i;
        o[0] = o[1] = o[2] = (uint8_t)g;
        if (g < lo) lo = g;
        if (g > hi) hi = g;
    }
    minmax[0] = lo; minmax[1] = hi;
}

/* ---- old/ historic functions on `long` samples (flat arrays instead of pointer trees; dimensions at run time) -------- */

void orc_old_parallel_convolution(const long *img_padded, const long kernel[9], long *convolved_inout, int rows, int cols, long gblur_norm)
{
    /* old/parallel_convolution.c:12-23: convolved[r][c][p] += img[r+kr][c+kc][p] * kernel[kr][kc]; then /= GBLUR_NORM */
    const size_t pitch = ((size_t)cols + 2) * 3;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            for (int p = 0; p < 3; ++p) {
                long *o = &convolved_inout[((size_t)r * cols + c) * 3 + p];
                for (int kr = 0; kr < 3; ++kr)
                    for (int kc = 0; kc < 3; ++kc) *o += img_padded[(size_t)(r + kr) * pitch + (size_t)(c + kc) * 3 + p] * kernel[kr * 3 + kc];
                *o /= gblur_norm;
            }
}

void orc_old_parallel_avg_pixel(const long *img, long *avgs, int rows, int cols)
{
    /* old/parallel_avg_pixel.c:14-41: every channel is added once per iteration of the inner `pixel` loop (3x) */
    long c1 = 0, c2 = 0, c3 = 0;
    const long count = (long)rows * cols * 3;
    for (long i = 0; i < (long)rows * cols; ++i)
        for (int pixel = 0; pixel < 3; ++pixel) { c1 += img[3 * i]; c2 += img[3 * i + 1]; c3 += img[3 * i + 2]; }
    avgs[0] += c1; avgs[1] += c2; avgs[2] += c3;
    for (int p = 0; p < 3; ++p) avgs[p] /= count;
}

void orc_old_parallel_to_grayscale(const long *img, long *gray_rgb_out, long *min_max_gray, int rows, int cols)
{
    /* old/parallel_to_grayscale.c:9-37 */
    int min_gray = 256, max_gray = -1;
    for (long i = 0; i < (long)rows * cols; ++i) {
        int sum = 0;
        for (int pixel = 0; pixel < 3; ++pixel) sum += img[3 * i + pixel];
        int g = sum / 3;
        gray_rgb_out[3 * i] = gray_rgb_out[3 * i + 1] = gray_rgb_out[3 * i + 2] = g;
        if (g < min_gray) min_gray = g;
        if (g > max_gray) max_gray = g;
    }
    min_max_gray[0] = min_gray;
    min_max_gray[1] = max_gray;
}

/* ---- synthetic frames ------------------------------------------------------------------ */

static inline uint64_t splitmix_word(uint64_t seed, uint64_t j)
{
    uint64_t z = seed + (j + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_fill_rand(uint8_t *buf, size_t nbytes, uint64_t seed)
{
    const size_t nwords = nbytes / 8;
#pragma omp parallel for schedule(static)
    for (long long j = 0; j < (long long)nwords; ++j) {
        uint64_t z = splitmix_word(seed, (uint64_t)j);
        memcpy(buf + 8 * (size_t)j, &z, 8);   /* little-endian host */
    }
    if (nbytes % 8) {
        uint64_t z = splitmix_word(seed, nwords);
        memcpy(buf + 8 * nwords, &z, nbytes % 8);
    }
}

void orc_fill_smooth(uint8_t *rgb, int width, int height)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            uint64_t i = (uint64_t)y * width + x;
            uint32_t n = (uint32_t)((i * 2654435761ull) & 0xFFFFFFFFull) >> 28;
            uint8_t *p = rgb + 3 * i;
            p[0] = (uint8_t)((x / 4 + y / 8 + n) & 255);
            p[1] = (uint8_t)((x / 8 + y / 4 + 2 * n) & 255);
            p[2] = (uint8_t)(((x + y) / 16 + n) & 255);
        }
    }
}

void orc_fill_cube(uint8_t *rgb)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (1ll << 24); ++i) {
        rgb[3 * i + 0] = (uint8_t)(i >> 16);
        rgb[3 * i + 1] = (uint8_t)(i >> 8);
        rgb[3 * i + 2] = (uint8_t)i;
    }
}

uint32_t orc_isqrt(uint32_t m)
{
    uint32_t r = 0, bit = 1u << 30;
    while (bit > m) bit >>= 2;
    while (bit) {
        if (m >= r + bit) { m -= r + bit; r = (r >> 1) + bit; }
        else r >>= 1;
        bit >>= 2;
    }
    return r;
}
