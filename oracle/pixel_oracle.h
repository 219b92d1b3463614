/*
 * pixel_oracle.h -- CPU restatement of the reference's per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (libgrayhip.so, bin/grayscale,
 * the Python host mirror) may include, link or call this.  It is used by tests/, by
 * __graft_entry__.smoke() as the checker, and by bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_*.py) against
 *   - the reference's only golden fixture, images/test_gray.png (via tests/golden/),
 *   - SHA-256 known answers produced by the compiled reference,
 *   - the reference's own sources compiled in place into oracle/_ref/ (see oracle/Makefile),
 *     when /root/reference is present.
 *
 * Reference citations (relative to /root/reference):
 *   orc_gray_inplace      <- monolithic/src/parallel_to_grayscale.c:5-17
 *   orc_sobel_interior    <- monolithic/src/sobel.c:6-25
 *   orc_extract_plane     <- monolithic/src/main_with_sobel.c:57-59
 *   orc_replicate_plane   <- monolithic/src/main_with_sobel.c:65-73
 *   orc_pipeline          <- monolithic/src/main_with_sobel.c:49-76 (and main.c:30-38)
 *   orc_conv3x3_rgb8      <- old/parallel_convolution.c:8-24   (restated on u8 frames)
 *   orc_sum_rgb8          <- old/parallel_avg_pixel.c:5-42     (restated on u8 frames)
 *   orc_mean_gray_minmax  <- old/parallel_to_grayscale.c:7-38  (restated on u8 frames)
 */
#ifndef PIXEL_ORACLE_H
#define PIXEL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* arithmetic mode of the luma expression */
enum { ORC_MODE_IEEE = 0, ORC_MODE_FMA = 1 };

/* one luma value, both modes; exposed so tests can sweep all 2^24 triples */
uint8_t orc_luma(uint8_t r, uint8_t g, uint8_t b, int mode);

/* a1: in-place interleaved RGB8/RGBA8 -> R=G=B=lum, alpha untouched; channels<3 is a no-op */
void orc_gray_inplace(uint8_t *data, int width, int height, int channels, int mode);

/* GPU-native variant of a1+a3: interleaved in, 1 B/px plane out */
void orc_gray_to_plane(const uint8_t *rgb, uint8_t *gray, int width, int height, int channels, int mode);

/* a2: writes only the interior 1<=x<w-1, 1<=y<h-1 of dst, exactly like the reference */
void orc_sobel_interior(const uint8_t *src, uint8_t *dst, int width, int height);

/* a2 with this build's border definition: whole dst written, border = 0 */
void orc_sobel_zero_border(const uint8_t *src, uint8_t *dst, int width, int height);

/* a3 / a4 */
void orc_extract_plane(const uint8_t *img, uint8_t *plane, size_t npix, int channels);
void orc_replicate_plane(const uint8_t *plane, uint8_t *img, size_t npix, int channels);

/* a5: `passes` x (gray in place -> extract -> sobel -> replicate); gray/edge are caller scratch
 * of width*height bytes; edge is zero-filled once on entry (border definition). */
void orc_pipeline(uint8_t *img, uint8_t *gray, uint8_t *edge, int width, int height, int channels,
                  int passes, int mode);

/* main.c:33-35 -- `passes` x gray in place */
void orc_gray_passes(uint8_t *img, int width, int height, int channels, int passes, int mode);

/* a6 on u8 RGB frames: zero border supplied by the kernel, out = clamp(trunc(sum/norm),0,255) */
void orc_conv3x3_rgb8(const uint8_t *src, uint8_t *dst, int width, int height, const int32_t k[9], int32_t norm);

/* a7 on u8 RGB frames: the three exact per-channel sums */
void orc_sum_rgb8(const uint8_t *rgb, size_t npix, uint64_t sums[3]);
/* the reference's own accumulate-then-divide formula: avgs = (avgs_in + 3*sum) / (3*N) */
void orc_avg_from_sums(const uint64_t sums[3], size_t npix, int64_t avgs_inout[3]);

/* a8 on u8 RGB frames: g=(r+g+b)/3 replicated into 3 channels; minmax[0]=min, [1]=max
 * (initial values 256 / -1 as in the reference, so an empty frame returns those) */
void orc_mean_gray_minmax(const uint8_t *rgb, uint8_t *out_rgb, size_t npix, int32_t minmax[2]);

/* the historic functions themselves on `long` samples, flat arrays, run-time dimensions:
 * old/parallel_convolution.c:8-24 (pre-padded [rows+2][cols+2][3] input, += then /= norm),
 * old/parallel_avg_pixel.c:5-42, old/parallel_to_grayscale.c:7-38 (gray_rgb_out = [rows][cols][3]) */
void orc_old_parallel_convolution(const long *img_padded, const long kernel[9], long *convolved_inout, int rows, int cols, long gblur_norm);
void orc_old_parallel_avg_pixel(const long *img, long *avgs, int rows, int cols);
void orc_old_parallel_to_grayscale(const long *img, long *gray_rgb_out, long *min_max_gray, int rows, int cols);

/* deterministic synthetic frames */
void orc_fill_rand(uint8_t *buf, size_t nbytes, uint64_t seed);      /* SplitMix64 byte stream */
void orc_fill_smooth(uint8_t *rgb, int width, int height);
void orc_fill_cube(uint8_t *rgb);                                    /* 4096x4096, every triple once */

/* integer square root used to cross-check the float sqrt model */
uint32_t orc_isqrt(uint32_t m);

#ifdef __cplusplus
}
#endif
#endif
