/*
 * grayhip.h -- C ABI of libgrayhip.so: the MI355X (gfx950) implementation of the reference's
 * per-pixel hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * Every entry point names the reference interface it replaces (paths relative to
 * PedemonteGiacomo/OpenMp-Parallel-Computing):
 *
 *   convert_to_grayscale / gh_gray_*      monolithic/include/parallel_to_grayscale.h:4,
 *                                         monolithic/src/parallel_to_grayscale.c:5-17
 *   sobel_edge / gh_sobel_*               monolithic/include/sobel.h:3-5, monolithic/src/sobel.c:6-25
 *   gh_extract_plane_dev                  monolithic/src/main_with_sobel.c:57-59
 *   gh_replicate_plane_dev                monolithic/src/main_with_sobel.c:65-73
 *   gh_pipeline_*                         monolithic/src/main_with_sobel.c:49-76, monolithic/src/main.c:30-38
 *   gh_conv3x3_rgb8_dev / _rows_dev       old/parallel_convolution.c:8-24   (u8 frames, zero border; row bands with halo rows)
 *   gh_sum_rgb8_dev                       old/parallel_avg_pixel.c:5-42     (exact u64 channel sums)
 *   gh_mean_gray_minmax_dev               old/parallel_to_grayscale.c:7-38  (u8 frames)
 *   gh_old_parallel_*                     old/parallel_convolution.c:8, old/parallel_avg_pixel.c:5,
 *                                         old/parallel_to_grayscale.c:7     (historic signatures, long samples)
 *
 * Conventions
 *   - *_dev functions take DEVICE pointers, enqueue work on `stream` (a hipStream_t passed as
 *     void*, NULL = the default stream) and return without synchronising.
 *   - *_host functions and the two reference-named symbols take HOST pointers, are synchronous
 *     and own their staging buffers (caller-owned image buffers, exactly like the reference).
 *   - Return value: 0 on success, a negative gh_status otherwise; gh_last_error() returns the
 *     message of the calling thread's last failure.  There is NO CPU fallback anywhere: if HIP
 *     or the GPU is unavailable the call fails.
 *   - `mode`: GH_MODE_IEEE reproduces the reference built with `-O3 -fopenmp` (this is the build that
 *     produced the reference's golden images/test_gray.png); GH_MODE_FMA reproduces the
 *     `-O3 -march=native -ffast-math` build on an FMA x86 host (monolithic/Makefile:2).
 *   - Sobel border: the reference never writes dst's border; the *_dev kernels write 0 there
 *     (what the reference CLI shows in practice), the host shim `sobel_edge` leaves it untouched.
 */
#ifndef GRAYHIP_H
#define GRAYHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GH_API __attribute__((visibility("default")))

typedef void *gh_stream_t;   /* hipStream_t */

enum gh_mode { GH_MODE_IEEE = 0, GH_MODE_FMA = 1 };

enum gh_status {
    GH_OK = 0,
    GH_ERR_HIP = -1,        /* a HIP runtime call failed (message has the hipError string) */
    GH_ERR_ARG = -2,        /* bad argument (NULL pointer, channels outside 1..4, src==dst, ...) */
    GH_ERR_NO_DEVICE = -3   /* no usable gfx950 device */
};

/* ---- runtime ------------------------------------------------------------------------------ */
GH_API int gh_init(int device);                 /* select device, warm the context; idempotent */
GH_API int gh_device_count(void);
GH_API const char *gh_last_error(void);
/* Name of the device-kernel form the calling thread's last Sobel / fused / convolution launch took, e.g. "k_sobel_dma2 regular" or
 * "k_fused_dma D=2" ("" before the first launch).  Diagnostic: lets a test assert that a shape runs the kernel it is meant to run. */
GH_API const char *gh_last_kernel(void);
GH_API const char *gh_version(void);
/* CU and XCD counts of the initialised device, as the launch rules use them (band heights and grid sizes are chosen in multiples of
 * the CU count; MI355X in SPX mode: 256 / 8).  Initialises the library if needed. */
GH_API int gh_device_topology(int *compute_units, int *xcds);
GH_API int gh_default_mode(void);               /* GRAYHIP_MODE=ieee|fma, default ieee */

GH_API int gh_malloc(void **dptr, size_t nbytes);
GH_API int gh_free(void *dptr);
GH_API int gh_memset_dev(void *dptr, int value, size_t nbytes, gh_stream_t stream);
GH_API int gh_copy_h2d(void *dst_dev, const void *src_host, size_t nbytes, gh_stream_t stream);
GH_API int gh_copy_d2h(void *dst_host, const void *src_dev, size_t nbytes, gh_stream_t stream);
GH_API int gh_stream_sync(gh_stream_t stream);

/* measurement aid ("confirm on the box with a copy-kernel ceiling"): linear nontemporal
 * copy of nbytes (16-byte aligned buffers and size), 1 R + 1 W -- the achievable HBM rate bench.py reports beside
 * the 8 TB/s spec peak.  Replaces nothing in the reference. */
GH_API int gh_copy_stream_dev(const void *src, void *dst, size_t nbytes, gh_stream_t stream);

/* ---- device-resident kernels (the measured hot path) ------------------------------------- */

/* RGB8/RGBA8 interleaved -> 1 B/px luma plane.  3 R + 1 W bytes per pixel (channels==3).
 * channels 1 or 2: plane = channel 0 (grey / grey+alpha input). */
GH_API int gh_gray_rgb8_to_plane_dev(const uint8_t *rgb, uint8_t *gray, size_t npix, int channels,
                                     int mode, gh_stream_t stream);

/* Reference ABI semantics on device memory: R=G=B=luma in place, alpha untouched.
 * channels<3 is a no-op (the reference reads out of bounds there). */
GH_API int gh_gray_inplace_dev(uint8_t *data, size_t npix, int channels, int mode, gh_stream_t stream);
/* `passes` applications of convert_to_grayscale in ONE launch (the driver loop of monolithic/src/main.c:33-35): passes
 * 2..N see a grey pixel (y, y, y) and map it to lum(y, y, y); they are independent per pixel, so they run in registers
 * between one load and one store -- the same arithmetic as N launches, bit-identical result. */
GH_API int gh_gray_inplace_passes_dev(uint8_t *data, size_t npix, int channels, int mode, int passes, gh_stream_t stream);

/* 3x3 Sobel magnitude, u8 plane -> u8 plane, whole frame; border pixels are written as 0.
 * width<3 or height<3: dst is all zeros.  Any width and alignment (1024 columns and more: LDS-DMA ring kernels). */
GH_API int gh_sobel_u8_dev(const uint8_t *src, uint8_t *dst, int width, int height, gh_stream_t stream);

/* Row band of a frame (row sharding): computes `rows` output rows.  src/dst point at the band's
 * first row.  If !top_is_border the row above the band (src - width) must be readable (halo
 * row); likewise below.  A band edge that IS the frame border is written as 0. */
GH_API int gh_sobel_u8_rows_dev(const uint8_t *src, uint8_t *dst, int width, int rows,
                                int top_is_border, int bottom_is_border, gh_stream_t stream);

/* Pass k >= 2 of the reference's gray + Sobel driver loop (monolithic/src/main_with_sobel.c:51-62 applied to what pass k - 1 wrote, the
 * replicated edge image (e, e, e)) in ONE kernel: src holds pass k - 1's edge plane and every byte goes through lum(e, e, e) -- `mode`
 * arithmetic; not the identity, 13 grey levels move in IEEE mode -- on its way into the 3x3 window.  1 R + 1 W byte per pixel, where
 * gh_gray_of_plane_dev + gh_sobel_u8_dev move 4.  Same band semantics and border rule as gh_sobel_u8_rows_dev; a halo row holds the
 * neighbouring EDGE row (the kernel applies the luma to it as well). */
GH_API int gh_sobel_of_gray_u8_dev(const uint8_t *src, uint8_t *dst, int width, int height, int mode, gh_stream_t stream);
GH_API int gh_sobel_of_gray_u8_rows_dev(const uint8_t *src, uint8_t *dst, int width, int rows, int mode,
                                        int top_is_border, int bottom_is_border, gh_stream_t stream);

/* Fused RGB -> luma -> Sobel; the luma plane never touches HBM.  Same band semantics as above
 * (halo rows are RGB rows).  3 R + 1 W bytes per pixel.  Any width and any alignment: frames of 1024 columns and more run the
 * LDS-DMA ring kernel whatever their row pitch and base address, narrower ones register-staged strips / the per-pixel kernel. */
GH_API int gh_gray_sobel_fused_rows_dev(const uint8_t *rgb, uint8_t *edge, int width, int rows, int channels,
                                        int mode, int top_is_border, int bottom_is_border, gh_stream_t stream);

GH_API int gh_extract_plane_dev(const uint8_t *img, uint8_t *plane, size_t npix, int channels, gh_stream_t stream);
GH_API int gh_replicate_plane_dev(const uint8_t *plane, uint8_t *img, size_t npix, int channels, gh_stream_t stream);

/* luma of an already grey plane: v -> lum(v,v,v) (what pass k+1 of the reference CLIs does
 * to pass k's output; not the identity for 13 grey levels in IEEE mode). */
GH_API int gh_gray_of_plane_dev(const uint8_t *src, uint8_t *dst, size_t npix, int mode, gh_stream_t stream);

/* generic 3x3 integer convolution on interleaved RGB8, zero border, out = clamp(trunc(sum/norm)); any width and alignment
 * (rows of 1024 bytes and more: LDS-DMA ring kernel; 0 < |norm| <= 4096 and sum|k| * 255 < 2^24, else the per-pixel kernel) */
GH_API int gh_conv3x3_rgb8_dev(const uint8_t *src, uint8_t *dst, int width, int height,
                               const int32_t kernel[9], int32_t norm, gh_stream_t stream);
/* Row band of a frame (row sharding of old/parallel_convolution.c:8-24): `rows` output rows, src/dst at the band's first row.  If
 * !top_is_border the RGB row above the band (src - 3*width) must be readable and is a real row of the frame (a halo row); likewise
 * below.  A band edge that IS the frame border sees the zero border there, as the whole-frame call does on all four sides. */
GH_API int gh_conv3x3_rgb8_rows_dev(const uint8_t *src, uint8_t *dst, int width, int rows, const int32_t kernel[9], int32_t norm,
                                    int top_is_border, int bottom_is_border, gh_stream_t stream);

/* exact per-channel sums of an RGB8 frame, ADDED into sums_dev[0..2] (device u64[3], zero it first) */
GH_API int gh_sum_rgb8_dev(const uint8_t *rgb, size_t npix, uint64_t *sums_dev, gh_stream_t stream);

/* g=(r+g+b)/3 replicated to 3 channels; minmax_dev[0]=min, [1]=max must be pre-set to 256 / -1 */
GH_API int gh_mean_gray_minmax_dev(const uint8_t *rgb, uint8_t *out_rgb, size_t npix, int32_t *minmax_dev,
                                   gh_stream_t stream);

/* `passes` x (gray -> sobel -> replicate) on a device-resident interleaved image, reference
 * semantics (main_with_sobel.c:51-74).  gray_ws / edge_ws: width*height bytes of scratch each (two distinct planes: the edge planes of
 * successive passes alternate between them, the last pass's ends in edge_ws).  Pass 1 is the fused kernel (4 B/px), every later pass
 * gh_sobel_of_gray_u8_dev (2 B/px). */
GH_API int gh_pipeline_gray_sobel_dev(uint8_t *img, int width, int height, int channels, int passes, int mode,
                                      uint8_t *gray_ws, uint8_t *edge_ws, gh_stream_t stream);

/* ---- host-pointer entry points (drop-in boundary) ---------------------------------------- */
GH_API int gh_convert_to_grayscale_host(unsigned char *data, int width, int height, int channels, int mode);
GH_API int gh_gray_passes_host(unsigned char *data, int width, int height, int channels, int passes, int mode,
                               double *kernel_seconds);
GH_API int gh_sobel_edge_host(const unsigned char *src, unsigned char *dst, int width, int height);
GH_API int gh_pipeline_gray_sobel_host(unsigned char *img, int width, int height, int channels, int passes,
                                       int mode, double *kernel_seconds);
GH_API int gh_sum_rgb8_host(const unsigned char *rgb, size_t npix, uint64_t sums[3]);

/* Streamed gray+Sobel of a frame that lives in HOST memory and may be larger than anyone wants to upload at once
 * (BASELINE configs[4]: 65536x65536 streamed tiles).  The frame is cut into bands of `tile_rows` rows (+ one RGB halo
 * row above and below, re-sent with each band, so bands are independent); three hipStreams and a 3-slot ring of
 * device buffers overlap H2D of band k+1, the fused kernel on band k and D2H of band k-1.  Output: the 1 B/px edge
 * plane (frame border = 0), identical to gh_gray_sobel_fused_rows_dev on the whole frame (any width).  Overlap is real only if
 * the host buffers are page-locked: pass host_buffers_pinned != 0 if they already are (hipHostMalloc /
 * hipHostRegister / torch pin_memory), 0 to let the call register and unregister them itself.
 * seconds (optional) = wall time of the pipelined loop, registration excluded.  PCIe-bound by construction. */
GH_API int gh_gray_sobel_streamed_host(const unsigned char *rgb, unsigned char *edge, int width, int height, int channels,
                                       int mode, int tile_rows, int host_buffers_pinned, double *seconds);
/* The same on a block of `height` rows of a taller host frame -- one rank's share when the frame is split over the GPUs
 * of a node (BASELINE configs[4] "8 GPUs": bands are independent, zero communication).  Where a side is not the frame
 * border, the RGB row just outside the block must be readable in host memory at rgb - pitch / rgb + height * pitch. */
GH_API int gh_gray_sobel_streamed_rows_host(const unsigned char *rgb, unsigned char *edge, int width, int height, int channels,
                                            int mode, int tile_rows, int host_buffers_pinned, int top_is_border,
                                            int bottom_is_border, double *seconds);

/* ---- row-sharded frames: one process per GPU, halo rows over RCCL (ncclSend/ncclRecv on xGMI) --------------
 * The reference parallelises rows with OpenMP threads over one shared image (monolithic/src/sobel.c:10,
 * parallel_to_grayscale.c:7); across GPUs each rank owns a contiguous block of rows and the 3x3 stencil needs the
 * row above and below the block from the neighbouring ranks.  RCCL is dlopen'ed by gh_comm_init (the rest of the
 * library does not depend on it).  Bootstrap: rank 0 calls gh_comm_unique_id, the 128 bytes reach the other ranks
 * by any means (torch.distributed broadcast, a file, MPI), every rank calls gh_comm_init on its own device. */
#define GH_COMM_ID_BYTES 128
typedef void *gh_comm_t;
GH_API int gh_comm_unique_id(uint8_t id[GH_COMM_ID_BYTES]);
GH_API int gh_comm_init(const uint8_t id[GH_COMM_ID_BYTES], int rank, int world, gh_comm_t *out);
GH_API int gh_comm_rank(gh_comm_t comm);
GH_API int gh_comm_world(gh_comm_t comm);
GH_API int gh_comm_destroy(gh_comm_t comm);
/* rehearsal hook for a one-GPU box: the sharded steps below treat this rank as one of the three kinds of shard a multi-GPU job has,
 * with ncclSend/ncclRecv to ITSELF standing in for the neighbours (a halo row becomes a copy of the shard's own first / last row):
 *   mode 1  a MIDDLE shard: neighbours above and below;
 *   mode 2  the TOP shard (rank 0 of N): the frame's zero border above (monolithic/src/sobel.c:11-12), ONE halo row, from below;
 *   mode 3  the BOTTOM shard (rank N-1): the mirror image;      mode 0: off. */
GH_API int gh_comm_set_loopback(gh_comm_t comm, int mode);
/* one ncclGroup on `stream`: send_up -> up_peer and its row -> recv_up; same downwards; peer < 0 = frame border */
GH_API int gh_halo_exchange_dev(gh_comm_t comm, int up_peer, const void *send_up, void *recv_up, int down_peer,
                                const void *send_down, void *recv_down, size_t bytes, gh_stream_t stream);
/* in-place sum of count u64 over all ranks (channel sums of old/parallel_avg_pixel.c on a sharded frame) */
GH_API int gh_comm_allreduce_sum_u64_dev(gh_comm_t comm, uint64_t *buf_dev, size_t count, gh_stream_t stream);
/* in place: minmax_dev[0] <- min over all ranks, minmax_dev[1] <- max over all ranks (ncclMin / ncclMax on one int32 each, one group):
 * the `reduction(min:min_gray) reduction(max:max_gray)` of old/parallel_to_grayscale.c:12,36-37 on a sharded frame; every rank runs
 * gh_mean_gray_minmax_dev on its rows first (a rank without rows keeps the initial 256 / -1, the neutral elements) */
GH_API int gh_comm_allreduce_minmax_i32_dev(gh_comm_t comm, int32_t *minmax_dev, gh_stream_t stream);
/* number of halo-row ncclGroups this communicator has issued so far (diagnostic: one per sharded step, one per batch of prefetched frames) */
GH_API unsigned long long gh_comm_exchange_count(gh_comm_t comm);
/* Halo-flag waits that gave up (see gh_sharded_*_step_dev): 0 in every healthy run.  Non-zero means an exchange took longer than
 * GH_HALO_WAIT_MS (default 10000) beside a launch whose edge bands were waiting for it; that step's edge rows are invalid and every
 * later halo-flag step on this communicator fails until gh_comm_set_step_form acknowledges it (the serial and event forms keep working). */
GH_API unsigned int gh_comm_halo_timeouts(gh_comm_t comm);
/* The check that belongs behind the synchronisation which hands a sharded step's output to the host (the steps are asynchronous, a wait
 * that gives up is only known once the launch has run): GH_OK, or GH_ERR_HIP when gh_comm_halo_timeouts() is non-zero -- the output of
 * that step must not be used.  Reads host-mapped memory; costs nothing.  Callers of the sharded steps through this C API call it after
 * their hipStreamSynchronize / gh_stream_sync (the Python runners' result() does). */
GH_API int gh_comm_check(gh_comm_t comm);
/* The halo-flag form reviews itself: its edge bands report how long they stood still in front of their halo row, the communicator looks
 * at the reports every 32 launches; once the average exceeds what the serial form pays instead -- one exchange with nothing beside
 * it, timed at gh_comm_init; threshold 1.25 x that + 2 us -- it swaps the side stream for a fresh one (another hardware queue; up to twice) and
 * then, unless a form is forced with gh_comm_set_step_form, goes back to the serial form: the exchange is then not running beside the launch (seen when the runtime puts the caller's stream
 * and the side stream on hardware queues of one command-processor pipe: 73 us per step instead of 33; serial: 40).
 * gh_comm_step_form_in_use: the form the next step takes on a frame that allows the halo-flag form (2 or 1).  gh_comm_halo_wait_stats: the
 * reports themselves -- waits that had to stand still, microseconds in all. */
GH_API int gh_comm_step_form_in_use(gh_comm_t comm);
/* how many times the review has swapped the side stream for a fresh one (another hardware queue) before judging again: 0 in a healthy
 * process, at most 2 -- after that the serial form */
GH_API int gh_comm_side_stream_replacements(gh_comm_t comm);
GH_API int gh_comm_halo_wait_stats(gh_comm_t comm, unsigned int *waits, double *microseconds);
/* one halo exchange with nothing beside it, in microseconds, as timed when the communicator met its neighbours (gh_comm_init) */
GH_API double gh_comm_exchange_alone_us(gh_comm_t comm);
/* Bound of the halo-flag form's device-side waits on this communicator in milliseconds (default: GH_HALO_WAIT_MS, else 10000); -1 = default */
GH_API int gh_comm_set_halo_wait_ms(gh_comm_t comm, int ms);
/* One step of a shard: buf holds rows+2 rows (row 0 / rows+1 = halo, rows 1..rows = the shard).  Default (GH_SHARD_STEP_FORM=2, shards
 * that run an LDS-DMA ring kernel: 1024 columns and more): the one-row exchange runs on the communicator's side stream BESIDE one launch
 * over the whole shard on `stream`; only the workgroups of the first / last band wait -- inside the kernel -- for a flag set behind the
 * ncclRecv, so the step costs max(kernel, exchange + one band) and the caller's stream sees one launch and no event.  Other shards, and
 * GH_SHARD_STEP_FORM=1: the exchange on `stream`, then one launch (serial).  (Probe builds keep form 0: exchange on the side stream ||
 * interior rows, then the first and last row behind an event -- two event hand-overs and two extra launches, the slowest form measured.)
 * `stream` should be a normal-priority stream: a caller's stream of the greatest priority may share the side stream's hardware queue, and
 * the step then takes the serial form by itself.  All device-side waits are bounded (gh_comm_set_halo_wait_ms); call gh_comm_check after
 * the synchronisation that hands the output to the host.
 * Neighbours are rank-1 (above) and rank+1 (below); rank 0 / world-1 own the frame border.
 *   fused: buf = interleaved RGB8/RGBA8 rows, the halo is one RGB row (the luma plane never exists in HBM)
 *   sobel: buf = luma rows (written by gh_gray_rgb8_to_plane_dev into rows 1..rows), the halo is one luma row */
/* form: 2 halo flag, 1 serial (0 events: probe builds only), -1 = GH_SHARD_STEP_FORM, else the self-reviewing default (2 until the edge
 * bands report that the exchange does not run beside the launch, then 1).  The forms differ only in how a rank orders its own exchange and
 * kernel; ranks of one job may use different forms.  Forcing a form also acknowledges waits that gave up earlier. */
GH_API int gh_comm_set_step_form(gh_comm_t comm, int form);
GH_API int gh_sharded_fused_step_dev(gh_comm_t comm, uint8_t *rgb_buf, uint8_t *edge, int width, int rows, int channels,
                                     int mode, gh_stream_t stream);
GH_API int gh_sharded_sobel_step_dev(gh_comm_t comm, uint8_t *luma_buf, uint8_t *edge, int width, int rows, gh_stream_t stream);
/*   sobel of gray (a later pass of the multi-pass pipeline): buf = the previous pass's EDGE rows in rows 1..rows, the halo is the
 *   neighbour's raw edge row; the kernel applies lum(e, e, e) on load (gh_sobel_of_gray_u8_rows_dev) */
GH_API int gh_sharded_sobel_of_gray_step_dev(gh_comm_t comm, uint8_t *edge_buf, uint8_t *edge, int width, int rows, int mode, gh_stream_t stream);
/*   conv3x3: buf = RGB8 rows, the halo is one RGB row, dst = rows * width * 3 bytes (old/parallel_convolution.c:8-24 on a shard; the
 *   frame's own top / bottom edge keeps the zero border) */
GH_API int gh_sharded_conv3x3_step_dev(gh_comm_t comm, uint8_t *rgb_buf, uint8_t *dst, int width, int rows, const int32_t kernel[9],
                                       int32_t norm, gh_stream_t stream);
/* `passes` x (gray -> extract -> Sobel -> replicate) on a row shard, in place in rows 1..rows of img_buf ((rows + 2) rows of
 * width * channels bytes): monolithic/src/main_with_sobel.c:49-76 on a sharded frame.  Pass k + 1 filters what pass k wrote, so there
 * is ONE halo exchange per pass: an RGB row for pass 1 (fused kernel), the neighbour's raw edge row for every later pass (the plane Sobel
 * applies lum(e, e, e) on load: one launch, 2 B/px per pass).  gray_ws and edge_ws: two distinct work planes of (rows + 2) * width bytes
 * each (rows 1..rows = a plane, rows 0 / rows + 1 = its halo rows); the edge planes of successive passes alternate between them and
 * the LAST pass's edge plane is in rows 1..rows of edge_ws (edge_ws + width) on return. */
GH_API int gh_sharded_pipeline_gray_sobel_dev(gh_comm_t comm, uint8_t *img_buf, int width, int rows, int channels, int passes, int mode,
                                              uint8_t *gray_ws, uint8_t *edge_ws, gh_stream_t stream);
/* Pipelined form for a STREAM of frames kept in a few such buffers: `cur` -- whose halo rows were requested by an
 * earlier call, where it was `prefetch` -- is processed by ONE launch over the whole shard while the halo rows of
 * `prefetch` (a frame that will be `cur` in a later call; its rows 1 / rows must already be in place, enqueued on
 * `stream`) are exchanged on the side stream.  The exchange latency, the cross-queue event latency and the two one-row
 * launches of the form above leave the critical path (request the halo two frames ahead: three buffers).  If cur's
 * halo was never requested (first calls, buffer changed) it is exchanged first; prefetch may be NULL or equal to cur.
 * At most 4 single requests are remembered. */
GH_API int gh_sharded_fused_step_pipelined_dev(gh_comm_t comm, uint8_t *cur_rgb_buf, uint8_t *prefetch_rgb_buf, uint8_t *edge,
                                               int width, int rows, int channels, int mode, gh_stream_t stream);
GH_API int gh_sharded_sobel_step_pipelined_dev(gh_comm_t comm, uint8_t *cur_luma_buf, uint8_t *prefetch_luma_buf, uint8_t *edge,
                                               int width, int rows, gh_stream_t stream);
/* The same stream of frames in the halo-flag way (the form to use on frames of 1024 columns and more): `cur` is filtered by one launch
 * whose edge bands wait -- inside the kernel, in front of their halo row -- for the exchange the PREVIOUS call started when cur was its
 * `next`; the halo rows of `next` (rows 1 / rows in place, enqueued on `stream`) travel on the side stream beside this launch.  No event
 * and no stream-level wait: an exchange has a whole launch to complete before anything needs it.  cur not announced by the previous call
 * (first call, another buffer): exchanged beside this launch, as in gh_sharded_fused_step_dev.  next may be NULL.  Narrow frames fall
 * back to gh_sharded_fused_step_pipelined_dev. */
GH_API int gh_sharded_fused_stream_step_dev(gh_comm_t comm, uint8_t *cur_rgb_buf, uint8_t *next_rgb_buf, uint8_t *edge, int width, int rows,
                                            int channels, int mode, gh_stream_t stream);
/* the same for the split pipeline: buffers of luma rows (written by gh_gray_rgb8_to_plane_dev into rows 1..rows), the halo is one luma row */
GH_API int gh_sharded_sobel_stream_step_dev(gh_comm_t comm, uint8_t *cur_luma_buf, uint8_t *next_luma_buf, uint8_t *edge, int width, int rows,
                                            gh_stream_t stream);
/* Batched prefetch for the pipelined steps: the halo rows of n (1..16) frame buffers -- (rows + 2) rows of row_bytes each, rows
 * 1 / rows in place (enqueued on `stream`) -- are exchanged in ONE ncclGroup on the side stream, one event behind all of them.
 * The pipelined steps that later run on these buffers (prefetch = NULL) find their halos requested: the first waits for the
 * event, the others launch their kernel and nothing else.  One exchange per n frames instead of one per frame: fewer, larger
 * collectives and fewer cross-queue hand-overs per step.  Up to 4 batches are remembered; all calls on one stream. */
GH_API int gh_sharded_prefetch_halos_dev(gh_comm_t comm, uint8_t *const *bufs, int n, size_t row_bytes, int rows, gh_stream_t stream);
/* Forgets every halo request that has not been consumed by a step.  The requests are remembered by buffer ADDRESS: a caller that
 * drops prefetched buffers and later puts other frames at the same addresses calls this in between (a remembered request would
 * otherwise be taken for the new frame's). */
GH_API int gh_sharded_prefetch_reset(gh_comm_t comm);

/* ---- the reference's historic entry points over `long` samples (old/parallel_*.c) --------------------------
 * Same argument meaning as the originals; the dimensions, compile-time macros of a utils.h that the reference does not
 * contain, are run-time arguments (DIM_RGB = 3, DIM_KERNEL = 3, PAD = 2), and a status is returned.
 *   img            row-major [rows][cols][3] longs (convolution: the PRE-PADDED [rows+2][cols+2][3] array; output (r, c)
 *                  reads input rows r..r+2 and columns c..c+2)                         old/parallel_convolution.c:8,17
 *   convolved_img  pointer tree [rows][cols] -> long[3]; the weighted sum is ADDED to what is there, then divided by
 *                  gblur_norm with C integer division                                  old/parallel_convolution.c:17,21
 *   avgs           long[3], accumulated into: avgs[c] = (avgs[c] + 3 * sum_c) / (rows * cols * 3)   old/parallel_avg_pixel.c:17-41
 *   grayscale_img  pointer tree, R = G = B = (int)(r + g + b) / 3; min_max_gray = {min(256, ..), max(-1, ..)}
 *                                                                                      old/parallel_to_grayscale.c:9-10,20-35
 * Byte-valued frames run on the hot-path u8 kernels (narrowed on upload), anything else that fits 30 bits on plain 64-bit
 * kernels; results equal the reference's arithmetic either way.  Parity status: unpinned (the old/ sources cannot be built here). */
GH_API int gh_old_parallel_convolution(const long *img, const long kernel[9], long ***convolved_img, int rows, int cols, long gblur_norm);
GH_API int gh_old_parallel_avg_pixel(const long *img, long *avgs, int rows, int cols);
GH_API int gh_old_parallel_to_grayscale(const long *img, long ***grayscale_img, long *min_max_gray, int rows, int cols);

/* The reference's own symbols (same prototypes as monolithic/include/ headers).  They return void in
 * the reference, so a failure is reported on stderr and the process is aborted -- never a
 * silent CPU fallback.  mode = gh_default_mode(). */
GH_API void convert_to_grayscale(unsigned char *data, int width, int height, int channels);
GH_API void sobel_edge(const unsigned char *src, unsigned char *dst, int width, int height);

/* ---- self tests that need the device (used by tests/ through the C ABI) ------------------- */
/* Runs every reachable Sobel gradient pair (gx, gy) in [-1020, 1020]^2 through one kernel family's
 * magnitude mapping on the device and counts results that differ from min(255, isqrt(gx^2 + gy^2)).
 * variant 0: strip-streaming packed-i16 core; 1: strip-streaming f32 core (fused kernel);
 * 2: LDS-tile / per-pixel kernels. */
GH_API int gh_selftest_sobel_mag(int variant, uint64_t *mismatches);

#ifdef __cplusplus
}
#endif
#endif
