// streamed.hip -- band-streamed gray+Sobel of a host-resident frame (BASELINE configs[4]).
//
// The reference holds the whole frame in RAM and runs its four OpenMP loops over it
// (monolithic/src/main_with_sobel.c:49-76).  Here the frame stays in host memory and flows through the GPU in
// bands of `tile_rows` rows: copy engine (H2D), compute (the fused RGB->luma->Sobel kernel on the band, halo rows
// included) and copy engine (D2H) run concurrently on three hipStreams over a 3-slot ring of device buffers, ordered
// by events only.  Each band carries its own RGB halo row above and below (re-sent: 2/tile_rows extra PCIe bytes), so
// bands are independent and restartable.  Bound: PCIe (3 B/px in + 1 B/px out per direction-shared link), not HBM.
#include "gh_common.h"

#include <chrono>

namespace {

constexpr int NSLOT = 3;

struct Ring {
    uint8_t *rgb[NSLOT] = {nullptr, nullptr, nullptr};
    uint8_t *edge[NSLOT] = {nullptr, nullptr, nullptr};
    hipEvent_t uploaded[NSLOT] = {}, computed[NSLOT] = {}, downloaded[NSLOT] = {};
    hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
    bool reg_in = false, reg_out = false;
    const void *reg_in_ptr = nullptr;
    void *reg_out_ptr = nullptr;

    ~Ring()
    {
        for (int i = 0; i < NSLOT; ++i) {
            if (rgb[i]) (void)hipFree(rgb[i]);
            if (edge[i]) (void)hipFree(edge[i]);
            if (uploaded[i]) (void)hipEventDestroy(uploaded[i]);
            if (computed[i]) (void)hipEventDestroy(computed[i]);
            if (downloaded[i]) (void)hipEventDestroy(downloaded[i]);
        }
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_comp) (void)hipStreamDestroy(s_comp);
        if (s_out) (void)hipStreamDestroy(s_out);
        if (reg_in) (void)hipHostUnregister(const_cast<void *>(reg_in_ptr));
        if (reg_out) (void)hipHostUnregister(reg_out_ptr);
    }
};

}  // namespace

extern "C" int gh_gray_sobel_streamed_host(const unsigned char *rgb, unsigned char *edge, int width, int height, int channels,
                                           int mode, int tile_rows, int host_buffers_pinned, double *seconds)
{
    return gh_gray_sobel_streamed_rows_host(rgb, edge, width, height, channels, mode, tile_rows, host_buffers_pinned, 1, 1, seconds);
}

// The same pipeline on a block of `height` rows of a taller frame (one rank's share when a host-resident frame is
// split over the GPUs of a node: bands are independent, no communication).  When a side is not the frame border the
// RGB row just outside the block is read from host memory next to it (rgb - pitch / rgb + height * pitch).
extern "C" int gh_gray_sobel_streamed_rows_host(const unsigned char *rgb, unsigned char *edge, int width, int height, int channels,
                                                int mode, int tile_rows, int host_buffers_pinned, int top_is_border, int bottom_is_border,
                                                double *seconds)
{
    GH_REQUIRE(width >= 0 && height >= 0, "negative size");
    GH_REQUIRE(channels == 3 || channels == 4, "streamed pipeline takes RGB8 or RGBA8");
    GH_REQUIRE(mode == GH_MODE_IEEE || mode == GH_MODE_FMA, "bad mode");
    GH_REQUIRE(tile_rows >= 1, "tile_rows must be >= 1");
    if (seconds) *seconds = 0.0;
    if (width == 0 || height == 0) return GH_OK;
    GH_REQUIRE(rgb && edge, "null pointer");
    if (int rc = gh_ensure_init()) return rc;

    const size_t row_in = (size_t)width * channels, row_out = (size_t)width;
    if (tile_rows > height) tile_rows = height;
    const int ntiles = (height + tile_rows - 1) / tile_rows;

    Ring r;
    if (!host_buffers_pinned) {
        const unsigned char *reg0 = rgb - (top_is_border ? 0 : row_in);
        const size_t reg_rows = (size_t)height + (top_is_border ? 0 : 1) + (bottom_is_border ? 0 : 1);
        GH_HIP_TRY(hipHostRegister(const_cast<unsigned char *>(reg0), row_in * reg_rows, hipHostRegisterDefault));
        r.reg_in = true; r.reg_in_ptr = reg0;
        GH_HIP_TRY(hipHostRegister(edge, row_out * height, hipHostRegisterDefault));
        r.reg_out = true; r.reg_out_ptr = edge;
    }
    GH_HIP_TRY(hipStreamCreateWithFlags(&r.s_in, hipStreamNonBlocking));
    GH_HIP_TRY(hipStreamCreateWithFlags(&r.s_comp, hipStreamNonBlocking));
    GH_HIP_TRY(hipStreamCreateWithFlags(&r.s_out, hipStreamNonBlocking));
    for (int i = 0; i < NSLOT; ++i) {
        GH_HIP_TRY(hipMalloc(&r.rgb[i], row_in * (tile_rows + 2)));
        GH_HIP_TRY(hipMalloc(&r.edge[i], row_out * tile_rows));
        GH_HIP_TRY(hipEventCreateWithFlags(&r.uploaded[i], hipEventDisableTiming));
        GH_HIP_TRY(hipEventCreateWithFlags(&r.computed[i], hipEventDisableTiming));
        GH_HIP_TRY(hipEventCreateWithFlags(&r.downloaded[i], hipEventDisableTiming));
    }

    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < ntiles; ++k) {
        const int slot = k % NSLOT;
        const int y0 = k * tile_rows;
        const int rows = (y0 + tile_rows <= height) ? tile_rows : height - y0;
        const bool top = y0 == 0 && top_is_border, bot = y0 + rows == height && bottom_is_border;
        // the slot's previous tenant (tile k - NSLOT) must have been computed (input buffer free) and downloaded (output free)
        if (k >= NSLOT) {
            GH_HIP_TRY(hipStreamWaitEvent(r.s_in, r.computed[slot], 0));
            GH_HIP_TRY(hipStreamWaitEvent(r.s_comp, r.downloaded[slot], 0));
        }
        // H2D: band rows plus the halo rows that exist; device layout = (rows + 2) rows, band at row 1
        const int first = top ? y0 : y0 - 1, last = bot ? y0 + rows : y0 + rows + 1;   // [first, last)
        GH_HIP_TRY(hipMemcpyAsync(r.rgb[slot] + (size_t)(first - (y0 - 1)) * row_in, rgb + (size_t)first * row_in,
                                  (size_t)(last - first) * row_in, hipMemcpyHostToDevice, r.s_in));
        GH_HIP_TRY(hipEventRecord(r.uploaded[slot], r.s_in));
        // compute
        GH_HIP_TRY(hipStreamWaitEvent(r.s_comp, r.uploaded[slot], 0));
        if (int rc = gh_gray_sobel_fused_rows_dev(r.rgb[slot] + row_in, r.edge[slot], width, rows, channels, mode, top ? 1 : 0,
                                                  bot ? 1 : 0, r.s_comp))
            return rc;
        GH_HIP_TRY(hipEventRecord(r.computed[slot], r.s_comp));
        // D2H
        GH_HIP_TRY(hipStreamWaitEvent(r.s_out, r.computed[slot], 0));
        GH_HIP_TRY(hipMemcpyAsync(edge + (size_t)y0 * row_out, r.edge[slot], (size_t)rows * row_out, hipMemcpyDeviceToHost, r.s_out));
        GH_HIP_TRY(hipEventRecord(r.downloaded[slot], r.s_out));
    }
    GH_HIP_TRY(hipStreamSynchronize(r.s_out));
    GH_HIP_TRY(hipStreamSynchronize(r.s_comp));
    GH_HIP_TRY(hipStreamSynchronize(r.s_in));
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return GH_OK;
}
