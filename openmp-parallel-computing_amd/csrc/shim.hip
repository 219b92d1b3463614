// shim.hip -- runtime, error reporting and the HOST-pointer side of the C ABI (include/grayhip.h):
// the reference's own symbols convert_to_grayscale() / sobel_edge() and the CLI's per-pass
// pipelines, implemented as H2D -> gfx950 kernels -> D2H.  No CPU fallback: if HIP or the GPU is
// missing every entry point fails (the two void reference symbols print to stderr and abort).
//
// Reference call sites this file stands in for:
//   monolithic/src/main.c:33-35             passes x convert_to_grayscale        -> gh_gray_passes_host
//   monolithic/src/main_with_sobel.c:51-74  passes x (gray, extract, sobel, replicate) -> gh_pipeline_gray_sobel_*
#include "gh_common.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>

namespace {

thread_local char g_err[768] = "";
thread_local const char *g_last_kernel = "";
std::once_flag g_init_once;
int g_init_rc = GH_OK;
int g_device = -1;
int g_num_cu = GH_REF_NUM_CU, g_num_xcd = GH_REF_NUM_XCD;
char g_init_err[768] = "";     // why initialisation failed, for threads other than the one that tried

// grow-only device scratch for the host-pointer entry points (caller threads are serialised)
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return GH_OK;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return gh_set_error(GH_ERR_HIP, "hipFree: %s", hipGetErrorString(e)); }
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { p = nullptr; return gh_set_error(GH_ERR_HIP, "hipMalloc(%zu): %s", n, hipGetErrorString(e)); }
        cap = n;
        return GH_OK;
    }
};
Scratch g_img, g_gray, g_edge;
std::mutex g_host_mu;


int do_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return gh_set_error(GH_ERR_NO_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return gh_set_error(GH_ERR_ARG, "device %d out of range (have %d)", device, n);
    hipDeviceProp_t prop;
    GH_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return gh_set_error(GH_ERR_NO_DEVICE, "device %d is %s; libgrayhip carries gfx950 code only", device, prop.gcnArchName);
    GH_HIP_TRY(hipSetDevice(device));
    GH_HIP_TRY(hipFree(nullptr));   // force context creation
    g_device = device;
    // topology for the launch rules (gh_common.h): what the device reports, not constants
    g_num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : GH_REF_NUM_CU;
    int xcc = 0;
    g_num_xcd = (hipDeviceGetAttribute(&xcc, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && xcc > 0) ? xcc : GH_REF_NUM_XCD;
    (void)hipGetLastError();
    return gh_sobel_init_tables(device);
}

// The two streams of the chunked host pipelines (upload | kernel + download), created once per process.  Both or neither: a call
// that found only one of them would run its other half on the NULL (blocking) stream without any error.
int host_pipeline_streams(hipStream_t *up, hipStream_t *dn)
{
    static hipStream_t s_up = nullptr, s_dn = nullptr;       // callers hold g_host_mu
    if (!s_up || !s_dn) {
        hipStream_t a = nullptr, b = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
        if (e != hipSuccess) {
            if (a) (void)hipStreamDestroy(a);
            return gh_set_error(GH_ERR_HIP, "hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
        }
        s_up = a;
        s_dn = b;
    }
    *up = s_up;
    *dn = s_dn;
    return GH_OK;
}

struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    int start(hipStream_t s)
    {
        GH_HIP_TRY(hipEventCreate(&a));
        GH_HIP_TRY(hipEventCreate(&b));
        GH_HIP_TRY(hipEventRecord(a, s));
        return GH_OK;
    }
    int stop(hipStream_t s, double *seconds)
    {
        GH_HIP_TRY(hipEventRecord(b, s));
        GH_HIP_TRY(hipEventSynchronize(b));
        float ms = 0.f;
        GH_HIP_TRY(hipEventElapsedTime(&ms, a, b));
        if (seconds) *seconds = ms * 1e-3;
        return GH_OK;
    }
    ~EventTimer()
    {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

}  // namespace

void gh_note_kernel(const char *name) { g_last_kernel = name; }

int gh_num_cu()
{
#ifdef GH_ENABLE_PROBES
    static const int forced = [] { const char *e = getenv("GH_NUM_CU_OVERRIDE"); return e ? atoi(e) : 0; }();      // launch-rule experiments
    if (forced > 0) return forced;
#endif
    return g_num_cu;
}
int gh_num_xcd() { return g_num_xcd; }

int gh_set_error(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    if (getenv("GRAYHIP_VERBOSE")) fprintf(stderr, "grayhip: %s\n", g_err);
    return status;
}

int gh_ensure_init()
{
    std::call_once(g_init_once, [] {
        const char *e = getenv("GRAYHIP_DEVICE");
        g_init_rc = do_init(e ? atoi(e) : 0);
        if (g_init_rc != GH_OK) snprintf(g_init_err, sizeof g_init_err, "%s", g_err);
    });
    if (g_init_rc != GH_OK && g_err[0] == 0)
        gh_set_error(g_init_rc, "%s", g_init_err[0] ? g_init_err : "HIP initialisation failed earlier in this process");
    // hipSetDevice is per host thread: a thread other than the one that initialised the library would otherwise launch on
    // device 0 (the CLIs initialise HIP on a helper thread while the main thread decodes the input image)
    static thread_local int t_device = -1;
    if (g_init_rc == GH_OK && t_device != g_device) {
        if (hipSetDevice(g_device) != hipSuccess) return gh_set_error(GH_ERR_HIP, "hipSetDevice(%d) failed on this thread", g_device);
        t_device = g_device;
    }
    return g_init_rc;
}

extern "C" {

int gh_init(int device)
{
    bool first = false;
    std::call_once(g_init_once, [&] {
        first = true;
        g_init_rc = do_init(device);
        if (g_init_rc != GH_OK) snprintf(g_init_err, sizeof g_init_err, "%s", g_err);
    });
    if (first || g_init_rc != GH_OK) return g_init_rc;
    if (device != g_device) {   // re-target an already initialised process
        int rc = do_init(device);
        if (rc != GH_OK) return rc;
    }
    return GH_OK;
}

int gh_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

const char *gh_last_error(void) { return g_err; }
const char *gh_last_kernel(void) { return g_last_kernel; }

const char *gh_version(void) { return "grayhip 0.1 (gfx950)"; }

int gh_device_topology(int *compute_units, int *xcds)
{
    if (int rc = gh_ensure_init()) return rc;
    if (compute_units) *compute_units = gh_num_cu();
    if (xcds) *xcds = gh_num_xcd();
    return GH_OK;
}

int gh_default_mode(void)
{
    const char *e = getenv("GRAYHIP_MODE");
    return (e && (strcmp(e, "fma") == 0 || strcmp(e, "FMA") == 0)) ? GH_MODE_FMA : GH_MODE_IEEE;
}

int gh_malloc(void **dptr, size_t nbytes)
{
    GH_REQUIRE(dptr, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    GH_HIP_TRY(hipMalloc(dptr, nbytes ? nbytes : 1));
    return GH_OK;
}

int gh_free(void *dptr)
{
    if (!dptr) return GH_OK;
    GH_HIP_TRY(hipFree(dptr));
    return GH_OK;
}

int gh_memset_dev(void *dptr, int value, size_t nbytes, gh_stream_t stream)
{
    if (nbytes == 0) return GH_OK;
    GH_REQUIRE(dptr, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    GH_HIP_TRY(hipMemsetAsync(dptr, value, nbytes, (hipStream_t)stream));
    return GH_OK;
}

int gh_copy_h2d(void *dst_dev, const void *src_host, size_t nbytes, gh_stream_t stream)
{
    if (nbytes == 0) return GH_OK;
    GH_REQUIRE(dst_dev && src_host, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    GH_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, nbytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return GH_OK;
}

int gh_copy_d2h(void *dst_host, const void *src_dev, size_t nbytes, gh_stream_t stream)
{
    if (nbytes == 0) return GH_OK;
    GH_REQUIRE(dst_host && src_dev, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    GH_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, nbytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return GH_OK;
}

int gh_stream_sync(gh_stream_t stream)
{
    if (int rc = gh_ensure_init()) return rc;
    GH_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GH_OK;
}

// ---- device-resident pipeline ------------------------------------------------------------------

int gh_pipeline_gray_sobel_dev(uint8_t *img, int width, int height, int channels, int passes, int mode, uint8_t *gray_ws,
                               uint8_t *edge_ws, gh_stream_t stream)
{
    GH_REQUIRE(width >= 0 && height >= 0, "negative size");
    GH_REQUIRE(channels >= 1 && channels <= 4, "channels must be 1..4");
    if (passes < 1) passes = 1;   // main_with_sobel.c:44-45
    const size_t npix = (size_t)width * height;
    if (npix == 0) return GH_OK;
    GH_REQUIRE(img && gray_ws && edge_ws, "null pointer");
    // pass p (0-based) writes plane[(passes - 1 - p) & 1]: the planes alternate and the last pass lands in edge_ws
    uint8_t *const plane[2] = {edge_ws, gray_ws};
    GH_REQUIRE(gray_ws != edge_ws, "gray_ws and edge_ws must differ");
    for (int p = 0; p < passes; ++p) {
        uint8_t *out = plane[(passes - 1 - p) & 1], *in = plane[(passes - p) & 1];
        int rc;
        if (p == 0 && channels >= 3) {
            // pass 1 in ONE kernel: RGB -> luma -> Sobel (the in-place luma write of the reference is dead: replicate overwrites
            // R,G,B; the luma plane is not needed afterwards -- later passes start from the edge plane)
            rc = gh_gray_sobel_fused_rows_dev(img, out, width, height, channels, mode, 1, 1, stream);
        } else if (p == 0) {                       // channels 1 / 2: the plane is channel 0, and every pass filters the plane as it is
            if ((rc = gh_gray_rgb8_to_plane_dev(img, in, npix, channels, mode, stream))) return rc;
            rc = gh_sobel_u8_dev(in, out, width, height, stream);
        } else if (channels >= 3) {
            // the image now holds (e,e,e): its luma is lum(e,e,e), not e (13 grey levels move in IEEE mode) -- applied on load by the
            // plane Sobel: one launch, 1 R + 1 W byte per pixel per pass (until round 4: a luma pass and a Sobel pass, 4 B/px)
            rc = gh_sobel_of_gray_u8_dev(in, out, width, height, mode, stream);
        } else {
            rc = gh_sobel_u8_dev(in, out, width, height, stream);
        }
        if (rc) return rc;
    }
    return gh_replicate_plane_dev(edge_ws, img, npix, channels, stream);
}

// ---- host-pointer entry points -----------------------------------------------------------------

int gh_gray_passes_host(unsigned char *data, int width, int height, int channels, int passes, int mode, double *kernel_seconds)
{
    GH_REQUIRE(width >= 0 && height >= 0, "negative size");
    GH_REQUIRE(channels >= 1 && channels <= 4, "channels must be 1..4");
    if (kernel_seconds) *kernel_seconds = 0.0;
    if (passes < 1) passes = 1;   // main.c:27-28
    const size_t npix = (size_t)width * height, nbytes = npix * channels;
    if (npix == 0 || channels < 3) return GH_OK;
    GH_REQUIRE(data, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (int rc = g_img.reserve(nbytes)) return rc;
    // Large frames: the pixels are independent, so the frame goes through in chunks -- an uploader thread copies chunk k+1
    // host -> device while this thread runs the kernel on chunk k and copies it back: PCIe is full duplex, the two directions
    // overlap (16384^2 RGB8: 29 ms -> see DESIGN.md section 5).  Small frames: one upload, one launch, one download.
    const size_t chunk_px = (size_t)GH_LAB_INT("GH_HOST_CHUNK_MPX", 16) << 20;          // multiple of 1024 px: whole wavefront groups
    if (nbytes >= ((size_t)GH_LAB_INT("GH_HOST_PIPELINE_MIN_MB", 64) << 20) && chunk_px > 0 && npix > 2 * chunk_px) {
        const int nchunk = (int)((npix + chunk_px - 1) / chunk_px);
        hipStream_t s_up = nullptr, s_dn = nullptr;
        if (int rc = host_pipeline_streams(&s_up, &s_dn)) return rc;
        std::vector<hipEvent_t> up(nchunk, nullptr), k0(nchunk, nullptr), k1(nchunk, nullptr);
        auto cleanup = [&] { for (auto *v : {&up, &k0, &k1}) for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e); };
        for (int k = 0; k < nchunk; ++k)
            if (hipEventCreateWithFlags(&up[k], hipEventDisableTiming) != hipSuccess || hipEventCreate(&k0[k]) != hipSuccess || hipEventCreate(&k1[k]) != hipSuccess) {
                cleanup();
                return gh_set_error(GH_ERR_HIP, "hipEventCreate failed");
            }
        uint8_t *dev = (uint8_t *)g_img.p;
        std::atomic<int> recorded{0};                 // chunks whose upload event has been RECORDED (a wait on an unrecorded event is a no-op)
        std::atomic<int> up_failed{0};
        hipError_t up_err = hipSuccess;
        std::thread uploader([&] {
            hipError_t ue = hipSetDevice(g_device);
            for (int k = 0; k < nchunk && ue == hipSuccess; ++k) {
                const size_t p0 = (size_t)k * chunk_px, n = std::min(chunk_px, npix - p0);
                ue = hipMemcpyAsync(dev + p0 * channels, data + p0 * channels, n * channels, hipMemcpyHostToDevice, s_up);
                if (ue == hipSuccess) ue = hipEventRecord(up[k], s_up);
                if (ue == hipSuccess) recorded.store(k + 1, std::memory_order_release);
            }
            if (ue != hipSuccess) { up_err = ue; up_failed.store(1, std::memory_order_release); }
        });
        int rc = GH_OK;
        hipError_t e = hipSuccess;
        int issued = 0;
        for (int k = 0; k < nchunk && rc == GH_OK && e == hipSuccess; ++k) {
            const size_t p0 = (size_t)k * chunk_px, n = std::min(chunk_px, npix - p0);
            while (recorded.load(std::memory_order_acquire) <= k && !up_failed.load(std::memory_order_acquire)) std::this_thread::yield();
            if (up_failed.load(std::memory_order_acquire)) break;
            e = hipStreamWaitEvent(s_dn, up[k], 0);
            if (e != hipSuccess) break;
            e = hipEventRecord(k0[k], s_dn);
            if (e != hipSuccess) break;
            rc = gh_gray_inplace_passes_dev(dev + p0 * channels, n, channels, mode, passes, s_dn);
            if (rc == GH_OK) e = hipEventRecord(k1[k], s_dn);
            if (rc == GH_OK && e == hipSuccess) e = hipMemcpyAsync(data + p0 * channels, dev + p0 * channels, n * channels, hipMemcpyDeviceToHost, s_dn);
            issued = k + 1;
        }
        uploader.join();
        if (e == hipSuccess) e = hipStreamSynchronize(s_dn);
        (void)hipStreamSynchronize(s_up);
        double secs = 0.0;
        for (int k = 0; k < issued && e == hipSuccess; ++k) { float ms = 0.f; if (hipEventElapsedTime(&ms, k0[k], k1[k]) == hipSuccess) secs += ms * 1e-3; }
        if (kernel_seconds) *kernel_seconds = secs;
        cleanup();
        if (rc != GH_OK) return rc;
        if (up_failed.load()) return gh_set_error(GH_ERR_HIP, "chunked upload failed: %s", hipGetErrorString(up_err));
        if (e != hipSuccess) return gh_set_error(GH_ERR_HIP, "chunked gray pipeline failed: %s", hipGetErrorString(e));
        return GH_OK;
    }
    GH_HIP_TRY(hipMemcpy(g_img.p, data, nbytes, hipMemcpyHostToDevice));
    EventTimer t;
    if (int rc = t.start(nullptr)) return rc;
    // all passes in one launch: they are independent per pixel (gh_gray_inplace_passes_dev)
    if (int rc = gh_gray_inplace_passes_dev((uint8_t *)g_img.p, npix, channels, mode, passes, nullptr)) return rc;
    if (int rc = t.stop(nullptr, kernel_seconds)) return rc;
    GH_HIP_TRY(hipMemcpy(data, g_img.p, nbytes, hipMemcpyDeviceToHost));
    return GH_OK;
}

int gh_convert_to_grayscale_host(unsigned char *data, int width, int height, int channels, int mode)
{
    return gh_gray_passes_host(data, width, height, channels, 1, mode, nullptr);
}

int gh_sobel_edge_host(const unsigned char *src, unsigned char *dst, int width, int height)
{
    GH_REQUIRE(width >= 0 && height >= 0, "negative size");
    if (width < 3 || height < 3) return GH_OK;   // no interior: the reference writes nothing
    GH_REQUIRE(src && dst && src != dst, "null or aliased pointers");
    if (int rc = gh_ensure_init()) return rc;
    const size_t npix = (size_t)width * height;
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (int rc = g_gray.reserve(npix)) return rc;
    if (int rc = g_edge.reserve(npix)) return rc;
    // dst's border must not change (sobel.c:11-12 never writes it).  A 2-D copy of the interior does that by itself but runs at
    // a quarter of the PCIe rate (16382-byte rows: 20 ms for 268 MB), so whole rows 1 .. h-2 are copied back in one piece per
    // chunk and the caller's first and last column are put back afterwards (2 bytes per row, saved here).
    std::vector<unsigned char> edge_cols(2 * (size_t)height);
    for (int y = 1; y < height - 1; ++y) { edge_cols[2 * y] = dst[(size_t)y * width]; edge_cols[2 * y + 1] = dst[(size_t)y * width + width - 1]; }
    auto restore_cols = [&] {
        for (int y = 1; y < height - 1; ++y) { dst[(size_t)y * width] = edge_cols[2 * y]; dst[(size_t)y * width + width - 1] = edge_cols[2 * y + 1]; }
    };
    // Large planes go through in row chunks like the gray driver above: an uploader thread copies chunk k+1 while this thread
    // filters the rows that chunk k completed -- output rows [kB - 1, (k+1)B - 1) need input rows up to (k+1)B - 1, i.e. chunks
    // 0..k -- and copies their interior back; both PCIe directions and the kernel overlap.
    const size_t chunk_px = (size_t)GH_LAB_INT("GH_HOST_CHUNK_MPX", 16) << 20;
    const int B = chunk_px ? (int)std::max<size_t>(64, chunk_px / (size_t)width) : 0;      // rows per chunk
    if (npix >= ((size_t)GH_LAB_INT("GH_HOST_PIPELINE_MIN_MB", 64) << 20) && B > 0 && height > 2 * B) {
        const int nchunk = (height + B - 1) / B;
        hipStream_t s_up = nullptr, s_dn = nullptr;
        if (int rc = host_pipeline_streams(&s_up, &s_dn)) return rc;
        std::vector<hipEvent_t> up(nchunk, nullptr);
        auto cleanup = [&] { for (hipEvent_t e : up) if (e) (void)hipEventDestroy(e); };
        for (int k = 0; k < nchunk; ++k)
            if (hipEventCreateWithFlags(&up[k], hipEventDisableTiming) != hipSuccess) { cleanup(); return gh_set_error(GH_ERR_HIP, "hipEventCreate failed"); }
        const uint8_t *din = (const uint8_t *)g_gray.p;
        uint8_t *dout = (uint8_t *)g_edge.p;
        std::atomic<int> recorded{0}, up_failed{0};
        hipError_t up_err = hipSuccess;
        std::thread uploader([&] {
            hipError_t ue = hipSetDevice(g_device);
            for (int k = 0; k < nchunk && ue == hipSuccess; ++k) {
                const size_t r0 = (size_t)k * B, n = std::min((size_t)B, (size_t)height - r0);
                ue = hipMemcpyAsync((uint8_t *)g_gray.p + r0 * width, src + r0 * width, n * width, hipMemcpyHostToDevice, s_up);
                if (ue == hipSuccess) ue = hipEventRecord(up[k], s_up);
                if (ue == hipSuccess) recorded.store(k + 1, std::memory_order_release);
            }
            if (ue != hipSuccess) { up_err = ue; up_failed.store(1, std::memory_order_release); }
        });
        int rc = GH_OK;
        hipError_t e = hipSuccess;
        for (int k = 0; k < nchunk && rc == GH_OK && e == hipSuccess; ++k) {
            while (recorded.load(std::memory_order_acquire) <= k && !up_failed.load(std::memory_order_acquire)) std::this_thread::yield();
            if (up_failed.load(std::memory_order_acquire)) break;
            e = hipStreamWaitEvent(s_dn, up[k], 0);
            if (e != hipSuccess) break;
            const int last = k == nchunk - 1;
            const int y0 = k == 0 ? 0 : k * B - 1, y1 = last ? height : (k + 1) * B - 1;       // output rows [y0, y1)
            rc = gh_sobel_u8_rows_dev(din + (size_t)y0 * width, dout + (size_t)y0 * width, width, y1 - y0, k == 0, last, s_dn);
            const int c0 = std::max(y0, 1), c1 = std::min(y1, height - 1);                      // never rows 0 and h-1
            if (rc == GH_OK && c1 > c0)
                e = hipMemcpyAsync(dst + (size_t)c0 * width, dout + (size_t)c0 * width, (size_t)(c1 - c0) * width, hipMemcpyDeviceToHost, s_dn);
        }
        uploader.join();
        if (e == hipSuccess) e = hipStreamSynchronize(s_dn);
        (void)hipStreamSynchronize(s_up);
        restore_cols();
        cleanup();
        if (rc != GH_OK) return rc;
        if (up_failed.load()) return gh_set_error(GH_ERR_HIP, "chunked upload failed: %s", hipGetErrorString(up_err));
        if (e != hipSuccess) return gh_set_error(GH_ERR_HIP, "chunked sobel pipeline failed: %s", hipGetErrorString(e));
        return GH_OK;
    }
    GH_HIP_TRY(hipMemcpy(g_gray.p, src, npix, hipMemcpyHostToDevice));
    if (int rc = gh_sobel_u8_dev((const uint8_t *)g_gray.p, (uint8_t *)g_edge.p, width, height, nullptr)) return rc;
    const hipError_t ce = hipMemcpy(dst + width, (const uint8_t *)g_edge.p + width, (size_t)(height - 2) * width, hipMemcpyDeviceToHost);
    restore_cols();
    if (ce != hipSuccess) return gh_set_error(GH_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(ce));
    return GH_OK;
}

int gh_pipeline_gray_sobel_host(unsigned char *img, int width, int height, int channels, int passes, int mode, double *kernel_seconds)
{
    GH_REQUIRE(width >= 0 && height >= 0, "negative size");
    GH_REQUIRE(channels >= 1 && channels <= 4, "channels must be 1..4");
    if (kernel_seconds) *kernel_seconds = 0.0;
    const size_t npix = (size_t)width * height, nbytes = npix * channels;
    if (npix == 0) return GH_OK;
    GH_REQUIRE(img, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (int rc = g_img.reserve(nbytes)) return rc;
    if (int rc = g_gray.reserve(npix)) return rc;
    if (int rc = g_edge.reserve(npix)) return rc;
    GH_HIP_TRY(hipMemcpy(g_img.p, img, nbytes, hipMemcpyHostToDevice));
    // (Replaying passes 2..N as a hipGraph was tried: 15.3 ms vs 15.2 ms for 1000 passes on a 1080p frame -- the cost is
    //  the ~7 us dispatch of each dependent kernel on the GPU side, not the host-side launches, so plain launches stay.)
    EventTimer t;
    if (int rc = t.start(nullptr)) return rc;
    if (int rc = gh_pipeline_gray_sobel_dev((uint8_t *)g_img.p, width, height, channels, passes, mode, (uint8_t *)g_gray.p,
                                            (uint8_t *)g_edge.p, nullptr))
        return rc;
    if (int rc = t.stop(nullptr, kernel_seconds)) return rc;
    GH_HIP_TRY(hipMemcpy(img, g_img.p, nbytes, hipMemcpyDeviceToHost));
    return GH_OK;
}

int gh_sum_rgb8_host(const unsigned char *rgb, size_t npix, uint64_t sums[3])
{
    GH_REQUIRE(sums, "null pointer");
    sums[0] = sums[1] = sums[2] = 0;
    if (npix == 0) return GH_OK;
    GH_REQUIRE(rgb, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (int rc = g_img.reserve(npix * 3)) return rc;
    if (int rc = g_gray.reserve(64)) return rc;
    GH_HIP_TRY(hipMemcpy(g_img.p, rgb, npix * 3, hipMemcpyHostToDevice));
    GH_HIP_TRY(hipMemset(g_gray.p, 0, 3 * sizeof(uint64_t)));
    if (int rc = gh_sum_rgb8_dev((const uint8_t *)g_img.p, npix, (uint64_t *)g_gray.p, nullptr)) return rc;
    GH_HIP_TRY(hipMemcpy(sums, g_gray.p, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return GH_OK;
}

// ---- the reference's own symbols ---------------------------------------------------------------

void convert_to_grayscale(unsigned char *data, int width, int height, int channels)
{
    if (gh_convert_to_grayscale_host(data, width, height, channels, gh_default_mode()) != GH_OK) {
        fprintf(stderr, "grayhip: convert_to_grayscale failed: %s\n", gh_last_error());
        abort();
    }
}

void sobel_edge(const unsigned char *src, unsigned char *dst, int width, int height)
{
    if (gh_sobel_edge_host(src, dst, width, height) != GH_OK) {
        fprintf(stderr, "grayhip: sobel_edge failed: %s\n", gh_last_error());
        abort();
    }
}

}  // extern "C"
