// gray.hip -- RGB8/RGBA8 -> luma kernels for gfx950 (MI355X), plus the plane extract / replicate
// byte shuffles of the reference's Sobel driver.
//
// Replaces (paths relative to the reference repository):
//   monolithic/src/parallel_to_grayscale.c:5-17   convert_to_grayscale  (k_gray_inplace*)
//   monolithic/src/main_with_sobel.c:57-59        plane extract         (k_extract*)
//   monolithic/src/main_with_sobel.c:65-73        plane replicate       (k_replicate*)
//
// Layout: a frame is one flat byte stream (pitch = width*channels, no padding, as in the
// reference), so the kernels are 1-D.  Every lane owns 16 consecutive pixels (48 B of RGB8), and a
// wavefront owns 1024 consecutive pixels = 3 KiB per iteration.
//
// RGB8 fast path (k_gray_rgb_to_plane_w / k_gray_inplace_rgb_w): the wave's 3 KiB are read with three
// FULLY COALESCED dwordx4 loads (lane i: 16 B at +16 i, +1024 + 16 i, +2048 + 16 i), written to a private
// 3 KiB LDS slab and read back as 48 contiguous bytes per lane (ds_read_b128 x3, conflict-free at a 48-byte
// stride).  Loading the 48 B per lane directly (stride-48 dwordx4) touches every 128-B line from three
// different instructions: 5.75 TB/s on MI355X, and `nt` makes it worse (the line is dropped between the
// three touches); the LDS de-interleave with nontemporal loads/stores reaches 6.3 TB/s.
// Bound: HBM (3 R + 1 W bytes per pixel for the plane variant, 3 R + 3 W in place); no MFMA (no contraction).
#include "gh_common.h"
#include "stream_core.h"

namespace {

constexpr int BLOCK = 256;
constexpr int PX_PER_LANE = 16;

template <int MODE>
__device__ __forceinline__ void luma16_rgb(const uint32_t (&w)[12], uint32_t (&lum)[16])
{
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int o = 3 * p;
        lum[p] = gh_luma<MODE>(gh_byte(w[o >> 2], o & 3), gh_byte(w[(o + 1) >> 2], (o + 1) & 3),
                               gh_byte(w[(o + 2) >> 2], (o + 2) & 3));
    }
}

// passes 2..N of the reference's driver loop (monolithic/src/main.c:33-35) see a grey pixel (y, y, y): pass k+1 maps it to
// lum(y, y, y), which is NOT the identity for every y in IEEE mode.  The passes are independent per pixel, so a launch
// can run all of them in registers (same arithmetic as N launches, no memory round trips in between).
template <int MODE>
__device__ __forceinline__ uint32_t luma_again(uint32_t y, int extra)
{
    for (int p = 0; p < extra; ++p) y = gh_luma<MODE>(y, y, y);
    return y;
}

__device__ __forceinline__ uint4 pack16(const uint32_t (&v)[16])
{
    uint4 o;
    o.x = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    o.y = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
    o.z = v[8] | (v[9] << 8) | (v[10] << 16) | (v[11] << 24);
    o.w = v[12] | (v[13] << 8) | (v[14] << 16) | (v[15] << 24);
    return o;
}

__device__ __forceinline__ void load48(const uint4 *p, uint32_t (&w)[12])
{
    const uint4 a = p[0], b = p[1], c = p[2];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
}

// ---- RGB8 -> plane, one wavefront per 1024 px, LDS de-interleave ---------------------------------
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_rgb_to_plane_w(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t nwavegroups)
{
    __shared__ __attribute__((aligned(16))) uint8_t slab[BLOCK / 64][3072];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * (BLOCK / 64);
    for (size_t g = (size_t)blockIdx.x * (BLOCK / 64) + wave; g < nwavegroups; g += stride) {
        uint32_t w[12], lum[16];
        stream_core::wave_load_rgb48(slab[wave], in + g * 192, lane, w);
        luma16_rgb<MODE>(w, lum);
        stream_core::store_stream(out + g * 64 + lane, pack16(lum));
    }
}

// byte k of the 48 replicated output bytes is v[k / 3]
__device__ __forceinline__ void replicate48(const uint32_t (&v)[16], uint32_t (&o)[12])
{
#pragma unroll
    for (int j = 0; j < 12; ++j)
        o[j] = v[(4 * j) / 3] | (v[(4 * j + 1) / 3] << 8) | (v[(4 * j + 2) / 3] << 16) | (v[(4 * j + 3) / 3] << 24);
}

// in place: same read path; the 48 replicated output bytes go back through the slab so the stores are coalesced too
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_inplace_rgb_w(uint4 *__restrict__ data, size_t nwavegroups, int extra)
{
    __shared__ __attribute__((aligned(16))) uint8_t slab[BLOCK / 64][3072];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * (BLOCK / 64);
    for (size_t g = (size_t)blockIdx.x * (BLOCK / 64) + wave; g < nwavegroups; g += stride) {
        uint32_t w[12], lum[16], o[12];
        stream_core::wave_load_rgb48(slab[wave], data + g * 192, lane, w);
        luma16_rgb<MODE>(w, lum);
        if (extra) {
#pragma unroll
            for (int k = 0; k < 16; ++k) lum[k] = luma_again<MODE>(lum[k], extra);
        }
        replicate48(lum, o);
        stream_core::wave_store_rgb48(slab[wave], data + g * 192, lane, o);
    }
}

// plane extract / replicate (main_with_sobel.c:57-59, 65-73), wavefront form
__global__ __launch_bounds__(BLOCK) void k_extract_rgb_w(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t nwavegroups)
{
    __shared__ __attribute__((aligned(16))) uint8_t slab[BLOCK / 64][3072];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * (BLOCK / 64);
    for (size_t g = (size_t)blockIdx.x * (BLOCK / 64) + wave; g < nwavegroups; g += stride) {
        uint32_t w[12], v[16];
        stream_core::wave_load_rgb48(slab[wave], in + g * 192, lane, w);
#pragma unroll
        for (int p = 0; p < 16; ++p) v[p] = gh_byte(w[(3 * p) >> 2], (3 * p) & 3);
        stream_core::store_stream(out + g * 64 + lane, pack16(v));
    }
}

__global__ __launch_bounds__(BLOCK) void k_replicate_rgb_w(const uint4 *__restrict__ plane, uint4 *__restrict__ img, size_t nwavegroups)
{
    __shared__ __attribute__((aligned(16))) uint8_t slab[BLOCK / 64][3072];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * (BLOCK / 64);
    for (size_t g = (size_t)blockIdx.x * (BLOCK / 64) + wave; g < nwavegroups; g += stride) {
        const uint4 q = stream_core::load_stream(plane + g * 64 + lane);
        const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
        uint32_t v[16], o[12];
#pragma unroll
        for (int p = 0; p < 16; ++p) v[p] = gh_byte(w4[p >> 2], p & 3);
        replicate48(v, o);
        stream_core::wave_store_rgb48(slab[wave], img + g * 192, lane, o);
    }
}

// ---- RGB8 -> plane, 16 px per lane (remainder of a frame that is not a multiple of 1024 px) ---------
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_rgb_to_plane(const uint4 *__restrict__ in, uint4 *__restrict__ out,
                                                             size_t ngroups)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < ngroups; g += stride) {
        uint32_t w[12], lum[16];
        load48(in + 3 * g, w);
        luma16_rgb<MODE>(w, lum);
        out[g] = pack16(lum);
    }
}

// ---- RGBA8 -> plane ----------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_rgba_to_plane(const uint4 *__restrict__ in, uint4 *__restrict__ out,
                                                              size_t ngroups)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < ngroups; g += stride) {
        uint32_t lum[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = in[4 * g + q];
            const uint32_t px[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) lum[4 * q + k] = gh_luma<MODE>(gh_byte(px[k], 0), gh_byte(px[k], 1), gh_byte(px[k], 2));
        }
        out[g] = pack16(lum);
    }
}

// ---- RGBA8, wave-level forms (monolithic/src/parallel_to_grayscale.c:9-15 with channels == 4) -----------------------
// A wavefront owns 1024 consecutive pixels = 4 KiB: four FULLY COALESCED nontemporal dwordx4 loads (lane i: vectors
// i, 64+i, 128+i, 192+i; one RGBA pixel = one dword, so a lane's vector is 4 whole pixels and the INPUT needs no
// de-interleave).  Plane output: the lane's four luma bytes of load j belong at plane byte 256 j + 4 i, i.e. one dword per
// lane and load; they go through a private 1 KiB LDS slab (ds_write_b32 x4 lane-linear, ds_read_b128 lane-linear: both
// conflict-free) so that the wave issues ONE coalesced dwordx4 store instead of four dword stores.
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_rgba_to_plane_w(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t nwavegroups)
{
    __shared__ __attribute__((aligned(16))) uint32_t slab[BLOCK / 64][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * (BLOCK / 64);
    for (size_t g = (size_t)blockIdx.x * (BLOCK / 64) + wave; g < nwavegroups; g += stride) {
        const uint4 *p = in + g * 256;
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = stream_core::load_stream(p + 64 * j + lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t px[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            uint32_t y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = gh_luma<MODE>(gh_byte(px[k], 0), gh_byte(px[k], 1), gh_byte(px[k], 2));
            slab[wave][64 * j + lane] = y[0] | (y[1] << 8) | (y[2] << 16) | (y[3] << 24);
        }
        __builtin_amdgcn_wave_barrier();
        const uint4 r = reinterpret_cast<const uint4 *>(slab[wave])[lane];
        __builtin_amdgcn_wave_barrier();
        stream_core::store_stream(out + g * 64 + lane, r);
    }
}

// in place: loads and stores are lane-linear already; one wave per 4 KiB, nontemporal both ways
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_inplace_rgba_w(uint4 *__restrict__ data, size_t nwavegroups, int extra)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * (BLOCK / 64);
    for (size_t g = (size_t)blockIdx.x * (BLOCK / 64) + wave; g < nwavegroups; g += stride) {
        uint4 *p = data + g * 256;
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = stream_core::load_stream(p + 64 * j + lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t px[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t y = luma_again<MODE>(gh_luma<MODE>(gh_byte(px[k], 0), gh_byte(px[k], 1), gh_byte(px[k], 2)), extra);
                px[k] = (px[k] & 0xff000000u) | (y * 0x00010101u);
            }
            stream_core::store_stream(p + 64 * j + lane, make_uint4(px[0], px[1], px[2], px[3]));
        }
    }
}

// ---- measurement aid: linear nontemporal copy (1 R + 1 W), the "achievable" HBM ceiling of SURVEY.md section 8(d) -------
__global__ __launch_bounds__(BLOCK) void k_copy_stream(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += stride) stream_core::store_stream(out + i, stream_core::load_stream(in + i));
}

// ---- in place, reference ABI: R=G=B=luma, alpha untouched --------------------------------------
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_inplace_rgb(uint4 *__restrict__ data, size_t ngroups, int extra)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < ngroups; g += stride) {
        uint32_t w[12], lum[16], o[12];
        load48(data + 3 * g, w);
        luma16_rgb<MODE>(w, lum);
        if (extra) {
#pragma unroll
            for (int k = 0; k < 16; ++k) lum[k] = luma_again<MODE>(lum[k], extra);
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) {   // output byte k of the 48 is lum[k / 3]
            o[j] = lum[(4 * j) / 3] | (lum[(4 * j + 1) / 3] << 8) | (lum[(4 * j + 2) / 3] << 16) |
                   (lum[(4 * j + 3) / 3] << 24);
        }
        data[3 * g + 0] = make_uint4(o[0], o[1], o[2], o[3]);
        data[3 * g + 1] = make_uint4(o[4], o[5], o[6], o[7]);
        data[3 * g + 2] = make_uint4(o[8], o[9], o[10], o[11]);
    }
}

template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_inplace_rgba(uint4 *__restrict__ data, size_t nquads, int extra)
{
    // one uint4 = 4 RGBA pixels
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t q = (size_t)blockIdx.x * BLOCK + threadIdx.x; q < nquads; q += stride) {
        uint4 v = data[q];
        uint32_t px[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t y = luma_again<MODE>(gh_luma<MODE>(gh_byte(px[k], 0), gh_byte(px[k], 1), gh_byte(px[k], 2)), extra);
            px[k] = (px[k] & 0xff000000u) | (y * 0x00010101u);
        }
        data[q] = make_uint4(px[0], px[1], px[2], px[3]);
    }
}

// ---- one pixel per lane: tails, unaligned buffers, channels 1/2 --------------------------------
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_to_plane_px(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                            size_t first, size_t npix, int channels)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = first + (size_t)blockIdx.x * BLOCK + threadIdx.x; i < npix; i += stride) {
        const uint8_t *p = in + i * channels;
        out[i] = channels >= 3 ? (uint8_t)gh_luma<MODE>(p[0], p[1], p[2]) : p[0];
    }
}

template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_inplace_px(uint8_t *__restrict__ data, size_t first, size_t npix,
                                                           int channels, int extra)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = first + (size_t)blockIdx.x * BLOCK + threadIdx.x; i < npix; i += stride) {
        uint8_t *p = data + i * channels;
        const uint8_t y = (uint8_t)luma_again<MODE>(gh_luma<MODE>(p[0], p[1], p[2]), extra);
        p[0] = y; p[1] = y; p[2] = y;
    }
}

// ---- luma of a grey plane: v -> lum(v,v,v) -----------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gray_of_plane(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                         size_t npix, int vec_ok)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    const size_t ngroups = vec_ok ? npix / 16 : 0;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < ngroups; g += stride) {
        const uint4 v = reinterpret_cast<const uint4 *>(src)[g];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t lum[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const uint32_t b = gh_byte(w[p >> 2], p & 3);
            lum[p] = gh_luma<MODE>(b, b, b);
        }
        reinterpret_cast<uint4 *>(dst)[g] = pack16(lum);
    }
    for (size_t i = ngroups * 16 + (size_t)blockIdx.x * BLOCK + threadIdx.x; i < npix; i += stride) {
        const uint32_t b = src[i];
        dst[i] = (uint8_t)gh_luma<MODE>(b, b, b);
    }
}

// ---- plane extract / replicate -----------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_extract_rgb(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t ngroups)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < ngroups; g += stride) {
        uint32_t w[12], v[16];
        load48(in + 3 * g, w);
#pragma unroll
        for (int p = 0; p < 16; ++p) v[p] = gh_byte(w[(3 * p) >> 2], (3 * p) & 3);
        out[g] = pack16(v);
    }
}

__global__ __launch_bounds__(BLOCK) void k_extract_px(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, size_t first,
                                                      size_t npix, int channels)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = first + (size_t)blockIdx.x * BLOCK + threadIdx.x; i < npix; i += stride) out[i] = in[i * channels];
}

__global__ __launch_bounds__(BLOCK) void k_replicate_rgb(const uint4 *__restrict__ plane, uint4 *__restrict__ img, size_t ngroups)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x; g < ngroups; g += stride) {
        const uint4 v = plane[g];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            uint32_t acc = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int p = (4 * j + b) / 3;
                acc |= gh_byte(w[p >> 2], p & 3) << (8 * b);
            }
            o[j] = acc;
        }
        img[3 * g + 0] = make_uint4(o[0], o[1], o[2], o[3]);
        img[3 * g + 1] = make_uint4(o[4], o[5], o[6], o[7]);
        img[3 * g + 2] = make_uint4(o[8], o[9], o[10], o[11]);
    }
}

__global__ __launch_bounds__(BLOCK) void k_replicate_px(const uint8_t *__restrict__ plane, uint8_t *__restrict__ img, size_t first,
                                                        size_t npix, int channels)
{
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = first + (size_t)blockIdx.x * BLOCK + threadIdx.x; i < npix; i += stride) {
        uint8_t *p = img + i * channels;
        const uint8_t v = plane[i];
        p[0] = v;
        if (channels >= 3) { p[1] = v; p[2] = v; }
    }
}

// grid for a grid-stride 1-D kernel: enough blocks to cover `items`, capped so the tail of a very
// large launch is absorbed by the stride loop (GH_GRAY_MAX_BLOCKS, tunable for experiments).
inline unsigned grid_for(size_t items)
{
    const int knob = GH_LAB_INT("GH_GRAY_MAX_BLOCKS", 0);       // tuning: cap the grid, the kernels grid-stride the rest
    const size_t cap = knob > 0 ? (size_t)knob : (size_t)1 << 20;
    size_t blocks = (items + BLOCK - 1) / BLOCK;
    if (blocks < 1) blocks = 1;
    if (blocks > cap) blocks = cap;
    return (unsigned)blocks;
}

}  // namespace

extern "C" {

int gh_gray_rgb8_to_plane_dev(const uint8_t *rgb, uint8_t *gray, size_t npix, int channels, int mode, gh_stream_t stream)
{
    GH_REQUIRE(channels >= 1 && channels <= 4, "channels must be 1..4");
    GH_REQUIRE(mode == GH_MODE_IEEE || mode == GH_MODE_FMA, "bad mode");
    if (npix == 0) return GH_OK;
    GH_REQUIRE(rgb && gray, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    hipStream_t s = (hipStream_t)stream;
    size_t done = 0;
    if ((channels == 3 || channels == 4) && gh_aligned16(rgb) && gh_aligned16(gray) && npix >= PX_PER_LANE) {
        const size_t ngroups = npix / PX_PER_LANE;
        auto in = reinterpret_cast<const uint4 *>(rgb);
        auto out = reinterpret_cast<uint4 *>(gray);
        if (channels == 3) {
            const size_t nwg = ngroups / 64;                 // whole wavefront groups of 1024 px
            if (nwg) {
                const unsigned gw = grid_for(nwg * 64);       // one wave per group, 4 waves per workgroup
                if (mode == GH_MODE_FMA) k_gray_rgb_to_plane_w<GH_MODE_FMA><<<gw, BLOCK, 0, s>>>(in, out, nwg);
                else k_gray_rgb_to_plane_w<GH_MODE_IEEE><<<gw, BLOCK, 0, s>>>(in, out, nwg);
                GH_LAUNCH_CHECK();
            }
            const size_t rest = ngroups - nwg * 64;           // < 64 lane groups left
            if (rest) {
                if (mode == GH_MODE_FMA) k_gray_rgb_to_plane<GH_MODE_FMA><<<1, BLOCK, 0, s>>>(in + nwg * 192, out + nwg * 64, rest);
                else k_gray_rgb_to_plane<GH_MODE_IEEE><<<1, BLOCK, 0, s>>>(in + nwg * 192, out + nwg * 64, rest);
            }
        } else {
            const size_t nwg = ngroups / 64;                 // whole wavefront groups of 1024 px (4 KiB in, 1 KiB out)
            if (nwg) {
                const unsigned gw = grid_for(nwg * 64);
                if (mode == GH_MODE_FMA) k_gray_rgba_to_plane_w<GH_MODE_FMA><<<gw, BLOCK, 0, s>>>(in, out, nwg);
                else k_gray_rgba_to_plane_w<GH_MODE_IEEE><<<gw, BLOCK, 0, s>>>(in, out, nwg);
                GH_LAUNCH_CHECK();
            }
            const size_t rest = ngroups - nwg * 64;
            if (rest) {
                if (mode == GH_MODE_FMA) k_gray_rgba_to_plane<GH_MODE_FMA><<<1, BLOCK, 0, s>>>(in + nwg * 256, out + nwg * 64, rest);
                else k_gray_rgba_to_plane<GH_MODE_IEEE><<<1, BLOCK, 0, s>>>(in + nwg * 256, out + nwg * 64, rest);
            }
        }
        GH_LAUNCH_CHECK();
        done = ngroups * PX_PER_LANE;
    }
    if (done < npix) {
        const unsigned grid = grid_for(npix - done);
        if (mode == GH_MODE_FMA) k_gray_to_plane_px<GH_MODE_FMA><<<grid, BLOCK, 0, s>>>(rgb, gray, done, npix, channels);
        else k_gray_to_plane_px<GH_MODE_IEEE><<<grid, BLOCK, 0, s>>>(rgb, gray, done, npix, channels);
        GH_LAUNCH_CHECK();
    }
    return GH_OK;
}

int gh_copy_stream_dev(const void *src, void *dst, size_t nbytes, gh_stream_t stream)
{
    if (nbytes == 0) return GH_OK;
    GH_REQUIRE(src && dst, "null pointer");
    GH_REQUIRE(gh_aligned16(src) && gh_aligned16(dst) && nbytes % 16 == 0, "16-byte aligned buffers and size only");
    if (int rc = gh_ensure_init()) return rc;
    const size_t n16 = nbytes / 16;
    k_copy_stream<<<grid_for(n16), BLOCK, 0, (hipStream_t)stream>>>(reinterpret_cast<const uint4 *>(src), reinterpret_cast<uint4 *>(dst), n16);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

int gh_gray_inplace_dev(uint8_t *data, size_t npix, int channels, int mode, gh_stream_t stream)
{
    return gh_gray_inplace_passes_dev(data, npix, channels, mode, 1, stream);
}

int gh_gray_inplace_passes_dev(uint8_t *data, size_t npix, int channels, int mode, int passes, gh_stream_t stream)
{
    GH_REQUIRE(channels >= 1 && channels <= 4, "channels must be 1..4");
    GH_REQUIRE(passes >= 1, "passes must be >= 1");
    const int extra = passes - 1;
    GH_REQUIRE(mode == GH_MODE_IEEE || mode == GH_MODE_FMA, "bad mode");
    if (npix == 0 || channels < 3) return GH_OK;   // channels 1/2: undefined in the reference, no-op here
    GH_REQUIRE(data, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    hipStream_t s = (hipStream_t)stream;
    size_t done = 0;
    if (gh_aligned16(data) && npix >= PX_PER_LANE) {
        auto p = reinterpret_cast<uint4 *>(data);
        if (channels == 3) {
            const size_t ngroups = npix / PX_PER_LANE;
            const size_t nwg = ngroups / 64;
            if (nwg) {
                const unsigned gw = grid_for(nwg * 64);
                if (mode == GH_MODE_FMA) k_gray_inplace_rgb_w<GH_MODE_FMA><<<gw, BLOCK, 0, s>>>(p, nwg, extra);
                else k_gray_inplace_rgb_w<GH_MODE_IEEE><<<gw, BLOCK, 0, s>>>(p, nwg, extra);
                GH_LAUNCH_CHECK();
            }
            const size_t rest = ngroups - nwg * 64;
            if (rest) {
                if (mode == GH_MODE_FMA) k_gray_inplace_rgb<GH_MODE_FMA><<<1, BLOCK, 0, s>>>(p + nwg * 192, rest, extra);
                else k_gray_inplace_rgb<GH_MODE_IEEE><<<1, BLOCK, 0, s>>>(p + nwg * 192, rest, extra);
            }
            done = ngroups * PX_PER_LANE;
        } else {
            const size_t nquads = npix / 4, nwg = nquads / 256;   // wavefront groups of 1024 px = 256 dwordx4
            if (nwg) {
                const unsigned gw = grid_for(nwg * 64);
                if (mode == GH_MODE_FMA) k_gray_inplace_rgba_w<GH_MODE_FMA><<<gw, BLOCK, 0, s>>>(p, nwg, extra);
                else k_gray_inplace_rgba_w<GH_MODE_IEEE><<<gw, BLOCK, 0, s>>>(p, nwg, extra);
                GH_LAUNCH_CHECK();
            }
            const size_t rest = nquads - nwg * 256;
            if (rest) {
                if (mode == GH_MODE_FMA) k_gray_inplace_rgba<GH_MODE_FMA><<<1, BLOCK, 0, s>>>(p + nwg * 256, rest, extra);
                else k_gray_inplace_rgba<GH_MODE_IEEE><<<1, BLOCK, 0, s>>>(p + nwg * 256, rest, extra);
            }
            done = nquads * 4;
        }
        GH_LAUNCH_CHECK();
    }
    if (done < npix) {
        const unsigned grid = grid_for(npix - done);
        if (mode == GH_MODE_FMA) k_gray_inplace_px<GH_MODE_FMA><<<grid, BLOCK, 0, s>>>(data, done, npix, channels, extra);
        else k_gray_inplace_px<GH_MODE_IEEE><<<grid, BLOCK, 0, s>>>(data, done, npix, channels, extra);
        GH_LAUNCH_CHECK();
    }
    return GH_OK;
}

int gh_gray_of_plane_dev(const uint8_t *src, uint8_t *dst, size_t npix, int mode, gh_stream_t stream)
{
    GH_REQUIRE(mode == GH_MODE_IEEE || mode == GH_MODE_FMA, "bad mode");
    if (npix == 0) return GH_OK;
    GH_REQUIRE(src && dst, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int vec_ok = gh_aligned16(src) && gh_aligned16(dst);
    const unsigned grid = grid_for(vec_ok ? (npix + 15) / 16 : npix);
    if (mode == GH_MODE_FMA) k_gray_of_plane<GH_MODE_FMA><<<grid, BLOCK, 0, s>>>(src, dst, npix, vec_ok);
    else k_gray_of_plane<GH_MODE_IEEE><<<grid, BLOCK, 0, s>>>(src, dst, npix, vec_ok);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

int gh_extract_plane_dev(const uint8_t *img, uint8_t *plane, size_t npix, int channels, gh_stream_t stream)
{
    GH_REQUIRE(channels >= 1 && channels <= 4, "channels must be 1..4");
    if (npix == 0) return GH_OK;
    GH_REQUIRE(img && plane, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    hipStream_t s = (hipStream_t)stream;
    size_t done = 0;
    if (channels == 3 && gh_aligned16(img) && gh_aligned16(plane) && npix >= PX_PER_LANE) {
        const size_t ngroups = npix / PX_PER_LANE, nwg = ngroups / 64, rest = ngroups - nwg * 64;
        auto in = reinterpret_cast<const uint4 *>(img);
        auto out = reinterpret_cast<uint4 *>(plane);
        if (nwg) k_extract_rgb_w<<<grid_for(nwg * 64), BLOCK, 0, s>>>(in, out, nwg);
        if (rest) k_extract_rgb<<<1, BLOCK, 0, s>>>(in + nwg * 192, out + nwg * 64, rest);
        GH_LAUNCH_CHECK();
        done = ngroups * PX_PER_LANE;
    }
    if (done < npix) {
        k_extract_px<<<grid_for(npix - done), BLOCK, 0, s>>>(img, plane, done, npix, channels);
        GH_LAUNCH_CHECK();
    }
    return GH_OK;
}

int gh_replicate_plane_dev(const uint8_t *plane, uint8_t *img, size_t npix, int channels, gh_stream_t stream)
{
    GH_REQUIRE(channels >= 1 && channels <= 4, "channels must be 1..4");
    if (npix == 0) return GH_OK;
    GH_REQUIRE(img && plane, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    hipStream_t s = (hipStream_t)stream;
    size_t done = 0;
    if (channels == 3 && gh_aligned16(img) && gh_aligned16(plane) && npix >= PX_PER_LANE) {
        const size_t ngroups = npix / PX_PER_LANE, nwg = ngroups / 64, rest = ngroups - nwg * 64;
        auto in = reinterpret_cast<const uint4 *>(plane);
        auto out = reinterpret_cast<uint4 *>(img);
        if (nwg) k_replicate_rgb_w<<<grid_for(nwg * 64), BLOCK, 0, s>>>(in, out, nwg);
        if (rest) k_replicate_rgb<<<1, BLOCK, 0, s>>>(in + nwg * 64, out + nwg * 192, rest);
        GH_LAUNCH_CHECK();
        done = ngroups * PX_PER_LANE;
    }
    if (done < npix) {
        k_replicate_px<<<grid_for(npix - done), BLOCK, 0, s>>>(plane, img, done, npix, channels);
        GH_LAUNCH_CHECK();
    }
    return GH_OK;
}

}  // extern "C"
