// reduce.hip -- streaming reductions over an RGB8 frame for gfx950 (MI355X).
//
//   k_sum_rgb      exact per-channel sums      <- old/parallel_avg_pixel.c:5-42 (the reference adds every
//                  channel 3x and divides by 3N: avgs = (avgs_in + 3*sum)/(3N); the sums are the
//                  whole information, the division is done by the caller -- see gh_sum_rgb8_host)
//   k_mean_minmax  g=(r+g+b)/3 + global min/max <- old/parallel_to_grayscale.c:7-38
//
// A wavefront streams 1024-pixel groups with three fully coalesced dwordx4 loads, re-sliced through a private LDS
// slab so that each lane owns 16 pixels (stream_core::wave_load_rgb48); lanes keep integer partials in registers,
// the 64-lane wavefront is folded with __shfl_down, the 4 waves of a workgroup through LDS, and one atomic per
// workgroup and channel reaches HBM.  Bound: HBM, 3 R bytes per pixel (+3 W for mean).
#include "gh_common.h"
#include "stream_core.h"

namespace {

constexpr int BLOCK = 256;

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_down(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_down(v, off, 64));
    return v;
}

__global__ __launch_bounds__(BLOCK) void k_sum_rgb(const uint8_t *__restrict__ rgb, size_t npix, int vec_ok,
                                                   unsigned long long *__restrict__ sums)
{
    __shared__ unsigned long long part[BLOCK / 64][3];
    __shared__ __attribute__((aligned(16))) uint8_t slab[BLOCK / 64][3072];
    unsigned long long acc[3] = {0, 0, 0};
    const size_t stride = (size_t)gridDim.x * BLOCK;
    const size_t nwg = vec_ok ? npix / 1024 : 0;             // whole wavefront groups of 1024 px
    const uint4 *in = reinterpret_cast<const uint4 *>(rgb);
    {
        const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
        const size_t wstride = (size_t)gridDim.x * (BLOCK / 64);
        for (size_t g = (size_t)blockIdx.x * (BLOCK / 64) + wave_; g < nwg; g += wstride) {
            uint32_t w[12];
            stream_core::wave_load_rgb48(slab[wave_], in + g * 192, lane_, w);
            uint32_t s[3] = {0, 0, 0};
#pragma unroll
            for (int k = 0; k < 48; ++k) s[k % 3] += gh_byte(w[k >> 2], k & 3);
            acc[0] += s[0]; acc[1] += s[1]; acc[2] += s[2];
        }
    }
    for (size_t i = nwg * 1024 + (size_t)blockIdx.x * BLOCK + threadIdx.x; i < npix; i += stride) {
        acc[0] += rgb[3 * i]; acc[1] += rgb[3 * i + 1]; acc[2] += rgb[3 * i + 2];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned long long v = wave_sum(acc[c]);
        if (lane == 0) part[wave][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long v = 0;
#pragma unroll
        for (int wv = 0; wv < BLOCK / 64; ++wv) v += part[wv][threadIdx.x];
        atomicAdd(&sums[threadIdx.x], v);
    }
}

__global__ __launch_bounds__(BLOCK) void k_mean_minmax(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ out, size_t npix,
                                                       int vec_ok, int *__restrict__ minmax)
{
    __shared__ int pmin[BLOCK / 64], pmax[BLOCK / 64];
    __shared__ __attribute__((aligned(16))) uint8_t slab[BLOCK / 64][3072];
    int lo = 256, hi = -1;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    const size_t nwg = vec_ok ? npix / 1024 : 0;
    const uint4 *in = reinterpret_cast<const uint4 *>(rgb);
    uint4 *o4 = reinterpret_cast<uint4 *>(out);
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    const size_t wstride = (size_t)gridDim.x * (BLOCK / 64);
    for (size_t g = (size_t)blockIdx.x * (BLOCK / 64) + wave_; g < nwg; g += wstride) {
        uint32_t w[12];
        stream_core::wave_load_rgb48(slab[wave_], in + g * 192, lane_, w);
        uint32_t m[16], o[12];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int k = 3 * p;
            const uint32_t s = gh_byte(w[k >> 2], k & 3) + gh_byte(w[(k + 1) >> 2], (k + 1) & 3) + gh_byte(w[(k + 2) >> 2], (k + 2) & 3);
            m[p] = s / 3u;
            lo = min(lo, (int)m[p]);
            hi = max(hi, (int)m[p]);
        }
#pragma unroll
        for (int j = 0; j < 12; ++j)
            o[j] = m[(4 * j) / 3] | (m[(4 * j + 1) / 3] << 8) | (m[(4 * j + 2) / 3] << 16) | (m[(4 * j + 3) / 3] << 24);
        stream_core::wave_store_rgb48(slab[wave_], o4 + g * 192, lane_, o);
    }
    for (size_t i = nwg * 1024 + (size_t)blockIdx.x * BLOCK + threadIdx.x; i < npix; i += stride) {
        const int g = ((int)rgb[3 * i] + rgb[3 * i + 1] + rgb[3 * i + 2]) / 3;
        out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = (uint8_t)g;
        lo = min(lo, g);
        hi = max(hi, g);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane == 0) { pmin[wave] = lo; pmax[wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int wv = 1; wv < BLOCK / 64; ++wv) { lo = min(lo, pmin[wv]); hi = max(hi, pmax[wv]); }
        // most workgroups cannot improve the running extrema: look before the (serialising) atomic
        if (lo < __atomic_load_n(&minmax[0], __ATOMIC_RELAXED)) atomicMin(&minmax[0], lo);
        if (hi > __atomic_load_n(&minmax[1], __ATOMIC_RELAXED)) atomicMax(&minmax[1], hi);
    }
}

// Workgroups per CU before the kernels grid-stride (measured on 16384^2, tools/ab_fused.py --op sum / minmax):
//   k_sum_rgb (reads only, 3 atomics per workgroup)  2: 87 %   4: 86 %   8: 82 %   16: 73 %   64: 43 % of 8 TB/s
//   k_mean_minmax (3 R + 3 W)                        8: 70-74 %   32: 80 %   64: 82 %   128: 81-82 %
// A read-only stream wants few long-lived workgroups, a read+write stream many short ones (as the gray kernels).
inline unsigned reduce_grid(size_t items, int wg_per_cu)
{
    size_t blocks = (items + BLOCK - 1) / BLOCK;
    const int knob = GH_LAB_INT("GH_REDUCE_WG_PER_CU", 0);
    const size_t cap = (size_t)gh_num_cu() * (size_t)(knob > 0 ? knob : wg_per_cu);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

extern "C" {

int gh_sum_rgb8_dev(const uint8_t *rgb, size_t npix, uint64_t *sums_dev, gh_stream_t stream)
{
    GH_REQUIRE(sums_dev, "null sums pointer");
    if (npix == 0) return GH_OK;
    GH_REQUIRE(rgb, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    const int vec_ok = gh_aligned16(rgb);
    k_sum_rgb<<<reduce_grid(vec_ok ? (npix + 15) / 16 : npix, 2), BLOCK, 0, (hipStream_t)stream>>>(
        rgb, npix, vec_ok, reinterpret_cast<unsigned long long *>(sums_dev));
    GH_LAUNCH_CHECK();
    return GH_OK;
}

int gh_mean_gray_minmax_dev(const uint8_t *rgb, uint8_t *out_rgb, size_t npix, int32_t *minmax_dev, gh_stream_t stream)
{
    GH_REQUIRE(minmax_dev, "null minmax pointer");
    if (npix == 0) return GH_OK;
    GH_REQUIRE(rgb && out_rgb, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    const int vec_ok = gh_aligned16(rgb) && gh_aligned16(out_rgb);
    k_mean_minmax<<<reduce_grid(vec_ok ? (npix + 15) / 16 : npix, 128), BLOCK, 0, (hipStream_t)stream>>>(rgb, out_rgb, npix, vec_ok,
                                                                                                   minmax_dev);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

}  // extern "C"
