// ubench.hip -- developer micro-benchmark for kernel variants (NOT part of the product, not shipped).
// Builds standalone: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -I<csrc> tools/ubench.hip -o bin/ubench
// Usage: bin/ubench [probe|copy|gray|sobel|all] [W H]
//
// Every variant is validated byte-for-byte against a naive one-pixel-per-lane kernel before it is timed;
// timings are interleaved rounds in one process (median and min of `ROUNDS` rounds of `ITERS` launches).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "gh_common.h"
#include "sobel_tile.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int gh_set_error(int s, const char *, ...) { return s; }
int gh_ensure_init() { return 0; }


typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NT> __device__ __forceinline__ uint4 ldg(const uint4 *p)
{
    if (NT) { const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
    return *p;
}
template <int NT> __device__ __forceinline__ void stg(uint4 *p, const uint4 v)
{
    if (NT) { u32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(p)); }
    else *p = v;
}

__device__ int g_remap = 1;     // XCD-aware tile order on/off (UB_REMAP=0|1)
__device__ __forceinline__ unsigned ub_remap(unsigned bid, unsigned n)
{
    if (!g_remap || (n % 8)) return bid;
    return (bid % 8) * (n / 8) + bid / 8;     // each XCD gets one contiguous run of tiles
}

static int ITERS = 20, ROUNDS = 7;

struct Variant { std::string name; std::function<void()> launch; double bytes; };

static void run_table(const char *title, std::vector<Variant> &vs, double npix)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<std::vector<float>> t(vs.size());
    for (auto &v : vs) { v.launch(); }
    CK(hipDeviceSynchronize());
    for (int r = 0; r < ROUNDS; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            CK(hipEventRecord(a));
            for (int k = 0; k < ITERS; ++k) vs[i].launch();
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            t[i].push_back(ms / ITERS);
        }
    printf("\n== %s ==\n%-44s %9s %9s %9s %8s %9s\n", title, "variant", "med ms", "min ms", "GB/s(med)", "%8TB/s", "Gpx/s");
    for (size_t i = 0; i < vs.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        const float med = t[i][t[i].size() / 2], mn = t[i][0];
        printf("%-44s %9.4f %9.4f %9.1f %8.1f %9.1f\n", vs[i].name.c_str(), med, mn, vs[i].bytes / med / 1e6, vs[i].bytes / med / 1e6 / 80.0,
               npix / med / 1e6);
    }
    fflush(stdout);
}

__global__ void k_count_diff(const uint8_t *a, const uint8_t *b, size_t n, unsigned long long *cnt)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) bad += a[i] != b[i];
    if (bad) atomicAdd(cnt, bad);
}
static unsigned long long count_diff(const uint8_t *a, const uint8_t *b, size_t n)
{
    unsigned long long *d, h = 0;
    CK(hipMalloc(&d, 8)); CK(hipMemset(d, 0, 8));
    k_count_diff<<<4096, 256>>>(a, b, n, d);
    CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); CK(hipFree(d));
    return h;
}

__global__ void k_fill(uint32_t *p, size_t nwords, uint32_t seed)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = (uint32_t)z;
    }
}

// =================================================================================================
// probe: semantics of v_cvt_pk_u8_f32 and of MODE.fp_round on this chip
// =================================================================================================
__global__ void k_probe(const float *in, int n, unsigned *out)
{
    int i = threadIdx.x;
    if (i >= n) return;
    const float f = in[i];
    out[i] = __builtin_amdgcn_cvt_pk_u8_f32(f, 0, 0u);
    __builtin_amdgcn_s_setreg(0x0801, 3);   // MODE[1:0] = f32 round toward zero
    out[n + i] = __builtin_amdgcn_cvt_pk_u8_f32(f, 0, 0u);
    out[3 * n + i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_sqrtf(f));
    __builtin_amdgcn_s_setreg(0x0801, 2);   // toward -inf
    out[2 * n + i] = __builtin_amdgcn_cvt_pk_u8_f32(f, 0, 0u);
    __builtin_amdgcn_s_setreg(0x0801, 0);
    out[4 * n + i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_sqrtf(f));
}

// raw v_sqrt_f32 on every perfect square k*k, k <= 2048: is it exact?
__global__ void k_probe_sqrt_squares(unsigned *bad_plain, unsigned *bad_half)
{
    unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > 2048) return;
    const float m = (float)(k * k);
    if ((unsigned)__builtin_amdgcn_sqrtf(m) != k) atomicAdd(bad_plain, 1u);
    if ((unsigned)__builtin_amdgcn_sqrtf(m + 0.5f) != k) atomicAdd(bad_half, 1u);
}

static void probe()
{
    const float vals[] = {-3.f, -0.4f, 0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.5f, 254.9f, 255.4f, 255.5f, 256.f, 300.f, 1e9f, 65025.f, 65024.5f};
    const int n = sizeof(vals) / sizeof(vals[0]);
    float *din; unsigned *dout;
    CK(hipMalloc(&din, sizeof(vals))); CK(hipMalloc(&dout, 5 * n * 4));
    CK(hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice));
    k_probe<<<1, 64>>>(din, n, dout);
    std::vector<unsigned> h(5 * n);
    CK(hipMemcpy(h.data(), dout, 5 * n * 4, hipMemcpyDeviceToHost));
    printf("== probe v_cvt_pk_u8_f32 ==\n%12s %8s %8s %8s   sqrt(RTZ) sqrt(RNE)\n", "f", "RNE", "RTZ", "RTN");
    for (int i = 0; i < n; ++i) {
        float a, b; memcpy(&a, &h[3 * n + i], 4); memcpy(&b, &h[4 * n + i], 4);
        printf("%12.2f %8u %8u %8u   %.6f %.6f\n", vals[i], h[i] & 255, h[n + i] & 255, h[2 * n + i] & 255, a, b);
    }
    unsigned *bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    k_probe_sqrt_squares<<<9, 256>>>(bad, bad + 1);
    unsigned hb[2]; CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
    printf("raw v_sqrt_f32 wrong on %u of 2049 perfect squares; sqrt(m+0.5) trunc wrong on %u\n", hb[0], hb[1]);
}

// =================================================================================================
// copy ceilings
// =================================================================================================
template <int NT>
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        uint4 v = ldg<NT>(in + i);
        stg<NT>(out + i, v);
    }
}
// 3 reads : 1 write, like the gray kernel but with trivial arithmetic
template <int NT>
__global__ __launch_bounds__(256) void k_read3_write1(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        uint4 a = ldg<NT>(in + 3 * i);
        uint4 b = ldg<NT>(in + 3 * i + 1);
        uint4 c = ldg<NT>(in + 3 * i + 2);
        uint4 v = make_uint4(a.x ^ b.x ^ c.x, a.y ^ b.y ^ c.y, a.z ^ b.z ^ c.z, a.w ^ b.w ^ c.w);
        stg<NT>(out + i, v);
    }
}
__global__ __launch_bounds__(256) void k_read_only(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        uint4 v = in[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc;
}

static void bench_copy(int W, int H)
{
    const size_t npix = (size_t)W * H;
    uint8_t *rgb, *plane;
    CK(hipMalloc(&rgb, npix * 3)); CK(hipMalloc(&plane, npix * 3));
    k_fill<<<4096, 256>>>((uint32_t *)rgb, npix * 3 / 4, 1);
    std::vector<Variant> vs;
    const size_t n16 = npix * 3 / 16;
    for (unsigned grid : {2048u, 8192u, 0u}) {
        const unsigned g = grid ? grid : (unsigned)((n16 + 255) / 256);
        vs.push_back({"copy 805MB->805MB grid=" + std::to_string(g), [=] { k_copy<0><<<g, 256>>>((const uint4 *)rgb, (uint4 *)plane, n16); }, 2.0 * npix * 3});
        vs.push_back({"copy nt grid=" + std::to_string(g), [=] { k_copy<1><<<g, 256>>>((const uint4 *)rgb, (uint4 *)plane, n16); }, 2.0 * npix * 3});
    }
    const size_t ng = npix / 16;
    for (unsigned grid : {2048u, 8192u, 0u}) {
        const unsigned g = grid ? grid : (unsigned)((ng + 255) / 256);
        vs.push_back({"read3:write1 grid=" + std::to_string(g), [=] { k_read3_write1<0><<<g, 256>>>((const uint4 *)rgb, (uint4 *)plane, ng); }, 4.0 * npix});
        vs.push_back({"read3:write1 nt grid=" + std::to_string(g), [=] { k_read3_write1<1><<<g, 256>>>((const uint4 *)rgb, (uint4 *)plane, ng); }, 4.0 * npix});
    }
    vs.push_back({"read-only 805MB grid=8192", [=] { k_read_only<<<8192, 256>>>((const uint4 *)rgb, (uint4 *)plane, n16); }, 3.0 * npix});
    vs.push_back({"read-only 805MB grid=all", [=] { k_read_only<<<(unsigned)((n16 + 255) / 256), 256>>>((const uint4 *)rgb, (uint4 *)plane, n16); }, 3.0 * npix});
    run_table("copy ceilings", vs, (double)npix);
    CK(hipFree(rgb)); CK(hipFree(plane));
}

// =================================================================================================
// gray variants
// =================================================================================================
__global__ void k_gray_naive(const uint8_t *in, uint8_t *out, size_t npix)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < npix; i += (size_t)gridDim.x * blockDim.x) out[i] = (uint8_t)gh_luma<0>(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
}

__device__ __forceinline__ uint4 luma16(const uint4 a, const uint4 b, const uint4 c)
{
    const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    uint32_t l[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int o = 3 * p;
        l[p] = gh_luma<0>(gh_byte(w[o >> 2], o & 3), gh_byte(w[(o + 1) >> 2], (o + 1) & 3), gh_byte(w[(o + 2) >> 2], (o + 2) & 3));
    }
    return make_uint4(l[0] | l[1] << 8 | l[2] << 16 | l[3] << 24, l[4] | l[5] << 8 | l[6] << 16 | l[7] << 24,
                      l[8] | l[9] << 8 | l[10] << 16 | l[11] << 24, l[12] | l[13] << 8 | l[14] << 16 | l[15] << 24);
}

// G groups per thread, consecutive groups strided by the block (keeps each load instruction's footprint contiguous per wave)
template <int G, int NT>
__global__ __launch_bounds__(256) void k_gray_v(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t ngroups)
{
    const size_t stride = (size_t)gridDim.x * 256 * G;
    for (size_t base = (size_t)blockIdx.x * 256 * G + threadIdx.x; base < ngroups; base += stride) {
        uint4 a[G], b[G], c[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const size_t g = base + (size_t)k * 256;
            if (g < ngroups) {
                a[k] = ldg<NT>(in + 3 * g);
                b[k] = ldg<NT>(in + 3 * g + 1);
                c[k] = ldg<NT>(in + 3 * g + 2);
            }
        }
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const size_t g = base + (size_t)k * 256;
            if (g < ngroups) {
                const uint4 v = luma16(a[k], b[k], c[k]);
                stg<NT>(out + g, v);
            }
        }
    }
}

// wave-contiguous loads: lane i loads 16 B at wave_base + k*1024 + 16*i (k=0..2), de-interleaved through LDS
template <int NT>
__global__ __launch_bounds__(256) void k_gray_lds(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t ngroups)
{
    __shared__ __attribute__((aligned(16))) uint8_t sm[4][3072 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t nwg = ngroups / 64;   // groups of 64 lanes x 16 px = 1024 px = 3 KiB per wave-iteration
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t wg = (size_t)blockIdx.x * 4 + wave; wg < nwg; wg += stride) {
        const uint4 *p = in + wg * 192;
        uint4 v0 = ldg<NT>(p + lane);
        uint4 v1 = ldg<NT>(p + 64 + lane);
        uint4 v2 = ldg<NT>(p + 128 + lane);
        uint4 *s = reinterpret_cast<uint4 *>(sm[wave]);
        s[lane] = v0; s[64 + lane] = v1; s[128 + lane] = v2;
        __builtin_amdgcn_wave_barrier();
        const uint4 *q = reinterpret_cast<const uint4 *>(sm[wave] + 48 * lane);
        const uint4 r = luma16(q[0], q[1], q[2]);
        __builtin_amdgcn_wave_barrier();
        stg<NT>(out + wg * 64 + lane, r);
    }
}

// 4 px per lane with 12-byte loads (lane-contiguous 768 B per wave-instruction), U-fold unrolled
struct __attribute__((packed, aligned(4))) u96 { uint32_t x, y, z; };
template <int U>
__global__ __launch_bounds__(256) void k_gray_x3(const u96 *__restrict__ in, uint32_t *__restrict__ out, size_t nquads)
{
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < nquads; base += stride) {
        u96 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { const size_t q = base + (size_t)k * 256; if (q < nquads) v[k] = in[q]; }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t q = base + (size_t)k * 256;
            if (q < nquads) {
                const uint32_t w[3] = {v[k].x, v[k].y, v[k].z};
                uint32_t l[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int o = 3 * p;
                    l[p] = gh_luma<0>(gh_byte(w[o >> 2], o & 3), gh_byte(w[(o + 1) >> 2], (o + 1) & 3), gh_byte(w[(o + 2) >> 2], (o + 2) & 3));
                }
                out[q] = l[0] | l[1] << 8 | l[2] << 16 | l[3] << 24;
            }
        }
    }
}

static void bench_gray(int W, int H)
{
    const size_t npix = (size_t)W * H, ng = npix / 16;
    uint8_t *rgb, *ref, *out;
    CK(hipMalloc(&rgb, npix * 3)); CK(hipMalloc(&ref, npix)); CK(hipMalloc(&out, npix));
    k_fill<<<4096, 256>>>((uint32_t *)rgb, npix * 3 / 4, 7);
    k_gray_naive<<<8192, 256>>>(rgb, ref, npix);
    std::vector<Variant> vs;
    auto add = [&](std::string name, std::function<void()> f) {
        CK(hipMemset(out, 0xEE, npix));
        f();
        CK(hipDeviceSynchronize());
        const unsigned long long bad = count_diff(ref, out, npix);
        if (bad) printf("!! %s: %llu wrong bytes\n", name.c_str(), bad);
        vs.push_back({name, f, 4.0 * npix});
    };
    const unsigned gall = (unsigned)((ng + 255) / 256);
    add("shipping (libgrayhip)", [=] { gh_gray_rgb8_to_plane_dev(rgb, out, npix, 3, 0, nullptr); });
    add("G1 grid=all", [=] { k_gray_v<1, 0><<<gall, 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    add("G1 nt grid=all", [=] { k_gray_v<1, 1><<<gall, 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    for (unsigned g : {2048u, 4096u, 8192u, 16384u}) {
        add("G1 grid=" + std::to_string(g), [=] { k_gray_v<1, 0><<<g, 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
        add("G1 nt grid=" + std::to_string(g), [=] { k_gray_v<1, 1><<<g, 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    }
    add("G2 grid=all/2", [=] { k_gray_v<2, 0><<<(gall + 1) / 2, 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    add("G2 nt grid=all/2", [=] { k_gray_v<2, 1><<<(gall + 1) / 2, 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    add("G2 grid=4096", [=] { k_gray_v<2, 0><<<4096, 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    add("G4 grid=all/4", [=] { k_gray_v<4, 0><<<(gall + 3) / 4, 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    add("G4 nt grid=all/4", [=] { k_gray_v<4, 1><<<(gall + 3) / 4, 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    add("LDS-transposed grid=all", [=] { k_gray_lds<0><<<(unsigned)((ng / 64 + 3) / 4), 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    add("LDS-transposed nt grid=all", [=] { k_gray_lds<1><<<(unsigned)((ng / 64 + 3) / 4), 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    add("LDS-transposed grid=4096", [=] { k_gray_lds<0><<<4096, 256>>>((const uint4 *)rgb, (uint4 *)out, ng); });
    add("x3 (12B/lane) U1 grid=all", [=] { k_gray_x3<1><<<(unsigned)((npix / 4 + 255) / 256), 256>>>((const u96 *)rgb, (uint32_t *)out, npix / 4); });
    add("x3 (12B/lane) U4 grid=all/4", [=] { k_gray_x3<4><<<(unsigned)((npix / 4 + 1023) / 1024), 256>>>((const u96 *)rgb, (uint32_t *)out, npix / 4); });
    run_table("gray_rgb8_to_plane variants (3R+1W B/px)", vs, (double)npix);
    CK(hipFree(rgb)); CK(hipFree(ref)); CK(hipFree(out));
}

// =================================================================================================
// sobel variants -- see tools/ubench_sobel.h
// =================================================================================================
#include "ubench_sobel.h"
#include "ubench_valu.h"
#include "ubench_stream.h"

int main(int argc, char **argv)
{
    const std::string what = argc > 1 ? argv[1] : "all";
    const int W = argc > 2 ? atoi(argv[2]) : 16384, H = argc > 3 ? atoi(argv[3]) : 16384;
    if (getenv("UB_ITERS")) ITERS = atoi(getenv("UB_ITERS"));
    if (getenv("UB_ROUNDS")) ROUNDS = atoi(getenv("UB_ROUNDS"));
    CK(hipSetDevice(0));
    { int r = getenv("UB_REMAP") ? atoi(getenv("UB_REMAP")) : 1; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_remap), &r, sizeof(r))); printf("XCD remap: %d\n", r); }
    if (what == "probe" || what == "all") probe();
    if (what == "copy" || what == "all") bench_copy(W, H);
    if (what == "gray" || what == "all") bench_gray(W, H);
    if (what == "sobel" || what == "all") bench_sobel(W, H);
    if (what == "valu" || what == "all") bench_valu();
    if (what == "stream" || what == "all") bench_stream(W, H);
    if (what == "s16" || what == "all") bench_stream16(W, H);
    if (what == "sc" || what == "all") bench_streamc(W, H);
    if (what == "pattern" || what == "all") bench_pattern(W, H);
    if (what == "pattern31" || what == "all") bench_pattern31(W, H);
    if (what == "pitch") bench_pitch(W, H);
    return 0;
}
