// ubench_sobel.h -- Sobel kernel variants for tools/ubench.hip (developer tool, not shipped).
#pragma once

namespace ub {

using sobel_tile::s16x2;
using sobel_tile::as_s16x2;

__global__ void k_sobel_naive(const uint8_t *src, uint8_t *dst, int w, int h)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)w * h;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / w), x = (int)(i % w);
        uint32_t o = 0;
        if (x > 0 && y > 0 && x < w - 1 && y < h - 1) {
            const uint8_t *m = src + i, *u = m - w, *d = m + w;
            const int gx = -(int)u[-1] - 2 * (int)m[-1] - (int)d[-1] + (int)u[1] + 2 * (int)m[1] + (int)d[1];
            const int gy = (int)u[-1] + 2 * (int)u[0] + (int)u[1] - (int)d[-1] - 2 * (int)d[0] - (int)d[1];
            const int mm = gx * gx + gy * gy;
            // exact integer sqrt
            uint32_t lo = 0, hi = 2047;
            while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (mid * mid <= (uint32_t)mm) lo = mid; else hi = mid - 1; }
            o = lo > 255 ? 255 : lo;
        }
        dst[i] = (uint8_t)o;
    }
}

// ---- optimised arithmetic core ------------------------------------------------------------------
__device__ __forceinline__ s16x2 pk_mad2(s16x2 a, s16x2 c)   // a*2 + c  (one v_pk_mad_u16)
{
    uint32_t d;
    const uint32_t two = 0x00020002u;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(__builtin_bit_cast(uint32_t, a)), "s"(two), "v"(__builtin_bit_cast(uint32_t, c)));
    return as_s16x2(d);
}

template <int MAD>
__device__ __forceinline__ void hstage2(const uint4 &qv, uint32_t ql, uint32_t qr, s16x2 (&h1)[8], s16x2 (&h2)[8])
{
    const uint32_t q[6] = {ql, qv.x, qv.y, qv.z, qv.w, qr};
    s16x2 o[9], e[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t cur = q[j + 1];
        e[2 * j] = as_s16x2(__builtin_amdgcn_perm(0u, cur, 0x0c010c00u));
        e[2 * j + 1] = as_s16x2(__builtin_amdgcn_perm(0u, cur, 0x0c030c02u));
        o[2 * j] = as_s16x2(__builtin_amdgcn_perm(cur, q[j], 0x0c040c03u));
        o[2 * j + 1] = as_s16x2(__builtin_amdgcn_perm(0u, cur, 0x0c020c01u));
    }
    o[8] = as_s16x2(__builtin_amdgcn_perm(q[5], q[4], 0x0c040c03u));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        h1[k] = o[k + 1] - o[k];
        if (MAD) h2[k] = pk_mad2(e[k], o[k] + o[k + 1]);
        else h2[k] = o[k] + o[k + 1] + e[k] * (short)2;
    }
}

// PACK: 0 = int squares, cvt_u32, min, shift/or (the shipping form)
//       1 = f32 squares (+0.5 folded in the fma), cvt_u32, min, shift/or
//       2 = f32 squares, floor, min_f32, v_cvt_pk_u8_f32
//       3 = f32 squares, floor, v_cvt_pk_u8_f32 (relies on saturation)
//       4 = f32 squares, v_cvt_pk_u8_f32 under MODE.fp_round = toward zero (set by the caller)
template <int MAD, int PACK>
__device__ __forceinline__ uint4 vstage2(const s16x2 (&h1u)[8], const s16x2 (&h1m)[8], const s16x2 (&h1d)[8],
                                         const s16x2 (&h2u)[8], const s16x2 (&h2d)[8])
{
    uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const s16x2 gx = MAD ? pk_mad2(h1m[k], h1u[k] + h1d[k]) : (h1u[k] + h1m[k] * (short)2 + h1d[k]);
        const s16x2 gy = h2u[k] - h2d[k];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int p = 2 * k + e;
            const int gxi = e ? gx.y : gx.x, gyi = e ? gy.y : gy.x;
            if (PACK == 0) {
                r[p >> 2] |= gh_sobel_mag(gxi * gxi + gyi * gyi) << (8 * (p & 3));
            } else {
                const float fx = (float)gxi, fy = (float)gyi;
                const float f = __builtin_amdgcn_sqrtf(__builtin_fmaf(fx, fx, __builtin_fmaf(fy, fy, 0.5f)));
                if (PACK == 1) {
                    uint32_t v = (uint32_t)f;
                    v = v > 255u ? 255u : v;
                    r[p >> 2] |= v << (8 * (p & 3));
                } else if (PACK == 2) {
                    r[p >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(__builtin_floorf(f), 255.0f), p & 3, r[p >> 2]);
                } else if (PACK == 3) {
                    r[p >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(f), p & 3, r[p >> 2]);
                } else {
                    r[p >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(f, p & 3, r[p >> 2]);
                }
            }
        }
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

// ---- V1: LDS tile (same staging as the shipping kernel), optimised core ---------------------------
template <int MAD, int PACK, int TH_, int RPW>
__global__ __launch_bounds__(256) void k_sobel_tiled2(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int tiles_x)
{
    constexpr int TW = 1024, LROW = TW + 32, LROWS = TH_ + 2, CH = LROW / 16, NIT = (LROWS * CH + 255) / 256;
    __shared__ __attribute__((aligned(16))) uint8_t tile[LROWS * LROW];
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int x0 = (int)(t % (unsigned)tiles_x) * TW, y0 = (int)(t / (unsigned)tiles_x) * TH_;
    uint4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int r = idx / CH, c = idx - r * CH;
        const int yy = y0 - 1 + r, gx = x0 - 16 + 16 * c;
        v[it] = make_uint4(0, 0, 0, 0);
        if (idx < LROWS * CH && gx >= 0 && gx < w && yy >= 0 && yy < rows) v[it] = *reinterpret_cast<const uint4 *>(src + (ptrdiff_t)yy * w + gx);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * 256;
        if (idx < LROWS * CH) *reinterpret_cast<uint4 *>(tile + idx * 16) = v[it];
    }
    __syncthreads();
    if (PACK == 4) __builtin_amdgcn_s_setreg(0x0801, 3);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVES_Y = TH_ / RPW;            // waves stacked vertically; remaining waves split x? (only WAVES_Y == 4 supported)
    static_assert(WAVES_Y == 4, "tile height must be 4 x rows-per-wave");
    const int x = x0 + 16 * lane, r0 = wave * RPW;
    if (y0 + r0 >= rows) return;
    const uint8_t *lp = tile + r0 * LROW + 16 + 16 * lane;
    s16x2 h1[3][8], h2[3][8];
#pragma unroll
    for (int i = 0; i < RPW + 2; ++i) {
        const uint4 qv = *reinterpret_cast<const uint4 *>(lp + i * LROW);
        const uint32_t ql = *reinterpret_cast<const uint32_t *>(lp + i * LROW - 4);
        const uint32_t qr = *reinterpret_cast<const uint32_t *>(lp + i * LROW + 16);
        hstage2<MAD>(qv, ql, qr, h1[i % 3], h2[i % 3]);
        if (i >= 2) {
            const int y = y0 + r0 + i - 2;
            uint4 out = vstage2<MAD, PACK>(h1[(i - 2) % 3], h1[(i - 1) % 3], h1[i % 3], h2[(i - 2) % 3], h2[i % 3]);
            if (x == 0) out.x &= 0xffffff00u;
            if (x + 15 == w - 1) out.w &= 0x00ffffffu;
            if (y == 0 || y == rows - 1) out = make_uint4(0, 0, 0, 0);
            if (x < w && y < rows) *reinterpret_cast<uint4 *>(dst + (size_t)y * w + x) = out;
        }
    }
}

// ---- V2: streaming rows through registers, halo by DPP (no LDS, no barrier) -------------------------
// one wave = 1024 px x R rows; the 4 waves of a workgroup sit side by side (4096 px of each row).
template <int MAD, int PACK, int R, int C>
__global__ __launch_bounds__(256) void k_sobel_stream(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int groups_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int xw = (gx_i * 4 + wave) * 1024;          // wave's first pixel
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    if (xw >= w) return;
    if (PACK == 4) __builtin_amdgcn_s_setreg(0x0801, 3);
    const bool in_frame = x < w;
    // halo byte owner lanes: lane 0 fetches pixel xw-1, lane 63 fetches pixel xw+1024
    const int hx = lane == 0 ? xw - 1 : xw + 1024;
    const bool hvalid = (lane == 0 || lane == 63) && hx >= 0 && hx < w;

    s16x2 h1[3][8], h2[3][8];
    uint4 q[C];
    uint32_t hb[C];
    auto load_chunk = [&](int cbase) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 - 1 + cbase + j;
            q[j] = make_uint4(0, 0, 0, 0);
            hb[j] = 0;
            if (yy >= 0 && yy < rows && cbase + j < R + 2) {
                if (in_frame) q[j] = *reinterpret_cast<const uint4 *>(src + (size_t)yy * w + x);
                if (hvalid) hb[j] = src[(size_t)yy * w + hx];
            }
        }
    };
    static_assert(C % 3 == 0, "chunk must be a multiple of the 3-row window");
    constexpr int NCH = (R + 2 + C - 1) / C;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        load_chunk(ch * C);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = ch * C + j;            // strip row index 0..R+1  (frame row y0-1+i)
            uint32_t ql = __builtin_amdgcn_update_dpp(0u, q[j].w, 0x138, 0xf, 0xf, false);   // lane l <- lane l-1
            uint32_t qr = __builtin_amdgcn_update_dpp(0u, q[j].x, 0x130, 0xf, 0xf, false);   // lane l <- lane l+1
            if (lane == 0) ql = hb[j] << 24;
            if (lane == 63) qr = hb[j];
            // C % 3 == 0, so the strip row i and the chunk slot j agree mod 3: static window indices
            hstage2<MAD>(q[j], ql, qr, h1[j % 3], h2[j % 3]);
            if (i >= 2 && i < R + 2) {
                const int y = y0 + i - 2;
                uint4 out = vstage2<MAD, PACK>(h1[(j + 1) % 3], h1[(j + 2) % 3], h1[j % 3], h2[(j + 1) % 3], h2[j % 3]);
                if (x == 0) out.x &= 0xffffff00u;
                if (x + 15 == w - 1) out.w &= 0x00ffffffu;
                if (y == 0 || y == rows - 1) out = make_uint4(0, 0, 0, 0);
                if (in_frame && y < rows) *reinterpret_cast<uint4 *>(dst + (size_t)y * w + x) = out;
            }
        }
    }
}

}  // namespace ub

static void bench_sobel(int W, int H)
{
    using namespace ub;
    const size_t npix = (size_t)W * H;
    uint8_t *src, *ref, *out;
    CK(hipMalloc(&src, npix)); CK(hipMalloc(&ref, npix)); CK(hipMalloc(&out, npix));
    // top half: smooth (small gradients, exercises magnitudes < 255); bottom half: noise
    {
        std::vector<uint8_t> h(npix);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < npix; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const int y = (int)(i / W), x = (int)(i % W);
            h[i] = y < H / 2 ? (uint8_t)((x / 3 + y / 5 + (int)(s & 7)) & 255) : (uint8_t)(s >> 24);
        }
        CK(hipMemcpy(src, h.data(), npix, hipMemcpyHostToDevice));
    }
    k_sobel_naive<<<8192, 256>>>(src, ref, W, H);
    CK(hipDeviceSynchronize());
    std::vector<Variant> vs;
    auto add = [&](std::string name, std::function<void()> f) {
        CK(hipMemset(out, 0xEE, npix));
        f();
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("!! %s: %s\n", name.c_str(), hipGetErrorString(e)); exit(1); }
        const unsigned long long bad = count_diff(ref, out, npix);
        if (bad) printf("!! %s: %llu wrong bytes\n", name.c_str(), bad);
        else vs.push_back({name, f, 2.0 * npix});
    };
    const int tiles_x = (W + 1023) / 1024;
    {
        const int ty = (H + 31) / 32;
        add("tiled TH32 shipping (libgrayhip)", [=] { gh_sobel_u8_dev(src, out, W, H, nullptr); });
        add("tiled TH32 mad=0 pack=0", [=] { k_sobel_tiled2<0, 0, 32, 8><<<tiles_x * ty, 256>>>(src, out, W, H, tiles_x); });
        add("tiled TH32 mad=1 pack=0", [=] { k_sobel_tiled2<1, 0, 32, 8><<<tiles_x * ty, 256>>>(src, out, W, H, tiles_x); });
        add("tiled TH32 mad=1 pack=1 (f32 sq)", [=] { k_sobel_tiled2<1, 1, 32, 8><<<tiles_x * ty, 256>>>(src, out, W, H, tiles_x); });
        add("tiled TH32 mad=1 pack=2 (floor,min,cvtpk)", [=] { k_sobel_tiled2<1, 2, 32, 8><<<tiles_x * ty, 256>>>(src, out, W, H, tiles_x); });
        add("tiled TH32 mad=1 pack=3 (floor,cvtpk)", [=] { k_sobel_tiled2<1, 3, 32, 8><<<tiles_x * ty, 256>>>(src, out, W, H, tiles_x); });
        add("tiled TH32 mad=1 pack=4 (RTZ,cvtpk)", [=] { k_sobel_tiled2<1, 4, 32, 8><<<tiles_x * ty, 256>>>(src, out, W, H, tiles_x); });
        const int ty64 = (H + 63) / 64;
        add("tiled TH64 mad=1 pack=1", [=] { k_sobel_tiled2<1, 1, 64, 16><<<tiles_x * ty64, 256>>>(src, out, W, H, tiles_x); });
        const int ty16 = (H + 15) / 16;
        add("tiled TH16 mad=1 pack=1", [=] { k_sobel_tiled2<1, 1, 16, 4><<<tiles_x * ty16, 256>>>(src, out, W, H, tiles_x); });
    }
    {
        const int gx = (W + 4095) / 4096;
        add("stream R64 C6 mad=1 pack=1", [=] { k_sobel_stream<1, 1, 64, 6><<<gx * ((H + 63) / 64), 256>>>(src, out, W, H, gx); });
        add("stream R32 C6 mad=1 pack=1", [=] { k_sobel_stream<1, 1, 32, 6><<<gx * ((H + 31) / 32), 256>>>(src, out, W, H, gx); });
        add("stream R16 C6 mad=1 pack=1", [=] { k_sobel_stream<1, 1, 16, 6><<<gx * ((H + 15) / 16), 256>>>(src, out, W, H, gx); });
        add("stream R32 C12 mad=1 pack=1", [=] { k_sobel_stream<1, 1, 32, 12><<<gx * ((H + 31) / 32), 256>>>(src, out, W, H, gx); });
        add("stream R32 C3 mad=1 pack=1", [=] { k_sobel_stream<1, 1, 32, 3><<<gx * ((H + 31) / 32), 256>>>(src, out, W, H, gx); });
        add("stream R32 C6 mad=1 pack=2", [=] { k_sobel_stream<1, 2, 32, 6><<<gx * ((H + 31) / 32), 256>>>(src, out, W, H, gx); });
        add("stream R32 C6 mad=1 pack=3", [=] { k_sobel_stream<1, 3, 32, 6><<<gx * ((H + 31) / 32), 256>>>(src, out, W, H, gx); });
        add("stream R32 C6 mad=1 pack=4", [=] { k_sobel_stream<1, 4, 32, 6><<<gx * ((H + 31) / 32), 256>>>(src, out, W, H, gx); });
        add("stream R128 C6 mad=1 pack=1", [=] { k_sobel_stream<1, 1, 128, 6><<<gx * ((H + 127) / 128), 256>>>(src, out, W, H, gx); });
    }
    run_table("sobel3x3_u8 variants (1R+1W B/px)", vs, (double)npix);
    CK(hipFree(src)); CK(hipFree(ref)); CK(hipFree(out));
}
