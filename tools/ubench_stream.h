// ubench_stream.h -- f32 "strip streaming" prototypes (developer tool): one wave owns a 1024-px wide
// strip and walks R rows; 3x3 reuse lives in registers, the horizontal halo comes from the neighbour
// lanes through DPP (wave_shr/wave_shl), no block barrier.  All arithmetic in full-rate f32 VALU ops
// (integers < 2^24, exact), magnitude = v_sqrt_f32 + v_cvt_pk_u8_f32 under MODE.fp_round = RTZ.
#pragma once

namespace ubs {

// opaque(): stops the optimiser from proving the float is a small integer and rewriting the float adds as
// (half-rate) integer/SDWA ops -- f32 add/sub/mul/fma are the full-rate VALU ops on gfx950.
__device__ __forceinline__ float opaque(float v) { asm("" : "+v"(v)); return v; }
__device__ __forceinline__ float ub0(uint32_t w) { return opaque((float)(w & 0xffu)); }
__device__ __forceinline__ float ub1(uint32_t w) { return opaque((float)((w >> 8) & 0xffu)); }
__device__ __forceinline__ float ub2(uint32_t w) { return opaque((float)((w >> 16) & 0xffu)); }
__device__ __forceinline__ float ub3(uint32_t w) { return opaque((float)(w >> 24)); }

struct SobelState {
    float h1p[16], cp[16], h2r[2][16];
};

// one input row (18 luma/plane samples p[0..17] = x-1 .. x+16) -> optional output row
// PAR = parity of the strip row index (static), EMIT = produce an output row (needs 2 previous rows)
template <int PAR>
__device__ __forceinline__ uint4 sobel_row(SobelState &st, const float (&p)[18], float half)
{
    float s[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) s[k] = p[k] + p[k + 1];
    uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float h1 = p[k + 2] - p[k];
        const float h2 = s[k] + s[k + 1];
        const float c = st.h1p[k] + h1;
        const float gx = st.cp[k] + c;
        const float gy = st.h2r[PAR][k] - h2;
        // floor(sqrt(m)) under round-to-nearest: sqrt(m + 0.5) is never within 9e-4 of an integer for results
        // <= 256, so RNE(sqrt(m + 0.5) - 0.5) == floor(sqrt(m)); v_cvt_pk_u8_f32 saturates at 255.
        float t;   // gy*gy + 0.5 as ONE v_fma_f32 (hipcc would otherwise emit v_mov 0.5 + v_fmac)
        asm("v_fma_f32 %0, %1, %1, %2" : "=v"(t) : "v"(gy), "v"(half));
        const float f = __builtin_amdgcn_sqrtf(__builtin_fmaf(gx, gx, t)) - half;
        r[k >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(f, k & 3, r[k >> 2]);
        st.h1p[k] = h1; st.cp[k] = c; st.h2r[PAR][k] = h2;
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

template <int R, int C, int NT, int PF = 0>
__global__ __launch_bounds__(256) void k_sobel_stream_f32(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int groups_x)
{
    static_assert(C % 2 == 0, "chunk must be even (h2 ring parity)");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int xw = (gx_i * 4 + wave) * 1024;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    if (xw >= w) return;
    float half = 0.5f;
    asm volatile("" : "+v"(half));          // keep 0.5 in a register: v_fma_f32 d,a,a,half instead of v_mov + v_fmac
    const bool in_frame = x < w;
    const int hx = lane == 0 ? xw - 1 : xw + 1024;
    const bool hvalid = (lane == 0 || lane == 63) && hx >= 0 && hx < w;

    SobelState st;
#pragma unroll
    for (int k = 0; k < 16; ++k) { st.h1p[k] = 0.f; st.cp[k] = 0.f; st.h2r[0][k] = 0.f; st.h2r[1][k] = 0.f; }

    constexpr int NCH = (R + 2 + C - 1) / C;
    auto load_chunk = [&](int ch, uint4 (&q)[C], uint32_t (&hb)[C]) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 - 1 + ch * C + j;
            q[j] = make_uint4(0, 0, 0, 0);
            hb[j] = 0;
            if (yy >= 0 && yy < rows && ch * C + j < R + 2) {
                if (in_frame) q[j] = ldg<NT>(reinterpret_cast<const uint4 *>(src + (size_t)yy * w + x));
                if (hvalid) hb[j] = src[(size_t)yy * w + hx];
            }
        }
    };
    auto compute_chunk = [&](int ch, const uint4 (&q)[C], const uint32_t (&hb)[C]) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = ch * C + j;
            uint32_t ql = __builtin_amdgcn_update_dpp(0u, q[j].w, 0x138, 0xf, 0xf, false);   // lane l <- lane l-1
            uint32_t qr = __builtin_amdgcn_update_dpp(0u, q[j].x, 0x130, 0xf, 0xf, false);   // lane l <- lane l+1
            if (lane == 0) ql = hb[j] << 24;
            if (lane == 63) qr = hb[j];
            float p[18];
            p[0] = ub3(ql);
            p[1] = ub0(q[j].x); p[2] = ub1(q[j].x); p[3] = ub2(q[j].x); p[4] = ub3(q[j].x);
            p[5] = ub0(q[j].y); p[6] = ub1(q[j].y); p[7] = ub2(q[j].y); p[8] = ub3(q[j].y);
            p[9] = ub0(q[j].z); p[10] = ub1(q[j].z); p[11] = ub2(q[j].z); p[12] = ub3(q[j].z);
            p[13] = ub0(q[j].w); p[14] = ub1(q[j].w); p[15] = ub2(q[j].w); p[16] = ub3(q[j].w);
            p[17] = ub0(qr);
            uint4 out = (j & 1) ? sobel_row<1>(st, p, half) : sobel_row<0>(st, p, half);
            if (i >= 2 && i < R + 2) {
                const int y = y0 + i - 2;
                if (x == 0) out.x &= 0xffffff00u;
                if (x + 15 == w - 1) out.w &= 0x00ffffffu;
                if (y == 0 || y == rows - 1) out = make_uint4(0, 0, 0, 0);
                if (in_frame && y < rows) stg<NT>(reinterpret_cast<uint4 *>(dst + (size_t)y * w + x), out);
            }
        }
    };
    if (PF) {
        // software prefetch: chunk ch+1 is in flight while chunk ch is computed (two static register sets)
        uint4 qa[C], qb[C];
        uint32_t ha[C], hb2[C];
        load_chunk(0, qa, ha);
#pragma unroll 1
        for (int ch = 0; ch < NCH; ch += 2) {
            load_chunk(ch + 1, qb, hb2);
            compute_chunk(ch, qa, ha);
            load_chunk(ch + 2, qa, ha);
            if (ch + 1 < NCH) compute_chunk(ch + 1, qb, hb2);
        }
    } else {
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
            uint4 q[C];
            uint32_t hb[C];
            load_chunk(ch, q, hb);
            compute_chunk(ch, q, hb);
        }
    }
}

// ---- fused RGB -> luma -> Sobel, strip streaming -------------------------------------------------
// per row the wave loads its 3 KiB of RGB fully coalesced (lane i: 16 B at +16 i, +1024, +2048),
// transposes through a private 3 KiB LDS slab so each lane owns 16 consecutive pixels, converts to
// luma (IEEE mode), then runs the same Sobel row step.
template <int MODE>
__device__ __forceinline__ float lumaf(uint32_t r, uint32_t g, uint32_t b)
{
#pragma clang fp contract(off)
    const float fr = (float)r, fg = (float)g, fb = (float)b;
    float y;
    if (MODE == 1) { float t = 0.587f * fg; t = __builtin_fmaf(0.299f, fr, t); y = __builtin_fmaf(0.114f, fb, t); }
    else { const float s = 0.299f * fr + 0.587f * fg; y = s + 0.114f * fb; }
    return opaque(__builtin_truncf(y));     // == (float)(unsigned char)y for 0 <= y <= 255
}

template <int R, int C, int NT, int PF = 0, int LB = 1>
__global__ __launch_bounds__(256, LB) void k_fused_stream_f32(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ dst, int w, int rows, int groups_x)
{
    static_assert(C % 2 == 0, "chunk must be even");
    __shared__ __attribute__((aligned(16))) uint8_t slab[4][3072];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int xw = (gx_i * 4 + wave) * 1024;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    if (xw >= w) return;
    float half = 0.5f;
    asm volatile("" : "+v"(half));
    const bool full = xw + 1024 <= w;
    const bool in_frame = x < w;
    const int hx = lane == 0 ? xw - 1 : xw + 1024;
    const bool hvalid = (lane == 0 || lane == 63) && hx >= 0 && hx < w;
    const int nvec = full ? 192 : (w - xw) * 3 / 16;

    SobelState st;
#pragma unroll
    for (int k = 0; k < 16; ++k) { st.h1p[k] = 0.f; st.cp[k] = 0.f; st.h2r[0][k] = 0.f; st.h2r[1][k] = 0.f; }
    uint4 *sw = reinterpret_cast<uint4 *>(slab[wave]);
    const uint4 *sr = reinterpret_cast<const uint4 *>(slab[wave] + 48 * lane);

    constexpr int NCH = (R + 2 + C - 1) / C;
    struct Chunk { uint4 a[C], b[C], c[C]; uint32_t hr[C], hg[C], hbl[C]; };
    auto load_chunk = [&](int ch, Chunk &k) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 - 1 + ch * C + j;
            k.a[j] = k.b[j] = k.c[j] = make_uint4(0, 0, 0, 0);
            k.hr[j] = k.hg[j] = k.hbl[j] = 0;
            if (yy >= 0 && yy < rows && ch * C + j < R + 2) {
                const uint4 *rowp = reinterpret_cast<const uint4 *>(rgb + ((size_t)yy * w + xw) * 3);
                if (lane < nvec) k.a[j] = ldg<NT>(rowp + lane);
                if (64 + lane < nvec) k.b[j] = ldg<NT>(rowp + 64 + lane);
                if (128 + lane < nvec) k.c[j] = ldg<NT>(rowp + 128 + lane);
                if (hvalid) { const uint8_t *hp = rgb + ((size_t)yy * w + hx) * 3; k.hr[j] = hp[0]; k.hg[j] = hp[1]; k.hbl[j] = hp[2]; }
            }
        }
    };
    auto compute_chunk = [&](int ch, const Chunk &k) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = ch * C + j;
            sw[lane] = k.a[j]; sw[64 + lane] = k.b[j]; sw[128 + lane] = k.c[j];
            __builtin_amdgcn_wave_barrier();
            const uint4 u0 = sr[0], u1 = sr[1], u2 = sr[2];
            __builtin_amdgcn_wave_barrier();
            const uint32_t wv[12] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w, u2.x, u2.y, u2.z, u2.w};
            float p[18];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int o = 3 * q;
                p[q + 1] = lumaf<0>(gh_byte(wv[o >> 2], o & 3), gh_byte(wv[(o + 1) >> 2], (o + 1) & 3), gh_byte(wv[(o + 2) >> 2], (o + 2) & 3));
            }
            const float hl = lumaf<0>(k.hr[j], k.hg[j], k.hbl[j]);
            float pl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(uint32_t, p[16]), 0x138, 0xf, 0xf, false));
            float pr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(uint32_t, p[1]), 0x130, 0xf, 0xf, false));
            if (lane == 0) pl = hl;
            if (lane == 63) pr = hl;
            p[0] = pl; p[17] = pr;
            uint4 out = (j & 1) ? sobel_row<1>(st, p, half) : sobel_row<0>(st, p, half);
            if (i >= 2 && i < R + 2) {
                const int y = y0 + i - 2;
                if (x == 0) out.x &= 0xffffff00u;
                if (x + 15 == w - 1) out.w &= 0x00ffffffu;
                if (y == 0 || y == rows - 1) out = make_uint4(0, 0, 0, 0);
                if (in_frame && y < rows) stg<NT>(reinterpret_cast<uint4 *>(dst + (size_t)y * w + x), out);
            }
        }
    };
    if (PF) {
        Chunk ka, kb;
        load_chunk(0, ka);
#pragma unroll 1
        for (int ch = 0; ch < NCH; ch += 2) {
            load_chunk(ch + 1, kb);
            compute_chunk(ch, ka);
            load_chunk(ch + 2, ka);
            if (ch + 1 < NCH) compute_chunk(ch + 1, kb);
        }
    } else {
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
            Chunk k;
            load_chunk(ch, k);
            compute_chunk(ch, k);
        }
    }
}

}  // namespace ubs

static void bench_stream(int W, int H)
{
    using namespace ubs;
    const size_t npix = (size_t)W * H;
    uint8_t *src, *ref, *out, *rgb, *gray;
    CK(hipMalloc(&src, npix)); CK(hipMalloc(&ref, npix)); CK(hipMalloc(&out, npix));
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
    ub::k_sobel_naive<<<8192, 256>>>(src, ref, W, H);
    CK(hipDeviceSynchronize());
    std::vector<Variant> vs;
    auto add = [&](std::string name, std::function<void()> f, const uint8_t *want, double bytes) {
        CK(hipMemset(out, 0xEE, npix));
        f();
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("!! %s: %s\n", name.c_str(), hipGetErrorString(e)); exit(1); }
        const unsigned long long bad = count_diff(want, out, npix);
        if (bad) printf("!! %s: %llu wrong bytes\n", name.c_str(), bad);
        else vs.push_back({name, f, bytes});
    };
    const int gx = (W + 4095) / 4096;
    add("sobel shipping (libgrayhip)", [=] { gh_sobel_u8_dev(src, out, W, H, nullptr); }, ref, 2.0 * npix);
    add("sobel i16 stream R32 C6 pack4", [=] { ub::k_sobel_stream<1, 4, 32, 6><<<gx * ((H + 31) / 32), 256>>>(src, out, W, H, gx); }, ref, 2.0 * npix);
#define SF(R_, C_, NT_) add("sobel f32 stream R" #R_ " C" #C_ " nt" #NT_, [=] { k_sobel_stream_f32<R_, C_, NT_><<<gx * ((H + R_ - 1) / R_), 256>>>(src, out, W, H, gx); }, ref, 2.0 * npix)
    SF(16, 2, 0); SF(16, 4, 0); SF(16, 6, 0); SF(32, 2, 0); SF(32, 4, 0); SF(32, 6, 0); SF(32, 8, 0); SF(64, 4, 0); SF(64, 8, 0);
    SF(16, 4, 1); SF(32, 4, 1); SF(32, 6, 1); SF(64, 4, 1);
#define SP(R_, C_, NT_) add("sobel f32 stream+prefetch R" #R_ " C" #C_ " nt" #NT_, [=] { k_sobel_stream_f32<R_, C_, NT_, 1><<<gx * ((H + R_ - 1) / R_), 256>>>(src, out, W, H, gx); }, ref, 2.0 * npix)
    SP(16, 2, 0); SP(32, 2, 0); SP(32, 4, 0); SP(64, 2, 0); SP(64, 4, 0); SP(32, 2, 1); SP(32, 4, 1); SP(64, 4, 1); SP(128, 4, 1);
    run_table("sobel3x3_u8: f32 strip streaming (1R+1W B/px)", vs, (double)npix);

    // fused
    vs.clear();
    CK(hipMalloc(&rgb, npix * 3)); CK(hipMalloc(&gray, npix));
    k_fill<<<4096, 256>>>((uint32_t *)rgb, npix * 3 / 4, 99);
    gh_gray_rgb8_to_plane_dev(rgb, gray, npix, 3, 0, nullptr);
    ub::k_sobel_naive<<<8192, 256>>>(gray, ref, W, H);
    CK(hipDeviceSynchronize());
    add("fused shipping (libgrayhip)", [=] { gh_gray_sobel_fused_rows_dev(rgb, out, W, H, 3, 0, 1, 1, nullptr); }, ref, 4.0 * npix);
    add("split shipping gray+sobel", [=] { gh_gray_rgb8_to_plane_dev(rgb, gray, npix, 3, 0, nullptr); gh_sobel_u8_dev(gray, out, W, H, nullptr); }, ref, 4.0 * npix);
#define FF(R_, C_, NT_) add("fused f32 stream R" #R_ " C" #C_ " nt" #NT_, [=] { k_fused_stream_f32<R_, C_, NT_><<<gx * ((H + R_ - 1) / R_), 256>>>(rgb, out, W, H, gx); }, ref, 4.0 * npix)
    FF(16, 2, 1); FF(32, 2, 1); FF(32, 4, 1); FF(64, 2, 1);
#define FP(R_, C_, NT_, PF_, LB_) add("fused f32 stream R" #R_ " C" #C_ " nt" #NT_ " pf" #PF_ " lb" #LB_, [=] { k_fused_stream_f32<R_, C_, NT_, PF_, LB_><<<gx * ((H + R_ - 1) / R_), 256>>>(rgb, out, W, H, gx); }, ref, 4.0 * npix)
    FP(32, 2, 1, 1, 1); FP(16, 2, 1, 1, 1); FP(64, 2, 1, 1, 1); FP(32, 4, 1, 1, 1); FP(32, 2, 1, 1, 2); FP(32, 2, 1, 0, 2); FP(32, 2, 1, 1, 3); FP(32, 2, 1, 0, 3); FP(32, 2, 0, 1, 1); FP(8, 2, 1, 1, 1);
    run_table("fused gray+sobel (3R+1W B/px)", vs, (double)npix);
    CK(hipFree(src)); CK(hipFree(ref)); CK(hipFree(out)); CK(hipFree(rgb)); CK(hipFree(gray));
}

// ---- i16-packed strip streaming, v2: compact rolling state (h1p, cp, h2 ring = 32 VGPRs), optional
// software prefetch, launch-bounds-controlled occupancy, RTZ tail ------------------------------------------
namespace ubs2 {
using sobel_tile::s16x2;
using sobel_tile::as_s16x2;

struct St16 { s16x2 h1p[8], cp[8], h2r[2][8]; };

template <int PAR>
__device__ __forceinline__ uint4 row16(St16 &st, const uint4 &qv, uint32_t ql, uint32_t qr, float half)
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
    uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const s16x2 h1 = o[k + 1] - o[k];
        const s16x2 h2 = ub::pk_mad2(e[k], o[k] + o[k + 1]);
        const s16x2 c = st.h1p[k] + h1;
        const s16x2 gx = st.cp[k] + c;
        const s16x2 gy = st.h2r[PAR][k] - h2;
        st.h1p[k] = h1; st.cp[k] = c; st.h2r[PAR][k] = h2;
#pragma unroll
        for (int el = 0; el < 2; ++el) {
            const int p = 2 * k + el;
            const float fx = (float)(int)(el ? gx.y : gx.x), fy = (float)(int)(el ? gy.y : gy.x);
            float t;
            asm("v_fma_f32 %0, %1, %1, %2" : "=v"(t) : "v"(fy), "v"(half));
            const float f = __builtin_amdgcn_sqrtf(__builtin_fmaf(fx, fx, t));
            r[p >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(f, p & 3, r[p >> 2]);    // MODE.fp_round = RTZ -> floor, saturating
        }
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

template <int R, int C, int NT, int PF, int LB>
__global__ __launch_bounds__(256, LB) void k_sobel_stream16(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int groups_x)
{
    static_assert(C % 2 == 0, "chunk must be even");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int xw = (gx_i * 4 + wave) * 1024;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    if (xw >= w) return;
    __builtin_amdgcn_s_setreg(0x0801, 3);
    float half = 0.5f;
    asm volatile("" : "+v"(half));
    const bool in_frame = x < w;
    const int hx = lane == 0 ? xw - 1 : xw + 1024;
    const bool hvalid = (lane == 0 || lane == 63) && hx >= 0 && hx < w;
    St16 st;
#pragma unroll
    for (int k = 0; k < 8; ++k) { st.h1p[k] = as_s16x2(0u); st.cp[k] = as_s16x2(0u); st.h2r[0][k] = as_s16x2(0u); st.h2r[1][k] = as_s16x2(0u); }
    constexpr int NCH = (R + 2 + C - 1) / C;
    auto load_chunk = [&](int ch, uint4 (&q)[C], uint32_t (&hb)[C]) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 - 1 + ch * C + j;
            q[j] = make_uint4(0, 0, 0, 0);
            hb[j] = 0;
            if (yy >= 0 && yy < rows && ch * C + j < R + 2) {
                if (in_frame) q[j] = ldg<NT>(reinterpret_cast<const uint4 *>(src + (size_t)yy * w + x));
                if (hvalid) hb[j] = src[(size_t)yy * w + hx];
            }
        }
    };
    auto compute_chunk = [&](int ch, const uint4 (&q)[C], const uint32_t (&hb)[C]) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = ch * C + j;
            uint32_t ql = __builtin_amdgcn_update_dpp(0u, q[j].w, 0x138, 0xf, 0xf, false);
            uint32_t qr = __builtin_amdgcn_update_dpp(0u, q[j].x, 0x130, 0xf, 0xf, false);
            if (lane == 0) ql = hb[j] << 24;
            if (lane == 63) qr = hb[j];
            uint4 out = (j & 1) ? row16<1>(st, q[j], ql, qr, half) : row16<0>(st, q[j], ql, qr, half);
            if (i >= 2 && i < R + 2) {
                const int y = y0 + i - 2;
                if (x == 0) out.x &= 0xffffff00u;
                if (x + 15 == w - 1) out.w &= 0x00ffffffu;
                if (y == 0 || y == rows - 1) out = make_uint4(0, 0, 0, 0);
                if (in_frame && y < rows) stg<NT>(reinterpret_cast<uint4 *>(dst + (size_t)y * w + x), out);
            }
        }
    };
    if (PF) {
        uint4 qa[C], qb[C];
        uint32_t ha[C], hb2[C];
        load_chunk(0, qa, ha);
#pragma unroll 1
        for (int ch = 0; ch < NCH; ch += 2) {
            load_chunk(ch + 1, qb, hb2);
            compute_chunk(ch, qa, ha);
            load_chunk(ch + 2, qa, ha);
            if (ch + 1 < NCH) compute_chunk(ch + 1, qb, hb2);
        }
    } else {
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
            uint4 q[C];
            uint32_t hb[C];
            load_chunk(ch, q, hb);
            compute_chunk(ch, q, hb);
        }
    }
}
}  // namespace ubs2

static void bench_stream16(int W, int H)
{
    using namespace ubs2;
    const size_t npix = (size_t)W * H;
    uint8_t *src, *ref, *out;
    CK(hipMalloc(&src, npix)); CK(hipMalloc(&ref, npix)); CK(hipMalloc(&out, npix));
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
    ub::k_sobel_naive<<<8192, 256>>>(src, ref, W, H);
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
    const int gx = (W + 4095) / 4096;
    add("i16 stream v1 R32 C6 pack4", [=] { ub::k_sobel_stream<1, 4, 32, 6><<<gx * ((H + 31) / 32), 256>>>(src, out, W, H, gx); });
#define S16(R_, C_, NT_, PF_, LB_) add("i16v2 R" #R_ " C" #C_ " nt" #NT_ " pf" #PF_ " lb" #LB_, [=] { k_sobel_stream16<R_, C_, NT_, PF_, LB_><<<gx * ((H + R_ - 1) / R_), 256>>>(src, out, W, H, gx); })
    S16(32, 2, 0, 0, 1); S16(32, 4, 0, 0, 1); S16(32, 6, 0, 0, 1); S16(32, 2, 0, 1, 1); S16(32, 4, 0, 1, 1);
    S16(32, 2, 0, 0, 5); S16(32, 4, 0, 0, 5); S16(32, 2, 0, 1, 5); S16(32, 4, 0, 1, 5);
    S16(32, 2, 0, 0, 6); S16(32, 4, 0, 0, 6); S16(32, 2, 0, 1, 6); S16(32, 2, 0, 0, 8); S16(32, 2, 0, 1, 8);
    S16(32, 4, 1, 0, 5); S16(32, 2, 1, 1, 5); S16(32, 4, 1, 1, 5); S16(16, 2, 1, 1, 5); S16(64, 2, 1, 1, 5); S16(64, 4, 1, 1, 5); S16(32, 2, 1, 1, 6);
    run_table("sobel3x3_u8: i16 strip streaming v2", vs, (double)npix);
    CK(hipFree(src)); CK(hipFree(ref)); CK(hipFree(out));
}

// ---- Design C: component-packed i16 (lo = h1/gx, hi = h2/gy per pixel), v_dot2_i32_i16 for gx^2+gy^2 ----
namespace ubc {
using sobel_tile::s16x2;

__device__ __forceinline__ uint32_t pk_mad_k(uint32_t a, uint32_t k_sgpr, uint32_t c)
{
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k_sgpr), "v"(c));
    return d;
}

struct StC { uint32_t ring[2][16]; };

// MAGV: 0 = cvt_f32_i32 + sqrt under RTZ ; 1 = same but with explicit floor (RNE-safe)
template <int PAR, int MAGV>
__device__ __forceinline__ uint4 rowc(StC &st, const uint4 &qv, uint32_t ql, uint32_t qr)
{
    const uint32_t q[4] = {qv.x, qv.y, qv.z, qv.w};
    uint32_t D[18];
    D[0] = __builtin_amdgcn_perm(0u, ql, 0x0c030c03u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        D[1 + 4 * j] = __builtin_amdgcn_perm(0u, q[j], 0x0c000c00u);
        D[2 + 4 * j] = __builtin_amdgcn_perm(0u, q[j], 0x0c010c01u);
        D[3 + 4 * j] = __builtin_amdgcn_perm(0u, q[j], 0x0c020c02u);
        D[4 + 4 * j] = __builtin_amdgcn_perm(0u, q[j], 0x0c030c03u);
    }
    D[17] = __builtin_amdgcn_perm(0u, qr, 0x0c000c00u);
    const uint32_t K1 = 0x0001FFFFu;   // (lo,hi) = (-1, +1)
    const uint32_t K2 = 0x00020000u;   // ( 0, +2)
    const uint32_t KB = 0x00000002u;   // (+2,  0)
    const uint32_t KD = 0xFFFF0001u;   // (+1, -1)
    uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t W = pk_mad_k(D[k], K1, D[k + 2]);          // (p[x+1]-p[x-1], p[x+1]+p[x-1])
        W = pk_mad_k(D[k + 1], K2, W);                       // (h1, h2)
        uint32_t G = pk_mad_k(st.ring[PAR ^ 1][k], KB, st.ring[PAR][k]);   // row i-2 + 2*(row i-1).lo
        G = pk_mad_k(W, KD, G);                              // (gx, gy)
        st.ring[PAR][k] = W;
        const int m = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, G), __builtin_bit_cast(s16x2, G), 0, false);
        float f = __builtin_amdgcn_sqrtf((float)m);
        if (MAGV == 1) f = __builtin_floorf(f);
        r[k >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(f, k & 3, r[k >> 2]);
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

template <int R, int C, int NT, int PF, int MAGV>
__global__ __launch_bounds__(256) void k_sobel_stream_c(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int groups_x)
{
    static_assert(C % 2 == 0, "chunk must be even");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int xw = (gx_i * 4 + wave) * 1024;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    if (xw >= w) return;
    if (MAGV == 0) __builtin_amdgcn_s_setreg(0x0801, 3);
    const bool in_frame = x < w;
    const int hx = lane == 0 ? xw - 1 : xw + 1024;
    const bool hvalid = (lane == 0 || lane == 63) && hx >= 0 && hx < w;
    StC st;
#pragma unroll
    for (int k = 0; k < 16; ++k) { st.ring[0][k] = 0; st.ring[1][k] = 0; }
    constexpr int NCH = (R + 2 + C - 1) / C;
    auto load_chunk = [&](int ch, uint4 (&q)[C], uint32_t (&hb)[C]) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 - 1 + ch * C + j;
            q[j] = make_uint4(0, 0, 0, 0);
            hb[j] = 0;
            if (yy >= 0 && yy < rows && ch * C + j < R + 2) {
                if (in_frame) q[j] = ldg<NT>(reinterpret_cast<const uint4 *>(src + (size_t)yy * w + x));
                if (hvalid) hb[j] = src[(size_t)yy * w + hx];
            }
        }
    };
    auto compute_chunk = [&](int ch, const uint4 (&q)[C], const uint32_t (&hb)[C]) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = ch * C + j;
            uint32_t ql = __builtin_amdgcn_update_dpp(0u, q[j].w, 0x138, 0xf, 0xf, false);
            uint32_t qr = __builtin_amdgcn_update_dpp(0u, q[j].x, 0x130, 0xf, 0xf, false);
            if (lane == 0) ql = hb[j] << 24;
            if (lane == 63) qr = hb[j];
            uint4 out = (j & 1) ? rowc<1, MAGV>(st, q[j], ql, qr) : rowc<0, MAGV>(st, q[j], ql, qr);
            if (i >= 2 && i < R + 2) {
                const int y = y0 + i - 2;
                if (x == 0) out.x &= 0xffffff00u;
                if (x + 15 == w - 1) out.w &= 0x00ffffffu;
                if (y == 0 || y == rows - 1) out = make_uint4(0, 0, 0, 0);
                if (in_frame && y < rows) stg<NT>(reinterpret_cast<uint4 *>(dst + (size_t)y * w + x), out);
            }
        }
    };
    if (PF) {
        uint4 qa[C], qb[C];
        uint32_t ha[C], hb2[C];
        load_chunk(0, qa, ha);
#pragma unroll 1
        for (int ch = 0; ch < NCH; ch += 2) {
            load_chunk(ch + 1, qb, hb2);
            compute_chunk(ch, qa, ha);
            load_chunk(ch + 2, qa, ha);
            if (ch + 1 < NCH) compute_chunk(ch + 1, qb, hb2);
        }
    } else {
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
            uint4 q[C];
            uint32_t hb[C];
            load_chunk(ch, q, hb);
            compute_chunk(ch, q, hb);
        }
    }
}
}  // namespace ubc

static void bench_streamc(int W, int H)
{
    using namespace ubc;
    const size_t npix = (size_t)W * H;
    uint8_t *src, *ref, *out;
    CK(hipMalloc(&src, npix)); CK(hipMalloc(&ref, npix)); CK(hipMalloc(&out, npix));
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
    ub::k_sobel_naive<<<8192, 256>>>(src, ref, W, H);
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
    const int gx = (W + 4095) / 4096;
    add("i16 stream v1 R32 C6 pack4", [=] { ub::k_sobel_stream<1, 4, 32, 6><<<gx * ((H + 31) / 32), 256>>>(src, out, W, H, gx); });
#define SC(R_, C_, NT_, PF_, MV_) add("designC R" #R_ " C" #C_ " nt" #NT_ " pf" #PF_ " mag" #MV_, [=] { k_sobel_stream_c<R_, C_, NT_, PF_, MV_><<<gx * ((H + R_ - 1) / R_), 256>>>(src, out, W, H, gx); })
    SC(32, 2, 0, 0, 0); SC(32, 4, 0, 0, 0); SC(32, 6, 0, 0, 0); SC(32, 2, 0, 1, 0); SC(32, 4, 0, 1, 0);
    SC(32, 2, 1, 1, 0); SC(32, 4, 1, 0, 0); SC(16, 4, 1, 0, 0); SC(64, 4, 1, 0, 0); SC(64, 2, 1, 1, 0);
    SC(32, 4, 0, 0, 1); SC(32, 2, 1, 1, 1);
    run_table("sobel3x3_u8: design C (component-packed i16 + dot2)", vs, (double)npix);
    CK(hipFree(src)); CK(hipFree(ref)); CK(hipFree(out));
}

// ---- access-pattern ceiling: the strip-streaming loop with NO arithmetic (row copy) --------------------
namespace ubp {
template <int R, int C, int NT, int WORK>
__global__ __launch_bounds__(256) void k_strip_copy(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int groups_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int xw = (gx_i * 4 + wave) * 1024;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    if (xw >= w || x >= w) return;
    constexpr int NCH = (R + C - 1) / C;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        uint4 q[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            q[j] = make_uint4(0, 0, 0, 0);
            if (yy < rows) q[j] = ldg<NT>(reinterpret_cast<const uint4 *>(src + (size_t)yy * w + x));
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            uint4 v = q[j];
#pragma unroll
            for (int k = 0; k < WORK; ++k) {   // WORK dependent-free VALU ops per dword to emulate compute load
                v.x = __builtin_amdgcn_perm(v.x, v.y, 0x07020500u + k); v.y = __builtin_amdgcn_perm(v.y, v.z, 0x03060104u + k);
                v.z = __builtin_amdgcn_perm(v.z, v.w, 0x07020500u + k); v.w = __builtin_amdgcn_perm(v.w, v.x, 0x03060104u + k);
            }
            if (yy < rows) stg<NT>(reinterpret_cast<uint4 *>(dst + (size_t)yy * w + x), v);
        }
    }
}

// P4: WPB waves side by side per workgroup (workgroup spans WPB KiB of each row; WPB=16 -> a whole 16 KiB row)
template <int R, int C, int NT, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_strip_copy_wide(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int groups_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int xw = (gx_i * WPB + wave) * 1024;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    if (xw >= w || x >= w) return;
    constexpr int NCH = (R + C - 1) / C;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        uint4 q[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            q[j] = make_uint4(0, 0, 0, 0);
            if (yy < rows) q[j] = ldg<NT>(reinterpret_cast<const uint4 *>(src + (size_t)yy * w + x));
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            if (yy < rows) stg<NT>(reinterpret_cast<uint4 *>(dst + (size_t)yy * w + x), q[j]);
        }
    }
}

// P1: one wave marches ALONG a band of R rows (x advances by 1 KiB per step, all R rows loaded per step);
// bands start at skewed x offsets so that concurrently running waves sit on different channels.
template <int R, int NT, int SKEW>
__global__ __launch_bounds__(256) void k_band_march(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int band = (int)t * 4 + wave;
    const int y0 = band * R;
    if (y0 >= rows) return;
    const int steps = w / 1024;
    const int start = SKEW ? (band * SKEW) % steps : 0;
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        int sx = start + s; if (sx >= steps) sx -= steps;
        const int x = sx * 1024 + 16 * lane;
        uint4 q[R];
#pragma unroll
        for (int j = 0; j < R; ++j) { q[j] = make_uint4(0, 0, 0, 0); if (y0 + j < rows) q[j] = ldg<NT>(reinterpret_cast<const uint4 *>(src + (size_t)(y0 + j) * w + x)); }
#pragma unroll
        for (int j = 0; j < R; ++j) if (y0 + j < rows) stg<NT>(reinterpret_cast<uint4 *>(dst + (size_t)(y0 + j) * w + x), q[j]);
    }
}

// sheared strips: strip column j starts its bands at row offset -(j*SK mod R), so concurrently running waves sit on
// different rows (different DRAM banks) instead of all on rows == const (mod R).
template <int R, int C, int NT, int SK>
__global__ __launch_bounds__(256) void k_strip_copy_shear(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int groups_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int col = gx_i * 4 + wave;
    const int xw = col * 1024;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R - (col * SK) % R;
    if (xw >= w || x >= w) return;
    constexpr int NCH = (R + C - 1) / C;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        uint4 q[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            q[j] = make_uint4(0, 0, 0, 0);
            if (yy >= 0 && yy < rows) q[j] = ldg<NT>(reinterpret_cast<const uint4 *>(src + (size_t)yy * w + x));
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            if (yy >= 0 && yy < rows) stg<NT>(reinterpret_cast<uint4 *>(dst + (size_t)yy * w + x), q[j]);
        }
    }
}

// 3R:1W pattern (fused shape): per row a wave reads 3 KiB contiguous RGB (3 coalesced dwordx4 per lane) and writes 1 KiB
template <int R, int C, int NT, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_pat31(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ dst, int w, int rows, int groups_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = ub_remap(blockIdx.x, gridDim.x);
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int xw = (gx_i * WPB + wave) * 1024;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    if (xw >= w || x >= w) return;
    constexpr int NCH = (R + C - 1) / C;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        uint4 a[C], b[C], c[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            a[j] = b[j] = c[j] = make_uint4(0, 0, 0, 0);
            if (yy < rows) {
                const uint4 *rowp = reinterpret_cast<const uint4 *>(rgb + ((size_t)yy * w + xw) * 3);
                a[j] = ldg<NT>(rowp + lane); b[j] = ldg<NT>(rowp + 64 + lane); c[j] = ldg<NT>(rowp + 128 + lane);
            }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            const uint4 v = make_uint4(a[j].x ^ b[j].y ^ c[j].z, a[j].y ^ b[j].z ^ c[j].w, a[j].z ^ b[j].w ^ c[j].x, a[j].w ^ b[j].x ^ c[j].y);
            if (yy < rows) stg<NT>(reinterpret_cast<uint4 *>(dst + (size_t)yy * w + x), v);
        }
    }
}

// pitch experiment: the same strip copy but rows are `pitch` bytes apart (pitch >= w): does a non power-of-two row stride help?
template <int R, int C, int NT>
__global__ __launch_bounds__(256) void k_strip_copy_pitch(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int groups_x, size_t pitch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = blockIdx.x;
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int xw = (gx_i * 4 + wave) * 1024;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    if (xw >= w || x >= w) return;
    constexpr int NCH = (R + C - 1) / C;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        uint4 q[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            q[j] = make_uint4(0, 0, 0, 0);
            if (yy < rows) q[j] = ldg<NT>(reinterpret_cast<const uint4 *>(src + (size_t)yy * pitch + x));
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            if (yy < rows) stg<NT>(reinterpret_cast<uint4 *>(dst + (size_t)yy * pitch + x), q[j]);
        }
    }
}

// cache-policy experiment: 3R:1W strip copy with buffer loads/stores and explicit cpol bits
//   aux bit0 = sc0, bit1 = nt, bit4 = sc1 (gfx940+ encoding of the LLVM cachepolicy immediate)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
template <int R, int C, int LAUX, int SAUX>
__global__ __launch_bounds__(256) void k_pat31_cpol(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ dst, int w, int rows, int groups_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = blockIdx.x;
    const int gx_i = (int)(t % (unsigned)groups_x), gy_i = (int)(t / (unsigned)groups_x);
    const int xw = (gx_i * 4 + wave) * 1024;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    if (xw >= w || x >= w) return;
    constexpr int NCH = (R + C - 1) / C;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        u32x4 a[C], b[C], c[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            const uint8_t *rowp = rgb + ((size_t)yy * w + xw) * 3;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(rowp, 3072);
            a[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, LAUX));
            b[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 1024 + lane * 16, 0, LAUX));
            c[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 2048 + lane * 16, 0, LAUX));
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 + ch * C + j;
            u32x4 v = a[j] ^ b[j] ^ c[j];
            const __amdgpu_buffer_rsrc_t rd = make_rsrc(dst + (size_t)yy * w + xw, 1024);
            __builtin_amdgcn_raw_buffer_store_b128(v, rd, lane * 16, 0, SAUX);
        }
    }
}
}  // namespace ubp

static void bench_pattern(int W, int H)
{
    using namespace ubp;
    const size_t npix = (size_t)W * H;
    uint8_t *src, *out;
    CK(hipMalloc(&src, npix)); CK(hipMalloc(&out, npix));
    k_fill<<<4096, 256>>>((uint32_t *)src, npix / 4, 5);
    std::vector<Variant> vs;
    const int gx = (W + 4095) / 4096;
    const size_t n16 = npix / 16;
    vs.push_back({"linear copy 268MB grid=all", [=] { k_copy<0><<<(unsigned)((n16 + 255) / 256), 256>>>((const uint4 *)src, (uint4 *)out, n16); }, 2.0 * npix});
    vs.push_back({"linear copy nt 268MB grid=all", [=] { k_copy<1><<<(unsigned)((n16 + 255) / 256), 256>>>((const uint4 *)src, (uint4 *)out, n16); }, 2.0 * npix});
#define PC(R_, C_, NT_, WK_) vs.push_back({"strip copy R" #R_ " C" #C_ " nt" #NT_ " work" #WK_, [=] { k_strip_copy<R_, C_, NT_, WK_><<<gx * ((H + R_ - 1) / R_), 256>>>(src, out, W, H, gx); }, 2.0 * npix})
    PC(32, 4, 0, 0); PC(32, 4, 1, 0); PC(16, 4, 1, 0); PC(64, 4, 1, 0); PC(32, 2, 1, 0); PC(32, 8, 1, 0);
    PC(32, 4, 1, 16);
#define PW(R_, C_, NT_, WPB_) vs.push_back({"wide strip copy R" #R_ " C" #C_ " nt" #NT_ " waves/blk" #WPB_, [=] { const int g = (W + 1024 * WPB_ - 1) / (1024 * WPB_); k_strip_copy_wide<R_, C_, NT_, WPB_><<<g * ((H + R_ - 1) / R_), 64 * WPB_>>>(src, out, W, H, g); }, 2.0 * npix})
    PW(32, 4, 1, 8); PW(32, 4, 1, 16); PW(16, 4, 1, 16); PW(8, 4, 1, 16); PW(64, 4, 1, 16); PW(32, 2, 1, 16); PW(32, 8, 1, 16); PW(32, 4, 0, 16); PW(16, 2, 1, 16); PW(8, 2, 1, 16); PW(4, 4, 1, 16);
#define PB(R_, NT_, SK_) vs.push_back({"band march R" #R_ " nt" #NT_ " skew" #SK_, [=] { const int bands = (H + R_ - 1) / R_; k_band_march<R_, NT_, SK_><<<(bands + 3) / 4, 256>>>(src, out, W, H); }, 2.0 * npix})
#define PS(R_, C_, NT_, SK_) vs.push_back({"sheared strip copy R" #R_ " C" #C_ " nt" #NT_ " shear" #SK_, [=] { k_strip_copy_shear<R_, C_, NT_, SK_><<<gx * ((H + R_ - 1) / R_ + 1), 256>>>(src, out, W, H, gx); }, 2.0 * npix})
    PS(32, 4, 1, 0); PS(32, 4, 1, 1); PS(32, 4, 1, 2); PS(32, 4, 1, 3); PS(32, 4, 1, 5); PS(32, 4, 1, 7); PS(32, 4, 1, 13);
    PS(16, 4, 1, 1); PS(16, 4, 1, 3); PS(16, 4, 1, 5); PS(64, 4, 1, 5); PS(64, 4, 1, 13); PS(32, 2, 1, 5); PS(32, 2, 1, 13); PS(32, 4, 0, 5); PS(64, 2, 1, 13); PS(128, 4, 1, 29);
    PB(8, 1, 0); PB(8, 1, 1); PB(8, 1, 3); PB(8, 1, 5); PB(4, 1, 1); PB(16, 1, 1); PB(16, 1, 5); PB(8, 0, 1);
    run_table("access-pattern ceiling: 1R+1W, 268 MB plane (fits Infinity Cache partially)", vs, (double)npix);
    CK(hipFree(src)); CK(hipFree(out));
}

static void bench_pattern31(int W, int H)
{
    using namespace ubp;
    const size_t npix = (size_t)W * H;
    uint8_t *rgb, *out;
    CK(hipMalloc(&rgb, npix * 3)); CK(hipMalloc(&out, npix));
    k_fill<<<4096, 256>>>((uint32_t *)rgb, npix * 3 / 4, 5);
    std::vector<Variant> vs;
    const size_t ng = npix / 16;
    vs.push_back({"linear read3:write1 grid=all", [=] { k_read3_write1<0><<<(unsigned)((ng + 255) / 256), 256>>>((const uint4 *)rgb, (uint4 *)out, ng); }, 4.0 * npix});
    vs.push_back({"gray LDS-transposed nt (real kernel)", [=] { k_gray_lds<1><<<(unsigned)((ng / 64 + 3) / 4), 256>>>((const uint4 *)rgb, (uint4 *)out, ng); }, 4.0 * npix});
#define P31(R_, C_, NT_, WPB_) vs.push_back({"strip 3R:1W R" #R_ " C" #C_ " nt" #NT_ " waves/blk" #WPB_, [=] { const int g = (W + 1024 * WPB_ - 1) / (1024 * WPB_); k_pat31<R_, C_, NT_, WPB_><<<g * ((H + R_ - 1) / R_), 64 * WPB_>>>(rgb, out, W, H, g); }, 4.0 * npix})
    P31(32, 2, 1, 4); P31(32, 4, 1, 4); P31(16, 2, 1, 4); P31(64, 2, 1, 4); P31(8, 2, 1, 4); P31(32, 2, 0, 4); P31(32, 1, 1, 4);
    P31(32, 2, 1, 16); P31(16, 2, 1, 16); P31(8, 2, 1, 16); P31(64, 2, 1, 16); P31(32, 2, 1, 8); P31(32, 2, 1, 2); P31(32, 2, 1, 1); P31(4, 2, 1, 16); P31(2, 2, 1, 16); P31(1, 1, 1, 16); P31(1, 1, 1, 4);
#define PCP(L_, S_) vs.push_back({"strip 3R:1W R32 C2 buffer ld aux" #L_ " st aux" #S_, [=] { const int g = (W + 4095) / 4096; k_pat31_cpol<32, 2, L_, S_><<<g * ((H + 31) / 32), 256>>>(rgb, out, W, H, g); }, 4.0 * npix})
    PCP(0, 0); PCP(2, 2); PCP(2, 0); PCP(0, 2); PCP(1, 2); PCP(3, 2); PCP(16, 2); PCP(18, 2); PCP(17, 2); PCP(19, 2); PCP(2, 18); PCP(2, 16); PCP(2, 3); PCP(2, 19); PCP(18, 18);
    // occupancy sensitivity: same strip copy, workgroups per CU limited by a dummy dynamic-LDS allocation
#define P31L(R_, C_, KB_) vs.push_back({"strip 3R:1W R" #R_ " C" #C_ " nt1 waves/blk4 LDS" #KB_ "KiB", [=] { const int g = (W + 4095) / 4096; k_pat31<R_, C_, 1, 4><<<g * ((H + R_ - 1) / R_), 256, KB_ * 1024>>>(rgb, out, W, H, g); }, 4.0 * npix})
    P31L(32, 2, 20); P31L(32, 2, 26); P31L(32, 2, 32); P31L(32, 2, 40); P31L(32, 2, 53); P31L(32, 2, 80); P31L(32, 4, 53); P31L(32, 6, 53); P31L(32, 4, 80); P31L(32, 8, 80);
    run_table("access-pattern ceiling: 3R+1W (fused shape)", vs, (double)npix);
    CK(hipFree(rgb)); CK(hipFree(out));
}

static void bench_pitch(int W, int H)
{
    using namespace ubp;
    const size_t maxpitch = (size_t)W + 8192;
    uint8_t *src, *out;
    CK(hipMalloc(&src, maxpitch * H + 4096)); CK(hipMalloc(&out, maxpitch * H + 4096 + 1024));
    k_fill<<<4096, 256>>>((uint32_t *)src, maxpitch * H / 4, 5);
    std::vector<Variant> vs;
    const int gx = (W + 4095) / 4096;
    const double npix = (double)W * H;
    for (size_t pad : {0, 256, 512, 1024, 2048, 4096, 4352, 8192, 128, 768}) {
        const size_t pitch = (size_t)W + pad;
        vs.push_back({"strip copy R32 C2 nt1 pitch=w+" + std::to_string(pad), [=] { k_strip_copy_pitch<32, 2, 1><<<gx * ((H + 31) / 32), 256>>>(src, out, W, H, gx, pitch); }, 2.0 * npix});
    }
    for (size_t pad : {0, 256, 1024}) {
        const size_t pitch = (size_t)W + pad;
        vs.push_back({"strip copy R16 C4 nt1 pitch=w+" + std::to_string(pad), [=] { k_strip_copy_pitch<16, 4, 1><<<gx * ((H + 15) / 16), 256>>>(src, out, W, H, gx, pitch); }, 2.0 * npix});
        vs.push_back({"strip copy R32 C2 dst+1024 pitch=w+" + std::to_string(pad), [=] { k_strip_copy_pitch<32, 2, 1><<<gx * ((H + 31) / 32), 256>>>(src, out + 1024, W, H, gx, pitch); }, 2.0 * npix});
    }
    // band-height experiment: is a power-of-two band stride (R * pitch) the problem?
#define PR(R_, C_) vs.push_back({"strip copy R" #R_ " C" #C_ " nt1 pitch=w", [=] { k_strip_copy_pitch<R_, C_, 1><<<gx * ((H + R_ - 1) / R_), 256>>>(src, out, W, H, gx, (size_t)W); }, 2.0 * npix})
    PR(32, 2); PR(30, 2); PR(34, 2); PR(31, 1); PR(33, 1); PR(28, 2); PR(36, 2); PR(62, 2); PR(64, 2); PR(66, 2); PR(60, 2); PR(48, 2); PR(46, 2); PR(50, 2); PR(24, 2); PR(22, 2); PR(26, 2); PR(16, 2); PR(14, 2); PR(18, 2); PR(126, 2); PR(128, 2); PR(130, 2); PR(8, 2); PR(10, 2); PR(6, 2);
    run_table("row-pitch experiment: 1R+1W strip copy", vs, npix);
    CK(hipFree(src)); CK(hipFree(out));
}
