// conv3x3.hip -- generic 3x3 integer convolution on interleaved RGB8 for gfx950 (MI355X).
//
// Replaces old/parallel_convolution.c:8-24 restated on u8 frames: per channel
//   out(r,c,p) = clamp( trunc( sum_{i,j} in(r+i-1, c+j-1, p) * k[i][j] / norm ), 0, 255 )
// with a zero border supplied by the kernel (the reference pre-pads its array) and C integer
// division (truncation toward zero).  The frame is treated as a byte plane of width 3*w: output
// byte xb depends on input bytes xb-3, xb, xb+3 of three rows.  Bound: HBM, 3 R + 3 W bytes per pixel.
//
//   k_conv3x3_stream  strip streaming (same walk as the Sobel kernels, stream_core.h): a wave owns 1024 bytes
//                     of every row, taps and sums in packed f32 on exact integers, two bytes per VALU op
//                     ("pair layout": byte m paired with byte m+8), rolling partial sums so that every input
//                     row is unpacked and multiplied once.  Needs 3*w % 16 == 0, 16-byte aligned frames,
//                     0 < norm <= 4096 and sum|k| * 255 < 2^24 (exactness conditions below).
//   k_conv3x3_px      one output byte per lane, any geometry / kernel: the fallback.
#include "gh_common.h"
#include "stream_core.h"

namespace {

constexpr int BLOCK = 256;

struct Taps { int k[9]; int norm; };

__global__ __launch_bounds__(BLOCK) void k_conv3x3_px(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int h, Taps t)
{
    const size_t rowb = (size_t)w * 3;
    const size_t n = rowb * h;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / rowb);
        const int xb = (int)(i - (size_t)y * rowb);
        int acc = 0;
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int yy = y + di - 1;
            if (yy < 0 || yy >= h) continue;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int xx = xb + 3 * (dj - 1);
                if (xx < 0 || xx >= (int)rowb) continue;
                acc += (int)src[(size_t)yy * rowb + xx] * t.k[di * 3 + dj];
            }
        }
        acc /= t.norm;
        acc = acc < 0 ? 0 : (acc > 255 ? 255 : acc);
        dst[i] = (uint8_t)acc;
    }
}

// ---- strip streaming -------------------------------------------------------------------------------------
// Exactness: every partial sum is an integer of magnitude < 2^24, so f32 adds / fmas are exact.  The quotient:
// q = trunc((acc + 0.5) * (1/norm)) under MODE.fp_round = toward zero.  For acc >= 0 the true (acc + 0.5)/norm is at
// least 0.5/norm away from an integer; the computed value is off by < 3 ulp, i.e. < 1e-4 for quotients < 300, which is
// below 0.5/norm for norm <= 4096 -- so the truncation is floor(acc/norm) whenever that is < 300, and anything larger
// saturates to 255 in v_cvt_pk_u8_f32 like the reference's clamp.  Negative acc gives a quotient <= 0 -> 0 either way.
struct TapsF { float k[9]; float inv, half_inv; };

template <int R, int C>
__global__ __launch_bounds__(stream_core::BLOCK) void k_conv3x3_stream(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                                      int wb, int h, int groups_x, int order, TapsF t)
{
    using namespace stream_core;
    constexpr int NROWS = R + 2;
    static_assert(NROWS % C == 0, "R + 2 must be a multiple of the chunk");
    constexpr int NCH = NROWS / C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    BandPlan plan{};                       // only the workgroup -> (band, column) mapping is used (see stream_core::band_of_block)
    plan.bands = (h + R - 1) / R;
    plan.order = order;
    int gx_i, gy_i;
    band_of_block(plan, blockIdx.x, groups_x, gx_i, gy_i);
    if (gy_i < 0) return;
    const int xw = (gx_i * WAVES_PER_BLOCK + wave) * STRIP_PX;   // first BYTE of this wave's strip
    if (xw >= wb) return;
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    __builtin_amdgcn_s_setreg(0x0801, 3);   // MODE.fp_round(f32) = toward zero (see above); nothing here is rounding-sensitive
    const bool in_frame = x < wb;
    const bool full = xw + STRIP_PX <= wb;
    // strip halo: lane 0 needs the 3 bytes before the strip (inside one aligned dword + ...): fetch the aligned 4 bytes
    // that END at the strip start, lane 63 the 4 bytes that START after the strip end.
    const int hxb = lane == 0 ? xw - 4 : xw + STRIP_PX;
    const bool hvalid = (lane == 0 || lane == 63) && hxb >= 0 && hxb + 4 <= wb;

    const f32x2 inv2 = {t.inv, t.inv}, hinv2 = {t.half_inv, t.half_inv};
    f32x2 kk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) kk[i] = f32x2{t.k[i], t.k[i]};

    // rolling partial sums for byte pairs (k, k+8): a0 = H0(row r-1) ... see the update below
    f32x2 a1[8], a0[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a1[k] = f32x2{0.f, 0.f}; a0[k] = f32x2{0.f, 0.f}; }

    auto load_chunk = [&](int ch, uint4 (&q)[C], uint32_t (&hb)[C]) {
        if (ch >= NCH) return;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = y0 - 1 + ch * C + j;
            q[j] = make_uint4(0, 0, 0, 0);
            hb[j] = 0;
            if (yy >= 0 && yy < h) {                       // rows outside the frame are zeros (zero border)
                const uint8_t *rowp = src + (size_t)yy * wb;
                if (full || in_frame) q[j] = load_stream(reinterpret_cast<const uint4 *>(rowp + x));
                if (hvalid) hb[j] = *reinterpret_cast<const uint32_t *>(rowp + hxb);
            }
        }
    };
    auto compute_chunk = [&](int ch, const uint4 (&q)[C], const uint32_t (&hb)[C]) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = ch * C + j;
            // bytes x-4..x-1 (left lane's q.w) and x+16..x+19 (right lane's q.x)
            uint32_t ql = __builtin_amdgcn_update_dpp(0u, q[j].w, 0x138, 0xf, 0xf, false);
            uint32_t qr = __builtin_amdgcn_update_dpp(0u, q[j].x, 0x130, 0xf, 0xf, false);
            if (lane == 0) ql = hb[j];
            if (lane == 63) qr = hb[j];
            // p[-3..18] as floats; pair layout G[m] = (p[m-3], p[m+5]), m = 0..13
            const uint32_t wv[6] = {ql, q[j].x, q[j].y, q[j].z, q[j].w, qr};   // wv[0] bytes = p[-4..-1]
            f32x2 G[14];
#pragma unroll
            for (int m = 0; m < 14; ++m) {
                const int b0 = m - 3 + 4, b1 = m + 5 + 4;                      // byte index into wv
                G[m] = f32x2{opaque((float)gh_byte(wv[b0 >> 2], b0 & 3)), opaque((float)gh_byte(wv[b1 >> 2], b1 & 3))};
            }
            uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f32x2 L = G[k], M = G[k + 3], Rr = G[k + 6];               // taps at xb-3, xb, xb+3
                const f32x2 h0 = __builtin_elementwise_fma(Rr, kk[2], __builtin_elementwise_fma(M, kk[1], L * kk[0]));
                const f32x2 h1 = __builtin_elementwise_fma(Rr, kk[5], __builtin_elementwise_fma(M, kk[4], L * kk[3]));
                const f32x2 h2 = __builtin_elementwise_fma(Rr, kk[8], __builtin_elementwise_fma(M, kk[7], L * kk[6]));
                // this input row r is: the BOTTOM row (kernel row 2) of output r-1, the MIDDLE row of output r, the TOP row of r+1
                const f32x2 acc = a1[k] + h2;                                  // output row r-1 complete
                a1[k] = a0[k] + h1;
                a0[k] = h0;
                const f32x2 qf = __builtin_elementwise_fma(acc, inv2, hinv2);  // (acc + 0.5) / norm
                r[k >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(qf.x, k & 3, r[k >> 2]);
                r[2 + (k >> 2)] = __builtin_amdgcn_cvt_pk_u8_f32(qf.y, k & 3, r[2 + (k >> 2)]);
            }
            if (i >= 2) {
                const int y = y0 + i - 2;
                if (in_frame && y < h) store_strip(dst + (size_t)y * wb + xw, lane, make_uint4(r[0], r[1], r[2], r[3]));
            }
        }
    };
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
}

}  // namespace

extern "C" int gh_conv3x3_rgb8_dev(const uint8_t *src, uint8_t *dst, int width, int height, const int32_t kernel[9], int32_t norm,
                                   gh_stream_t stream)
{
    GH_REQUIRE(width >= 0 && height >= 0, "negative size");
    GH_REQUIRE(kernel, "null kernel");
    GH_REQUIRE(norm != 0, "norm must be non-zero");
    if (width == 0 || height == 0) return GH_OK;
    GH_REQUIRE(src && dst && src != dst, "null or aliased pointers");
    long long ksum = 0;
    for (int i = 0; i < 9; ++i) ksum += kernel[i] < 0 ? -(long long)kernel[i] : kernel[i];
    GH_REQUIRE(ksum * 255 < (1ll << 31), "kernel taps too large for 32-bit accumulation");
    if (int rc = gh_ensure_init()) return rc;
    hipStream_t s = (hipStream_t)stream;
    // trunc(acc / norm) == trunc(-acc / -norm): normalise to a positive divisor
    const int sgn = norm < 0 ? -1 : 1;
    const long long wb = (long long)width * 3;
    const bool fast = (wb % 16 == 0) && gh_aligned16(src) && gh_aligned16(dst) && (long long)sgn * norm <= 4096 && ksum * 255 < (1ll << 24) &&
                      !getenv("GH_CONV_IMPL");
    if (fast) {
        TapsF t;
        for (int i = 0; i < 9; ++i) t.k[i] = (float)(sgn * kernel[i]);
        const float n = (float)(sgn * norm);
        t.inv = 1.0f / n;
        t.half_inv = 0.5f / n;
        // band height: 64 rows whenever that still gives every CU two workgroups (the same rule as the Sobel kernels:
        // 64-row bands measured 5-10 % faster than 32 on MI355X), else 32, else 16
        const int span = stream_core::STRIP_PX * stream_core::WAVES_PER_BLOCK;
        const int groups_x = (int)((wb + span - 1) / span);
        int R = GH_TUNE_INT("GH_CONV_R", 0);
        if (R != 16 && R != 32 && R != 64) {
            const long long nb64 = (long long)groups_x * ((height + 63) / 64), nb32 = (long long)groups_x * ((height + 31) / 32);
            R = nb64 >= 2 * GH_NUM_CU ? 64 : (nb32 >= 2 * GH_NUM_CU ? 32 : 16);
        }
        stream_core::BandPlan plan{};
        plan.bands = (height + R - 1) / R;
        plan.order = stream_core::sanitize_order(GH_TUNE_INT("GH_CONV_ORDER", 6) % 1000, plan.bands, groups_x);   // band-per-XCD, as the fused kernel
        const long long nb = stream_core::plan_grid(plan, groups_x);
        GH_REQUIRE(nb < (1ll << 31), "frame too large for one launch");
        if (R == 64) k_conv3x3_stream<64, 2><<<(unsigned)nb, stream_core::BLOCK, 0, s>>>(src, dst, (int)wb, height, groups_x, plan.order, t);
        else if (R == 32) k_conv3x3_stream<32, 2><<<(unsigned)nb, stream_core::BLOCK, 0, s>>>(src, dst, (int)wb, height, groups_x, plan.order, t);
        else k_conv3x3_stream<16, 2><<<(unsigned)nb, stream_core::BLOCK, 0, s>>>(src, dst, (int)wb, height, groups_x, plan.order, t);
    } else {
        Taps t;
        for (int i = 0; i < 9; ++i) t.k[i] = kernel[i];
        t.norm = norm;
        size_t blocks = ((size_t)width * height * 3 + BLOCK - 1) / BLOCK;
        if (blocks > (1u << 20)) blocks = 1u << 20;
        k_conv3x3_px<<<(unsigned)blocks, BLOCK, 0, s>>>(src, dst, width, height, t);
    }
    GH_LAUNCH_CHECK();
    return GH_OK;
}
