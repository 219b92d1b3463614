// conv3x3.hip -- generic 3x3 integer convolution on interleaved RGB8 for gfx950 (MI355X).
//
// Replaces old/parallel_convolution.c:8-24 restated on u8 frames: per channel
//   out(r,c,p) = clamp( trunc( sum_{i,j} in(r+i-1, c+j-1, p) * k[i][j] / norm ), 0, 255 )
// with a zero border supplied by the kernel (the reference pre-pads its array) and C integer
// division (truncation toward zero).  The frame is treated as a byte plane of width 3*w: output
// byte xb depends on input bytes xb-3, xb, xb+3 of three rows.  Bound: HBM, 3 R + 3 W bytes per pixel.
//
//   k_conv3x3_dma     default: the strip walk below with the rows brought HBM -> LDS by LDS-DMA (the ring of k_sobel_dma), any
//                     row length >= 1024 bytes and any alignment; same arithmetic and exactness conditions as k_conv3x3_stream.
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

// ylo / yhi: first and last readable input row relative to the band's first row (-1 / h where a real halo row lies beyond the band's
// edge -- row sharding --, 0 / h - 1 where the band edge is the frame border and the row beyond it is the zero border).
__global__ __launch_bounds__(BLOCK) void k_conv3x3_px(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int h, int ylo, int yhi, Taps t)
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
            if (yy < ylo || yy > yhi) continue;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int xx = xb + 3 * (dj - 1);
                if (xx < 0 || xx >= (int)rowb) continue;
                acc += (int)src[(ptrdiff_t)yy * (ptrdiff_t)rowb + xx] * t.k[di * 3 + dj];
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
                                                                      int wb, int h, int ylo, int yhi, int groups_x, int order, TapsF t)
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
            if (yy >= ylo && yy <= yhi) {                  // rows outside the frame are zeros (zero border)
                const uint8_t *rowp = src + (ptrdiff_t)yy * wb;
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
            uint32_t ql = __builtin_amdgcn_update_dpp(0u, q[j].w, 0x138, 0xf, 0xf, true);
            uint32_t qr = __builtin_amdgcn_update_dpp(0u, q[j].x, 0x130, 0xf, 0xf, true);
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

// ---- LDS-DMA ring form (any row length >= 1024 bytes, any alignment) -----------------------------------------------------------
// The same row ring as k_sobel_dma (sobel.hip; waits counted by hand, checked on the generated code by tests/test_isa_vmcnt.py):
// a wave's 1 KiB of each row of the 3w-byte plane goes HBM -> LDS directly, D rows deep; rows of frames whose pitch or base is not
// a multiple of 16 start at the aligned address before the strip and land dword-shifted (UNAL, see k_sobel_dma).  The last strip of
// a row is shifted left to end at the row's last byte, so every strip is full.  Rows outside the frame and bytes outside a row are
// zeros (the reference pre-pads its array).
// REG = 1: regular bands (every band R + 2 input rows in whole chunks of 6: the branch-free form of k_sobel_dma2 / k_sobel_dma).
template <int D, int UNAL, int REG = 0>
__global__ __launch_bounds__(stream_core::BLOCK) void k_conv3x3_dma(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int wb, int h,
                                                                   int ylo, int yhi, int groups_x, stream_core::BandPlan plan, TapsF t)
{
    using namespace stream_core;
    constexpr int PRE = UNAL ? 16 : 0;
    constexpr int SLOT = UNAL ? 16 + STRIP_PX + 32 : STRIP_PX + 16;
    constexpr int CHK = (D % 2 == 0) ? D : 2 * D;
    static_assert(3 * (D - 1) <= 60, "vmcnt is 6 bits");
    __shared__ __attribute__((aligned(16))) uint8_t ring[WAVES_PER_BLOCK][D][SLOT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int gx_i, gy_i;
    band_of_block(plan, blockIdx.x, groups_x, gx_i, gy_i);
    int xw = (gx_i * WAVES_PER_BLOCK + wave) * STRIP_PX;        // first BYTE of this wave's strip
    if (xw >= wb) return;
    if (xw + STRIP_PX > wb) xw = wb - STRIP_PX;
    if (gy_i < 0) return;
    const int R = plan.R[0];
    int y0 = gy_i * R;
    if (REG && y0 + R > h) y0 = h - R;          // regular bands: the last band is shifted up to end at the frame edge (see k_sobel_dma)
    const int nrows = min(R, h - y0) + 2;
    const int nch = (nrows + CHK - 1) / CHK;
    __builtin_amdgcn_s_setreg(0x0801, 3);   // MODE.fp_round(f32) = toward zero (see the exactness note above)
    const bool has_right = xw + STRIP_PX < wb;
    // bytes of the row left / right of the strip that the taps can reach: 3 each, fewer where the row ends first (a shifted last
    // strip may start at byte 1 or 2, the strip before it may end 1 or 2 bytes before the row does); what lies beyond is zero
    const int nl = min(3, xw), nr = min(3, wb - xw - STRIP_PX);
    const uint32_t voff = (uint32_t)xw + (uint32_t)lane * 16u;
    // aligned frames: halo dwords = the 4 bytes before / after the strip (16-byte aligned rows: always inside the row when they exist)
    uint32_t hoff = lane == 0 ? (uint32_t)xw - 4u : (uint32_t)(xw + STRIP_PX);
    if ((lane == 0 && xw == 0) || (lane != 0 && !has_right)) hoff = 0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)&ring[wave][0][0];

    const f32x2 inv2 = {t.inv, t.inv}, hinv2 = {t.half_inv, t.half_inv};
    f32x2 kk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) kk[i] = f32x2{t.k[i], t.k[i]};
    f32x2 a1[8], a0[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a1[k] = f32x2{0.f, 0.f}; a0[k] = f32x2{0.f, 0.f}; }

    auto strip_row = [&](int i) {                                // first byte of the strip in band row i, clamped into the frame (wave-uniform)
        int yy = y0 - 1 + i;
        yy = yy < ylo ? ylo : (yy > yhi ? yhi : yy);
        return uniform_ptr(src + (ptrdiff_t)yy * wb + xw);
    };
    auto issue_row = [&](int i, int slot) {
        const uint32_t d = uniform_u32(lds0 + (uint32_t)slot * SLOT);
        if (UNAL) {
            const uint8_t *A = strip_row(i);
            const uint32_t a = (uint32_t)(uintptr_t)A & 15u;
            const uint32_t ap = a < (uint32_t)nl ? a + 16u : a;                    // the piece carries the nl bytes left of the strip
            // tail blocks: fetched only where they hold bytes of the strip or of its nr right neighbours -- every block read contains a byte of the row
            const uint32_t t0 = (ap == 0u && !has_right) ? (uint32_t)STRIP_PX - 16u : (uint32_t)STRIP_PX;
            const uint32_t t1 = (ap > 16u || (nr > 0 && ap + (uint32_t)nr > 16u)) ? (uint32_t)STRIP_PX + 16u : t0;
            const uint32_t dm = uniform_u32(d + PRE + (a & 3u) - ap);
            glds_row1k_tail(dm, dm + STRIP_PX, uniform_ptr(A - ap), (uint32_t)lane * 16u, lane == 0 ? t0 : t1, slot);
        } else {
            glds_row1k_halo<1>(d, d + STRIP_PX, strip_row(i) - xw, voff, hoff, slot);
        }
    };
#pragma unroll
    for (int i = 0; i < D - 1; ++i) issue_row(i, i);

    // row step in the three statically different forms of k_sobel_dma: 0 generic, 1 transition, 2 steady (exact count 3 (D-1))
    auto row_step = [&](auto kind_c, int ch, int j) {
        constexpr int KIND = decltype(kind_c)::value;
        const int i = ch * CHK + j;
        if (KIND == 0 && i >= nrows) return true;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool issue = KIND == 0 ? i + D - 1 < nrows : (KIND == 4 ? j + D - 1 < CHK : true);     // KIND 3 / 4: first / last chunk of a regular band
        if (issue) {
            issue_row(i + D - 1, (j + D - 1) % D);
            if (KIND == 2 || KIND == 4) wait_vm<3 * (D - 1)>(j % D); else wait_vm<2 * (D - 1)>(j % D);
        } else {
            wait_vm<0>(j % D);
        }
        const uint8_t *slotp = &ring[wave][j % D][0];
        uint4 q = *reinterpret_cast<const uint4 *>(slotp + PRE + 16 * lane);
        uint32_t ql = __builtin_amdgcn_update_dpp(0u, q.w, 0x138, 0xf, 0xf, true);   // wave_shr:1  lane l <- lane l-1
        uint32_t qr = __builtin_amdgcn_update_dpp(0u, q.x, 0x130, 0xf, 0xf, true);   // wave_shl:1  lane l <- lane l+1
        if (UNAL) {
            const uint32_t fl = *reinterpret_cast<const uint32_t *>(slotp + PRE - 4);
            const uint2 fr = *reinterpret_cast<const uint2 *>(slotp + PRE + STRIP_PX);
            if (lane == 0) ql = fl;
            if (lane == 63) qr = fr.x;
            if (UNAL == 2) {                                         // the strip starts at byte sh of the lane grid
                const uint32_t sh = (uint32_t)(uintptr_t)strip_row(i) & 3u;
                uint32_t qr2 = __builtin_amdgcn_update_dpp(0u, q.y, 0x130, 0xf, 0xf, true);
                if (lane == 63) qr2 = fr.y;
                ql = __builtin_amdgcn_alignbyte(q.x, ql, sh);
                q.x = __builtin_amdgcn_alignbyte(q.y, q.x, sh);
                q.y = __builtin_amdgcn_alignbyte(q.z, q.y, sh);
                q.z = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
                q.w = __builtin_amdgcn_alignbyte(qr, q.w, sh);
                qr = __builtin_amdgcn_alignbyte(qr2, qr, sh);
            }
        } else {
            const uint2 hh = *reinterpret_cast<const uint2 *>(slotp + STRIP_PX);
            if (lane == 0) ql = hh.x;
            if (lane == 63) qr = hh.y;
        }
        // zeros outside the row and outside the frame: ql = bytes x-4 .. x-1 of which the last nl are in the row, qr = bytes x+16 .. x+19
        // of which the first nr are
        if (lane == 0 && nl < 3) ql = nl == 0 ? 0u : ql & (~0u << (8 * (4 - nl)));
        if (lane == 63 && nr < 3) qr = nr <= 0 ? 0u : qr & (~0u >> (8 * (4 - nr)));
        const int yin = y0 - 1 + i;
        // rows beyond the frame (the zero border): only a band's first and last input row can be one -- in the regular form two statically
        // known steps -- and the (wave-uniform) test is a scalar branch, not six selects on every row
        if (!REG || (KIND == 3 && j == 0) || (KIND == 4 && j == CHK - 1)) {
            if (__builtin_expect(yin < ylo || yin > yhi, 0)) {
                asm volatile("" ::: "memory");
                q = make_uint4(0, 0, 0, 0); ql = 0; qr = 0;
            }
        }
        // p[-3..18] as floats; pair layout G[m] = (p[m-3], p[m+5]), m = 0..13
        const uint32_t wv[6] = {ql, q.x, q.y, q.z, q.w, qr};                   // wv[0] bytes = p[-4..-1]
        f32x2 G[14];
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            const int b0 = m - 3 + 4, b1 = m + 5 + 4;
            G[m] = f32x2{opaque((float)gh_byte(wv[b0 >> 2], b0 & 3)), opaque((float)gh_byte(wv[b1 >> 2], b1 & 3))};
        }
        uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 L = G[k], M = G[k + 3], Rr = G[k + 6];
            const f32x2 h0 = __builtin_elementwise_fma(Rr, kk[2], __builtin_elementwise_fma(M, kk[1], L * kk[0]));
            const f32x2 h1 = __builtin_elementwise_fma(Rr, kk[5], __builtin_elementwise_fma(M, kk[4], L * kk[3]));
            const f32x2 h2 = __builtin_elementwise_fma(Rr, kk[8], __builtin_elementwise_fma(M, kk[7], L * kk[6]));
            const f32x2 acc = a1[k] + h2;
            a1[k] = a0[k] + h1;
            a0[k] = h0;
            const f32x2 qf = __builtin_elementwise_fma(acc, inv2, hinv2);
            r[k >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(qf.x, k & 3, r[k >> 2]);
            r[2 + (k >> 2)] = __builtin_amdgcn_cvt_pk_u8_f32(qf.y, k & 3, r[2 + (k >> 2)]);
        }
        if (KIND == 3 ? j >= 2 : (KIND != 0 || i >= 2))              // y = y0 + i - 2 < h by construction of nrows
            store_strip_p<2>(dst + (size_t)(y0 + i - 2) * wb + xw, lane, make_uint4(r[0], r[1], r[2], r[3]));
        return false;
    };
    auto chunk = [&](auto kind_c, int ch) {
#pragma unroll
        for (int j = 0; j < CHK; ++j)
            if (row_step(kind_c, ch, j)) return true;
        return false;
    };
    if (REG) {
        chunk(std::integral_constant<int, 3>{}, 0);
        int chr = 1;
        if (CHK - 2 < D - 1) { chunk(std::integral_constant<int, 1>{}, 1); chr = 2; }
#pragma unroll 1
        for (; chr < nch - 1; ++chr) chunk(std::integral_constant<int, 2>{}, chr);
        chunk(std::integral_constant<int, 4>{}, nch - 1);
        return;
    }
    const int n_full = (nrows - D + 1) / CHK;
    if (chunk(std::integral_constant<int, 0>{}, 0)) return;
    int ch = 1;
    if (n_full >= 2) {
        chunk(std::integral_constant<int, 1>{}, 1);
#pragma unroll 1
        for (ch = 2; ch < n_full; ++ch) chunk(std::integral_constant<int, 2>{}, ch);
    }
#pragma unroll 1
    for (; ch < nch; ++ch)
        if (chunk(std::integral_constant<int, 0>{}, ch)) return;
}

}  // namespace

// Band form (row sharding): `height` output rows starting at src / dst; where an edge of the band is not the
// frame border the input row beyond it (src - pitch / src + height * pitch) is a real row of the frame -- a halo row received from
// the neighbouring rank -- instead of the zero border.  The whole-frame entry point is the band with both edges on the border.
extern "C" int gh_conv3x3_rgb8_rows_dev(const uint8_t *src, uint8_t *dst, int width, int height, const int32_t kernel[9], int32_t norm,
                                        int top_is_border, int bottom_is_border, gh_stream_t stream)
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
    bool exact = (long long)sgn * norm <= 4096 && ksum * 255 < (1ll << 24);     // conditions of the f32 forms
#ifdef GH_ENABLE_PROBES
    if (getenv("GH_CONV_IMPL")) exact = false;      // probe builds: GH_CONV_IMPL=px sends every frame to the per-pixel integer kernel
#endif
    const bool aligned = (wb % 16 == 0) && gh_aligned16(src) && gh_aligned16(dst);
    const int ylo = top_is_border ? 0 : -1, yhi = bottom_is_border ? height - 1 : height;   // readable input rows, relative to the band
    TapsF t;
    for (int i = 0; i < 9; ++i) t.k[i] = (float)(sgn * kernel[i]);
    {
        const float n = (float)(sgn * norm);
        t.inv = 1.0f / n;
        t.half_inv = 0.5f / n;
    }
    // LDS-DMA ring kernel: any row of 1024 bytes and more, any alignment (probe builds, GH_CONV_DMA=0: the register-staged strips, aligned frames only)
    if (exact && wb >= stream_core::STRIP_PX && wb < (1ll << 31) && GH_LAB_INT("GH_CONV_DMA", 1)) {
        const int span = stream_core::STRIP_PX * stream_core::WAVES_PER_BLOCK;
        const int groups_x = (int)((wb + span - 1) / span);
        // Band height: 16 rows.  (This kernel had inherited the Sobel kernels' 64 / 32 / 16 rule untested.  Measured, tools/sweep.py r2_bb, bc:
        // 16384^2 0.2954 ms with 16-row bands against 0.3137 with 64, 16370 x 16384 0.3332-0.3391 vs 0.3559, 12288 x 4096 0.0606 vs 0.0796,
        // 7680 x 4352 0.0426 vs 0.0490; only 16384 x 4096 (0.0770 vs 0.0746) and 32768 x 8192 (32-row bands 0.2894, 16: 0.2974, 64: 0.3119)
        // have a slightly better height.  At 97 registers five workgroups fit a CU, so short bands do not starve it of waves.)
        // Round 3 (tools/sweep.py conv_bands): 22-row bands -- 24 input rows = four whole chunks of the regular form, and band starts 22 rows
        // apart fall on many channel phases where 16 x 48 KiB fall on four -- beat 16 on every shape tried: 16384^2 0.2851 vs 0.2956 ms
        // (70.6 % vs 68.1 %), 16370 x 16384 0.3287 vs 0.3379, 16384 x 4096 0.0759 vs 0.0779, 7680 x 4320 0.0440 vs 0.0455, 12288^2 equal; 28 /
        // 34 / 40 / 64 rows lose 3-7 %.  Where 22-row starts would alias and 16-row ones do not (stream_core::prefer_short_bands) 16 stays.
        int R = GH_LAB_INT("GH_CONV_R", 0);
        if (R <= 0) R = (height >= 22 && !(GH_LAB_INT("GH_BAND_ALIAS_RULE", 1) && stream_core::prefer_short_bands(wb, 22, 16))) ? 22 : 16;
        stream_core::BandPlan plan = stream_core::make_band_plan(height, R, -1, 0, -1, 0);
        // pairs-per-XCD dispatch order (8): 0.290 ms against 0.293 (band-per-XCD) / 0.294 (plain) on 16384^2, ahead on 16368 / 16370 too
        plan.order = stream_core::sanitize_order(GH_LAB_INT("GH_CONV_ORDER", 8) % 1000, plan.bands, groups_x, gh_num_xcd());
        const long long nb = stream_core::plan_grid(plan, groups_x);
        GH_REQUIRE(nb < (1ll << 31), "frame too large for one launch");
        const int unal = aligned ? 0 : ((wb % 4 == 0 && ((uintptr_t)src & 3) == 0) ? 1 : 2);
        const unsigned g = (unsigned)nb, b = stream_core::BLOCK;
        // regular-band form: bands whose R + 2 input rows are whole chunks of 6 and into which the rows split (16- and 64-row bands)
        const bool reg = GH_LAB_INT("GH_CONV_REG", 1) && (R + 2) % 6 == 0 && R >= 16 && height >= R && plan.order < 1000 &&
                         !stream_core::shifted_band_overflows(groups_x, height, R, 5, gh_num_cu());
        gh_note_kernel(reg ? (unal ? "k_conv3x3_dma regular unaligned" : "k_conv3x3_dma regular") : (unal ? "k_conv3x3_dma unaligned" : "k_conv3x3_dma"));
        if (reg && unal == 0) k_conv3x3_dma<3, 0, 1><<<g, b, 0, s>>>(src, dst, (int)wb, height, ylo, yhi, groups_x, plan, t);
        else if (reg && unal == 1) k_conv3x3_dma<3, 1, 1><<<g, b, 0, s>>>(src, dst, (int)wb, height, ylo, yhi, groups_x, plan, t);
        else if (reg) k_conv3x3_dma<3, 2, 1><<<g, b, 0, s>>>(src, dst, (int)wb, height, ylo, yhi, groups_x, plan, t);
        else if (unal == 0) k_conv3x3_dma<3, 0><<<g, b, 0, s>>>(src, dst, (int)wb, height, ylo, yhi, groups_x, plan, t);
        else if (unal == 1) k_conv3x3_dma<3, 1><<<g, b, 0, s>>>(src, dst, (int)wb, height, ylo, yhi, groups_x, plan, t);
        else k_conv3x3_dma<3, 2><<<g, b, 0, s>>>(src, dst, (int)wb, height, ylo, yhi, groups_x, plan, t);
        GH_LAUNCH_CHECK();
        return GH_OK;
    }
    const bool fast = exact && aligned;
    gh_note_kernel(fast ? "k_conv3x3_stream" : "k_conv3x3_px");
    if (fast) {
        // band height: 64 rows whenever that still gives every CU two workgroups (the same rule as the Sobel kernels:
        // 64-row bands measured 5-10 % faster than 32 on MI355X), else 32, else 16
        const int span = stream_core::STRIP_PX * stream_core::WAVES_PER_BLOCK;
        const int groups_x = (int)((wb + span - 1) / span);
        int R = GH_LAB_INT("GH_CONV_R", 0);
        if (R != 16 && R != 32 && R != 64) {
            const long long nb64 = (long long)groups_x * ((height + 63) / 64), nb32 = (long long)groups_x * ((height + 31) / 32);
            R = nb64 >= 2 * gh_num_cu() ? 64 : (nb32 >= 2 * gh_num_cu() ? 32 : 16);
        }
        stream_core::BandPlan plan{};
        plan.bands = (height + R - 1) / R;
        plan.order = stream_core::sanitize_order(GH_LAB_INT("GH_CONV_ORDER", 6) % 1000, plan.bands, groups_x, gh_num_xcd());   // band-per-XCD, as the fused kernel
        const long long nb = stream_core::plan_grid(plan, groups_x);
        GH_REQUIRE(nb < (1ll << 31), "frame too large for one launch");
        if (R == 64) k_conv3x3_stream<64, 2><<<(unsigned)nb, stream_core::BLOCK, 0, s>>>(src, dst, (int)wb, height, ylo, yhi, groups_x, plan.order, t);
        else if (R == 32) k_conv3x3_stream<32, 2><<<(unsigned)nb, stream_core::BLOCK, 0, s>>>(src, dst, (int)wb, height, ylo, yhi, groups_x, plan.order, t);
        else k_conv3x3_stream<16, 2><<<(unsigned)nb, stream_core::BLOCK, 0, s>>>(src, dst, (int)wb, height, ylo, yhi, groups_x, plan.order, t);
    } else {
        Taps t;
        for (int i = 0; i < 9; ++i) t.k[i] = kernel[i];
        t.norm = norm;
        size_t blocks = ((size_t)width * height * 3 + BLOCK - 1) / BLOCK;
        if (blocks > (1u << 20)) blocks = 1u << 20;
        k_conv3x3_px<<<(unsigned)blocks, BLOCK, 0, s>>>(src, dst, width, height, ylo, yhi, t);
    }
    GH_LAUNCH_CHECK();
    return GH_OK;
}

extern "C" int gh_conv3x3_rgb8_dev(const uint8_t *src, uint8_t *dst, int width, int height, const int32_t kernel[9], int32_t norm,
                                   gh_stream_t stream)
{
    return gh_conv3x3_rgb8_rows_dev(src, dst, width, height, kernel, norm, 1, 1, stream);
}
