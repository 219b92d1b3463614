// fused.hip -- RGB8/RGBA8 -> luma -> 3x3 Sobel in one pass for gfx950 (MI355X).
//
// Replaces one pass of the reference driver's pipeline, monolithic/src/main_with_sobel.c:51-62
// (convert_to_grayscale, plane extract, sobel_edge), without ever writing the luma plane to HBM.
// HBM traffic: 3 R + 1 W bytes per pixel instead of 6 + 4 + 2 for the reference's three loops.  Halo
// luma (rows above/below a band, pixels left/right of a strip) is recomputed from RGB, which is also the
// zero-communication alternative to exchanging luma halo rows between shards.
//
//   k_fused_dma         default for RGB8 / RGBA8, width >= 1024: strip streaming (stream_core.h) with an LDS row ring filled by
//                       LDS-DMA (global_load_lds_dwordx4, three rows deep): a wave's 3 KiB of a row lands lane-linear
//                       in LDS, every lane reads back its 16 consecutive pixels (48 B), converts to luma in f32 and
//                       feeds the f32 Sobel row step directly (no byte round trip).
//   k_fused_stream      frames narrower than 1024 columns: the same strip streaming, rows staged through registers
//                       (coalesced dwordx4 loads -> ds_write -> ds_read de-interleave).
//   k_gray_sobel_fused  the LDS-tile form (sobel_tile.h): probe builds only (`make probes`), like every GH_FUSED_* tuning knob
#include "stream_core.h"
#ifdef GH_ENABLE_PROBES
#include "sobel_tile.h"
#endif

#include <type_traits>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

#ifdef GH_ENABLE_PROBES
using namespace sobel_tile;
#endif
using stream_core::row_readable;
using stream_core::BandPlan;
using stream_core::make_band_plan;

using stream_core::luma_f2;      // (stream_core.h: shared with the plane Sobel's luma-on-load form)

template <int MODE>
__device__ __forceinline__ float luma_f(uint32_t r, uint32_t g, uint32_t b)
{
    const stream_core::f32x2 y = luma_f2<MODE>(stream_core::f32x2{(float)r, 0.f}, stream_core::f32x2{(float)g, 0.f}, stream_core::f32x2{(float)b, 0.f});
    return y.x;
}

// One strip row of the fused kernels: this lane's 16 pixels (wv: 48 or 64 interleaved bytes) + the strip-halo pixel
// of lanes 0 / 63 (hw: R,G,B in bytes 0,1,2) -> luma in f32 -> one step of the rolling Sobel window (rowf2).
template <int MODE, int CH, int PAR>
__device__ __forceinline__ uint4 fused_row(stream_core::StateF2 &st, const uint32_t (&wv)[CH * 4], uint32_t hw, int lane, stream_core::f32x2 half2)
{
    using namespace stream_core;
    // pair layout: G[m] = (luma[m-1], luma[m+7]), m = 1..8 from this lane's pixels (m-1, m+7)
    f32x2 G[10];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x2 r2, g2, b2;
        if (CH == 3) {
            const int o0 = 3 * m, o1 = 3 * (m + 8);
            r2 = f32x2{(float)gh_byte(wv[o0 >> 2], o0 & 3), (float)gh_byte(wv[o1 >> 2], o1 & 3)};
            g2 = f32x2{(float)gh_byte(wv[(o0 + 1) >> 2], (o0 + 1) & 3), (float)gh_byte(wv[(o1 + 1) >> 2], (o1 + 1) & 3)};
            b2 = f32x2{(float)gh_byte(wv[(o0 + 2) >> 2], (o0 + 2) & 3), (float)gh_byte(wv[(o1 + 2) >> 2], (o1 + 2) & 3)};
        } else {
            r2 = f32x2{(float)gh_byte(wv[m], 0), (float)gh_byte(wv[m + 8], 0)};
            g2 = f32x2{(float)gh_byte(wv[m], 1), (float)gh_byte(wv[m + 8], 1)};
            b2 = f32x2{(float)gh_byte(wv[m], 2), (float)gh_byte(wv[m + 8], 2)};
        }
        G[m + 1] = luma_f2<MODE>(r2, g2, b2);
    }
    const float hl = luma_f<MODE>(gh_byte(hw, 0), gh_byte(hw, 1), gh_byte(hw, 2));
    // strip halo: pixel x-1 is the left lane's pixel 15 (G[8].y), pixel x+16 the right lane's pixel 0 (G[1].x)
    // (copy the vector elements to scalars first: __builtin_bit_cast on an ext-vector element expression reads
    //  element 0 of the vector with this hipcc, which silently turned G[8].y into G[8].x)
    const float own15 = G[8].y, own0 = G[1].x;
    float pl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(uint32_t, own15), 0x138, 0xf, 0xf, true));
    float pr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(uint32_t, own0), 0x130, 0xf, 0xf, true));
    if (lane == 0) pl = hl;
    if (lane == 63) pr = hl;
    G[0] = f32x2{pl, G[8].x};        // (p[-1], p[7])
    G[9] = f32x2{G[1].y, pr};        // (p[8],  p[16])
    return rowf2<PAR>(st, G, half2);
}

// R output rows per strip, C rows per load chunk
template <int MODE, int CH, int R, int C, int ROLL = 0>
__global__ __launch_bounds__(stream_core::BLOCK) void k_fused_stream(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ edge, int w,
                                                                    int rows, int groups_x, int zero_top, int zero_bot, int ws)
{
    // w: frame width (row pitch = w * CH bytes; rows and buffers need not be 16-byte aligned: the vector accesses are then
    // unaligned, which gfx950 global memory instructions handle); ws <= w, a multiple of 16: the columns covered by
    // strips -- columns [ws, w) of a frame whose width is not a multiple of 16 are left to k_fused_px_cols.
    using namespace stream_core;
    static_assert(C % 2 == 0, "chunk must be even: the h2 ring slot is the row parity");
    constexpr int SLAB = STRIP_PX * CH;                          // bytes of one strip row
    constexpr int NV = SLAB / 1024;                              // dwordx4 loads per lane per row (3 or 4)
    __shared__ __attribute__((aligned(16))) uint8_t slab[WAVES_PER_BLOCK][SLAB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx_i = (int)(blockIdx.x % (unsigned)groups_x), gy_i = (int)(blockIdx.x / (unsigned)groups_x);
    int xw = (gx_i * WAVES_PER_BLOCK + wave) * STRIP_PX;
    if (xw >= ws) return;
    if (xw + STRIP_PX > ws && ws >= STRIP_PX && ws < w) xw = ws - STRIP_PX;   // unaligned frames: last strip shifted left, all strips full
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    const f32x2 half2 = half2_reg();
    const bool in_frame = x < ws;
    const bool full = xw + STRIP_PX <= ws;
    const int nvec = full ? NV * 64 : (ws - xw) * CH / 16;                    // 16-byte vectors of this strip row inside the frame
    // strip halo: lane 0 fetches the 16 bytes that END at the strip's first byte (pixel xw-1 is in there), lane 63 the 16
    // bytes that START after its last byte (pixel xw+1024).  One aligned dwordx4 per row for two lanes; nothing is
    // done with it until it is used, so it does not force an early s_waitcnt in the middle of the row loads.
    const int hx = lane == 0 ? xw - 1 : xw + STRIP_PX;
    const bool hvalid = (lane == 0 || lane == 63) && hx >= 0 && hx < w;
    const ptrdiff_t hoff = lane == 0 ? (ptrdiff_t)xw * CH - 16 : (ptrdiff_t)(xw + STRIP_PX) * CH;   // byte offset in the row

    StateF2 st;
    st.clear();
    uint4 *sw = reinterpret_cast<uint4 *>(slab[wave]);
    const uint4 *sr = reinterpret_cast<const uint4 *>(slab[wave] + 16 * CH * lane);

    constexpr int NROWS = R + 2;
    static_assert(NROWS % C == 0, "R + 2 must be a multiple of the chunk");
    constexpr int NCH = NROWS / C;

    // ROLL = 1: rolling prefetch, C rows deep, at no register cost: the staging registers of strip row i are dead as
    // soon as they have been written to the LDS slab, so the loads of row i + C are issued into the same registers
    // right there, before row i is computed.  ROLL = 0: the C rows of a chunk are loaded together, then computed.
    // Measured on MI355X (16384^2): ROLL = 0 is 7 % faster -- more rows in flight per wave make the row-strided
    // HBM pattern worse, exactly as in the no-compute strip-copy probe (tools/ubench, "pattern").
    uint4 v[C][NV];
    uint4 hp[C];
    auto issue_row = [&](int i, uint4 (&vr)[NV], uint4 &hr) {
        const int yy = clamp_row(y0 - 1 + i, rows, zero_top, zero_bot);   // always a readable row
        const uint8_t *rowb = rgb + ((ptrdiff_t)yy * w) * CH;
        const uint4 *rowp = reinterpret_cast<const uint4 *>(rowb + (size_t)xw * CH);
        if (full) {                                   // wave-uniform: the whole strip row is inside the frame
#pragma unroll
            for (int k = 0; k < NV; ++k) vr[k] = load_stream(rowp + 64 * k + lane);
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                vr[k] = make_uint4(0, 0, 0, 0);
                if (64 * k + lane < nvec) vr[k] = load_stream(rowp + 64 * k + lane);
            }
        }
        hr = make_uint4(0, 0, 0, 0);
        if (hvalid) {
            if (lane == 0 || (hx + 6) <= w) hr = *reinterpret_cast<const uint4 *>(rowb + hoff);      // 16 bytes = 5.3 (4) pixels: inside the row
            else {                           // right halo pixel within the last 5 columns of an unaligned frame: no 16-byte read past the row
                const uint8_t *q = rowb + hoff;
                hr.x = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
            }
        }
    };
    if (ROLL) {
#pragma unroll
        for (int j = 0; j < C; ++j) issue_row(j, v[j], hp[j]);
    }

#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        if (!ROLL) {
#pragma unroll
            for (int j = 0; j < C; ++j) issue_row(ch * C + j, v[j], hp[j]);
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = ch * C + j;
            // de-interleave: the wave's row (lane-linear vectors) -> LDS -> 16 consecutive pixels per lane
#pragma unroll
            for (int k = 0; k < NV; ++k) sw[64 * k + lane] = v[j][k];
            const uint4 hq = hp[j];
            if (ROLL && ch + 1 < NCH) issue_row(i + C, v[j], hp[j]);   // prefetch row i + C into the registers just freed
            __builtin_amdgcn_wave_barrier();     // same wave, DS ops execute in order: only a scheduling fence
            uint32_t wv[4 * NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) { const uint4 u = sr[k]; wv[4 * k] = u.x; wv[4 * k + 1] = u.y; wv[4 * k + 2] = u.z; wv[4 * k + 3] = u.w; }
            __builtin_amdgcn_wave_barrier();
            // halo pixel bytes: lane 0 -> last pixel of the left chunk, lane 63 -> first pixel of the right chunk
            const uint32_t hw = lane == 0 ? (CH == 3 ? (hq.w >> 8) : hq.w) : hq.x;      // R,G,B in bytes 0,1,2
            uint4 out = (j & 1) ? fused_row<MODE, CH, 1>(st, wv, hw, lane, half2) : fused_row<MODE, CH, 0>(st, wv, hw, lane, half2);
            if (i >= 2 && i < NROWS) {
                const int y = y0 + i - 2;
                if (x == 0) out.x &= 0xffffff00u;
                if (x + 15 == w - 1) out.w &= 0x00ffffffu;
                if ((y == 0 && zero_top) || (y == rows - 1 && zero_bot)) out = make_uint4(0, 0, 0, 0);
                if (in_frame && y < rows) store_strip(edge + (size_t)y * w + xw, lane, out);
            }
        }
    }
}

// Per-pixel form for what the strip kernels do not cover: the last w % 16 columns of a frame whose width is not a multiple
// of 16, and frames narrower than 16 pixels (x0 = 0).  Same arithmetic: luma of the 3x3 neighbourhood, integer Sobel.
template <int MODE>
__global__ __launch_bounds__(stream_core::BLOCK) void k_fused_px_cols(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ edge, int w, int rows,
                                                                     int ch, int x0, int zero_top, int zero_bot)
{
    const int nc = w - x0;
    const size_t n = (size_t)nc * rows, stride = (size_t)gridDim.x * stream_core::BLOCK;
    for (size_t i = (size_t)blockIdx.x * stream_core::BLOCK + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / nc), x = x0 + (int)(i - (size_t)y * nc);
        uint32_t out = 0;
        const bool border = x == 0 || x == w - 1 || (y == 0 && zero_top) || (y == rows - 1 && zero_bot);
        if (!border) {
            int l[3][3];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const uint8_t *p = rgb + ((ptrdiff_t)(y + dy) * w + (x + dx)) * ch;
                    l[dy + 1][dx + 1] = (int)gh_luma<MODE>(p[0], p[1], p[2]);
                }
            const int gx = -l[0][0] - 2 * l[1][0] - l[2][0] + l[0][2] + 2 * l[1][2] + l[2][2];
            const int gy = l[0][0] + 2 * l[0][1] + l[0][2] - l[2][0] - 2 * l[2][1] - l[2][2];
            out = gh_sobel_mag(gx * gx + gy * gy);
        }
        edge[(size_t)y * w + x] = (uint8_t)out;
    }
}

// ---- LDS-DMA form (RGB8 / RGBA8, width >= 1024): rows go HBM -> LDS directly (global_load_lds_dwordx4), D rows deep -----------
// The register-staged kernel above holds C rows of loads in VGPRs (24 + 8 registers at C = 2) and pays three ds_write per
// row to de-interleave them.  Here a wave owns a ring of D row slots in LDS; one row = three 1 KiB LDS-DMA pieces + one
// dword piece carrying the two strip-halo pixels (lanes 0 and 1 fetch the dword left / right of the strip).  The DMA
// writes lane-linear (piece base + lane * 16), which is exactly the slab image the de-interleaving ds_read_b128 x3 want.
// hipcc does not count asm memory operations, so the waits are ours: vmcnt is in-order on gfx9, and at the wait for row i
// the only loads younger than row i's are those of rows i+1 .. i+D-1 (4 instructions each); output stores in between
// only make the count conservative.  The last strip of a row is shifted left to end at the frame edge (all strips are
// full; the overlap is recomputed and stored twice with identical bytes), so no lane of a DMA piece is ever masked.
// UNAL != 0 (frames whose rows or base are not 16-byte aligned): every row's pieces start at the 16-byte aligned address at or
// before the strip (far enough before it to carry the left halo pixel) and land dword-shifted in LDS; UNAL = 2 re-aligns the
// lane's bytes with v_alignbyte_b32 -- see k_sobel_dma in sobel.hip for the scheme and the counters behind it.
// REG = 1 ("regular bands", see k_sobel_dma2 in sobel.hip): every band has R + 2 input rows and that is a multiple of the unroll, so
// which steps issue a row and which store is known per unrolled step and a band runs without a data-dependent branch.
template <int MODE, int D, int SP, int CH = 3, int UNAL = 0, int REG = 0>
__global__ __launch_bounds__(stream_core::BLOCK)
void k_fused_dma(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ edge, int w, int rows, int groups_x, BandPlan plan, int zero_top,
                 int zero_bot, GhHaloWait hw)
{
    using namespace stream_core;
    constexpr int PRE = UNAL ? 16 : 0;                            // UNAL: room in front of the lane grid (pieces land up to 16 bytes early)
    constexpr int SLOT = UNAL ? 16 + STRIP_PX * CH + 32 : STRIP_PX * CH + 16;   // one strip row + the halo dword pair / pre-area + 32-byte tail
    constexpr int CHK = (D % 2 == 0) ? D : 2 * D;                // unroll: slot = j % D and h2 ring parity = j & 1 are static
    constexpr int NLD = CH + 1;                                   // DMA pieces per row: CH x 1 KiB + the halo dword
    static_assert((NLD + 1) * (D - 1) <= 60, "vmcnt is 6 bits");
    __shared__ __attribute__((aligned(16))) uint8_t ring[WAVES_PER_BLOCK][D][SLOT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int gx_i, gy_i;
    band_of_block(plan, blockIdx.x, groups_x, gx_i, gy_i);
    int xw = (gx_i * WAVES_PER_BLOCK + wave) * STRIP_PX;
    if (xw >= w) return;
    if (xw + STRIP_PX > w) xw = w - STRIP_PX;                    // last strip: shifted left, overlaps its neighbour
    const int x = xw + 16 * lane;
    // band plan: n[0] bands of R[0] rows, then n[1] of R[1], then bands of R[2] (shorter bands last: a short drain)
    if (gy_i < 0) return;
    int R = plan.R[0], y0 = gy_i * plan.R[0];
    if (plan.order / 1000 == 1) {            // alternating heights R-1, R+1: band starts alternate between even and odd rows
        y0 = (gy_i >> 1) * 2 * plan.R[0] + ((gy_i & 1) ? plan.R[0] - 1 : 0);
        R = (gy_i & 1) ? plan.R[0] + 1 : plan.R[0] - 1;
        if (gy_i == plan.bands - 1) R = rows - y0;        // the last band takes what is left
        if (y0 >= rows) return;
    } else if (gy_i >= plan.n[0]) {
        const int g1 = gy_i - plan.n[0];
        R = plan.R[1]; y0 = plan.y[1] + g1 * plan.R[1];
        if (g1 >= plan.n[1]) { R = plan.R[2]; y0 = plan.y[2] + (g1 - plan.n[1]) * plan.R[2]; }
    }
    if (REG && y0 + R > rows) y0 = rows - R;                     // regular bands: the last band is shifted up to end at the frame edge (see k_sobel_dma)
    const int nrows = min(R, rows - y0) + 2;                     // input rows of this band (clipped to the frame) incl. halo
    const int nch = (nrows + CHK - 1) / CHK;
    // Sharded step, halo-flag form (gh_common.h): a band that reads a halo row waits -- right before it requests that row -- until the
    // exchange beside this launch has landed.  The shard's FIRST band walks upward (the rolling window is symmetric under the reversal),
    // so that its halo row, like the last band's, is the last row it needs: by then the exchange has normally completed and the wait
    // costs nothing (waiting at the start of the band would delay the whole band, and with every workgroup resident at once a band
    // lasts as long as the launch).
    const bool up = band_walks_up(plan, gy_i) || (hw.flag && !zero_top && y0 == 0 && y0 + nrows - 2 < rows);
    if (hw.flag) gh_halo_started(hw);
    const f32x2 half2 = half2_reg();
    const uint32_t voff = (uint32_t)xw * CH + (uint32_t)lane * 16u;            // byte offset of this lane's piece in a row
    // halo dwords: lane 0 -> the 4 bytes ending at the strip's first byte (pixel xw-1 = its bytes 1..3),
    //              lane 1 -> the 4 bytes after the strip's last byte (pixel xw+1024 = its bytes 0..2); clamped when outside
    uint32_t hoff = lane == 0 ? (uint32_t)xw * CH - 4u : (uint32_t)(xw + STRIP_PX) * CH;   // RGBA: the left dword IS pixel xw-1
    if ((lane == 0 && xw == 0) || (lane != 0 && xw + STRIP_PX >= w)) hoff = 0;
    // widths that are not multiples of 16 (RGB8): when the right halo pixel is the LAST pixel of the row, the dword behind it
    // would reach one byte past the row; it is fetched one byte earlier instead and shifted back after the read
    const int over = (xw + STRIP_PX < w) ? max(0, (xw + STRIP_PX) * CH + 4 - w * CH) : 0;   // wave-uniform: 0 or 1 (RGB8), 0 (RGBA8)
    if (lane != 0) hoff -= (uint32_t)over;
    const uint32_t rsh = 8u * (uint32_t)over;
    // and on the left: a shifted last strip of an RGB8 frame may start at column 1 (w = 1025): "xw * 3 - 4" would be -1 (offsets
    // are unsigned); the dword then starts at the row's first byte and is shifted up after the read
    const int under = xw > 0 ? max(0, 4 - xw * CH) : 0;                                  // wave-uniform: 0 or 1
    if (lane == 0) hoff += (uint32_t)under;
    const uint32_t lsh = 8u * (uint32_t)under;
    const uint32_t lds0 = (uint32_t)(uintptr_t)&ring[wave][0][0];

    StateF2 st;
    st.clear();
    const uint32_t first_col_mask = x == 0 ? 0xffffff00u : 0xffffffffu;
    const uint32_t last_col_mask = x + 15 == w - 1 ? 0x00ffffffu : 0xffffffffu;

    const bool has_right = xw + STRIP_PX < w;                    // (wave-uniform) a pixel right of the strip exists
    auto strip_row = [&](int i) {                                // first byte of the strip in band row i (wave-uniform)
        const int yy = clamp_row(up ? y0 + nrows - 2 - i : y0 - 1 + i, rows, zero_top, zero_bot);
        return uniform_ptr(rgb + ((ptrdiff_t)yy * w + xw) * CH);
    };
    auto issue_row = [&](int i, int slot) {
        if (hw.flag) {                                           // (wave-uniform) halo row of a shard: has the exchange landed?
            const int raw = up ? y0 + nrows - 2 - i : y0 - 1 + i;
            if ((raw < 0 && !zero_top) || (raw >= rows && !zero_bot)) gh_halo_wait(hw);
        }
        const uint32_t dst = uniform_u32(lds0 + (uint32_t)slot * SLOT);
        // always exactly NLD vector-memory instructions, one statement
        if (UNAL) {
            constexpr uint32_t ROWB = STRIP_PX * CH;
            const uint8_t *A = strip_row(i);
            const uint32_t a = (uint32_t)(uintptr_t)A & 15u;
            const uint32_t ap = (a < (uint32_t)CH && xw > 0) ? a + 16u : a;        // the pieces start a' bytes before the strip: the left pixel is in them
            // tail blocks (relative to the first piece): the block behind the pieces holds the strip's last a' bytes (a' >= 1) or the
            // right halo pixel; the second one is fetched only where it holds bytes of the strip (a' > 16) or of that pixel --
            // every block fetched contains a byte of the frame, so no access leaves the frame's pages
            const uint32_t t0 = (ap == 0u && !has_right) ? ROWB - 16u : ROWB;
            const uint32_t t1 = (ap > 16u || (has_right && ap + (uint32_t)CH > 16u)) ? ROWB + 16u : t0;
            const uint32_t dm = uniform_u32(dst + PRE + (a & 3u) - ap);
            if (CH == 3) glds_row3k_tail(dm, dm + ROWB, uniform_ptr(A - ap), (uint32_t)lane * 16u, lane == 0 ? t0 : t1, slot);
            else glds_row4k_tail(dm, dm + ROWB, uniform_ptr(A - ap), (uint32_t)lane * 16u, lane == 0 ? t0 : t1, slot);
        } else {
            const uint8_t *rowb = strip_row(i) - (ptrdiff_t)xw * CH;
            if (CH == 3) glds_row3k_halo<(SP >= 100 ? SP / 100 - 1 : 1)>(dst, dst + STRIP_PX * CH, rowb, voff, hoff, slot);
            else glds_row4k_halo(dst, dst + STRIP_PX * CH, rowb, voff, hoff, slot);
        }
    };
    // nrows >= 3 > D - 1 for D <= 3; for deeper rings the prologue may run past a very short band: clamp_row keeps it readable
#pragma unroll
    for (int i = 0; i < D - 1; ++i) issue_row(i, i);

    // One row step in three statically different forms (see k_sobel_dma in sobel.hip and tests/test_isa_vmcnt.py):
    //   KIND 0 generic (rows may be missing, the first two rows produce no output; only the NLD DMA instructions of each of the
    //   D-1 younger rows are counted), 1 transition (every row exists, issues and stores; same conservative count), 2 steady
    //   (as 1, and the D-1 previous steps are statically known to have stored: exact count (NLD + 1) (D-1)).
    auto row_step = [&](auto kind_c, int ch, int j) {
        constexpr int KIND = decltype(kind_c)::value;
        const int i = ch * CHK + j;
        if (KIND == 0 && i >= nrows) return true;                    // past the band: so is every later step -- the wave is done
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the slot about to be refilled has been read
        // KIND 3 / 4: first / last chunk of a regular band (j is a constant in the unrolled chunk)
        const bool issue = KIND == 0 ? i + D - 1 < nrows : (KIND == 4 ? j + D - 1 < CHK : true);
        if (issue) {
            issue_row(i + D - 1, (j + D - 1) % D);
            if (KIND == 2 || KIND == 4) wait_vm<(NLD + 1) * (D - 1)>(j % D); else wait_vm<NLD * (D - 1)>(j % D);
        } else {
            wait_vm<0>(j % D);
        }
        const uint8_t *slotp = &ring[wave][j % D][0];
        const uint4 *sr = reinterpret_cast<const uint4 *>(slotp + PRE + 16 * CH * lane);
        uint32_t wv[4 * CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) { const uint4 u = sr[k]; wv[4 * k] = u.x; wv[4 * k + 1] = u.y; wv[4 * k + 2] = u.z; wv[4 * k + 3] = u.w; }
        uint32_t hw;
        if (UNAL) {
            // the dword in front of the lane grid and the two behind it (wave-uniform addresses)
            const uint32_t fl = *reinterpret_cast<const uint32_t *>(slotp + PRE - 4);
            const uint2 fr = *reinterpret_cast<const uint2 *>(slotp + PRE + STRIP_PX * CH);
            uint32_t hl = fl, hr = fr.x;
            if (UNAL == 2) {                                         // the strip starts at byte sh of the lane grid
                const uint32_t sh = (uint32_t)(uintptr_t)strip_row(i) & 3u;
                uint32_t nx = __builtin_amdgcn_update_dpp(0u, wv[0], 0x130, 0xf, 0xf, true);   // wave_shl:1  lane l <- lane l+1
                if (lane == 63) nx = fr.x;
                hl = __builtin_amdgcn_alignbyte(wv[0], fl, sh);
                hr = __builtin_amdgcn_alignbyte(fr.y, fr.x, sh);
#pragma unroll
                for (int k = 0; k < 4 * CH - 1; ++k) wv[k] = __builtin_amdgcn_alignbyte(wv[k + 1], wv[k], sh);
                wv[4 * CH - 1] = __builtin_amdgcn_alignbyte(nx, wv[4 * CH - 1], sh);
            }
            hw = lane == 0 ? (CH == 3 ? hl >> 8 : hl) : hr;
        } else {
            const uint2 hh = *reinterpret_cast<const uint2 *>(slotp + STRIP_PX * CH);
            hw = lane == 0 ? (CH == 3 ? ((hh.x << lsh) >> 8) : hh.x) : (hh.y >> rsh);
        }
        uint4 out;
        if ((SP % 100) >= 10) out = make_uint4(wv[0] ^ wv[4] ^ wv[8] ^ hw, wv[1] ^ wv[5] ^ wv[9], wv[2] ^ wv[6] ^ wv[10], wv[3] ^ wv[7] ^ wv[4 * CH - 1]);   // probe: no compute
        else out = (j & 1) ? fused_row<MODE, CH, 1>(st, wv, hw, lane, half2) : fused_row<MODE, CH, 0>(st, wv, hw, lane, half2);
        if (KIND == 3 ? j >= 2 : (KIND != 0 || i >= 2)) {           // y = y0 + i - 2 < rows by construction of nrows
            const int y = up ? y0 + nrows - 1 - i : y0 + i - 2;
            out.x &= first_col_mask;                                 // the frame's first / last column is border: written as 0
            out.w &= last_col_mask;
            zero_if((y == 0 && zero_top) || (y == rows - 1 && zero_bot), out);
            if ((SP % 100) >= 20)      // access-pattern probe: same reads, but this workgroup's output rows written back to back
                store_strip_p<(SP % 100) % 10>(edge + ((size_t)blockIdx.x * R + (i - 2)) * (WAVES_PER_BLOCK * STRIP_PX) + wave * STRIP_PX, lane, out);
            else
                store_strip_p<(SP % 100) % 10>(edge + (size_t)y * w + xw, lane, out);
        }
        return false;
    };
    auto chunk = [&](auto kind_c, int ch) {          // true: the band is finished
#pragma unroll
        for (int j = 0; j < CHK; ++j)
            if (row_step(kind_c, ch, j)) return true;
        return false;
    };
    if (REG) {                                       // nch >= 3 chunks, all of them full (host: rows >= R, (R + 2) % CHK == 0, R + 2 >= 3 CHK)
        chunk(std::integral_constant<int, 3>{}, 0);
        int chr = 1;
        if (CHK - 2 < D - 1) { chunk(std::integral_constant<int, 1>{}, 1); chr = 2; }
#pragma unroll 1
        for (; chr < nch - 1; ++chr) chunk(std::integral_constant<int, 2>{}, chr);
        chunk(std::integral_constant<int, 4>{}, nch - 1);
        return;
    }
    // chunks 0 .. n_full-1 hold only rows that exist and still issue a row: (ch + 1) * CHK - 1 + D - 1 < nrows
    const int n_full = (nrows - D + 1) / CHK;
    if (chunk(std::integral_constant<int, 0>{}, 0)) return;
    int ch = 1;
    if (n_full >= 2) {
        chunk(std::integral_constant<int, 1>{}, 1);              // rows CHK .. 2 CHK - 1: all of them store (CHK >= 2)
#pragma unroll 1
        for (ch = 2; ch < n_full; ++ch) chunk(std::integral_constant<int, 2>{}, ch);
    }
#pragma unroll 1
    for (; ch < nch; ++ch)
        if (chunk(std::integral_constant<int, 0>{}, ch)) return;
}

#ifdef GH_ENABLE_PROBES
// ---- the LDS-tile form (sobel_tile.h): the first version measured (0.232 ms against 0.184); GH_FUSED_IMPL=tile in probe builds ----
template <int MODE, int CH>
__device__ __forceinline__ uint4 luma_chunk(const uint8_t *__restrict__ p)
{
    uint32_t lum[16];
    if (CH == 3) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
        const uint4 a = q[0], b = q[1], c = q[2];
        const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = 3 * i;
            lum[i] = gh_luma<MODE>(gh_byte(w[o >> 2], o & 3), gh_byte(w[(o + 1) >> 2], (o + 1) & 3),
                                   gh_byte(w[(o + 2) >> 2], (o + 2) & 3));
        }
    } else {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = q[j];
            const uint32_t px[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) lum[4 * j + k] = gh_luma<MODE>(gh_byte(px[k], 0), gh_byte(px[k], 1), gh_byte(px[k], 2));
        }
    }
    uint4 o;
    o.x = lum[0] | (lum[1] << 8) | (lum[2] << 16) | (lum[3] << 24);
    o.y = lum[4] | (lum[5] << 8) | (lum[6] << 16) | (lum[7] << 24);
    o.z = lum[8] | (lum[9] << 8) | (lum[10] << 16) | (lum[11] << 24);
    o.w = lum[12] | (lum[13] << 8) | (lum[14] << 16) | (lum[15] << 24);
    return o;
}


template <int MODE, int CH>
__global__ __launch_bounds__(BLOCK) void k_gray_sobel_fused(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ edge, int w,
                                                            int rows, int tiles_x, int zero_top, int zero_bot)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[LDS_BYTES];

    const unsigned t = blockIdx.x;
    const int x0 = (int)(t % (unsigned)tiles_x) * TW;
    const int y0 = (int)(t / (unsigned)tiles_x) * TH;

#pragma unroll 3
    for (int it = 0; it < STAGE_ITERS; ++it) {
        const int idx = threadIdx.x + it * BLOCK;
        const int r = idx / CHUNKS, c = idx - r * CHUNKS;
        const int yy = y0 - 1 + r;
        const int gx = x0 - 16 + 16 * c;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (idx < LDS_ROWS * CHUNKS && gx >= 0 && gx < w && row_readable(yy, rows, zero_top, zero_bot))
            v = luma_chunk<MODE, CH>(rgb + ((ptrdiff_t)yy * w + gx) * CH);
        if (idx < LDS_ROWS * CHUNKS) *reinterpret_cast<uint4 *>(tile + idx * 16) = v;
    }
    __syncthreads();

    compute_tile(tile, edge, w, rows, x0, y0, zero_top, zero_bot);
}

#endif   // GH_ENABLE_PROBES

// launch arguments of k_fused_dma, and the channel / mode dispatch
struct DmaLaunch {
    const uint8_t *rgb; uint8_t *edge; int width, rows, groups_x; BandPlan plan; int zt, zb, channels, mode; unsigned grid; size_t pad; hipStream_t s;
    GhHaloWait hw;
};
template <int D, int SP, int UNAL = 0, int REG = 0>
void launch_fused_dma(const DmaLaunch &L)
{
    const unsigned b = stream_core::BLOCK;
    if (L.channels == 4) {
        if (L.mode == GH_MODE_FMA) k_fused_dma<GH_MODE_FMA, D, SP % 10, 4, UNAL, REG><<<L.grid, b, L.pad, L.s>>>(L.rgb, L.edge, L.width, L.rows, L.groups_x, L.plan, L.zt, L.zb, L.hw);
        else k_fused_dma<GH_MODE_IEEE, D, SP % 10, 4, UNAL, REG><<<L.grid, b, L.pad, L.s>>>(L.rgb, L.edge, L.width, L.rows, L.groups_x, L.plan, L.zt, L.zb, L.hw);
    } else {
        if (L.mode == GH_MODE_FMA) k_fused_dma<GH_MODE_FMA, D, SP, 3, UNAL, REG><<<L.grid, b, L.pad, L.s>>>(L.rgb, L.edge, L.width, L.rows, L.groups_x, L.plan, L.zt, L.zb, L.hw);
        else k_fused_dma<GH_MODE_IEEE, D, SP, 3, UNAL, REG><<<L.grid, b, L.pad, L.s>>>(L.rgb, L.edge, L.width, L.rows, L.groups_x, L.plan, L.zt, L.zb, L.hw);
    }
}

}  // namespace

extern "C" int gh_gray_sobel_fused_rows_dev(const uint8_t *rgb, uint8_t *edge, int width, int rows, int channels, int mode,
                                            int top_is_border, int bottom_is_border, gh_stream_t stream)
{
    return gh_fused_rows_halo_wait(rgb, edge, width, rows, channels, mode, top_is_border, bottom_is_border, stream, nullptr);
}

// hw != nullptr: the halo-flag form of the sharded step (gh_common.h) -- only the LDS-DMA ring kernel carries the wait; a frame that
// would run another form is refused with GH_NO_FLAG_FORM before anything is launched.
// GH_LAB_INT knobs are compile-time constants in the release library (gh_common.h); what they select beyond their defaults -- other ring
// depths, store / load policies, the no-arithmetic probes, the LDS-tile form, tapered band plans -- exists in probe builds only.
int gh_fused_rows_halo_wait(const uint8_t *rgb, uint8_t *edge, int width, int rows, int channels, int mode, int top_is_border,
                            int bottom_is_border, gh_stream_t stream, const GhHaloWait *hw)
{
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    GH_REQUIRE(channels == 3 || channels == 4, "fused kernel takes RGB8 or RGBA8");
    GH_REQUIRE(mode == GH_MODE_IEEE || mode == GH_MODE_FMA, "bad mode");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(rgb && edge, "null pointer");
    if (int rc = gh_ensure_init()) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int zt = top_is_border ? 1 : 0, zb = bottom_is_border ? 1 : 0;
    // Fast paths want 16-pixel columns and 16-byte aligned buffers.  Anything else: the register-staged strip kernel on the
    // first ws = width & ~15 columns with unaligned vector accesses + the per-pixel kernel on the last width % 16 columns
    // (large frames), or the per-pixel kernel alone (small ones).
    const bool aligned = width % 16 == 0 && gh_aligned16(rgb) && gh_aligned16(edge);
    const int ws = width & ~15;
    auto launch_px_cols = [&](int x0) {
        size_t blocks = ((size_t)(width - x0) * rows + stream_core::BLOCK - 1) / stream_core::BLOCK;
        if (blocks > 8192) blocks = 8192;
        if (mode == GH_MODE_FMA) k_fused_px_cols<GH_MODE_FMA><<<(unsigned)blocks, stream_core::BLOCK, 0, s>>>(rgb, edge, width, rows, channels, x0, zt, zb);
        else k_fused_px_cols<GH_MODE_IEEE><<<(unsigned)blocks, stream_core::BLOCK, 0, s>>>(rgb, edge, width, rows, channels, x0, zt, zb);
    };
    // (unaligned strips need ws >= 1024: then every strip is full -- the last one is shifted left -- and the pixel right of
    //  column ws - 1 arrives through lane 63's halo load; a ragged last strip would need column ws from beyond its loads)
    // The LDS-DMA kernel takes any width >= 1024 and any base alignment (see sobel.hip: DMA pieces need no alignment on the
    // global side, every strip is full).
    const int dma_knob = GH_LAB_INT("GH_FUSED_DMA", -1);      // ring depth D (-1 = by grid size; probe builds: 0 = register-staged kernel, 2 / 3 / 4)
    const int dma_unal = GH_LAB_INT("GH_FUSED_DMA_UNALIGNED", 2);    // 2 aligned-down pieces (UNAL), 1 pieces from the strip's own address, 0 strips + tail
    const bool dma_ok = dma_knob != 0 && width >= stream_core::STRIP_PX && (aligned || dma_unal);
    bool use_tile = false;
#ifdef GH_ENABLE_PROBES
    static const bool use_tile_env = [] { const char *e = getenv("GH_FUSED_IMPL"); return e && strcmp(e, "tile") == 0; }();
    use_tile = use_tile_env && aligned;
#endif
    if (hw && (!dma_ok || use_tile)) return GH_NO_FLAG_FORM;
    if (hw && hw->pad) return GH_OK;             // probe: the LDS-DMA ring kernel will run
    if (!aligned && !dma_ok && (ws < stream_core::STRIP_PX || (long long)width * rows < GH_LAB_INT("GH_FUSED_UNALIGNED_MIN_PX", 1 << 20))) {
        gh_note_kernel("k_fused_px_cols");
        launch_px_cols(0);
        GH_LAUNCH_CHECK();
        return GH_OK;
    }
#ifdef GH_ENABLE_PROBES
    if (use_tile) {
        const int tiles_x = (width + TW - 1) / TW;
        const int tiles_y = (rows + TH - 1) / TH;
        const long long nt = (long long)tiles_x * tiles_y;
        GH_REQUIRE(nt < (1ll << 31), "frame too large for one launch");
        const unsigned grid = (unsigned)nt;
        gh_note_kernel("k_gray_sobel_fused");
        if (channels == 3) {
            if (mode == GH_MODE_FMA) k_gray_sobel_fused<GH_MODE_FMA, 3><<<grid, BLOCK, 0, s>>>(rgb, edge, width, rows, tiles_x, zt, zb);
            else k_gray_sobel_fused<GH_MODE_IEEE, 3><<<grid, BLOCK, 0, s>>>(rgb, edge, width, rows, tiles_x, zt, zb);
        } else {
            if (mode == GH_MODE_FMA) k_gray_sobel_fused<GH_MODE_FMA, 4><<<grid, BLOCK, 0, s>>>(rgb, edge, width, rows, tiles_x, zt, zb);
            else k_gray_sobel_fused<GH_MODE_IEEE, 4><<<grid, BLOCK, 0, s>>>(rgb, edge, width, rows, tiles_x, zt, zb);
        }
        GH_LAUNCH_CHECK();
        return GH_OK;
    }
#endif
    const int span = stream_core::STRIP_PX * stream_core::WAVES_PER_BLOCK;
    const int groups_x = ((dma_ok ? width : ws) + span - 1) / span;
    const unsigned b = stream_core::BLOCK;
    if (!dma_ok) {
        // register-staged strips: frames narrower than 1024 columns (rolling prefetch, two rows deep)
#define GH_FUSED_LAUNCH(R_, C_, RL_)                                                                                         \
        do {                                                                                                              \
            const long long nb = (long long)groups_x * ((rows + (R_) - 1) / (R_));                                         \
            GH_REQUIRE(nb < (1ll << 31), "frame too large for one launch");                                                \
            const unsigned g = (unsigned)nb;                                                                              \
            if (channels == 3) {                                                                                          \
                if (mode == GH_MODE_FMA) k_fused_stream<GH_MODE_FMA, 3, R_, C_, RL_><<<g, b, 0, s>>>(rgb, edge, width, rows, groups_x, zt, zb, ws);  \
                else k_fused_stream<GH_MODE_IEEE, 3, R_, C_, RL_><<<g, b, 0, s>>>(rgb, edge, width, rows, groups_x, zt, zb, ws);                     \
            } else {                                                                                                      \
                if (mode == GH_MODE_FMA) k_fused_stream<GH_MODE_FMA, 4, R_, C_, RL_><<<g, b, 0, s>>>(rgb, edge, width, rows, groups_x, zt, zb, ws);  \
                else k_fused_stream<GH_MODE_IEEE, 4, R_, C_, RL_><<<g, b, 0, s>>>(rgb, edge, width, rows, groups_x, zt, zb, ws);                     \
            }                                                                                                             \
        } while (0)
        gh_note_kernel(ws < width ? "k_fused_stream + k_fused_px_cols" : "k_fused_stream");
        int pick = GH_LAB_INT("GH_FUSED_VARIANT", 0);
        if (pick == 0) {
            const long long nb64 = (long long)groups_x * ((rows + 63) / 64), nb32 = (long long)groups_x * ((rows + 31) / 32);
            pick = nb64 >= 4 * gh_num_cu() ? 64 : (nb32 >= 4 * gh_num_cu() ? 32 : 16);     // >= 4 workgroups per CU
        }
        switch (pick) {
        case 64: GH_FUSED_LAUNCH(64, 2, 1); break;
        case 16: GH_FUSED_LAUNCH(16, 2, 1); break;
#ifdef GH_ENABLE_PROBES
        case 1: GH_FUSED_LAUNCH(32, 2, 0); break;      // chunked loads instead of the rolling prefetch
#endif
        default: GH_FUSED_LAUNCH(32, 2, 1); break;
        }
#undef GH_FUSED_LAUNCH
        GH_LAUNCH_CHECK();
        if (ws < width) {
            launch_px_cols(ws);
            GH_LAUNCH_CHECK();
        }
        return GH_OK;
    }

    // ---- LDS-DMA ring kernel.  Tuning found on MI355X (tools/ab_fused.py, 16384-wide frames):
    //  * band height 64 is the sweet spot whenever it still gives every CU a workgroup (69-73 % of 8 TB/s for 4096 ..
    //    16384-row frames); 32- and 128-row bands are 5-15 % slower at the same frame, 16-row bands pay 12.5 % halo rows;
    //  * band-per-XCD dispatch order (order 6: XCD x owns the bands k = x mod 8 with all their column groups) is 0.5-2.5 %
    //    faster than the plain band-major order on every frame shape and shard size tried (16384^2: 72.9 % vs 72.1 %);
    //  * two resident workgroups per CU beat three (fewer row streams in flight at once: 71.9 % vs 69.5 %), so a large
    //    launch uses the 2-deep ring plus an unused dynamic-LDS pad that caps residency at two.
    int dma = dma_knob;
    int rd = GH_LAB_INT("GH_FUSED_DMA_R", 0);
    if (rd <= 0)
        rd = stream_core::fused_band_rows(groups_x, rows, (long long)width * channels, gh_num_cu(), GH_LAB_INT("GH_BAND_ALIAS_RULE", 1) != 0,
                                          hw != nullptr && GH_LAB_INT("GH_SHARD_BAND_RULE", 1) != 0);
    const int sp = GH_LAB_INT("GH_FUSED_DMA_ST", 2);       // 2 = nontemporal stores; other values: probes
    const int R_ = rd;
    // (taper knobs, probe builds: GH_FUSED_DMA_N0 bands of R_, then GH_FUSED_DMA_N1 of GH_FUSED_DMA_R1, the rest GH_FUSED_DMA_R2)
    BandPlan plan = make_band_plan(rows, R_, GH_LAB_INT("GH_FUSED_DMA_N0", -1), GH_LAB_INT("GH_FUSED_DMA_R1", 0),
                                         GH_LAB_INT("GH_FUSED_DMA_N1", -1), GH_LAB_INT("GH_FUSED_DMA_R2", 0));
    plan.order = stream_core::sanitize_order(GH_LAB_INT("GH_FUSED_DMA_ORDER", 6), plan.bands, groups_x, gh_num_xcd());
    GH_REQUIRE(stream_core::plan_grid(plan, groups_x) < (1ll << 31), "frame too large for one launch");
    const unsigned g = (unsigned)stream_core::plan_grid(plan, groups_x);
    int pad_kb = GH_LAB_INT("GH_FUSED_DMA_PAD_KB", -1);    // unused dynamic LDS: caps the workgroups resident per CU
    if (dma < 0) {
        const bool big = g > 2u * gh_num_cu();
        dma = big ? 2 : 3;
        if (pad_kb < 0) pad_kb = big ? 30 : 0;               // 24.7 KiB ring + 30 KiB > 160 KiB / 3: two per CU
    }
    if (pad_kb < 0) pad_kb = 0;
    const size_t pad = (size_t)pad_kb * 1024;
    const DmaLaunch L{rgb, edge, width, rows, groups_x, plan, zt, zb, channels, mode, g, pad, s, hw ? *hw : GhHaloWait{}};
    // frames with unaligned rows: aligned-down pieces, landing dword-shifted (1) or byte-shifted + v_alignbyte (2); 0: aligned, or (probe
    // builds, GH_FUSED_DMA_UNALIGNED=1) pieces from the strip's own unaligned address
    const int unal = aligned || dma_unal < 2 || dma == 4 ? 0
                     : (((long long)width * channels % 4 == 0 && ((uintptr_t)rgb & 3) == 0) ? 1 : 2);
    // regular-band form (see k_sobel_dma): 16-row bands whose rows split evenly -- the mid-size and small frames, 0-5 % there
    // (3840 x 2160 0.0150 vs 0.0157 ms, 1920 x 1088 0.0136 vs 0.0143, 8192 x 2048 0.0176 vs 0.0183; tools/sweep.py r2_ba); aligned frames only
    const int reg_knob = GH_LAB_INT("GH_FUSED_REG", 1);     // (2, probe builds: also tall bands whose R + 2 rows are whole chunks of 6, NOTES.md R3.5)
    const bool reg = reg_knob && sp == 2 && dma == 3 && unal == 0 && (R_ == 16 || (reg_knob == 2 && (R_ + 2) % 6 == 0)) && rows >= R_ && plan.n[1] == 0 &&
                     plan.order < 1000 && !stream_core::shifted_band_overflows(groups_x, rows, R_, 3, gh_num_cu());
    gh_note_kernel(reg ? (R_ == 16 ? "k_fused_dma regular R=16" : "k_fused_dma regular") : (unal ? (dma == 2 ? "k_fused_dma D=2 unaligned" : "k_fused_dma D=3 unaligned") : (dma == 2 ? "k_fused_dma D=2" : (dma == 4 ? "k_fused_dma D=4" : "k_fused_dma D=3"))));
    if (reg) {
        launch_fused_dma<3, 2, 0, 1>(L);
    } else if (sp == 2 && unal) {
        if (dma == 2) { if (unal == 1) launch_fused_dma<2, 2, 1>(L); else launch_fused_dma<2, 2, 2>(L); }
        else { if (unal == 1) launch_fused_dma<3, 2, 1>(L); else launch_fused_dma<3, 2, 2>(L); }
    } else if (sp == 2 && dma != 4) {   // what ships: nontemporal output stores, ring depth 2 (large grids) or 3
        if (dma == 2) launch_fused_dma<2, 2>(L);
        else launch_fused_dma<3, 2>(L);
    } else {
#ifdef GH_ENABLE_PROBES
        // SP = 100 * (load policy + 1) + 10 * probe + store policy -- see k_fused_dma; RGB8 only
        switch (dma * 1000 + sp) {
        case 4002: launch_fused_dma<4, 2>(L); break;
        case 2000: launch_fused_dma<2, 0>(L); break;
        case 2001: launch_fused_dma<2, 1>(L); break;
        case 2102: launch_fused_dma<2, 102>(L); break;
        case 2302: launch_fused_dma<2, 302>(L); break;
        case 2402: launch_fused_dma<2, 402>(L); break;
        case 2502: launch_fused_dma<2, 502>(L); break;
        case 3000: launch_fused_dma<3, 0>(L); break;
        case 3001: launch_fused_dma<3, 1>(L); break;
        case 3010: launch_fused_dma<3, 10>(L); break;
        case 3012: launch_fused_dma<3, 12>(L); break;
        case 3022: launch_fused_dma<3, 22>(L); break;
        case 4000: launch_fused_dma<4, 0>(L); break;
        case 4001: launch_fused_dma<4, 1>(L); break;
        default: GH_REQUIRE(false, "unknown GH_FUSED_DMA / GH_FUSED_DMA_ST probe combination");
        }
#else
        return gh_set_error(GH_ERR_ARG, "unreachable: the release library has ring depths 2 / 3 and one store policy");
#endif
    }
    GH_LAUNCH_CHECK();
    return GH_OK;
}
