// sobel.hip -- 3x3 Sobel magnitude, u8 plane -> u8 plane, for gfx950 (MI355X).
//
// Replaces monolithic/src/sobel.c:6-25 (sobel_edge).  Bound: HBM, 1 R + 1 W byte per pixel, no MFMA.
//
//   k_sobel_dma2   : large aligned frames whose width is a multiple of 8192: two neighbouring 1 KiB strips per wave, regular bands
//   k_sobel_dma    : every other frame of 1024 columns and more, any alignment: strip streaming -- 3x3 reuse in registers, halo
//                    through DPP (stream_core.h) -- with rows brought HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), three rows deep
//   k_sobel_stream : the same strip streaming with register-staged loads: frames narrower than 1024 columns (first width & ~15
//                    columns; k_sobel_px takes the remaining columns)
//   k_sobel_px     : one pixel per lane, any geometry / alignment (small unaligned frames, widths < 16, column tails)
//   every one of them also with LUMA = 1: lum(e, e, e) applied on load -- pass k >= 2 of the reference's gray + Sobel loop
//   (gh_sobel_of_gray_u8_dev; stream_core::plane_row_pairs)
// Probe builds only (-DGH_ENABLE_PROBES, `make probes`): k_sobel_dma4, k_sobel_tiled (the LDS-tile form, sobel_tile.h), other ring depths
// and store policies, the access-pattern probes, and the GH_SOBEL_* knobs that select them.
#include "stream_core.h"
#ifdef GH_ENABLE_PROBES
#include "sobel_tile.h"
#endif

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

#ifdef GH_ENABLE_PROBES
using namespace sobel_tile;
#endif

// ---- strip streaming ---------------------------------------------------------------------------------
// R output rows per strip, C rows per load chunk; chunk ch+1 is in flight while chunk ch is computed.
// CORE 0: packed-i16 row step (stream_core::row16); CORE 1: packed-f32 pair-layout row step (rowf2).  Both run with
// MODE.fp_round = toward zero: this kernel has no rounding-sensitive float op (all sums are exact integers).
template <int R, int C, int CORE, int LUMA = 0>
__global__ __launch_bounds__(stream_core::BLOCK) void k_sobel_stream(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w,
                                                                    int rows, int groups_x, int zero_top, int zero_bot, int ws, const uint8_t *lum_lut)
{
    __shared__ uint8_t lut[LUMA ? 256 : 4];
    if (LUMA) stream_core::load_lum_lut(lut, lum_lut);
    // w: frame width = row pitch (rows need not be 16-byte aligned: the vector accesses below are then unaligned, which
    // gfx950 global memory instructions handle); ws <= w, a multiple of 16: the columns covered by strips.  Columns
    // [ws, w) of a frame whose width is not a multiple of 16 are left to k_sobel_px_cols.
    using namespace stream_core;
    static_assert(C % 2 == 0, "chunk must be even: the h2 ring slot is the row parity");
    constexpr int NROWS = R + 2;
    static_assert(NROWS % C == 0, "R + 2 must be a multiple of the chunk");
    constexpr int NCH = NROWS / C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx_i = (int)(blockIdx.x % (unsigned)groups_x), gy_i = (int)(blockIdx.x / (unsigned)groups_x);
    int xw = (gx_i * WAVES_PER_BLOCK + wave) * STRIP_PX;         // first pixel of this wave's strip
    if (xw >= ws) return;
    if (xw + STRIP_PX > ws && ws >= STRIP_PX) xw = ws - STRIP_PX;   // last strip shifted left onto its neighbour: every strip is full
    const int x = xw + 16 * lane;
    const int y0 = gy_i * R;
    static_assert(LUMA == 0 || CORE == 1, "luma on load: packed-f32 core only");
    __builtin_amdgcn_s_setreg(0x0801, 3);   // MODE.fp_round(f32) = toward zero for the whole kernel (see mag_pack_rtz)
    const float half = half_reg();
    const f32x2 half2 = {half, half};
    const bool in_frame = x < ws;
    const bool full = xw + STRIP_PX <= ws;                       // wave-uniform: no lane predication needed
    const int hx = lane == 0 ? xw - 1 : xw + STRIP_PX;           // pixel just outside the strip, fetched by lanes 0 / 63
    const bool hvalid = (lane == 0 || lane == 63) && hx >= 0 && hx < w;

    State16 st16;
    StateF2 stf;
    if (CORE == 0) st16.clear(); else stf.clear();

    auto load_chunk = [&](int ch, uint4 (&q)[C], uint32_t (&hb)[C]) {
        if (ch >= NCH) return;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int yy = clamp_row(y0 - 1 + ch * C + j, rows, zero_top, zero_bot);   // always a readable row
            const uint8_t *rowp = src + (ptrdiff_t)yy * w;
            if (full) q[j] = load_stream(reinterpret_cast<const uint4 *>(rowp + x));
            else {
                q[j] = make_uint4(0, 0, 0, 0);
                if (in_frame) q[j] = load_stream(reinterpret_cast<const uint4 *>(rowp + x));
                else if (x == ws && ws < w) q[j].x = rowp[ws];     // column ws exists (w % 16 != 0): right neighbour of the last strip column
            }
            hb[j] = 0;
            if (hvalid) hb[j] = rowp[hx];
        }
    };
    auto compute_chunk = [&](int ch, const uint4 (&q)[C], const uint32_t (&hb)[C]) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = ch * C + j;
            uint32_t ql = __builtin_amdgcn_update_dpp(0u, q[j].w, 0x138, 0xf, 0xf, true);   // wave_shr:1  lane l <- lane l-1
            uint32_t qr = __builtin_amdgcn_update_dpp(0u, q[j].x, 0x130, 0xf, 0xf, true);   // wave_shl:1  lane l <- lane l+1
            if (lane == 0) ql = hb[j] << 24;
            if (lane == 63) qr = hb[j];
            uint4 out;
            if (CORE == 0) {
                out = (j & 1) ? row16<1>(st16, q[j], ql, qr, half) : row16<0>(st16, q[j], ql, qr, half);
            } else {
                f32x2 G[10];
                plane_row_pairs<LUMA>(q[j], ql, qr, G, lut);
                out = (j & 1) ? rowf2<1, 1>(stf, G, half2) : rowf2<0, 1>(stf, G, half2);
            }
            if (i >= 2) {
                const int y = y0 + i - 2;
                if (x == 0) out.x &= 0xffffff00u;            // frame column 0
                if (x + 15 == w - 1) out.w &= 0x00ffffffu;   // frame column w-1
                if ((y == 0 && zero_top) || (y == rows - 1 && zero_bot)) out = make_uint4(0, 0, 0, 0);
                if (in_frame && y < rows) store_strip(dst + (size_t)y * w + xw, lane, out);
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

// ---- LDS-DMA form (width >= 1024): a wave's 1 KiB of each row goes HBM -> LDS directly, D rows deep -------------------
// Same ring as fused.hip's k_fused_dma (one global_load_lds_dwordx4 + one halo dword per row, own vmcnt accounting),
// but the prefetch depth costs LDS (1 KiB per row and wave) instead of registers, so it can be deep: the plane kernel
// moves only 2 B/px, a wave's row is consumed in ~550 cycles and the register-staged form cannot keep enough bytes
// in flight at 3 waves per SIMD.  The last strip of a row is shifted left to end at the frame edge (all strips full).
//
// UNAL (frames whose row pitch or base address is not a multiple of 16): a global_load_lds_dwordx4 whose lanes' 16 bytes are
// not 16-byte aligned costs the texture-address / L1 path four times the tag accesses of an aligned one (rocprofv3
// TCP_TOTAL_CACHE_ACCESSES 3.4x, TA_BUSY 1.9x on a 16370-wide plane; HBM traffic unchanged) and the kernel 7-9 % of its time.
// UNAL != 0 therefore starts every row's piece at the 16-byte aligned address G = A - a' at or before the strip's first byte A
// (a' = A mod 16, or 16 when that is 0 and a left neighbour exists, so the piece always carries the left halo pixel) and
// fetches the 16 or 32 bytes behind the piece with the row's two-lane instruction.  An LDS-DMA destination needs only DWORD
// alignment (measured: tools/glds_align_test.hip -- byte offsets are rounded down), so the piece is landed 4 * (a' / 4) bytes
// early and the strip's first pixel sits at byte s = A mod 4 of the slot's lane grid: every ds_read stays naturally aligned.
// UNAL = 1: s is 0 for every row (pitch and base multiples of 4): nothing else changes.  UNAL = 2: the lane's 16 pixels and
// its two neighbours are re-aligned in registers, six v_alignbyte_b32 with the row's (wave-uniform) s.
template <int D, int PROBE = 0, int SP = 2, int LP = 1, int UNAL = 0, int REG = 0, int LUMA = 0>
__global__ __launch_bounds__(stream_core::BLOCK) void k_sobel_dma(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows,
                                                                 int groups_x, stream_core::BandPlan plan, int zero_top, int zero_bot, GhHaloWait hw,
                                                                 const uint8_t *lum_lut)
{
    using namespace stream_core;
    __shared__ uint8_t lut[LUMA ? 256 : 4];
    if (LUMA) load_lum_lut(lut, lum_lut);
    constexpr int PRE = UNAL ? 16 : 0;                           // UNAL: 16 bytes in front of the lane grid (the piece may land up to 16 early)
    constexpr int SLOT = UNAL ? 16 + STRIP_PX + 32 : STRIP_PX + 16;   // one strip row + the halo dword pair / + pre-area and 32-byte tail
    constexpr int CHK = (D % 2 == 0) ? D : 2 * D;                // unroll: slot = j % D and h2 ring parity = j & 1 are static
    static_assert(3 * (D - 1) <= 60, "vmcnt is 6 bits");
    __shared__ __attribute__((aligned(16))) uint8_t ring[WAVES_PER_BLOCK][D][SLOT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int gx_i, gy_i;
    band_of_block(plan, blockIdx.x, groups_x, gx_i, gy_i);
    int xw = (gx_i * WAVES_PER_BLOCK + wave) * STRIP_PX;
    if (xw >= w) return;
    if (xw + STRIP_PX > w) xw = w - STRIP_PX;
    const int x = xw + 16 * lane;
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
    if (REG && y0 + R > rows) y0 = rows - R;    // regular bands: the last band is shifted UP to end at the frame edge (rows >= R): every band is
                                                // full, the rows it shares with the band above are computed and stored twice with identical bytes
    const int nrows = min(R, rows - y0) + 2;
    const int nch = (nrows + CHK - 1) / CHK;
    // sharded step, halo-flag form (gh_common.h; see k_fused_dma): the shard's first band walks upward, so that a band's halo row is the
    // LAST row it requests, and the wait for the exchange sits right in front of that request
    const bool up = band_walks_up(plan, gy_i) || (hw.flag && !zero_top && y0 == 0 && y0 + nrows - 2 < rows);
    if (hw.flag) gh_halo_started(hw);
    __builtin_amdgcn_s_setreg(0x0801, 3);   // MODE.fp_round(f32) = toward zero for the whole kernel (see mag_pack_rtz)
    const f32x2 half2 = half2_reg();
    const uint32_t voff = (uint32_t)xw + (uint32_t)lane * 16u;
    // halo dwords: lane 0 -> the 4 bytes ending at the strip's first byte (pixel xw-1 = its byte 3),
    //              lane 1 -> the 4 bytes after the strip's last byte (pixel xw+1024 = its byte 0); clamped when outside
    uint32_t hoff = lane == 0 ? (uint32_t)xw - 4u : (uint32_t)(xw + STRIP_PX);
    if ((lane == 0 && xw == 0) || (lane != 0 && xw + STRIP_PX >= w)) hoff = 0;
    // widths that are not multiples of 16: the right halo pixel may be within 3 bytes of the end of the row; the dword is then
    // fetched `over` bytes earlier (it never reaches past the row, hence never past the plane) and shifted back after the read
    const int over = xw + STRIP_PX < w ? max(0, xw + STRIP_PX + 4 - w) : 0;             // wave-uniform, 0 for aligned frames
    if (lane != 0) hoff -= (uint32_t)over;
    const uint32_t rsh = 8u * (uint32_t)over;
    // likewise on the left: a shifted last strip may start at column 1..3 (w = 1025..1027); its left dword then starts at the
    // row's first byte instead of before it (offsets are unsigned: "xw - 4" would be 4 GiB away) and is shifted up after the read
    const int under = xw > 0 ? max(0, 4 - xw) : 0;                                      // wave-uniform, 0 for aligned frames
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
        return uniform_ptr(src + (ptrdiff_t)yy * w + xw);
    };
    auto issue_row = [&](int i, int slot) {
        if (hw.flag) {                                           // (wave-uniform) halo row of a shard: has the exchange landed?
            const int raw = up ? y0 + nrows - 2 - i : y0 - 1 + i;
            if ((raw < 0 && !zero_top) || (raw >= rows && !zero_bot)) gh_halo_wait(hw);
        }
        const uint32_t d = uniform_u32(lds0 + (uint32_t)slot * SLOT);
        if (UNAL) {
            const uint8_t *A = strip_row(i);
            const uint32_t a = (uint32_t)(uintptr_t)A & 15u;
            const uint32_t ap = (a == 0u && xw > 0) ? 16u : a;
            // tail blocks (relative to G): the one behind the piece holds the strip's last a' pixels (a' >= 1) or the right halo
            // pixel (a' = 0); neither -> the piece's last block again.  The second one is fetched only where it holds the right
            // halo pixel (a' = 16).  Every block fetched thus contains a byte of the frame: no access leaves its pages.
            const uint32_t t0 = (ap == 0u && !has_right) ? (uint32_t)STRIP_PX - 16u : (uint32_t)STRIP_PX;
            const uint32_t t1 = (ap == 16u && has_right) ? (uint32_t)STRIP_PX + 16u : t0;
            const uint32_t dm = uniform_u32(d + PRE + (a & 3u) - ap);
            glds_row1k_tail(dm, dm + STRIP_PX, uniform_ptr(A - ap), (uint32_t)lane * 16u, lane == 0 ? t0 : t1, slot);
        } else {
            if (PROBE == 5) glds_row1k_nohalo(d, d + STRIP_PX, strip_row(i) - xw, voff, hoff, slot);      // probe: what do the two 4-byte halo requests per row cost?
            else glds_row1k_halo<LP>(d, d + STRIP_PX, strip_row(i) - xw, voff, hoff, slot);      // always exactly 2 vector-memory instructions
        }
    };
#pragma unroll
    for (int i = 0; i < D - 1; ++i) issue_row(i, i);

    // One row step, in three statically different forms (so that the generated code can be CHECKED: tests/test_isa_vmcnt.py):
    //   KIND 0  generic: rows may be missing at the end of the band, the first two rows produce no output.  Only the DMA
    //           instructions of the D-1 younger rows are counted (2 each); stores in between make the wait conservative.
    //   KIND 1  transition: every row of the chunk exists, issues a row and stores one; same conservative count.
    //   KIND 2  steady: as 1, and the D-1 previous steps are statically known to have stored (they ran as KIND 1 or 2),
    //           so D-1 more operations are younger than row i's DMA: exact count 3 (D-1).
    auto row_step = [&](auto kind_c, int ch, int j) {
        constexpr int KIND = decltype(kind_c)::value;
        const int i = ch * CHK + j;
        if (KIND == 0 && i >= nrows) return true;                    // past the band: so is every later step -- the wave is done
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the slot about to be refilled has been read
        if (PROBE == 3 || PROBE == 4) __builtin_amdgcn_s_barrier();  // probe: the workgroup's four strips advance row by row together (every wave must be live)
        const bool issue = KIND == 0 ? i + D - 1 < nrows : (KIND == 4 ? j + D - 1 < CHK : true);     // KIND 3 / 4: regular bands, see k_sobel_dma2
        if (issue) {
            issue_row(i + D - 1, (j + D - 1) % D);
            constexpr int NLD = PROBE == 5 ? 1 : 2;                   // vector-memory instructions per ring row
            if (KIND == 2 || KIND == 4) wait_vm<(NLD + 1) * (D - 1)>(j % D); else wait_vm<NLD * (D - 1)>(j % D);
        } else {
            wait_vm<0>(j % D);
        }
        const uint8_t *slotp = &ring[wave][j % D][0];
        uint4 q = *reinterpret_cast<const uint4 *>(slotp + PRE + 16 * lane);
        uint2 hh;
        if (UNAL) hh = make_uint2(*reinterpret_cast<const uint32_t *>(slotp + PRE - 4), *reinterpret_cast<const uint32_t *>(slotp + PRE + STRIP_PX));
        else hh = *reinterpret_cast<const uint2 *>(slotp + STRIP_PX);
        uint32_t ql = __builtin_amdgcn_update_dpp(0u, q.w, 0x138, 0xf, 0xf, true);   // wave_shr:1  lane l <- lane l-1
        uint32_t qr = __builtin_amdgcn_update_dpp(0u, q.x, 0x130, 0xf, 0xf, true);   // wave_shl:1  lane l <- lane l+1
        if (lane == 0) ql = UNAL ? hh.x : hh.x << lsh;
        if (lane == 63) qr = UNAL ? hh.y : hh.y >> rsh;
        if (UNAL == 2) {                                             // the strip starts at byte s of the lane grid
            const uint32_t sh = (uint32_t)(uintptr_t)strip_row(i) & 3u;
            ql = __builtin_amdgcn_alignbyte(q.x, ql, sh);
            q.x = __builtin_amdgcn_alignbyte(q.y, q.x, sh);
            q.y = __builtin_amdgcn_alignbyte(q.z, q.y, sh);
            q.z = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
            q.w = __builtin_amdgcn_alignbyte(qr, q.w, sh);
            qr >>= 8u * sh;
        }
        f32x2 G[10];
        plane_row_pairs<LUMA>(q, ql, qr, G, lut);
        uint4 out = (j & 1) ? rowf2<1, 1, PROBE == 1>(st, G, half2) : rowf2<0, 1, PROBE == 1>(st, G, half2);
        if (PROBE == 2 || PROBE == 4 || PROBE == 5) out = make_uint4(q.x ^ hh.x, q.y, q.z, q.w ^ hh.y);   // timing probe: the kernel's access pattern without its arithmetic
        if (KIND == 3 ? j >= 2 : (KIND != 0 || i >= 2)) {           // y = y0 + i - 2 < rows by construction of nrows
            const int y = up ? y0 + nrows - 1 - i : y0 + i - 2;
            out.x &= first_col_mask;                                 // the frame's first / last column is border: written as 0
            out.w &= last_col_mask;
            zero_if((y == 0 && zero_top) || (y == rows - 1 && zero_bot), out);
            store_strip_p<SP>(dst + (size_t)y * w + xw, lane, out);
        }
        return false;
    };
    auto chunk = [&](auto kind_c, int ch) {          // true: the band is finished
#pragma unroll
        for (int j = 0; j < CHK; ++j)
            if (row_step(kind_c, ch, j)) return true;
        return false;
    };
    if (REG) {                                       // regular bands: see k_sobel_dma2
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

// ---- two strips per wave (width >= 2048): a wave owns 2 KiB of every row ----------------------------------------------------------
// The no-arithmetic skeletons run faster the more contiguous bytes ONE wave reads and writes per row (1 KiB:
// 67-69 %, 2 KiB: 68-69 %, 4 KiB: 72 % of 8 TB/s).  The rolling Sobel state is 64 VGPRs per 1,024 columns, so two strips per wave
// is what fits in 256 registers: the same row step run on two neighbouring 1 KiB strips out of one ring slot (two DMA pieces + the
// halo dword per row, two stores), half as many waves, the per-row scalar work shared.  Between the two strips the neighbour
// pixels come from the other strip's registers (v_readlane), not from LDS.
// REG = 1 ("regular bands"): every band has exactly nrows = R + 2 input rows and nrows is a multiple of the unroll CHK (the host
// checks rows >= R and (R + 2) % CHK == 0: R = 64 or 16 with D = 3; a last band that would be ragged is shifted up instead).  Then which steps issue a row and which store is known
// per unrolled step -- first chunk: steps 0, 1 store nothing; last chunk: the last D - 1 steps issue nothing -- and the band runs
// without a single data-dependent branch.  That is what keeps this kernel inside 256 registers at ring depth 3: the generic form's
// early exits cost 70 more (allocated as AGPR spills in the tail, but they halve the residency of the whole kernel).
// UNAL: rows of frames whose pitch or base is not a multiple of 16 start at the aligned address before the strip pair and land
// dword-shifted (1), plus the byte re-alignment of both strips (2) -- exactly k_sobel_dma's scheme on a 2 KiB row.
template <int D, int LB = 1, int REG = 0, int UNAL = 0, int LUMA = 0>
__global__ __launch_bounds__(stream_core::BLOCK, LB) void k_sobel_dma2(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows,
                                                                  int groups_x, stream_core::BandPlan plan, int zero_top, int zero_bot, GhHaloWait hw,
                                                                  const uint8_t *lum_lut)
{
    using namespace stream_core;
    __shared__ uint8_t lut[LUMA ? 256 : 4];
    if (LUMA) load_lum_lut(lut, lum_lut);
    constexpr int SW = 2 * STRIP_PX;
    constexpr int PRE = UNAL ? 16 : 0;
    constexpr int SLOT = UNAL ? 16 + SW + 32 : SW + 16;
    constexpr int CHK = (D % 2 == 0) ? D : 2 * D;
    static_assert(5 * (D - 1) <= 60, "vmcnt is 6 bits");
    __shared__ __attribute__((aligned(16))) uint8_t ring[WAVES_PER_BLOCK][D][SLOT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int gx_i, gy_i;
    band_of_block(plan, blockIdx.x, groups_x, gx_i, gy_i);
    int xw = (gx_i * WAVES_PER_BLOCK + wave) * SW;
    if (xw >= w) return;
    if (xw + SW > w) xw = w - SW;                                // last pair of strips: shifted left to end at the frame edge
    if (gy_i < 0) return;
    const int R = plan.R[0];
    int y0 = gy_i * R;
    if (REG && y0 + R > rows) y0 = rows - R;    // regular bands: the last band is shifted up to end at the frame edge (see k_sobel_dma)
    const int nrows = min(R, rows - y0) + 2;
    const int nch = (nrows + CHK - 1) / CHK;
    // sharded step, halo-flag form (gh_common.h; see k_fused_dma): first band upward, the wait in front of the halo row's request
    const bool up = hw.flag && !zero_top && y0 == 0 && y0 + nrows - 2 < rows;
    if (hw.flag) gh_halo_started(hw);
    __builtin_amdgcn_s_setreg(0x0801, 3);   // MODE.fp_round(f32) = toward zero for the whole kernel (see mag_pack_rtz)
    const f32x2 half2 = half2_reg();
    const uint32_t voff = (uint32_t)xw + (uint32_t)lane * 16u;
    uint32_t hoff = lane == 0 ? (uint32_t)xw - 4u : (uint32_t)(xw + SW);
    if ((lane == 0 && xw == 0) || (lane != 0 && xw + SW >= w)) hoff = 0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)&ring[wave][0][0];

    StateF2 st0, st1;
    st0.clear();
    st1.clear();
    const uint32_t first_col_mask = (xw + 16 * lane == 0) ? 0xffffff00u : 0xffffffffu;
    const uint32_t last_col_mask = (xw + STRIP_PX + 16 * lane + 15 == w - 1) ? 0x00ffffffu : 0xffffffffu;

    const bool has_right = xw + SW < w;
    auto strip_row = [&](int i) {                                // first byte of the strip pair in band row i (wave-uniform)
        const int yy = clamp_row(up ? y0 + nrows - 2 - i : y0 - 1 + i, rows, zero_top, zero_bot);
        return uniform_ptr(src + (ptrdiff_t)yy * w + xw);
    };
    auto issue_row = [&](int i, int slot) {                      // always exactly 3 vector-memory instructions
        if (hw.flag) {                                           // (wave-uniform) halo row of a shard: has the exchange landed?
            const int raw = up ? y0 + nrows - 2 - i : y0 - 1 + i;
            if ((raw < 0 && !zero_top) || (raw >= rows && !zero_bot)) gh_halo_wait(hw);
        }
        const uint32_t d = uniform_u32(lds0 + (uint32_t)slot * SLOT);
        if (UNAL) {                                              // see k_sobel_dma: a', the tail blocks, the landing offset
            const uint8_t *A = strip_row(i);
            const uint32_t a = (uint32_t)(uintptr_t)A & 15u;
            const uint32_t ap = (a == 0u && xw > 0) ? 16u : a;
            const uint32_t t0 = (ap == 0u && !has_right) ? (uint32_t)SW - 16u : (uint32_t)SW;
            const uint32_t t1 = (ap == 16u && has_right) ? (uint32_t)SW + 16u : t0;
            const uint32_t dm = uniform_u32(d + PRE + (a & 3u) - ap);
            glds_row2k_tail(dm, dm + SW, uniform_ptr(A - ap), (uint32_t)lane * 16u, lane == 0 ? t0 : t1, slot);
        } else {
            glds_row2k_halo(d, d + SW, strip_row(i) - xw, voff, hoff, slot);
        }
    };
#pragma unroll
    for (int i = 0; i < D - 1; ++i) issue_row(i, i);

    // row step in the statically different forms of k_sobel_dma (0 generic, 1 transition, 2 steady) plus, for regular bands, 3 = first
    // chunk and 4 = last chunk; per row 3 DMA instructions and 2 stores: the generic, transition and first forms count the DMA
    // instructions of the D-1 younger rows (3 each), the steady and last forms also their stores: 5 (D-1)
    auto row_step = [&](auto kind_c, int ch, int j) {
        constexpr int KIND = decltype(kind_c)::value;
        const int i = ch * CHK + j;
        if (KIND == 0 && i >= nrows) return true;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool issue = KIND == 0 ? i + D - 1 < nrows : (KIND == 4 ? j + D - 1 < CHK : true);     // (j is a constant in the unrolled chunk)
        if (issue) {
            issue_row(i + D - 1, (j + D - 1) % D);
            if (KIND == 2 || KIND == 4) wait_vm<5 * (D - 1)>(j % D); else wait_vm<3 * (D - 1)>(j % D);
        } else {
            wait_vm<0>(j % D);
        }
        const uint8_t *slotp = &ring[wave][j % D][0];
        const uint4 q0 = *reinterpret_cast<const uint4 *>(slotp + PRE + 16 * lane);
        const uint4 q1 = *reinterpret_cast<const uint4 *>(slotp + PRE + STRIP_PX + 16 * lane);
        uint2 hh;                                                    // the dword in front of the lane grid, the dword behind it
        if (UNAL) hh = make_uint2(*reinterpret_cast<const uint32_t *>(slotp + PRE - 4), *reinterpret_cast<const uint32_t *>(slotp + PRE + SW));
        else hh = *reinterpret_cast<const uint2 *>(slotp + SW);
        const uint32_t sh = UNAL == 2 ? (uint32_t)(uintptr_t)strip_row(i) & 3u : 0u;     // the pair starts at byte sh of the lane grid
        const int y = up ? y0 + nrows - 1 - i : y0 + i - 2;
        // a frame-border row can only be a band's first or last output row: in the regular form those are two statically known steps
        // (the steady chunks never see one), so the select below disappears from the steady loop
        // (a band that walks upward -- a shard's first band with a real halo row above -- touches neither border row)
        const bool may_be_border_row = !REG || (KIND == 3 && j == 2) || (KIND == 4 && j == CHK - 1);
        const bool zero_row = may_be_border_row && ((y == 0 && zero_top) || (y == rows - 1 && zero_bot));
        auto strip = [&](auto sub_c, StateF2 &st, uint4 q, uint32_t left_of_lane0, uint32_t right_of_lane63) {
            constexpr int SUB = decltype(sub_c)::value;
            uint32_t ql = __builtin_amdgcn_update_dpp(0u, q.w, 0x138, 0xf, 0xf, true);   // wave_shr:1  lane l <- lane l-1
            uint32_t qr = __builtin_amdgcn_update_dpp(0u, q.x, 0x130, 0xf, 0xf, true);   // wave_shl:1  lane l <- lane l+1
            if (lane == 0) ql = left_of_lane0;
            if (lane == 63) qr = right_of_lane63;
            if (UNAL == 2) {
                ql = __builtin_amdgcn_alignbyte(q.x, ql, sh);
                q.x = __builtin_amdgcn_alignbyte(q.y, q.x, sh);
                q.y = __builtin_amdgcn_alignbyte(q.z, q.y, sh);
                q.z = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
                q.w = __builtin_amdgcn_alignbyte(qr, q.w, sh);
                qr >>= 8u * sh;
            }
            f32x2 G[10];
            plane_row_pairs<LUMA>(q, ql, qr, G, lut);
            uint4 out = (j & 1) ? rowf2<1, 1>(st, G, half2) : rowf2<0, 1>(st, G, half2);
            if (KIND == 3 ? j >= 2 : (KIND != 0 || i >= 2)) {       // y < rows by construction of nrows
                if (SUB == 0) out.x &= first_col_mask;               // the frame's first / last column is border: written as 0
                if (SUB == 1) out.w &= last_col_mask;
                zero_if(zero_row, out);
                store_strip_p<2>(dst + (size_t)y * w + xw + SUB * STRIP_PX, lane, out);
            }
        };
        // the pixel right of strip 0 is strip 1's first (lane 0, byte 0 of q1.x), the pixel left of strip 1 strip 0's last (lane 63, byte 3 of q0.w)
        const uint32_t first_of_1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q1.x);
        const uint32_t last_of_0 = (uint32_t)__builtin_amdgcn_readlane((int)q0.w, 63);
        strip(std::integral_constant<int, 0>{}, st0, q0, hh.x, first_of_1);
        strip(std::integral_constant<int, 1>{}, st1, q1, last_of_0, hh.y);
        return false;
    };
    auto chunk = [&](auto kind_c, int ch) {
#pragma unroll
        for (int j = 0; j < CHK; ++j)
            if (row_step(kind_c, ch, j)) return true;
        return false;
    };
    if (REG) {                                       // nch >= 3 chunks, all of them full
        // the exact count of a chunk's first step needs the D - 1 steps before it to have stored: true from chunk 1 on when the first
        // chunk's last D - 1 steps store (CHK - 2 >= D - 1), else from chunk 2 on
        chunk(std::integral_constant<int, 3>{}, 0);
        int ch = 1;
        if (CHK - 2 < D - 1) { chunk(std::integral_constant<int, 1>{}, 1); ch = 2; }
#pragma unroll 1
        for (; ch < nch - 1; ++ch) chunk(std::integral_constant<int, 2>{}, ch);
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

#ifdef GH_ENABLE_PROBES
// ======== measurement lab: variants that lost their A/B (kept selectable in probe builds, `make probes`) and access-pattern probes ========
// ---- four strips per wave: 4 KiB of every row, packed-i16 state ------------------------------------------------------------------------
// The no-arithmetic skeletons (DESIGN.md / NOTES.md) say a wave that owns 4 KiB of every row reads closer to a linear stream than one that
// owns 1 or 2 KiB; the f32 pair state of k_sobel_dma2 (64 VGPRs per strip) stops at two strips.  The packed-i16 state (State16, 32 VGPRs
// per strip) fits four: CORE 0 = row16 (scalar f32 magnitude, 10 VALU instructions per pixel), CORE 1 = row16p (packed f32 magnitude, 9.1)
// against 8.4 for k_sobel_dma2.  Regular bands only (every band R + 2 input rows in whole chunks, the last one shifted up), aligned
// frames at least 4096 columns wide; per row 5 DMA instructions (4 pieces + the halo dword) and 4 stores.
template <int D, int CORE = 1>
__global__ __launch_bounds__(stream_core::BLOCK) void k_sobel_dma4(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows,
                                                                  int groups_x, stream_core::BandPlan plan, int zero_top, int zero_bot, GhHaloWait hw)
{
    using namespace stream_core;
    constexpr int NS = 4, SW = NS * STRIP_PX;
    constexpr int SLOT = SW + 16;
    constexpr int CHK = (D % 2 == 0) ? D : 2 * D;
    static_assert((2 * NS + 1) * (D - 1) <= 60, "vmcnt is 6 bits");
    __shared__ __attribute__((aligned(16))) uint8_t ring[WAVES_PER_BLOCK][D][SLOT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int gx_i, gy_i;
    band_of_block(plan, blockIdx.x, groups_x, gx_i, gy_i);
    int xw = (gx_i * WAVES_PER_BLOCK + wave) * SW;
    if (xw >= w) return;
    if (xw + SW > w) xw = w - SW;                                // last group of strips: shifted left to end at the frame edge
    if (gy_i < 0) return;
    const int R = plan.R[0];
    int y0 = gy_i * R;
    if (y0 + R > rows) y0 = rows - R;           // the last band is shifted up to end at the frame edge (see k_sobel_dma)
    const int nrows = R + 2;
    const int nch = nrows / CHK;
    const bool up = hw.flag && !zero_top && y0 == 0 && y0 + nrows - 2 < rows;      // sharded step, halo-flag form: see k_fused_dma
    if (hw.flag) gh_halo_started(hw);
    __builtin_amdgcn_s_setreg(0x0801, 3);   // MODE.fp_round(f32) = toward zero for the whole kernel (see mag_pack_rtz)
    const float half = half_reg();
    const f32x2 half2 = half2_reg();
    const uint32_t voff = (uint32_t)xw + (uint32_t)lane * 16u;
    uint32_t hoff = lane == 0 ? (uint32_t)xw - 4u : (uint32_t)(xw + SW);
    if ((lane == 0 && xw == 0) || (lane != 0 && xw + SW >= w)) hoff = 0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)&ring[wave][0][0];

    State16 st[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) st[k].clear();

    auto strip_row = [&](int i) {                                // first byte of the wave's 4 KiB in band row i (wave-uniform)
        const int yy = clamp_row(up ? y0 + nrows - 2 - i : y0 - 1 + i, rows, zero_top, zero_bot);
        return uniform_ptr(src + (ptrdiff_t)yy * w + xw);
    };
    auto issue_row = [&](int i, int slot) {                      // always exactly 5 vector-memory instructions
        if (hw.flag) {                                           // (wave-uniform) halo row of a shard: has the exchange landed?
            const int raw = up ? y0 + nrows - 2 - i : y0 - 1 + i;
            if ((raw < 0 && !zero_top) || (raw >= rows && !zero_bot)) gh_halo_wait(hw);
        }
        const uint32_t d = uniform_u32(lds0 + (uint32_t)slot * SLOT);
        glds_row4k_halo(d, d + SW, strip_row(i) - xw, voff, hoff, slot);
    };
#pragma unroll
    for (int i = 0; i < D - 1; ++i) issue_row(i, i);

    // row step in the forms of a regular band (see k_sobel_dma2): 3 = first chunk (steps 0 and 1 store nothing), 1 = transition,
    // 2 = steady, 4 = last chunk (its last D - 1 steps issue nothing).  Counted: the first and transition forms the DMA instructions of the
    // D - 1 younger rows (5 each), the steady and last forms also their stores: 9 (D - 1).
    auto row_step = [&](auto kind_c, int ch, int j) {
        constexpr int KIND = decltype(kind_c)::value;
        const int i = ch * CHK + j;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool issue = KIND == 4 ? j + D - 1 < CHK : true;       // (j is a constant in the unrolled chunk)
        if (issue) {
            issue_row(i + D - 1, (j + D - 1) % D);
            if (KIND == 2 || KIND == 4) wait_vm<(2 * NS + 1) * (D - 1)>(j % D); else wait_vm<(NS + 1) * (D - 1)>(j % D);
        } else {
            wait_vm<0>(j % D);
        }
        const uint8_t *slotp = &ring[wave][j % D][0];
        uint4 q[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) q[k] = *reinterpret_cast<const uint4 *>(slotp + k * STRIP_PX + 16 * lane);
        const uint2 hh = *reinterpret_cast<const uint2 *>(slotp + SW);   // .x: the dword before the wave's bytes, .y: the dword behind them
        const int y = up ? y0 + nrows - 1 - i : y0 + i - 2;
        const bool zero_row = (y == 0 && zero_top) || (y == rows - 1 && zero_bot);
        const bool store = KIND == 3 ? j >= 2 : true;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            // the pixel left of strip k is strip k-1's last (lane 63, top byte of q.w), the pixel right of it strip k+1's first (lane 0, low byte of q.x)
            const uint32_t left_of_lane0 = k == 0 ? hh.x : (uint32_t)__builtin_amdgcn_readlane((int)q[k > 0 ? k - 1 : 0].w, 63);
            const uint32_t right_of_lane63 = k == NS - 1 ? hh.y : (uint32_t)__builtin_amdgcn_readfirstlane((int)q[k < NS - 1 ? k + 1 : 0].x);
            uint32_t ql = __builtin_amdgcn_update_dpp(0u, q[k].w, 0x138, 0xf, 0xf, true);   // wave_shr:1  lane l <- lane l-1
            uint32_t qr = __builtin_amdgcn_update_dpp(0u, q[k].x, 0x130, 0xf, 0xf, true);   // wave_shl:1  lane l <- lane l+1
            if (lane == 0) ql = left_of_lane0;
            if (lane == 63) qr = right_of_lane63;
            uint4 out;
            if (CORE == 0) out = (j & 1) ? row16<1>(st[k], q[k], ql, qr, half) : row16<0>(st[k], q[k], ql, qr, half);
            else out = (j & 1) ? row16p<1>(st[k], q[k], ql, qr, half2) : row16p<0>(st[k], q[k], ql, qr, half2);
            if (store) {
                const int x = xw + k * STRIP_PX + 16 * lane;
                if (k == 0 && x == 0) out.x &= 0xffffff00u;
                if (k == NS - 1 && x + 15 == w - 1) out.w &= 0x00ffffffu;
                if (zero_row) out = make_uint4(0, 0, 0, 0);
                store_strip_p<2>(dst + (size_t)y * w + xw + k * STRIP_PX, lane, out);
            }
        }
    };
    auto chunk = [&](auto kind_c, int ch) {
#pragma unroll
        for (int j = 0; j < CHK; ++j) row_step(kind_c, ch, j);
    };
    chunk(std::integral_constant<int, 3>{}, 0);
    int ch = 1;
    if (CHK - 2 < D - 1) { chunk(std::integral_constant<int, 1>{}, 1); ch = 2; }
#pragma unroll 1
    for (; ch < nch - 1; ++ch) chunk(std::integral_constant<int, 2>{}, ch);
    chunk(std::integral_constant<int, 4>{}, nch - 1);
}

// ---- access-pattern probe: wider strips -- a wave owns PIECES KiB of every row (PIECES LDS-DMA pieces + the halo dword per row step) ----
// The fused kernel (3 KiB of RGB per wave and row) sits closer to its linear-copy ceiling than the plane Sobel (1 KiB): is it the
// size of the contiguous piece a wave reads per row?  Same ring, waits and stores as k_sobel_dma's no-arithmetic skeleton
// (GH_SOBEL_DMA=33), strips PIECES times as wide, a quarter / half as many waves.  Output = a copy of the input rows (not a Sobel result).
// OUTDIV = 3: the fused kernel's shape -- PIECES KiB in, PIECES / 3 KiB out per wave and row (run it on a "plane" three times as wide).
template <int PIECES, int D, int WPB = 4, int OUTDIV = 1>
__global__ __launch_bounds__(64 * WPB) void k_wide_strip_probe(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows,
                                                                        int groups_x, stream_core::BandPlan plan)
{
    using namespace stream_core;
    constexpr int SW = STRIP_PX * PIECES;
    constexpr int SLOT = SW + 16;
    __shared__ __attribute__((aligned(16))) uint8_t ring[WPB][D][SLOT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int gx_i, gy_i;
    band_of_block(plan, blockIdx.x, groups_x, gx_i, gy_i);
    if (gy_i < 0) return;
    const int xw = (gx_i * WPB + wave) * SW;
    if (xw + SW > w) return;
    const int R = plan.R[0], y0 = gy_i * R;
    const int nrows = min(R, rows - y0) + 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)&ring[wave][0][0];
    const uint32_t voff = (uint32_t)xw + (uint32_t)lane * 16u;
    uint32_t hoff = lane == 0 ? (uint32_t)xw - 4u : (uint32_t)(xw + SW);
    if ((lane == 0 && xw == 0) || (lane != 0 && xw + SW >= w)) hoff = 0;
    auto issue = [&](int i, int slot) {
        int yy = y0 - 1 + i; yy = yy < 0 ? 0 : (yy >= rows ? rows - 1 : yy);
        const uint8_t *rowb = uniform_ptr(src + (ptrdiff_t)yy * w);
        const uint32_t d = uniform_u32(lds0 + (uint32_t)slot * SLOT);
        const uint32_t dh = d + SW;
        uint32_t keep; uint64_t keepx;
#pragma unroll
        for (int p = 0; p < PIECES; ++p)          // one instruction per statement; volatile asm keeps their order
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff + (uint32_t)p * STRIP_PX), "s"(rowb), "s"(d + (uint32_t)p * STRIP_PX) : "memory");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, 3\n\tglobal_load_lds_dword %4, %2\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(keepx) : "s"(rowb), "s"(dh), "v"(hoff) : "memory");
    };
#pragma unroll
    for (int i = 0; i < D - 1; ++i) issue(i, i);
#pragma unroll 1
    for (int i0 = 0; i0 < nrows; i0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int i = i0 + j;
            if (i >= nrows) return;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (i + D - 1 < nrows) {
                issue(i + D - 1, (j + D - 1) % D);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PIECES + 1) * (D - 1)) : "memory");      // conservative: stores in between not counted
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const uint8_t *slotp = &ring[wave][j][0];
            const uint2 hh = *reinterpret_cast<const uint2 *>(slotp + SW);
            if (i >= 2) {
                const int y = y0 + i - 2;
#pragma unroll
                for (int p = 0; p < PIECES / OUTDIV; ++p) {
                    uint4 q = *reinterpret_cast<const uint4 *>(slotp + p * OUTDIV * STRIP_PX + 16 * lane);
#pragma unroll
                    for (int o = 1; o < OUTDIV; ++o) {
                        const uint4 t = *reinterpret_cast<const uint4 *>(slotp + (p * OUTDIV + o) * STRIP_PX + 16 * lane);
                        q.x ^= t.x; q.y ^= t.y; q.z ^= t.z; q.w ^= t.w;
                    }
                    q.x ^= hh.x & 0u; q.w ^= hh.y & 0u;
                    store_strip_p<2>(dst + (size_t)y * (w / OUTDIV) + xw / OUTDIV + p * STRIP_PX, lane, q);
                }
            } else {
                asm volatile("" ::"v"(hh.x), "v"(hh.y));
            }
        }
    }
}

// ---- access-pattern probe: one loader wave per workgroup -- wave 0 brings the workgroup's whole 4 KiB of a row in (four pieces +
// one halo dword, back to back), one s_barrier per row hands it to the four waves, each of which reads and stores its own 1 KiB.
// Separates "4 KiB per row issued by ONE wave" (the wide-strip probe's gain) from "4 KiB stored by one wave".  Output = input rows.
template <int D>
__global__ __launch_bounds__(stream_core::BLOCK) void k_loader_probe(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows,
                                                                    int groups_x, stream_core::BandPlan plan)
{
    using namespace stream_core;
    constexpr int SW = STRIP_PX * WAVES_PER_BLOCK;
    constexpr int SLOT = SW + 16;
    __shared__ __attribute__((aligned(16))) uint8_t ring[D][SLOT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int gx_i, gy_i;
    band_of_block(plan, blockIdx.x, groups_x, gx_i, gy_i);
    if (gy_i < 0) return;
    const int xw = gx_i * SW;                      // workgroup-uniform
    if (xw + SW > w) return;
    const int R = plan.R[0], y0 = gy_i * R;
    const int nrows = min(R, rows - y0) + 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)&ring[0][0];
    const uint32_t voff = (uint32_t)xw + (uint32_t)lane * 16u;
    uint32_t hoff = lane == 0 ? (uint32_t)xw - 4u : (uint32_t)(xw + SW);
    if ((lane == 0 && xw == 0) || (lane != 0 && xw + SW >= w)) hoff = 0;
    auto issue = [&](int i, int slot) {            // wave 0 only
        int yy = y0 - 1 + i; yy = yy < 0 ? 0 : (yy >= rows ? rows - 1 : yy);
        const uint8_t *rowb = uniform_ptr(src + (ptrdiff_t)yy * w);
        const uint32_t d = uniform_u32(lds0 + (uint32_t)slot * SLOT);
        const uint32_t dh = d + SW;
        uint32_t keep; uint64_t keepx;
#pragma unroll
        for (int p = 0; p < WAVES_PER_BLOCK; ++p)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff + (uint32_t)p * STRIP_PX), "s"(rowb), "s"(d + (uint32_t)p * STRIP_PX) : "memory");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, 3\n\tglobal_load_lds_dword %4, %2\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(keepx) : "s"(rowb), "s"(dh), "v"(hoff) : "memory");
    };
    // rows 0 .. D-2 are in flight when the loop starts; at step i the loader has issued rows up to i + D - 2
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < D - 1; ++i) issue(i, i);
    }
#pragma unroll 1
    for (int i0 = 0; i0 < nrows; i0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int i = i0 + j;
            if (i >= nrows) return;
            // row i has landed: younger than its five instructions are rows i+1 .. i+D-2 (5 each) and wave 0's own stores (uncounted: conservative)
            if (wave == 0) {
                if (i + D - 2 < nrows) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (D - 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's reads of the previous row's slot are done
            __builtin_amdgcn_s_barrier();
            if (wave == 0 && i + D - 1 < nrows) issue(i + D - 1, (j + D - 1) % D);     // into the slot every wave read in step i - 1
            const uint8_t *slotp = &ring[j][0];
            const uint2 hh = *reinterpret_cast<const uint2 *>(slotp + SW);
            if (i >= 2) {
                const int y = y0 + i - 2;
                uint4 q = *reinterpret_cast<const uint4 *>(slotp + wave * STRIP_PX + 16 * lane);
                q.x ^= hh.x & 0u; q.w ^= hh.y & 0u;
                store_strip_p<2>(dst + (size_t)y * w + xw + wave * STRIP_PX, lane, q);
            } else {
                asm volatile("" ::"v"(hh.x), "v"(hh.y));
            }
        }
    }
}

// ---- access-pattern probe: "band march" -- a wave owns RW rows and marches ALONG them, 1 KiB per step -------------------------
// The strip kernels above walk DOWN a 1 KiB-wide strip; their no-arithmetic skeleton runs as fast as the real kernel, i.e. the
// traversal, not the arithmetic, is the limit.  This probe moves the same bytes (RW + 2 input rows in, RW rows out per wave and
// step, LDS-DMA ring of two steps) with the other traversal.  Output = a copy of the middle rows (not a Sobel result).
template <int RW>
__global__ __launch_bounds__(stream_core::BLOCK) void k_march_probe(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int skew)
{
    using namespace stream_core;
    constexpr int NR = RW + 2;
    __shared__ __attribute__((aligned(16))) uint8_t ring[WAVES_PER_BLOCK][2][NR][STRIP_PX];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y0 = ((int)blockIdx.x * WAVES_PER_BLOCK + wave) * RW;
    if (y0 >= rows) return;
    const int nsteps = w / STRIP_PX;
    const int start = (int)(((unsigned)blockIdx.x * (unsigned)skew) % (unsigned)nsteps);
    const uint32_t lds0 = (uint32_t)(uintptr_t)&ring[wave][0][0][0];
    auto issue = [&](int step, int slot) {
        int sx = start + step; if (sx >= nsteps) sx -= nsteps;
        const uint32_t voff = (uint32_t)sx * STRIP_PX + (uint32_t)lane * 16u;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            int yy = y0 - 1 + r; yy = yy < 0 ? 0 : (yy >= rows ? rows - 1 : yy);
            const uint8_t *rowb = uniform_ptr(src + (ptrdiff_t)yy * w);
            const uint32_t d = uniform_u32(lds0 + (uint32_t)(slot * NR + r) * STRIP_PX);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(rowb), "s"(d) : "memory");
        }
    };
    issue(0, 0);
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
        const int slot = step & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (step + 1 < nsteps) {
            if (slot == 0) issue(step + 1, 1); else issue(step + 1, 0);
            // younger than this step's DMA: the next step's NR pieces + (from the second step on) the RW stores of the previous step
            if (step >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NR + RW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NR) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        int sx = start + step; if (sx >= nsteps) sx -= nsteps;
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const uint4 q = *reinterpret_cast<const uint4 *>(&ring[wave][slot][r][16 * lane]);
            if (r == 0 || r == NR - 1) { acc.x ^= q.x & 1u; continue; }       // halo rows: read, keep alive, not stored
            uint4 o = q;
            o.x ^= acc.x & 0u;
            const int y = y0 + r - 1;
            if (y < rows) store_strip(dst + (size_t)y * w + (size_t)sx * STRIP_PX, lane, o);
            else asm volatile("s_nop 0" ::: "memory");
        }
    }
}

// ---- the LDS-tile form (tile + one-pixel halo staged in LDS, sobel_tile.h): the first version measured, 0.134 ms against 0.095 -----------
using stream_core::row_readable;

__global__ __launch_bounds__(BLOCK) void k_sobel_tiled(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w,
                                                       int rows, int tiles_x, int zero_top, int zero_bot)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[LDS_BYTES];

    const unsigned t = blockIdx.x;
    const int x0 = (int)(t % (unsigned)tiles_x) * TW;
    const int y0 = (int)(t / (unsigned)tiles_x) * TH;

    // ---- stage (TH+2) x (TW+32) bytes: issue every global load first, then the LDS writes ----
    uint4 v[STAGE_ITERS];
#pragma unroll
    for (int it = 0; it < STAGE_ITERS; ++it) {
        const int idx = threadIdx.x + it * BLOCK;
        const int r = idx / CHUNKS, c = idx - r * CHUNKS;
        const int yy = y0 - 1 + r;
        const int gx = x0 - 16 + 16 * c;
        v[it] = make_uint4(0, 0, 0, 0);
        if (idx < LDS_ROWS * CHUNKS && gx >= 0 && gx < w && row_readable(yy, rows, zero_top, zero_bot))
            v[it] = *reinterpret_cast<const uint4 *>(src + (ptrdiff_t)yy * w + gx);
    }
#pragma unroll
    for (int it = 0; it < STAGE_ITERS; ++it) {
        const int idx = threadIdx.x + it * BLOCK;
        if (idx < LDS_ROWS * CHUNKS) *reinterpret_cast<uint4 *>(tile + idx * 16) = v[it];
    }
    __syncthreads();

    compute_tile(tile, dst, w, rows, x0, y0, zero_top, zero_bot);
}

#endif   // GH_ENABLE_PROBES

// One pixel per lane; rows above/below the band are read only when they exist.  x0 > 0: only the columns [x0, w) (the last w % 16 columns of a
// frame whose width is not a multiple of 16: the strip kernel covers the rest).  LUMA = 1 / 2: every byte read goes through lum(e, e, e)
// first (stream_core::plane_row_pairs explains why), here with the scalar gh_luma of gh_common.h itself (IEEE / FMA arithmetic).
template <int LUMA>
__global__ __launch_bounds__(stream_core::BLOCK) void k_sobel_px(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int rows, int x0,
                                                                int zero_top, int zero_bot)
{
    const int nc = w - x0;
    const size_t n = (size_t)nc * rows;
    const size_t stride = (size_t)gridDim.x * stream_core::BLOCK;
    auto px = [](const uint8_t *p) -> int {
        const uint32_t v = *p;
        return LUMA ? (int)gh_luma<LUMA ? LUMA - 1 : 0>(v, v, v) : (int)v;
    };
    for (size_t i = (size_t)blockIdx.x * stream_core::BLOCK + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / nc), x = x0 + (int)(i - (size_t)y * nc);
        uint32_t out = 0;
        const bool border = x == 0 || x == w - 1 || (y == 0 && zero_top) || (y == rows - 1 && zero_bot);
        if (!border) {
            const uint8_t *m = src + (ptrdiff_t)y * w + x;
            const uint8_t *u = m - w, *d = m + w;
            const int gx = -px(u - 1) - 2 * px(m - 1) - px(d - 1) + px(u + 1) + 2 * px(m + 1) + px(d + 1);
            const int gy = px(u - 1) + 2 * px(u) + px(u + 1) - px(d - 1) - 2 * px(d) - px(d + 1);
            out = gh_sobel_mag(gx * gx + gy * gy);
        }
        dst[(size_t)y * w + x] = (uint8_t)out;
    }
}

// every reachable gradient pair (gx, gy) in [-1020, 1020]^2 through the magnitude mapping of one core
//   variant 0: stream_core::mag_pack_rtz (packed-i16 core, MODE = RTZ)   variant 1: stream_core::mag_pack_rne (f32 core)
//   variant 2: gh_sobel_mag on m = gx^2 + gy^2 (LDS-tile and per-pixel kernels)
__global__ void k_selftest_mag(int variant, unsigned long long *mismatches)
{
    if (variant == 0) __builtin_amdgcn_s_setreg(0x0801, 3);
    const float half = stream_core::half_reg();
    constexpr int SPAN = 2041;
    const size_t n = (size_t)SPAN * SPAN;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int gx = (int)(i / SPAN) - 1020, gy = (int)(i % SPAN) - 1020;
        const uint32_t m = (uint32_t)(gx * gx + gy * gy);
        uint32_t lo = 0, hi = 2047;   // exact integer square root by bisection
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (mid * mid <= m) lo = mid; else hi = mid - 1;
        }
        const uint32_t want = lo > 255 ? 255 : lo;
        uint32_t got;
        if (variant == 0) got = stream_core::mag_pack_rtz((float)gx, (float)gy, half, 0, 0u) & 0xffu;
        else if (variant == 1) got = stream_core::mag_pack_rne((float)gx, (float)gy, half, 0, 0u) & 0xffu;
        else got = gh_sobel_mag((int)m);
        if (got != want) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// ---- the luma-on-load table: lum(v, v, v) for v = 0..255, row 0 in IEEE arithmetic, row 1 in FMA arithmetic -- filled ON THE DEVICE by
// gh_luma itself (the function every gray kernel uses, pinned by the compiled reference over all 2^24 triples), once per device
__device__ uint8_t g_lum_lut[2][256];

__global__ void k_fill_lum_lut()
{
    const uint32_t v = threadIdx.x & 255u;
    g_lum_lut[0][v] = (uint8_t)gh_luma<GH_MODE_IEEE>(v, v, v);
    g_lum_lut[1][v] = (uint8_t)gh_luma<GH_MODE_FMA>(v, v, v);
}

}  // namespace

extern "C" {

int gh_sobel_u8_rows_dev(const uint8_t *src, uint8_t *dst, int width, int rows, int top_is_border, int bottom_is_border,
                         gh_stream_t stream)
{
    return gh_sobel_rows_halo_wait(src, dst, width, rows, top_is_border, bottom_is_border, stream, nullptr, 0);
}

// Pass k >= 2 of the reference's gray + Sobel driver loop in ONE kernel (monolithic/src/main_with_sobel.c:51-62 applied to pass k - 1's
// replicated edge image): src holds the previous pass's edge plane e, every byte goes through lum(e, e, e) on its way into the Sobel
// window.  1 R + 1 W byte per pixel instead of the 2 + 2 of gh_gray_of_plane_dev followed by gh_sobel_u8_rows_dev.
int gh_sobel_of_gray_u8_rows_dev(const uint8_t *src, uint8_t *dst, int width, int rows, int mode, int top_is_border, int bottom_is_border,
                                 gh_stream_t stream)
{
    GH_REQUIRE(mode == GH_MODE_IEEE || mode == GH_MODE_FMA, "bad mode");
    return gh_sobel_rows_halo_wait(src, dst, width, rows, top_is_border, bottom_is_border, stream, nullptr, 1 + mode);
}

int gh_sobel_of_gray_u8_dev(const uint8_t *src, uint8_t *dst, int width, int height, int mode, gh_stream_t stream)
{
    return gh_sobel_of_gray_u8_rows_dev(src, dst, width, height, mode, 1, 1, stream);
}

}  // extern "C"

// Called by gh_init for the device it selects (shim.hip): fills the table there, on the null stream, and remembers its address.
static const uint8_t *g_lum_lut_addr[64] = {};
int gh_sobel_init_tables(int device)
{
    uint8_t *base = nullptr;
    GH_HIP_TRY(hipGetSymbolAddress((void **)&base, HIP_SYMBOL(g_lum_lut)));
    // On a stream of its own, NOT the null stream: a library that is the first to launch on the null stream of a process changes which
    // hardware queue that stream is given, and with it how the caller's stream and the communicator's side stream share the command
    // processor's pipes -- measured: the halo-flag step of a 2048-row shard 73 us per step instead of 33 (NOTES.md R4.4).
    hipStream_t ts = nullptr;
    GH_HIP_TRY(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking));
    k_fill_lum_lut<<<1, 256, 0, ts>>>();
    const hipError_t le = hipGetLastError();
    const hipError_t se = le == hipSuccess ? hipStreamSynchronize(ts) : le;
    (void)hipStreamDestroy(ts);
    if (se != hipSuccess) return gh_set_error(GH_ERR_HIP, "filling the luma table failed: %s", hipGetErrorString(se));
    if (device >= 0 && device < 64) g_lum_lut_addr[device] = base;
    return GH_OK;
}

// device address of the 256-byte table row for `luma` (1 IEEE, 2 FMA) on the current device
static int lum_lut_row(int luma, const uint8_t **row)
{
    int dev = 0;
    GH_HIP_TRY(hipGetDevice(&dev));
    GH_REQUIRE(dev >= 0 && dev < 64 && g_lum_lut_addr[dev], "the luma table of this device was never initialised (gh_init selects the device)");
    *row = g_lum_lut_addr[dev] + 256 * (luma - 1);
    return GH_OK;
}

// The launcher of every plane-Sobel entry point.
// hw != nullptr: the halo-flag form of the sharded step (gh_common.h) -- only the LDS-DMA ring kernels carry the wait; a frame that would
// run another form (or the two-launch split of k_sobel_dma2) is refused with GH_NO_FLAG_FORM before anything is launched.
// luma: 0 = the plane as it is, 1 / 2 = lum(e, e, e) applied on load in IEEE / FMA arithmetic (stream_core::plane_row_pairs).
// GH_LAB_INT knobs are compile-time constants in the release library (gh_common.h); what they select beyond their defaults exists in
// probe builds only.
int gh_sobel_rows_halo_wait(const uint8_t *src, uint8_t *dst, int width, int rows, int top_is_border, int bottom_is_border, gh_stream_t stream,
                            const GhHaloWait *hw, int luma)
{
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    GH_REQUIRE(luma >= 0 && luma <= 2, "luma must be 0 (none), 1 (IEEE) or 2 (FMA)");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(src && dst, "null pointer");
    GH_REQUIRE(src != dst, "src and dst must differ");
    if (int rc = gh_ensure_init()) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int zt = top_is_border ? 1 : 0, zb = bottom_is_border ? 1 : 0;
    const uint8_t *lut = nullptr;          // luma on load: the 256-byte table row of the arithmetic mode (device memory)
    if (luma)
        if (int rc = lum_lut_row(luma, &lut)) return rc;
    const bool aligned = width % 16 == 0 && gh_aligned16(src) && gh_aligned16(dst);
    bool use_tile = false;
    int core = 1;
#ifdef GH_ENABLE_PROBES
    static const bool tile_env = [] { const char *e = getenv("GH_SOBEL_IMPL"); return e && strcmp(e, "tile") == 0; }();
    static const int core_env = [] { const char *e = getenv("GH_SOBEL_CORE"); return (e && strcmp(e, "i16") == 0) ? 0 : 1; }();
    use_tile = tile_env && !luma;       // (the lab's other cores carry no luma-on-load form)
    core = luma ? 1 : core_env;
#endif
    const unsigned b = stream_core::BLOCK;
    const char *const sfx = luma ? " luma-on-load" : "";
    static thread_local char kname[96];
    auto note = [&](const char *name) { snprintf(kname, sizeof kname, "%s%s", name, sfx); gh_note_kernel(kname); };
    auto launch_px = [&](int x0) {
        size_t blocks = ((size_t)(width - x0) * rows + b - 1) / b;
        const size_t cap = x0 ? 4096 : (1u << 20);
        if (blocks > cap) blocks = cap;
        if (luma == 1) k_sobel_px<1><<<(unsigned)blocks, b, 0, s>>>(src, dst, width, rows, x0, zt, zb);
        else if (luma == 2) k_sobel_px<2><<<(unsigned)blocks, b, 0, s>>>(src, dst, width, rows, x0, zt, zb);
        else k_sobel_px<0><<<(unsigned)blocks, b, 0, s>>>(src, dst, width, rows, x0, zt, zb);
    };
    // Widths that are not a multiple of 16 (or unaligned planes): the register-staged strip kernel runs on the first
    // ws = width & ~15 columns with unaligned vector accesses (rows start at arbitrary bytes), the per-pixel kernel takes
    // the last width % 16 columns.  (All per-pixel: 0.69 ms on 16370 x 16384; this way: 0.125 ms.)
    const int ws = width & ~15;
    // (small unaligned frames stay on the single per-pixel launch: 1921 x 1081 takes 7 us that way, 11 us as strips + tail)
    // (luma on load exists for the shipped forms only: in probe builds it ignores the ring-depth and store-policy knobs)
    const int dma = luma ? 3 : GH_LAB_INT("GH_SOBEL_DMA", 3);      // LDS-DMA ring depth (probe builds: 0 = register-staged kernel, other depths, probes)
    // The LDS-DMA kernel takes ANY width >= 1024 and any base alignment: a DMA piece is 64 lanes x 16 B from consecutive global
    // addresses that need no alignment (only the LDS side is lane-linear), every strip is full (the last one is shifted left
    // to end at the frame edge), so a row pitch that is not a multiple of 16 changes nothing but the address arithmetic --
    // no per-pixel column tail.
    const int dma_unal = GH_LAB_INT("GH_SOBEL_DMA_UNALIGNED", 2);      // 2 aligned-down pieces (UNAL), 1 pieces from the strip's own address, 0 strips + tail
    const bool use_dma = !use_tile && dma && core == 1 && width >= stream_core::STRIP_PX && (aligned || dma_unal);
    if (hw && (!use_dma || dma != 3)) return GH_NO_FLAG_FORM;      // (dma != 3: the probe forms)
    const GhHaloWait hwv = hw ? *hw : GhHaloWait{};
    const bool strips = !use_tile && ws >= 16 &&
                        (aligned || use_dma || (GH_LAB_INT("GH_SOBEL_UNALIGNED", 1) && (long long)width * rows >= GH_LAB_INT("GH_SOBEL_UNALIGNED_MIN_PX", 1 << 21)));
    if (!strips) {
#ifdef GH_ENABLE_PROBES
        if (aligned && !luma) {           // GH_SOBEL_IMPL=tile
            const int tiles_x = (width + sobel_tile::TW - 1) / sobel_tile::TW, tiles_y = (rows + sobel_tile::TH - 1) / sobel_tile::TH;
            const long long nt = (long long)tiles_x * tiles_y;
            GH_REQUIRE(nt < (1ll << 31), "frame too large for one launch");
            gh_note_kernel("k_sobel_tiled");
            k_sobel_tiled<<<(unsigned)nt, sobel_tile::BLOCK, 0, s>>>(src, dst, width, rows, tiles_x, zt, zb);
            GH_LAUNCH_CHECK();
            return GH_OK;
        }
#endif
        note("k_sobel_px");
        launch_px(0);
        GH_LAUNCH_CHECK();
        return GH_OK;
    }
    const int r_env = GH_LAB_INT("GH_SOBEL_R", 0);
    const int span = stream_core::STRIP_PX * stream_core::WAVES_PER_BLOCK;
    const int groups_x = ((use_dma ? width : ws) + span - 1) / span;
    // strip height: taller strips re-read fewer halo rows from HBM (2 per strip: 12.5 % at R = 16, 3 % at R = 64)
    // but a shard must still give every CU a few workgroups.  Power-of-two heights only: band starts aligned to
    // 2^k rows measured 10-15 % faster than 30 / 34 / 48 / 60-row bands in the no-compute strip-copy probe.
    int R = r_env;
    // (the LDS-DMA kernel takes any band height -- a lab knob; the register-staged one is instantiated for 16 / 32 / 64)
    if (!(use_dma && R > 0) && R != 16 && R != 32 && R != 64)
        R = stream_core::sobel_band_rows(groups_x, rows, width, gh_num_cu(), use_dma, GH_LAB_INT("GH_BAND_ALIAS_RULE", 1) != 0, GH_LAB_INT("GH_SOBEL_REG", 1) != 0);
    const long long nb = (long long)groups_x * ((rows + R - 1) / R);
    GH_REQUIRE(nb < (1ll << 31), "frame too large for one launch");
    if (!use_dma) {
        // register-staged strips: frames narrower than 1024 columns
        const unsigned g = (unsigned)nb;
        note(ws < width ? "k_sobel_stream + k_sobel_px" : "k_sobel_stream");
#define GH_SOBEL_STREAM(R_)                                                                                                        \
        do {                                                                                                                         \
            if (luma) k_sobel_stream<R_, 2, 1, 1><<<g, b, 0, s>>>(src, dst, width, rows, groups_x, zt, zb, ws, lut);                   \
            else if (GH_LAB && core == 0) k_sobel_stream<R_, 2, GH_LAB ? 0 : 1, 0><<<g, b, 0, s>>>(src, dst, width, rows, groups_x, zt, zb, ws, nullptr);   \
            else k_sobel_stream<R_, 2, 1, 0><<<g, b, 0, s>>>(src, dst, width, rows, groups_x, zt, zb, ws, nullptr);                   \
        } while (0)
        switch (R) {
        case 64: GH_SOBEL_STREAM(64); break;
        case 32: GH_SOBEL_STREAM(32); break;
        default: GH_SOBEL_STREAM(16); break;
        }
#undef GH_SOBEL_STREAM
        GH_LAUNCH_CHECK();
        if (ws < width) {
            launch_px(ws);
            GH_LAUNCH_CHECK();
        }
        return GH_OK;
    }

    // ---- LDS-DMA ring kernels ---------------------------------------------------------------------------------------------------------
    // dispatch order 8 (pairs-per-XCD: XCD x owns the band pairs p = x mod 8): +1-2 % on 16384^2 in four interleaved
    // A/B runs, +14 % on 32768 x 8192, neutral on shards (tools/ab_fused.py --op sobel)
    const int order_knob = GH_LAB_INT("GH_SOBEL_ORDER", 8);
    stream_core::BandPlan plan = stream_core::make_band_plan(rows, R, -1, 0, -1, 0);
    plan.order = stream_core::sanitize_order(order_knob, plan.bands, groups_x, gh_num_xcd());
    const unsigned g = (unsigned)stream_core::plan_grid(plan, groups_x);
    const size_t pad = (size_t)GH_LAB_INT("GH_SOBEL_PAD_KB", 0) * 1024;   // occupancy probe: unused dynamic LDS
    // Output store policy: nontemporal.  sc1 / plain stores LOOK 8-12 % faster when the same 268 MB output plane is
    // rewritten launch after launch (74.7 % vs 66.7 %): the 256 MB Infinity Cache absorbs the writes and they never
    // reach HBM.  With the output rotating over six planes it is the other way round (sc1 62.6 %, plain 60.7 %,
    // nt 66.8 %; tools/ab_fused.py --out-ring 6), and nt performs the same in both settings.
    const int sp = luma ? 2 : GH_LAB_INT("GH_SOBEL_ST", 2);        // 2 nt; probe builds: 0 plain, 1 sc1
    // Two strips per wave (k_sobel_dma2, regular-band form): frames whose rows split into whole bands of 64 or 16 rows
    // and that are wide and tall enough to give every SIMD a wave.  Measured against the one-strip kernel (tools/sweep.py r2_ai,
    // sweep.py r2_aj): 16384^2 0.0952 vs 0.1000 ms, 32768 x 8192 0.0944 vs 0.0972, shards of 8192 / 4096 / 2048 rows 0.0498 vs
    // 0.0521 / 0.0262 vs 0.0309 / 0.0166 vs 0.0185, 8192^2 0.0270 vs 0.0288, 7680 x 4352 0.0230 vs 0.0246; narrow or small
    // frames lose (4096 x 16384: two of a workgroup's four waves idle; 4096^2, 16384 x 1024: too few waves) and stay where they were.
    // 64-row bands when they still give every CU a workgroup, else 16.  (GH_SOBEL_NS, probe builds: 1 never, 2 whenever the geometry
    // allows, 4 the four-strip kernel that lost its A/B: 59.7-65.0 % against 68.8 %, NOTES.md R3.4.)
    const int ns_knob = GH_LAB_INT("GH_SOBEL_NS", 0);
#ifdef GH_ENABLE_PROBES
    if (ns_knob == 4 && !luma && aligned && dma == 3 && sp == 2 && width >= 4 * stream_core::STRIP_PX) {
        const int d4 = GH_LAB_INT("GH_SOBEL_NS4_D", 3), core4 = GH_LAB_INT("GH_SOBEL_NS4_CORE", 1);
        const int R4 = r_env > 0 ? r_env : 64, chk4 = d4 == 3 ? 6 : d4;
        const int quads = (width + 4 * stream_core::STRIP_PX - 1) / (4 * stream_core::STRIP_PX), gx4 = (quads + stream_core::WAVES_PER_BLOCK - 1) / stream_core::WAVES_PER_BLOCK;
        if ((d4 == 2 || d4 == 3 || d4 == 6) && (R4 + 2) % chk4 == 0 && (R4 + 2) / chk4 >= 3 && rows >= R4) {
            if (hw && hw->pad) return GH_OK;
            stream_core::BandPlan p4 = stream_core::make_band_plan(rows, R4, -1, 0, -1, 0);
            p4.order = stream_core::sanitize_order(order_knob % 1000, p4.bands, gx4, gh_num_xcd());
            const unsigned g4 = (unsigned)stream_core::plan_grid(p4, gx4);
            gh_note_kernel(core4 ? "k_sobel_dma4 packed-f32 magnitude" : "k_sobel_dma4");
            if (d4 == 2) { if (core4) k_sobel_dma4<2, 1><<<g4, b, pad, s>>>(src, dst, width, rows, gx4, p4, zt, zb, hwv); else k_sobel_dma4<2, 0><<<g4, b, pad, s>>>(src, dst, width, rows, gx4, p4, zt, zb, hwv); }
            else if (d4 == 6) { if (core4) k_sobel_dma4<6, 1><<<g4, b, pad, s>>>(src, dst, width, rows, gx4, p4, zt, zb, hwv); else k_sobel_dma4<6, 0><<<g4, b, pad, s>>>(src, dst, width, rows, gx4, p4, zt, zb, hwv); }
            else { if (core4) k_sobel_dma4<3, 1><<<g4, b, pad, s>>>(src, dst, width, rows, gx4, p4, zt, zb, hwv); else k_sobel_dma4<3, 0><<<g4, b, pad, s>>>(src, dst, width, rows, gx4, p4, zt, zb, hwv); }
            GH_LAUNCH_CHECK();
            return GH_OK;
        }
    }
#endif
    // 0 aligned rows, 1 rows that start on a dword (pieces land dword-shifted), 2 any byte (+ v_alignbyte); -1: no aligned-down form asked for
    const int unal = aligned ? 0 : (dma_unal < 2 ? -1 : ((width % 4 == 0 && ((uintptr_t)src & 3) == 0) ? 1 : 2));
    if (unal >= 0 && dma == 3 && sp == 2 && ns_knob != 1 && width >= 2 * stream_core::STRIP_PX) {
        const int pairs = (width + 2 * stream_core::STRIP_PX - 1) / (2 * stream_core::STRIP_PX), gx2 = (pairs + stream_core::WAVES_PER_BLOCK - 1) / stream_core::WAVES_PER_BLOCK;
        // 64-row bands from one workgroup per CU up -- below two per CU only when they spread evenly over the CUs (16384 x 12288: 384
        // workgroups, 0.0863 ms against 0.0749 with 16-row bands)
        const long long nb64_2 = (long long)gx2 * (rows / 64);
        const bool even64 = nb64_2 >= 2 * gh_num_cu() || (nb64_2 >= gh_num_cu() && stream_core::wg_balance(nb64_2, gh_num_cu()) >= 0.97);
        int R2 = r_env > 0 ? r_env : (even64 ? 64 : 16);
        if (r_env <= 0 && R2 == 64 && GH_LAB_INT("GH_BAND_ALIAS_RULE", 1) && stream_core::prefer_short_bands(width, 64, 16)) R2 = 16;
        const bool band_ok = R2 >= 16 && (R2 + 2) % 6 == 0;                             // R2 + 2 input rows = whole chunks of 6
        // (widths that are multiples of 8192 only: that is where the row streams stay in step and the wider pieces pay -- 12288 x 8192 and
        //  7680 x 4352 run 0.0380 / 0.0198 ms on the one-strip regular-band form below and 0.0433 / 0.0230 with two strips; frames with
        //  unaligned rows only on request in probe builds: 16370 / 16380 x 16384 measure 0.1116 / 0.1108 ms this way and 0.1099 / 0.1084 with
        //  one strip per wave)
        const bool wanted = (GH_LAB && ns_knob == 2) || (unal == 0 && width % 8192 == 0 && (long long)pairs * (rows / R2) >= 4 * gh_num_cu());
        // rows that do not split into whole bands: the kernel shifts the last band up -- unless the whole bands fill the 512 resident
        // slots exactly; then the whole bands here and the ragged rest (fewer than R2 rows, real halo row above) in a second launch
        // (the second launch costs ~10 us: only where the first one is long -- 16384 x 16390 0.1076 ms this way, 0.1146 on the one-strip kernel;
        //  16384 x 8200 would be 0.068 against 0.059, so smaller frames fall through to the one-strip kernel instead)
        const bool overflow2 = band_ok && wanted && rows > R2 && stream_core::shifted_band_overflows(gx2, rows, R2, 2, gh_num_cu());
        if (overflow2 && (long long)width * rows >= (long long)GH_LAB_INT("GH_SOBEL_SPLIT_MIN_PX", 1 << 28)) {
            if (hw) return GH_NO_FLAG_FORM;
            const int main_rows = rows - rows % R2;
            if (int rc = gh_sobel_rows_halo_wait(src, dst, width, main_rows, top_is_border, 0, stream, nullptr, luma)) return rc;
            return gh_sobel_rows_halo_wait(src + (size_t)main_rows * width, dst + (size_t)main_rows * width, width, rows - main_rows, 0, bottom_is_border, stream, nullptr, luma);
        }
        if (band_ok && rows >= R2 && !overflow2 && wanted) {                            // (a ragged last band is shifted up inside the kernel)
            if (hw && hw->pad) return GH_OK;      // probe: k_sobel_dma2 will run
            stream_core::BandPlan p2 = stream_core::make_band_plan(rows, R2, -1, 0, -1, 0);
            p2.order = stream_core::sanitize_order(order_knob % 1000, p2.bands, gx2, gh_num_xcd());
            const unsigned g2 = (unsigned)stream_core::plan_grid(p2, gx2);
            bool launched = false;
#ifdef GH_ENABLE_PROBES
            const int nsd = luma ? 33 : GH_LAB_INT("GH_SOBEL_NS_D", 33);          // other ring depths / the generic band form (register counts: NOTES.md)
            launched = true;
            if (nsd == 2) k_sobel_dma2<2><<<g2, b, pad, s>>>(src, dst, width, rows, gx2, p2, zt, zb, hwv, nullptr);
            else if (nsd == 3) k_sobel_dma2<3><<<g2, b, pad, s>>>(src, dst, width, rows, gx2, p2, zt, zb, hwv, nullptr);
            else if (nsd == 4) k_sobel_dma2<4><<<g2, b, pad, s>>>(src, dst, width, rows, gx2, p2, zt, zb, hwv, nullptr);
            else if (nsd == 22 && (R2 + 2) / 2 >= 4) k_sobel_dma2<2, 1, 1><<<g2, b, pad, s>>>(src, dst, width, rows, gx2, p2, zt, zb, hwv, nullptr);     // regular-band form, other depths
            else if (nsd == 66 && (R2 + 2) / 6 >= 4) k_sobel_dma2<6, 1, 1><<<g2, b, pad, s>>>(src, dst, width, rows, gx2, p2, zt, zb, hwv, nullptr);
            else if (unal == 1 && !luma) k_sobel_dma2<3, 1, 1, 1><<<g2, b, pad, s>>>(src, dst, width, rows, gx2, p2, zt, zb, hwv, nullptr);
            else if (unal == 2 && !luma) k_sobel_dma2<3, 1, 1, 2><<<g2, b, pad, s>>>(src, dst, width, rows, gx2, p2, zt, zb, hwv, nullptr);
            else launched = false;
            if (launched) gh_note_kernel(R2 == 64 ? "k_sobel_dma2 regular R=64" : "k_sobel_dma2 regular R=16");
#endif
            if (!launched && unal == 0) {
                note(R2 == 64 ? "k_sobel_dma2 regular R=64" : "k_sobel_dma2 regular R=16");
                if (luma) k_sobel_dma2<3, 1, 1, 0, 1><<<g2, b, pad, s>>>(src, dst, width, rows, gx2, p2, zt, zb, hwv, lut);
                else k_sobel_dma2<3, 1, 1, 0, 0><<<g2, b, pad, s>>>(src, dst, width, rows, gx2, p2, zt, zb, hwv, nullptr);
                launched = true;
            }
            if (launched) {
                GH_LAUNCH_CHECK();
                return GH_OK;
            }
        }
    }
    if (hw && hw->pad) return GH_OK;              // probe: k_sobel_dma will run
    const int u1 = unal < 0 ? 0 : unal;           // (dma_unal == 1, probe builds: pieces from the strip's own unaligned address, UNAL = 0 code)
    // Regular-band (branch-free) form of the one-strip kernel, 128-134 VGPRs: every band has R + 2 input rows in whole chunks of 6.
    // With 16-row bands -- the mid-size and small frames, where a wave's row step is latency-bound and the generic form's scalar
    // branches count -- it is 7-20 % faster (7680 x 4352 0.0198 vs 0.0249 ms, 4096^2 0.0115 vs 0.0129, 16384 x 1024 0.0117 vs 0.0134,
    // 3840 x 2160 0.0108 vs 0.0119, 1920 x 1088 0.0100 vs 0.0107); with 64-row bands on large frames it only lets a third workgroup
    // onto each CU and loses 1.5-5 % (16384^2, 32768 x 8192), so those keep the generic form.
    const bool reg1 = GH_LAB_INT("GH_SOBEL_REG", 1) && dma == 3 && sp == 2 && R == 16 && rows >= R && plan.n[1] == 0 && plan.order < 1000 &&
                      !stream_core::shifted_band_overflows(groups_x, rows, R, 3, gh_num_cu());
    auto launch_dma = [&](auto unal_c, auto reg_c) {
        constexpr int U = decltype(unal_c)::value, RG = decltype(reg_c)::value;
        if (luma) k_sobel_dma<3, 0, 2, 1, U, RG, 1><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, lut);
        else k_sobel_dma<3, 0, 2, 1, U, RG, 0><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);
    };
    using std::integral_constant;
    if (dma == 3 && sp == 2) {                    // what ships: ring depth 3, nontemporal stores
        note(reg1 ? (u1 ? "k_sobel_dma regular R=16 unaligned" : "k_sobel_dma regular R=16") : (u1 ? "k_sobel_dma unaligned" : "k_sobel_dma"));
        if (reg1) {
            if (u1 == 0) launch_dma(integral_constant<int, 0>{}, integral_constant<int, 1>{});
            else if (u1 == 1) launch_dma(integral_constant<int, 1>{}, integral_constant<int, 1>{});
            else launch_dma(integral_constant<int, 2>{}, integral_constant<int, 1>{});
        } else {
            if (u1 == 0) launch_dma(integral_constant<int, 0>{}, integral_constant<int, 0>{});
            else if (u1 == 1) launch_dma(integral_constant<int, 1>{}, integral_constant<int, 0>{});
            else launch_dma(integral_constant<int, 2>{}, integral_constant<int, 0>{});
        }
        GH_LAUNCH_CHECK();
        return GH_OK;
    }
#ifdef GH_ENABLE_PROBES
    GH_REQUIRE(!luma, "luma on load: ring depth 3 and nontemporal stores only");
    gh_note_kernel(u1 ? "k_sobel_dma unaligned" : "k_sobel_dma");
    if (u1 && sp == 0) {
        if (u1 == 1) k_sobel_dma<3, 0, 0, 1, 1><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);
        else k_sobel_dma<3, 0, 0, 1, 2><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);
    }
    else if (dma == 3 && sp == 1) k_sobel_dma<3, 0, 1><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);
    else if (dma == 3 && sp == 0) k_sobel_dma<3, 0, 0><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);
    else if (dma >= 100 && dma < 200 && width % stream_core::STRIP_PX == 0) {     // band-march probe: dma = 100 + RW, GH_SOBEL_SKEW
        const int rw = dma - 100, skew = GH_LAB_INT("GH_SOBEL_SKEW", 3);
        const unsigned gm = (unsigned)((rows + 4 * rw - 1) / (4 * rw));
        if (rw == 4) k_march_probe<4><<<gm, b, 0, s>>>(src, dst, width, rows, skew);
        else if (rw == 6) k_march_probe<6><<<gm, b, 0, s>>>(src, dst, width, rows, skew);
        else if (rw == 8) k_march_probe<8><<<gm, b, 0, s>>>(src, dst, width, rows, skew);
        else if (rw == 14) k_march_probe<14><<<gm, b, 0, s>>>(src, dst, width, rows, skew);
        else k_march_probe<2><<<gm, b, 0, s>>>(src, dst, width, rows, skew);
    }
    else if (dma >= 200 && dma < 1800 && (dma % 100 >= 2 && dma % 100 <= 4) && width % (dma >= 1700 ? 4096 : 1024 * (dma / 100) * (dma >= 1600 ? 1 : (dma >= 800 ? 2 : 4))) == 0 && rows % R == 0) {
        // wide-strip probe: dma = 100 * KiB per wave and row + ring depth (8 and 16 KiB: two / one wave per workgroup)
        const int pieces = dma >= 1700 ? 1 : dma / 100, wpb = pieces >= 16 ? 1 : (pieces >= 8 ? 2 : 4), gxw = width / (1024 * pieces * wpb);
        stream_core::BandPlan pw = stream_core::make_band_plan(rows, R, -1, 0, -1, 0);
        pw.order = stream_core::sanitize_order(order_knob, pw.bands, gxw, gh_num_xcd());
        const unsigned gw = (unsigned)stream_core::plan_grid(pw, gxw);
        if (dma == 203) k_wide_strip_probe<2, 3><<<gw, 256, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 202) k_wide_strip_probe<2, 2><<<gw, 256, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 403) k_wide_strip_probe<4, 3><<<gw, 256, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 402) k_wide_strip_probe<4, 2><<<gw, 256, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 803) k_wide_strip_probe<8, 3, 2><<<gw, 128, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 802) k_wide_strip_probe<8, 2, 2><<<gw, 128, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 1703) k_loader_probe<3><<<gw, 256, 0, s>>>(src, dst, width, rows, gxw, pw);      // 17xx: the loader-wave probe (4 KiB per workgroup and row)
        else if (dma == 1704) k_loader_probe<4><<<gw, 256, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 1603) k_wide_strip_probe<16, 3, 1><<<gw, 64, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 1602) k_wide_strip_probe<16, 2, 1><<<gw, 64, 0, s>>>(src, dst, width, rows, gxw, pw);
        else GH_REQUIRE(false, "wide-strip probe: unknown size / depth");
    }
    else if (dma >= 30000 && dma < 40000 && width % 12288 == 0 && rows % R == 0) {
        // 3:1 wide-strip probe on a plane of 3 x 16384 bytes per row: dma = 30000 + 100 * KiB read per wave and row + ring depth
        const int pieces = (dma - 30000) / 100, depth = dma % 100, wpb = pieces >= 12 ? (depth == 3 ? 1 : 2) : (pieces >= 6 ? (depth == 3 ? 2 : 4) : 4);
        const int gxw = width / (1024 * pieces * wpb);
        stream_core::BandPlan pw = stream_core::make_band_plan(rows, R, -1, 0, -1, 0);
        pw.order = stream_core::sanitize_order(GH_LAB_INT("GH_SOBEL_ORDER", 6), pw.bands, gxw, gh_num_xcd());
        const unsigned gw = (unsigned)stream_core::plan_grid(pw, gxw);
        if (dma == 30302) k_wide_strip_probe<3, 2, 4, 3><<<gw, 256, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 30303) k_wide_strip_probe<3, 3, 4, 3><<<gw, 256, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 30602) k_wide_strip_probe<6, 2, 4, 3><<<gw, 256, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 30603) k_wide_strip_probe<6, 3, 2, 3><<<gw, 128, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 31202) k_wide_strip_probe<12, 2, 2, 3><<<gw, 128, 0, s>>>(src, dst, width, rows, gxw, pw);
        else if (dma == 31203) k_wide_strip_probe<12, 3, 1, 3><<<gw, 64, 0, s>>>(src, dst, width, rows, gxw, pw);
        else GH_REQUIRE(false, "3:1 wide-strip probe: unknown size / depth");
    }
    else if (dma == 13) k_sobel_dma<3, 1><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);        // timing probe: no sqrt (wrong results)
    else if (dma == 33) k_sobel_dma<3, 2><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);        // timing probe: no arithmetic (wrong results)
    else if (dma == 63) k_sobel_dma<3, 5><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);        // the same without the halo dword requests (what do they cost?)
    else if (dma == 43 && width % 4096 == 0 && rows % 64 == 0) k_sobel_dma<3, 3><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);   // probe: row-synchronous workgroup
    else if (dma == 53 && width % 4096 == 0 && rows % 64 == 0) k_sobel_dma<3, 4><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);   // the same without arithmetic
    else if (dma == 23) k_sobel_dma<3, 0, 2, 0><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);  // probe: default-policy DMA loads
    else if (dma == 22) k_sobel_dma<2, 0, 2, 0><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);
    else if (dma <= 2) k_sobel_dma<2><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);
    else if (dma == 4) k_sobel_dma<4><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);
    else if (dma <= 6) k_sobel_dma<6><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);
    else k_sobel_dma<8><<<g, b, pad, s>>>(src, dst, width, rows, groups_x, plan, zt, zb, hwv, nullptr);
    GH_LAUNCH_CHECK();
    return GH_OK;
#else
    return gh_set_error(GH_ERR_ARG, "unreachable: the release library has one ring depth and one store policy");
#endif
}

extern "C" {

int gh_sobel_u8_dev(const uint8_t *src, uint8_t *dst, int width, int height, gh_stream_t stream)
{
    return gh_sobel_u8_rows_dev(src, dst, width, height, 1, 1, stream);
}

int gh_selftest_sobel_mag(int variant, uint64_t *mismatches)
{
    GH_REQUIRE(mismatches, "null pointer");
    GH_REQUIRE(variant >= 0 && variant <= 2, "variant must be 0, 1 or 2");
    if (int rc = gh_ensure_init()) return rc;
    unsigned long long *d = nullptr;
    GH_HIP_TRY(hipMalloc(&d, sizeof(*d)));
    GH_HIP_TRY(hipMemset(d, 0, sizeof(*d)));
    k_selftest_mag<<<1024, 256>>>(variant, d);
    GH_LAUNCH_CHECK();
    unsigned long long h = 0;
    GH_HIP_TRY(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    GH_HIP_TRY(hipFree(d));
    *mismatches = h;
    return GH_OK;
}

}  // extern "C"
