// jpeg_decode.cpp -- JPEG decoder for the drop-in CLIs (host side): Huffman-coded baseline, extended-sequential and
// progressive DCT frames (ITU-T T.81 SOF0 / SOF1 / SOF2), 8-bit samples, 1 / 3 / 4 components, restart intervals.
//
// The reference drivers decode their input with the third-party stb_image.h they vendor (monolithic/src/main.c:21 ->
// stbi_load(path, &w, &h, &channels, 0)), and the reference's golden images/test_gray.png was made from THOSE pixels.  A
// JPEG stream fixes the quantised DCT coefficients, not the pixels: the inverse DCT, the chroma upsampling filter and the
// colour conversion are the decoder's choice, and stb's differ from libjpeg's by up to +-3 per channel on images/test.jpg.
// So this file has two kinds of code:
//
//  * Everything that the STANDARD defines -- marker and segment syntax (T.81 Annex B), Huffman table generation and
//    decoding (Annex C, F.2.2.3), the sequential block procedure (F.2.2), the progressive procedures (G.1.2, Figure G.7),
//    restart handling (F.2.2.4 / E.2.4) -- is written here from the standard: own bit reader, canonical decoding by
//    MINCODE / MAXCODE / VALPTR with an 8-bit first-level lookup, own structures and error texts.
//
//  * The three stages where pixel parity with the reference REQUIRES stb's arithmetic are restatements of it, cited:
//      idct8x8()      fixed-point "islow" inverse DCT: 12-bit constants, column pass rounded to 2 extra bits (+512 >> 10),
//                     row pass +65536 (+128 level shift) >> 17, clamp                 monolithic/include/stb_image.h:2425-2520
//      upsample_row() "fancy" chroma upsampling: 3:1 vertical / horizontal taps, 9:3:3:1 for 2x2 (two 3:1 passes, +8 >> 4),
//                     edge samples not interpolated, nearest-neighbour for other factors; which two source rows feed an
//                     output row                                                       stb_image.h:3464-3526, 3646-3655, 3835-3880
//      ycc_to_rgb()   20-bit fixed-point YCbCr -> RGB, the Cb term of G truncated to a multiple of 2^16
//                                                                                       stb_image.h:3658-3685
//      + the colour-space decisions (RGB-tagged components, Adobe APP14 transform, CMYK / YCCK) stb_image.h:3812-3815, 3895-3925
//    (stb_image v2.30 is public-domain / MIT software by Sean Barrett and contributors.)
//
// tests/test_cli_image_io.py compares the output pixel for pixel with the reference's own decoder, compiled in place in
// the build container, on the reference's JPEGs and on a set of generated streams (baseline / progressive, 4:4:4 / 4:2:2 /
// 4:2:0, greyscale, restart markers, optimised tables, CMYK).
#include "image_io.h"

#include <cstdlib>
#include <cstring>
#include <memory>

namespace {

// zig-zag position -> index in the 8x8 block (T.81 Figure A.6)
const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

enum : uint8_t {
    M_SOF0 = 0xC0, M_SOF1 = 0xC1, M_SOF2 = 0xC2, M_DHT = 0xC4, M_RST0 = 0xD0, M_RST7 = 0xD7, M_SOI = 0xD8, M_EOI = 0xD9,
    M_SOS = 0xDA, M_DQT = 0xDB, M_DNL = 0xDC, M_DRI = 0xDD, M_APP0 = 0xE0, M_APP14 = 0xEE, M_APP15 = 0xEF, M_COM = 0xFE
};

// ---- entropy-coded segment reader (T.81 B.1.1.5, F.2.2.5) -------------------------------------------------------------
// Bits are taken MSB first.  Inside entropy-coded data 0xFF is followed by a stuffed 0x00; 0xFF followed by anything else
// (after optional 0xFF fill bytes) is a marker: it ends the segment, is remembered, and the reader hands out zero bits
// from there on (what a decoder does at that point is its choice; zeros decode to short, harmless codes).
class EntropyReader {
public:
    void open(const uint8_t *at, const uint8_t *end) { p_ = at; end_ = end; acc_ = 0; have_ = 0; marker_ = 0; }
    const uint8_t *position() const { return p_; }
    uint8_t pending_marker() const { return marker_; }
    void clear_marker() { marker_ = 0; }
    void byte_align() { acc_ = 0; have_ = 0; }

    uint32_t peek16()
    {
        while (have_ < 16) pull();
        return (uint32_t)(acc_ >> 48);
    }
    void drop(int n) { acc_ <<= n; have_ -= n; }
    uint32_t take(int n)               // 1 <= n <= 16
    {
        while (have_ < n) pull();
        const uint32_t v = (uint32_t)(acc_ >> (64 - n));
        drop(n);
        return v;
    }
    int bit() { return (int)take(1); }
    // RECEIVE + EXTEND (T.81 F.2.2.1, Figure F.12): an n-bit two's-complement-like magnitude category value
    int take_signed(int n)
    {
        if (n == 0) return 0;
        const int v = (int)take(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }

private:
    void pull()
    {
        uint32_t b = 0;
        if (!marker_ && p_ < end_) {
            b = *p_++;
            if (b == 0xFF) {
                const uint8_t *q = p_;
                while (q < end_ && *q == 0xFF) ++q;           // fill bytes
                const uint8_t follow = q < end_ ? *q++ : (uint8_t)M_EOI;   // a stream that stops inside a marker: treat as end
                p_ = q;
                if (follow != 0) { marker_ = follow; b = 0; }
            }
        }
        acc_ |= (uint64_t)b << (56 - have_);
        have_ += 8;
    }
    const uint8_t *p_ = nullptr, *end_ = nullptr;
    uint64_t acc_ = 0;
    int have_ = 0;
    uint8_t marker_ = 0;
};

// ---- Huffman tables (T.81 Annex C generation, F.2.2.3 decoding) --------------------------------------------------------
struct HuffmanTable {
    bool defined = false;
    uint8_t symbol[256];
    int nsym = 0;
    int32_t mincode[17], maxcode[17], valptr[17];        // per code length 1..16; maxcode = -1: no code of that length
    uint16_t quick[256];                                  // first 8 bits -> (length << 8) | symbol, 0 = longer than 8 bits

    // BITS (counts per length) + HUFFVAL.  Rejects over-subscribed length sets (a code would not fit its length).
    bool define(const uint8_t counts[16], const uint8_t *values, int nvalues)
    {
        int total = 0;
        for (int l = 0; l < 16; ++l) total += counts[l];
        if (total != nvalues || total > 256) return false;
        memcpy(symbol, values, (size_t)total);
        nsym = total;
        memset(quick, 0, sizeof quick);
        int32_t code = 0;
        int k = 0;
        for (int len = 1; len <= 16; ++len) {
            const int n = counts[len - 1];
            if (n == 0) { mincode[len] = 0; maxcode[len] = -1; valptr[len] = 0; }
            else {
                if (code + n > (1 << len)) return false;
                valptr[len] = k;
                mincode[len] = code;
                maxcode[len] = code + n - 1;
                if (len <= 8)
                    for (int c = code; c < code + n; ++c) {
                        const int first = c << (8 - len), span = 1 << (8 - len);
                        for (int f = first; f < first + span; ++f) quick[f] = (uint16_t)((len << 8) | symbol[k + (c - code)]);
                    }
                k += n;
                code += n;
            }
            code <<= 1;
        }
        defined = true;
        return true;
    }

    // DECODE (Figure F.16): returns the symbol, or -1 when the next 16 bits are no code of this table
    int decode(EntropyReader &in) const
    {
        const uint32_t w = in.peek16();
        const uint16_t q = quick[w >> 8];
        if (q) { in.drop(q >> 8); return q & 0xFF; }
        for (int len = 9; len <= 16; ++len) {
            const int32_t c = (int32_t)(w >> (16 - len));
            if (c <= maxcode[len] && c >= mincode[len] && maxcode[len] >= 0) {
                in.drop(len);
                return symbol[valptr[len] + (c - mincode[len])];
            }
        }
        return -1;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;        // frame header
    int td = 0, ta = 0;                      // tables selected by the current scan
    int px_w = 0, px_h = 0;                  // samples that carry image content: ceil(X * h / hmax), ceil(Y * v / vmax)
    int stride = 0, rows = 0;                // plane size, padded to whole MCUs
    int blocks_w = 0, blocks_h = 0;          // stride / 8, rows / 8
    int dc_pred = 0;
    std::vector<uint8_t> plane;              // reconstructed samples
    std::vector<int16_t> coef;               // progressive only: blocks_w * blocks_h * 64 coefficients, block-natural order
};

class JpegDecoder {
public:
    JpegDecoder(const uint8_t *data, size_t size, std::string &err) : p_(data), end_(data + size), err_(err) {}
    bool run(GhImage &out);

private:
    bool fail(const char *what) { if (err_.empty()) err_ = std::string("jpeg: ") + what; return false; }
    // segment reading
    bool need(size_t n) const { return (size_t)(end_ - p_) >= n; }
    int u8() { return p_ < end_ ? *p_++ : 0; }
    int u16() { const int hi = u8(); return (hi << 8) | u8(); }
    bool next_marker(uint8_t &m);
    bool read_tables_or_misc(uint8_t m);
    bool read_frame_header(uint8_t m);
    bool read_scan_header();
    bool decode_scan();
    bool restart(int &countdown);
    // block procedures
    bool block_sequential(Component &c, int16_t *blk);
    bool block_dc_progressive(Component &c, int16_t *blk);
    bool block_ac_first(Component &c, int16_t *blk);
    bool block_ac_refine(Component &c, int16_t *blk);
    bool code_block(Component &c, int bx, int by);
    void reconstruct_progressive();
    bool produce(GhImage &out);

    const uint8_t *p_, *end_;
    std::string &err_;
    EntropyReader bits_;
    uint16_t quant_[4][64];                  // zig-zag order, as transmitted
    bool quant_defined_[4] = {false, false, false, false};
    HuffmanTable dc_[4], ac_[4];
    int width_ = 0, height_ = 0, ncomp_ = 0, hmax_ = 1, vmax_ = 1, mcus_x_ = 0, mcus_y_ = 0;
    bool progressive_ = false, have_frame_ = false, jfif_ = false;
    int adobe_transform_ = -1, rgb_tagged_ = 0;
    int restart_interval_ = 0;
    Component comp_[4];
    // current scan
    int scan_n_ = 0, scan_comp_[4] = {0, 0, 0, 0}, ss_ = 0, se_ = 63, ah_ = 0, al_ = 0;
    int eobrun_ = 0;
};

// ---- inverse DCT: stb_image.h:2425-2520 (STBI__IDCT_1D / stbi__idct_block), restated ------------------------------------
inline int fx(double v) { return (int)(v * 4096 + 0.5); }       // stbi__f2f: 12-bit fixed point constants

// All sums and products wrap modulo 2^32 (unsigned arithmetic, reinterpreted as signed for the shifts): valid streams stay
// far inside 32 bits, and for hostile ones (16-bit quantisers times large coefficients) this is what the reference's plain
// `int` code does on a two's-complement machine, without the undefined behaviour.
struct Idct1D { uint32_t x0, x1, x2, x3, t0, t1, t2, t3; };

// one 8-point pass: even part into x0..x3, odd part into t0..t3 (results are x_k +- t_{3-k})
inline Idct1D idct_pass(int i0, int i1, int i2, int i3, int i4, int i5, int i6, int i7)
{
    static const uint32_t c0541 = (uint32_t)fx(0.5411961f), c1847 = (uint32_t)fx(-1.847759065f), c0765 = (uint32_t)fx(0.765366865f),
                          c1175 = (uint32_t)fx(1.175875602f), c0298 = (uint32_t)fx(0.298631336f), c2053 = (uint32_t)fx(2.053119869f),
                          c3072 = (uint32_t)fx(3.072711026f), c1501 = (uint32_t)fx(1.501321110f), c0899 = (uint32_t)fx(-0.899976223f),
                          c2562 = (uint32_t)fx(-2.562915447f), c1961 = (uint32_t)fx(-1.961570560f), c0390 = (uint32_t)fx(-0.390180644f);
    const uint32_t s0 = (uint32_t)i0, s1 = (uint32_t)i1, s2 = (uint32_t)i2, s3 = (uint32_t)i3, s4 = (uint32_t)i4, s5 = (uint32_t)i5,
                   s6 = (uint32_t)i6, s7 = (uint32_t)i7;
    Idct1D r;
    const uint32_t e = (s2 + s6) * c0541;
    const uint32_t even_lo = e + s6 * c1847, even_hi = e + s2 * c0765;
    const uint32_t sum = (s0 + s4) * 4096u, dif = (s0 - s4) * 4096u;
    r.x0 = sum + even_hi; r.x3 = sum - even_hi; r.x1 = dif + even_lo; r.x2 = dif - even_lo;
    const uint32_t a = s7 + s3, b = s5 + s1, c = s7 + s1, d = s5 + s3;
    const uint32_t z = (a + b) * c1175;
    const uint32_t pc = z + c * c0899, pd = z + d * c2562, pa = a * c1961, pb = b * c0390;
    r.t0 = s7 * c0298 + pc + pa;
    r.t1 = s5 * c2053 + pd + pb;
    r.t2 = s3 * c3072 + pd + pa;
    r.t3 = s1 * c1501 + pc + pb;
    return r;
}

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline int sar(uint32_t v, int n) { return (int)(int32_t)v >> n; }        // arithmetic shift of the wrapped value

void idct8x8(uint8_t *out, int stride, const int16_t in[64])
{
    int mid[64];
    for (int col = 0; col < 8; ++col) {                  // columns: keep 2 extra bits (>> 10 of the 12)
        const int16_t *s = in + col;
        Idct1D r = idct_pass(s[0], s[8], s[16], s[24], s[32], s[40], s[48], s[56]);
        r.x0 += 512; r.x1 += 512; r.x2 += 512; r.x3 += 512;
        int *m = mid + col;
        m[0] = sar(r.x0 + r.t3, 10);  m[56] = sar(r.x0 - r.t3, 10);
        m[8] = sar(r.x1 + r.t2, 10);  m[48] = sar(r.x1 - r.t2, 10);
        m[16] = sar(r.x2 + r.t1, 10); m[40] = sar(r.x2 - r.t1, 10);
        m[24] = sar(r.x3 + r.t0, 10); m[32] = sar(r.x3 - r.t0, 10);
    }
    for (int row = 0; row < 8; ++row) {                  // rows: 12 + 2 + 3 bits to remove, rounding and the +128 level shift folded in
        const int *m = mid + 8 * row;
        Idct1D r = idct_pass(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
        const uint32_t bias = 65536u + (128u << 17);
        r.x0 += bias; r.x1 += bias; r.x2 += bias; r.x3 += bias;
        uint8_t *o = out + (size_t)row * stride;
        o[0] = clamp8(sar(r.x0 + r.t3, 17)); o[7] = clamp8(sar(r.x0 - r.t3, 17));
        o[1] = clamp8(sar(r.x1 + r.t2, 17)); o[6] = clamp8(sar(r.x1 - r.t2, 17));
        o[2] = clamp8(sar(r.x2 + r.t1, 17)); o[5] = clamp8(sar(r.x2 - r.t1, 17));
        o[3] = clamp8(sar(r.x3 + r.t0, 17)); o[4] = clamp8(sar(r.x3 - r.t0, 17));
    }
}

// ---- chroma upsampling: stb_image.h:3464-3526 (resample_row_*), restated ------------------------------------------------
// near / far: the two source rows this output row lies between (near = the closer one); w = source samples; returns the
// row to read (the source row itself when nothing is to be done)
const uint8_t *upsample_row(uint8_t *dst, const uint8_t *near, const uint8_t *far, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; ++i) dst[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
        return dst;
    }
    if (hs == 2 && vs == 1) {
        if (w == 1) { dst[0] = dst[1] = near[0]; return dst; }
        dst[0] = near[0];
        dst[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
        for (int i = 1; i < w - 1; ++i) {
            const int n = 3 * near[i] + 2;
            dst[2 * i] = (uint8_t)((n + near[i - 1]) >> 2);
            dst[2 * i + 1] = (uint8_t)((n + near[i + 1]) >> 2);
        }
        dst[2 * (w - 1)] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
        dst[2 * (w - 1) + 1] = near[w - 1];
        return dst;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { dst[0] = dst[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return dst; }
        int cur = 3 * near[0] + far[0];                 // vertical 3:1 first, at 4x scale ...
        dst[0] = (uint8_t)((cur + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            const int prev = cur;
            cur = 3 * near[i] + far[i];
            dst[2 * i - 1] = (uint8_t)((3 * prev + cur + 8) >> 4);      // ... then horizontal 3:1, at 16x
            dst[2 * i] = (uint8_t)((3 * cur + prev + 8) >> 4);
        }
        dst[2 * w - 1] = (uint8_t)((cur + 2) >> 2);
        return dst;
    }
    for (int i = 0; i < w; ++i)                           // any other factor: sample replication, nearest source row
        for (int j = 0; j < hs; ++j) dst[i * hs + j] = near[i];
    return dst;
}

// ---- colour conversion: stb_image.h:3658-3685 (stbi__YCbCr_to_RGB_row), restated -----------------------------------------
inline int fx20(double v) { return ((int)(v * 4096.0f + 0.5f)) << 8; }      // stbi__float2fixed

void ycc_to_rgb(uint8_t *out, const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int count, int step)
{
    static const int kRCr = fx20(1.40200f), kGCr = fx20(0.71414f), kGCb = fx20(0.34414f), kBCb = fx20(1.77200f);
    for (int i = 0; i < count; ++i) {
        const int yy = (y[i] << 20) + (1 << 19);
        const int dr = cr[i] - 128, db = cb[i] - 128;
        int r = yy + dr * kRCr;
        int g = yy + dr * -kGCr + (int)((uint32_t)(db * -kGCb) & 0xffff0000u);
        int b = yy + db * kBCb;
        out[0] = clamp8(r >> 20);
        out[1] = clamp8(g >> 20);
        out[2] = clamp8(b >> 20);
        out += step;
    }
}

inline uint8_t mul255(uint8_t a, uint8_t b)              // rounded a * b / 255 (stb_image.h:3790-3795)
{
    const unsigned t = (unsigned)a * b + 128;
    return (uint8_t)((t + (t >> 8)) >> 8);
}

// ---- markers and segments (T.81 Annex B) ----------------------------------------------------------------------------------
bool JpegDecoder::next_marker(uint8_t &m)
{
    // markers may be preceded by any number of 0xFF fill bytes; bytes that are not part of a marker are skipped
    while (p_ < end_) {
        if (*p_++ != 0xFF) continue;
        while (p_ < end_ && *p_ == 0xFF) ++p_;
        if (p_ >= end_) break;
        const uint8_t c = *p_++;
        if (c != 0) { m = c; return true; }
    }
    return false;
}

bool JpegDecoder::read_tables_or_misc(uint8_t m)
{
    if (!need(2)) return fail("segment length missing");
    int len = u16();
    if (len < 2 || !need((size_t)len - 2)) return fail("segment longer than the file");
    const uint8_t *seg_end = p_ + (len - 2);
    switch (m) {
    case M_DQT:
        while (p_ < seg_end) {
            const int pq_tq = u8(), precision = pq_tq >> 4, id = pq_tq & 15;
            if (precision > 1) return fail("quantization table precision must be 8 or 16 bits");
            if (id > 3) return fail("quantization table id out of range");
            if ((size_t)(seg_end - p_) < (size_t)(precision ? 128 : 64)) return fail("quantization table cut short");
            for (int k = 0; k < 64; ++k) quant_[id][k] = (uint16_t)(precision ? u16() : u8());
            quant_defined_[id] = true;
        }
        break;
    case M_DHT:
        while (p_ < seg_end) {
            if (seg_end - p_ < 17) return fail("Huffman table header cut short");
            const int tc_th = u8(), cls = tc_th >> 4, id = tc_th & 15;
            if (cls > 1 || id > 3) return fail("Huffman table class / id out of range");
            uint8_t counts[16];
            int total = 0;
            for (int l = 0; l < 16; ++l) { counts[l] = (uint8_t)u8(); total += counts[l]; }
            if (total > 256 || seg_end - p_ < total) return fail("Huffman table values cut short");
            HuffmanTable &t = cls ? ac_[id] : dc_[id];
            if (!t.define(counts, p_, total)) return fail("Huffman code lengths do not form a prefix code");
            p_ += total;
        }
        break;
    case M_DRI:
        if (len != 4) return fail("DRI segment must hold one 16-bit interval");
        restart_interval_ = u16();
        break;
    case M_APP0:
        if (len >= 7 && memcmp(p_, "JFIF\0", 5) == 0) jfif_ = true;
        break;
    case M_APP14:
        // "Adobe" + version(2) flags0(2) flags1(2) transform(1)
        if (len >= 14 && memcmp(p_, "Adobe", 5) == 0) adobe_transform_ = p_[11];
        break;
    default:
        break;                                           // APPn, COM, DNL, ...: skipped
    }
    p_ = seg_end;
    return true;
}

bool JpegDecoder::read_frame_header(uint8_t m)
{
    if (have_frame_) return fail("more than one frame");
    if (!need(2)) return fail("frame header missing");
    const int len = u16();
    if (len < 8 || !need((size_t)len - 2)) return fail("frame header cut short");
    progressive_ = m == M_SOF2;
    if (u8() != 8) return fail("only 8-bit sample precision is supported");
    height_ = u16();
    width_ = u16();
    ncomp_ = u8();
    if (width_ == 0 || height_ == 0) return fail("frame with zero width or height (DNL-defined height is not supported)");
    if (ncomp_ != 1 && ncomp_ != 3 && ncomp_ != 4) return fail("frames with 1, 3 or 4 components only");
    if (len != 8 + 3 * ncomp_) return fail("frame header length does not match its component count");
    if ((uint64_t)width_ * height_ * ncomp_ > (1ull << 31)) return fail("frame too large");
    static const char tags[3] = {'R', 'G', 'B'};
    rgb_tagged_ = 0;
    for (int i = 0; i < ncomp_; ++i) {
        Component &c = comp_[i];
        c.id = u8();
        if (ncomp_ == 3 && c.id == tags[i]) ++rgb_tagged_;
        const int hv = u8();
        c.h = hv >> 4;
        c.v = hv & 15;
        c.tq = u8();
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return fail("sampling factor outside 1..4");
        if (c.tq > 3) return fail("component refers to quantization table > 3");
        if (c.h > hmax_) hmax_ = c.h;
        if (c.v > vmax_) vmax_ = c.v;
    }
    for (int i = 0; i < ncomp_; ++i)
        if (hmax_ % comp_[i].h || vmax_ % comp_[i].v) return fail("sampling factors that do not divide the largest one");
    mcus_x_ = (width_ + 8 * hmax_ - 1) / (8 * hmax_);
    mcus_y_ = (height_ + 8 * vmax_ - 1) / (8 * vmax_);
    for (int i = 0; i < ncomp_; ++i) {
        Component &c = comp_[i];
        c.px_w = (width_ * c.h + hmax_ - 1) / hmax_;
        c.px_h = (height_ * c.v + vmax_ - 1) / vmax_;
        c.blocks_w = mcus_x_ * c.h;
        c.blocks_h = mcus_y_ * c.v;
        c.stride = c.blocks_w * 8;
        c.rows = c.blocks_h * 8;
        c.plane.assign((size_t)c.stride * c.rows, 0);
        if (progressive_) c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
    }
    have_frame_ = true;
    return true;
}

bool JpegDecoder::read_scan_header()
{
    if (!have_frame_) return fail("scan before any frame header");
    if (!need(2)) return fail("scan header missing");
    const int len = u16();
    if (len < 6 || !need((size_t)len - 2)) return fail("scan header cut short");
    scan_n_ = u8();
    if (scan_n_ < 1 || scan_n_ > 4 || scan_n_ > ncomp_) return fail("scan component count out of range");
    if (len != 6 + 2 * scan_n_) return fail("scan header length does not match its component count");
    for (int i = 0; i < scan_n_; ++i) {
        const int id = u8(), tables = u8();
        int which = -1;
        for (int k = 0; k < ncomp_; ++k)
            if (comp_[k].id == id) { which = k; break; }
        if (which < 0) return fail("scan names a component the frame does not have");
        for (int k = 0; k < i; ++k)
            if (scan_comp_[k] == which) return fail("scan names a component twice");
        comp_[which].td = tables >> 4;
        comp_[which].ta = tables & 15;
        if (comp_[which].td > 3 || comp_[which].ta > 3) return fail("scan refers to Huffman table > 3");
        scan_comp_[i] = which;
    }
    ss_ = u8();
    se_ = u8();
    const int a = u8();
    ah_ = a >> 4;
    al_ = a & 15;
    if (progressive_) {
        if (ss_ > 63 || se_ > 63 || ss_ > se_ || ah_ > 13 || al_ > 13) return fail("spectral selection / successive approximation out of range");
        if (ss_ == 0 && se_ != 0) return fail("a progressive scan holds DC or AC coefficients, not both");
        if (ss_ != 0 && scan_n_ != 1) return fail("progressive AC scans carry one component");
    } else {
        if (ss_ != 0 || ah_ != 0 || al_ != 0) return fail("sequential scan with progressive parameters");
        se_ = 63;
    }
    return true;
}

// ---- block procedures ---------------------------------------------------------------------------------------------------
bool JpegDecoder::block_sequential(Component &c, int16_t *blk)          // T.81 F.2.2.1 / F.2.2.2
{
    const HuffmanTable &dct = dc_[c.td], &act = ac_[c.ta];
    if (!dct.defined || !act.defined) return fail("scan uses a Huffman table that was never defined");
    const uint16_t *q = quant_[c.tq];
    memset(blk, 0, 64 * sizeof(int16_t));
    const int t = dct.decode(bits_);
    if (t < 0 || t > 15) return fail("invalid DC difference category");
    c.dc_pred = (int)((uint32_t)c.dc_pred + (uint32_t)bits_.take_signed(t));
    blk[0] = (int16_t)((uint32_t)c.dc_pred * q[0]);
    for (int k = 1; k < 64;) {
        const int rs = act.decode(bits_);
        if (rs < 0) return fail("invalid AC code");
        const int run = rs >> 4, size = rs & 15;
        if (size == 0) {
            if (run != 15) break;                         // EOB
            k += 16;                                      // ZRL
            continue;
        }
        k += run;
        if (k > 63) return fail("AC run passes the end of the block");
        blk[kNatural[k]] = (int16_t)((uint32_t)bits_.take_signed(size) * q[k]);
        ++k;
    }
    return true;
}

bool JpegDecoder::block_dc_progressive(Component &c, int16_t *blk)      // T.81 G.1.2.1
{
    if (ah_ == 0) {
        const HuffmanTable &dct = dc_[c.td];
        if (!dct.defined) return fail("scan uses a Huffman table that was never defined");
        const int t = dct.decode(bits_);
        if (t < 0 || t > 15) return fail("invalid DC difference category");
        c.dc_pred = (int)((uint32_t)c.dc_pred + (uint32_t)bits_.take_signed(t));
        blk[0] = (int16_t)((uint32_t)c.dc_pred << al_);
    } else if (bits_.bit()) {
        blk[0] = (int16_t)(blk[0] + (1 << al_));
    }
    return true;
}

bool JpegDecoder::block_ac_first(Component &c, int16_t *blk)            // T.81 G.1.2.2
{
    if (eobrun_ > 0) { --eobrun_; return true; }
    const HuffmanTable &act = ac_[c.ta];
    if (!act.defined) return fail("scan uses a Huffman table that was never defined");
    for (int k = ss_; k <= se_;) {
        const int rs = act.decode(bits_);
        if (rs < 0) return fail("invalid AC code");
        const int run = rs >> 4, size = rs & 15;
        if (size == 0) {
            if (run < 15) {                               // EOBn: this band ends here for 2^run + extra blocks, this one included
                eobrun_ = (1 << run) - 1;
                if (run) eobrun_ += (int)bits_.take(run);
                break;
            }
            k += 16;
            continue;
        }
        k += run;
        if (k > 63) return fail("AC run passes the end of the block");
        blk[kNatural[k]] = (int16_t)((uint32_t)bits_.take_signed(size) << al_);
        ++k;
    }
    return true;
}

bool JpegDecoder::block_ac_refine(Component &c, int16_t *blk)           // T.81 G.1.2.3, Figure G.7
{
    const int plus = 1 << al_, minus = -plus;
    // a coefficient that is already nonzero receives one correction bit whenever the scan position passes it
    auto correct = [&](int16_t &v) {
        if (bits_.bit() && (v & plus) == 0) v = (int16_t)(v + (v > 0 ? plus : minus));
    };
    int k = ss_;
    if (eobrun_ == 0) {
        const HuffmanTable &act = ac_[c.ta];
        if (!act.defined) return fail("scan uses a Huffman table that was never defined");
        while (k <= se_) {
            const int rs = act.decode(bits_);
            if (rs < 0) return fail("invalid AC code");
            int zeros = rs >> 4;
            const int size = rs & 15;
            int fresh = 0;
            if (size == 0) {
                if (zeros < 15) {                         // EOBn: the rest of this block only gets correction bits
                    eobrun_ = 1 << zeros;
                    if (zeros) eobrun_ += (int)bits_.take(zeros);
                    break;
                }                                         // ZRL: pass 16 zero-history coefficients
            } else {
                if (size != 1) return fail("refinement scan introduces a coefficient larger than one");
                fresh = bits_.bit() ? plus : minus;
            }
            // pass `zeros` coefficients whose history is zero (correcting the nonzero ones on the way); the next
            // zero-history position receives the fresh value (or, for ZRL, is simply the 16th one passed)
            for (; k <= se_; ++k) {
                int16_t &v = blk[kNatural[k]];
                if (v != 0) { correct(v); continue; }
                if (zeros == 0) {
                    if (fresh) v = (int16_t)fresh;
                    ++k;
                    break;
                }
                --zeros;
            }
        }
    }
    if (eobrun_ > 0) {                                    // inside an end-of-band run (possibly opened just above)
        for (; k <= se_; ++k) {
            int16_t &v = blk[kNatural[k]];
            if (v != 0) correct(v);
        }
        --eobrun_;
    }
    return true;
}

bool JpegDecoder::code_block(Component &c, int bx, int by)
{
    if (!progressive_) {
        int16_t blk[64];
        if (!block_sequential(c, blk)) return false;
        if (!quant_defined_[c.tq]) return fail("component uses a quantization table that was never defined");
        idct8x8(c.plane.data() + (size_t)by * 8 * c.stride + (size_t)bx * 8, c.stride, blk);
        return true;
    }
    int16_t *blk = c.coef.data() + ((size_t)by * c.blocks_w + bx) * 64;
    if (ss_ == 0) return block_dc_progressive(c, blk);
    return ah_ == 0 ? block_ac_first(c, blk) : block_ac_refine(c, blk);
}

// restart interval boundary (T.81 E.2.4, F.2.2.4): byte-align, expect RSTm, reset predictions and the end-of-band run.
// A stream that does not carry the marker there simply ends the scan (what is decoded so far stays).
bool JpegDecoder::restart(int &countdown)
{
    bits_.byte_align();
    uint8_t m = bits_.pending_marker();
    if (!m) {
        const uint8_t *q = bits_.position();
        if (end_ - q >= 2 && q[0] == 0xFF && q[1] >= M_RST0 && q[1] <= M_RST7) { m = q[1]; bits_.open(q + 2, end_); }
    } else if (m >= M_RST0 && m <= M_RST7) {
        bits_.clear_marker();
    }
    if (m < M_RST0 || m > M_RST7) return false;
    for (int i = 0; i < ncomp_; ++i) comp_[i].dc_pred = 0;
    eobrun_ = 0;
    countdown = restart_interval_;
    return true;
}

bool JpegDecoder::decode_scan()
{
    bits_.open(p_, end_);
    for (int i = 0; i < ncomp_; ++i) comp_[i].dc_pred = 0;
    eobrun_ = 0;
    int countdown = restart_interval_ ? restart_interval_ : 0x7fffffff;
    bool more = true;
    if (scan_n_ == 1) {
        // a single-component scan is not interleaved: its data units are the blocks that hold image samples, row by row
        Component &c = comp_[scan_comp_[0]];
        const int bw = (c.px_w + 7) >> 3, bh = (c.px_h + 7) >> 3;
        for (int by = 0; by < bh && more; ++by)
            for (int bx = 0; bx < bw && more; ++bx) {
                if (!code_block(c, bx, by)) return false;
                if (--countdown <= 0) more = restart(countdown);
            }
    } else {
        for (int my = 0; my < mcus_y_ && more; ++my)
            for (int mx = 0; mx < mcus_x_ && more; ++mx) {
                for (int s = 0; s < scan_n_; ++s) {
                    Component &c = comp_[scan_comp_[s]];
                    for (int v = 0; v < c.v; ++v)
                        for (int h = 0; h < c.h; ++h)
                            if (!code_block(c, mx * c.h + h, my * c.v + v)) return false;
                }
                if (--countdown <= 0) more = restart(countdown);
            }
    }
    // continue with the segments behind the entropy-coded data
    p_ = bits_.position();
    if (bits_.pending_marker()) p_ -= 2;                   // the reader stopped right behind a marker's two bytes (fill bytes included:
                                                          // it consumes 0xFF runs up to the code): hand the marker back to the segment loop
    return true;
}

void JpegDecoder::reconstruct_progressive()
{
    for (int i = 0; i < ncomp_; ++i) {
        Component &c = comp_[i];
        const uint16_t *q = quant_[c.tq];
        const int bw = (c.px_w + 7) >> 3, bh = (c.px_h + 7) >> 3;
        for (int by = 0; by < bh; ++by)
            for (int bx = 0; bx < bw; ++bx) {
                int16_t *blk = c.coef.data() + ((size_t)by * c.blocks_w + bx) * 64;
                for (int k = 0; k < 64; ++k) blk[kNatural[k]] = (int16_t)((uint32_t)(int)blk[kNatural[k]] * q[k]);
                idct8x8(c.plane.data() + (size_t)by * 8 * c.stride + (size_t)bx * 8, c.stride, blk);
            }
    }
}

// ---- planes -> interleaved pixels (stb_image.h:3798-3960: which rows feed an output row, colour-space decisions) ---------
bool JpegDecoder::produce(GhImage &out)
{
    const int nout = ncomp_ >= 3 ? 3 : 1;                 // stbi_load(..., req_comp = 0): grey stays grey, everything else is RGB
    out.width = width_;
    out.height = height_;
    out.channels = nout;
    out.pixels.assign((size_t)width_ * height_ * nout, 0);
    const bool plain_rgb = ncomp_ == 3 && (rgb_tagged_ == 3 || (adobe_transform_ == 0 && !jfif_));
    std::vector<uint8_t> line[4];
    int hs[4], vs[4], lores_w[4];
    for (int k = 0; k < ncomp_; ++k) {
        hs[k] = hmax_ / comp_[k].h;
        vs[k] = vmax_ / comp_[k].v;
        lores_w[k] = (width_ + hs[k] - 1) / hs[k];
        line[k].assign((size_t)width_ + 8, 0);            // upsampling may write up to hs - 1 samples past the row
    }
    const uint8_t *row[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int y = 0; y < height_; ++y) {
        for (int k = 0; k < ncomp_; ++k) {
            // Output row y of a component subsampled vs times lies between two source rows.  With t = y + vs/2:
            // t / vs rows have been completed; the pair is (t/vs - 1, t/vs), clamped to the rows that hold image content;
            // the nearer one is the second of the pair in the first half of a step, the first one after it.
            const Component &c = comp_[k];
            const int t = y + (vs[k] >> 1), done = t / vs[k], phase = t % vs[k];
            const int last = c.px_h - 1;
            const int r0 = done - 1 < 0 ? 0 : (done - 1 > last ? last : done - 1), r1 = done > last ? last : done;
            const bool second_is_near = phase >= (vs[k] >> 1);
            const uint8_t *p0 = c.plane.data() + (size_t)r0 * c.stride, *p1 = c.plane.data() + (size_t)r1 * c.stride;
            row[k] = upsample_row(line[k].data(), second_is_near ? p1 : p0, second_is_near ? p0 : p1, lores_w[k], hs[k], vs[k]);
        }
        uint8_t *o = out.pixels.data() + (size_t)y * width_ * nout;
        if (ncomp_ == 1) {
            memcpy(o, row[0], (size_t)width_);
        } else if (ncomp_ == 3) {
            if (plain_rgb)
                for (int x = 0; x < width_; ++x) { o[3 * x] = row[0][x]; o[3 * x + 1] = row[1][x]; o[3 * x + 2] = row[2][x]; }
            else
                ycc_to_rgb(o, row[0], row[1], row[2], width_, 3);
        } else {                                          // 4 components: CMYK family, by the Adobe transform flag
            if (adobe_transform_ == 0) {
                for (int x = 0; x < width_; ++x) {
                    const uint8_t k = row[3][x];
                    o[3 * x] = mul255(row[0][x], k); o[3 * x + 1] = mul255(row[1][x], k); o[3 * x + 2] = mul255(row[2][x], k);
                }
            } else {
                ycc_to_rgb(o, row[0], row[1], row[2], width_, 3);
                if (adobe_transform_ == 2)
                    for (int x = 0; x < width_; ++x) {
                        const uint8_t k = row[3][x];
                        o[3 * x] = mul255((uint8_t)(255 - o[3 * x]), k);
                        o[3 * x + 1] = mul255((uint8_t)(255 - o[3 * x + 1]), k);
                        o[3 * x + 2] = mul255((uint8_t)(255 - o[3 * x + 2]), k);
                    }
            }
        }
    }
    return true;
}

bool JpegDecoder::run(GhImage &out)
{
    uint8_t m = 0;
    if (!need(2) || p_[0] != 0xFF || p_[1] != M_SOI) return fail("does not start with SOI");
    p_ += 2;
    bool saw_scan = false;
    while (next_marker(m)) {
        if (m == M_EOI) break;
        if (m == M_SOF0 || m == M_SOF1 || m == M_SOF2) {
            if (!read_frame_header(m)) return false;
        } else if ((m >= 0xC3 && m <= 0xCF && m != M_DHT && m != 0xC8) ) {
            return fail("lossless, hierarchical and arithmetic-coded frames are not supported");
        } else if (m == M_SOS) {
            if (!read_scan_header() || !decode_scan()) return false;
            saw_scan = true;
        } else if (m >= M_RST0 && m <= M_RST7) {
            continue;                                     // a stray restart marker between segments
        } else if (m == M_SOI || m == 0x01) {
            continue;                                     // markers without a segment
        } else {
            if (!read_tables_or_misc(m)) return false;
        }
    }
    if (!have_frame_) return fail("no frame header");
    if (!saw_scan) return fail("no scan");
    for (int i = 0; i < ncomp_; ++i)
        if (!quant_defined_[comp_[i].tq]) return fail("component uses a quantization table that was never defined");
    if (progressive_) reconstruct_progressive();
    return produce(out);
}

}  // namespace

bool gh_jpeg_decode(const uint8_t *data, size_t size, GhImage &out, std::string &err)
{
    err.clear();
    std::unique_ptr<JpegDecoder> d(new JpegDecoder(data, size, err));
    if (!d->run(out)) {
        if (err.empty()) err = "jpeg: corrupt stream";
        return false;
    }
    return true;
}
