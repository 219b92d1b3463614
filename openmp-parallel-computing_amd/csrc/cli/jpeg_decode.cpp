// jpeg_decode.cpp -- baseline + progressive JPEG decoder for the drop-in CLIs (host side).
//
// The reference drivers decode their input with the third-party stb_image.h they vendor
// (monolithic/src/main.c:21 -> stbi_load(path, &w, &h, &channels, 0)).  stb's JPEG path is not
// bit-identical to libjpeg (up to +-3 per channel on images/test.jpg), and the
// reference's golden images/test_gray.png was produced from stb-decoded pixels, so for FILE-level
// parity this decoder reproduces the same published arithmetic choices:
//   * integer "islow" IDCT with 12-bit constants, column pass >> 10 (rounding 512), row pass >> 17
//     (rounding 65536, +128 level shift folded in), clamp to 0..255;
//   * "fancy" chroma upsampling: 3:1 taps for h2 / v2, 9:3:3:1 (via two 3:1 passes, >> 4, +8) for h2v2,
//     the first/last sample of a row not interpolated;
//   * YCbCr -> RGB in 20-bit fixed point with constants round(c * 4096) << 8, the Cb contribution to G
//     truncated to a multiple of 2^16, +0.5 rounding, >> 20, clamp.
// tests/test_cli_image_io.py checks it pixel-for-pixel against the reference's own decoder (compiled in
// place in the build container) on the reference's three JPEGs (baseline 4:4:4, progressive 4:2:0).
// Written from the JPEG standard (ITU T.81) and the behaviour above; no third-party code is included.
#include "image_io.h"

#include <cstdlib>
#include <cstring>

namespace {

const uint8_t kDezigzag[64 + 15] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};   // padding: corrupt streams may run past 63

struct Huff {
    uint8_t fast[512];        // 9-bit lookup: symbol index or 255
    uint16_t code[256];
    uint8_t values[256];
    uint8_t size[257];
    unsigned maxcode[18];
    int delta[17];
    bool valid = false;

    bool build(const int *count)
    {
        int k = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < count[i]; ++j) { size[k++] = (uint8_t)(i + 1); if (k >= 257) return false; }
        size[k] = 0;
        int code_v = 0;
        k = 0;
        for (int j = 1; j <= 16; ++j) {
            delta[j] = k - code_v;
            if (size[k] == j) {
                while (size[k] == j) code[k++] = (uint16_t)(code_v++);
                if (code_v - 1 >= (1 << j)) return false;
            }
            maxcode[j] = (unsigned)code_v << (16 - j);
            code_v <<= 1;
        }
        maxcode[17] = 0xffffffff;
        memset(fast, 255, sizeof(fast));
        for (int i = 0; i < k; ++i) {
            const int s = size[i];
            if (s <= 9) {
                const int c = code[i] << (9 - s), m = 1 << (9 - s);
                for (int j = 0; j < m; ++j) fast[c + j] = (uint8_t)i;
            }
        }
        valid = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;
    std::vector<uint8_t> data;      // w2 * h2 samples
    std::vector<int16_t> coeff;     // progressive: 64 per block
    int coeff_w = 0, coeff_h = 0;
};

struct Decoder {
    const uint8_t *p = nullptr, *end = nullptr;
    std::string *err = nullptr;

    Huff huff_dc[4], huff_ac[4];
    uint16_t dequant[4][64];
    Component comp[4];
    int img_x = 0, img_y = 0, img_n = 0;
    int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0, mcu_w = 0, mcu_h = 0;
    bool progressive = false, jfif = false;
    int app14_transform = -1, rgb = 0;
    int restart_interval = 0, todo = 0;
    int scan_n = 0, order[4] = {0, 0, 0, 0};
    int spec_start = 0, spec_end = 0, succ_high = 0, succ_low = 0, eob_run = 0;

    // entropy-coded segment reader
    uint32_t code_buffer = 0;
    int code_bits = 0;
    uint8_t marker = 0xff;    // 0xff = none pending
    bool nomore = false;

    bool fail(const char *m) { if (err && err->empty()) *err = m; return false; }
    int get8() { return p < end ? *p++ : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }

    void grow()
    {
        do {
            unsigned b = nomore ? 0u : (unsigned)get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8();      // fill bytes
                if (c != 0) { marker = (uint8_t)c; nomore = true; return; }
            }
            code_buffer |= b << (24 - code_bits);
            code_bits += 8;
        } while (code_bits <= 24);
    }

    int decode(const Huff &h)
    {
        if (code_bits < 16) grow();
        const int c = (code_buffer >> (32 - 9)) & 511;
        int k = h.fast[c];
        if (k < 255) {
            const int s = h.size[k];
            if (s > code_bits) return -1;
            code_buffer <<= s;
            code_bits -= s;
            return h.values[k];
        }
        const unsigned temp = code_buffer >> 16;
        for (k = 10; temp >= h.maxcode[k]; ++k) {}
        if (k == 17) { code_bits -= 16; return -1; }
        if (k > code_bits) return -1;
        const int idx = (int)((code_buffer >> (32 - k)) & ((1u << k) - 1)) + h.delta[k];
        if (idx < 0 || idx >= 256) return -1;
        code_bits -= k;
        code_buffer <<= k;
        return h.values[idx];
    }

    int receive_extend(int n)     // T.81 F.2.2.1 RECEIVE + EXTEND
    {
        if (n == 0) return 0;
        if (code_bits < n) grow();
        if (code_bits < n) return 0;
        const int v = (int)(code_buffer >> (32 - n));
        code_buffer <<= n;
        code_bits -= n;
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int get_bits(int n)
    {
        if (n == 0) return 0;
        if (code_bits < n) grow();
        if (code_bits < n) return 0;
        const int v = (int)(code_buffer >> (32 - n));
        code_buffer <<= n;
        code_bits -= n;
        return v;
    }
    int get_bit()
    {
        if (code_bits < 1) grow();
        if (code_bits < 1) return 0;
        const int v = (int)(code_buffer >> 31);
        code_buffer <<= 1;
        --code_bits;
        return v;
    }

    void reset_entropy()
    {
        code_bits = 0;
        code_buffer = 0;
        nomore = false;
        for (int i = 0; i < 4; ++i) comp[i].dc_pred = 0;
        marker = 0xff;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        eob_run = 0;
    }

    // ---- block decoders ----------------------------------------------------------------------------
    bool block_baseline(int16_t *data, const Huff &hdc, const Huff &hac, int b, const uint16_t *dq)
    {
        if (code_bits < 16) grow();
        const int t = decode(hdc);
        if (t < 0 || t > 15) return fail("bad huffman code");
        memset(data, 0, 64 * sizeof(int16_t));
        const int diff = t ? receive_extend(t) : 0;
        const int dc = comp[b].dc_pred + diff;
        comp[b].dc_pred = dc;
        data[0] = (int16_t)(dc * dq[0]);
        int k = 1;
        do {
            const int rs = decode(hac);
            if (rs < 0) return fail("bad huffman code");
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += r;
                const unsigned zig = kDezigzag[k++];
                data[zig] = (int16_t)(receive_extend(s) * dq[zig]);
            }
        } while (k < 64);
        return true;
    }

    bool block_prog_dc(int16_t *data, const Huff &hdc, int b)
    {
        if (spec_end != 0) return fail("can't merge dc and ac");
        if (code_bits < 16) grow();
        if (succ_high == 0) {
            memset(data, 0, 64 * sizeof(int16_t));
            const int t = decode(hdc);
            if (t < 0 || t > 15) return fail("can't merge dc and ac");
            const int diff = t ? receive_extend(t) : 0;
            const int dc = comp[b].dc_pred + diff;
            comp[b].dc_pred = dc;
            data[0] = (int16_t)(dc * (1 << succ_low));
        } else if (get_bit()) {
            data[0] = (int16_t)(data[0] + (1 << succ_low));
        }
        return true;
    }

    bool block_prog_ac(int16_t *data, const Huff &hac)
    {
        if (spec_start == 0) return fail("can't merge dc and ac");
        if (succ_high == 0) {
            const int shift = succ_low;
            if (eob_run) { --eob_run; return true; }
            int k = spec_start;
            do {
                const int rs = decode(hac);
                if (rs < 0) return fail("bad huffman code");
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = 1 << r;
                        if (r) eob_run += get_bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const unsigned zig = kDezigzag[k++];
                    data[zig] = (int16_t)(receive_extend(s) * (1 << shift));
                }
            } while (k <= spec_end);
        } else {
            const int16_t bit = (int16_t)(1 << succ_low);
            if (eob_run) {
                --eob_run;
                for (int k = spec_start; k <= spec_end; ++k) {
                    int16_t *q = &data[kDezigzag[k]];
                    if (*q != 0 && get_bit() && (*q & bit) == 0) *q = (int16_t)(*q > 0 ? *q + bit : *q - bit);
                }
            } else {
                int k = spec_start;
                do {
                    const int rs = decode(hac);
                    if (rs < 0) return fail("bad huffman code");
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) {
                            eob_run = (1 << r) - 1;
                            if (r) eob_run += get_bits(r);
                            r = 64;   // run to the end of the band, refining non-zero coefficients
                        }
                    } else {
                        if (s != 1) return fail("bad huffman code");
                        s = get_bit() ? bit : -bit;
                    }
                    while (k <= spec_end) {
                        int16_t *q = &data[kDezigzag[k++]];
                        if (*q != 0) {
                            if (get_bit() && (*q & bit) == 0) *q = (int16_t)(*q > 0 ? *q + bit : *q - bit);
                        } else {
                            if (r == 0) { *q = (int16_t)s; break; }
                            --r;
                        }
                    }
                } while (k <= spec_end);
            }
        }
        return true;
    }

    // ---- inverse DCT (integer, "islow") ------------------------------------------------------------
    static inline uint8_t clamp8(int x) { return (unsigned)x > 255 ? (x < 0 ? 0 : 255) : (uint8_t)x; }
    static constexpr int f2f(double x) { return (int)(x * 4096 + 0.5); }

#define GH_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                 \
    int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;        \
    p2 = s2; p3 = s6;                                              \
    p1 = (p2 + p3) * f2f(0.5411961f);                              \
    t2 = p1 + p3 * f2f(-1.847759065f);                             \
    t3 = p1 + p2 * f2f(0.765366865f);                              \
    p2 = s0; p3 = s4;                                              \
    t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                  \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;        \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                            \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;        \
    p5 = (p3 + p4) * f2f(1.175875602f);                            \
    t0 = t0 * f2f(0.298631336f); t1 = t1 * f2f(2.053119869f);      \
    t2 = t2 * f2f(3.072711026f); t3 = t3 * f2f(1.501321110f);      \
    p1 = p5 + p1 * f2f(-0.899976223f); p2 = p5 + p2 * f2f(-2.562915447f); \
    p3 = p3 * f2f(-1.961570560f); p4 = p4 * f2f(-0.390180644f);    \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

    static void idct_block(uint8_t *out, int stride, const int16_t *d)
    {
        int val[64];
        for (int i = 0; i < 8; ++i) {
            const int16_t *c = d + i;
            int *v = val + i;
            if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
                const int dcterm = c[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
            } else {
                GH_IDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
                x0 += 512; x1 += 512; x2 += 512; x3 += 512;
                v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
                v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
                v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
                v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
            }
        }
        for (int i = 0; i < 8; ++i) {
            const int *v = val + 8 * i;
            uint8_t *o = out + (size_t)i * stride;
            GH_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
            x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
            o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
            o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
            o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
            o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
        }
    }
#undef GH_IDCT_1D

    // ---- markers -----------------------------------------------------------------------------------
    int next_marker()
    {
        if (marker != 0xff) { const int m = marker; marker = 0xff; return m; }
        int x = get8();
        if (x != 0xff) return 0xff;
        while (x == 0xff) x = get8();
        return x;
    }

    bool process_marker(int m)
    {
        switch (m) {
        case 0xff: return fail("expected marker");
        case 0xDD:
            if (get16() != 4) return fail("bad DRI len");
            restart_interval = get16();
            return true;
        case 0xDB: {
            int L = get16() - 2;
            while (L > 0) {
                const int q = get8();
                const int prec = q >> 4, t = q & 15;
                if (prec > 1) return fail("bad DQT type");
                if (t > 3) return fail("bad DQT table");
                for (int i = 0; i < 64; ++i) dequant[t][kDezigzag[i]] = (uint16_t)(prec ? get16() : get8());
                L -= prec ? 129 : 65;
            }
            return L == 0 ? true : fail("bad DQT len");
        }
        case 0xC4: {
            int L = get16() - 2;
            while (L > 0) {
                int sizes[16], n = 0;
                const int q = get8();
                const int tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) return fail("bad DHT header");
                for (int i = 0; i < 16; ++i) { sizes[i] = get8(); n += sizes[i]; }
                if (n > 256) return fail("bad DHT header");
                L -= 17;
                Huff &h = tc == 0 ? huff_dc[th] : huff_ac[th];
                if (!h.build(sizes)) return fail("bad code lengths");
                for (int i = 0; i < n; ++i) h.values[i] = (uint8_t)get8();
                L -= n;
            }
            return L == 0 ? true : fail("bad DHT len");
        }
        default: break;
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
            int L = get16();
            if (L < 2) return fail(m == 0xFE ? "bad COM len" : "bad APP len");
            L -= 2;
            if (m == 0xE0 && L >= 5) {
                static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; ++i) if (get8() != tag[i]) ok = false;
                L -= 5;
                if (ok) jfif = true;
            } else if (m == 0xEE && L >= 12) {
                static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; ++i) if (get8() != tag[i]) ok = false;
                L -= 6;
                if (ok) { get8(); get16(); get16(); app14_transform = get8(); L -= 6; }
            }
            if (p + L > end) { p = end; return true; }
            p += L;
            return true;
        }
        return fail("unknown marker");
    }

    bool process_sof()
    {
        const int Lf = get16();
        if (Lf < 11) return fail("bad SOF len");
        if (get8() != 8) return fail("only 8-bit");
        img_y = get16();
        img_x = get16();
        if (img_y == 0 || img_x == 0) return fail("0 width/height");
        img_n = get8();
        if (img_n != 1 && img_n != 3 && img_n != 4) return fail("bad component count");
        if (Lf != 8 + 3 * img_n) return fail("bad SOF len");
        rgb = 0;
        for (int i = 0; i < img_n; ++i) {
            static const uint8_t rgbid[3] = {'R', 'G', 'B'};
            comp[i].id = get8();
            if (img_n == 3 && comp[i].id == rgbid[i]) ++rgb;
            const int q = get8();
            comp[i].h = q >> 4;
            comp[i].v = q & 15;
            if (!comp[i].h || comp[i].h > 4 || !comp[i].v || comp[i].v > 4) return fail("bad sampling factor");
            comp[i].tq = get8();
            if (comp[i].tq > 3) return fail("bad TQ");
        }
        h_max = v_max = 1;
        for (int i = 0; i < img_n; ++i) { if (comp[i].h > h_max) h_max = comp[i].h; if (comp[i].v > v_max) v_max = comp[i].v; }
        for (int i = 0; i < img_n; ++i) if (h_max % comp[i].h || v_max % comp[i].v) return fail("bad sampling factor");
        if ((uint64_t)img_x * img_y * img_n > (1ull << 31)) return fail("too large");
        mcu_w = h_max * 8;
        mcu_h = v_max * 8;
        mcu_x = (img_x + mcu_w - 1) / mcu_w;
        mcu_y = (img_y + mcu_h - 1) / mcu_h;
        for (int i = 0; i < img_n; ++i) {
            Component &c = comp[i];
            c.x = (img_x * c.h + h_max - 1) / h_max;
            c.y = (img_y * c.v + v_max - 1) / v_max;
            c.w2 = mcu_x * c.h * 8;
            c.h2 = mcu_y * c.v * 8;
            c.data.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) {
                c.coeff_w = c.w2 / 8;
                c.coeff_h = c.h2 / 8;
                c.coeff.assign((size_t)c.w2 * c.h2, 0);
            }
        }
        return true;
    }

    bool process_sos()
    {
        const int Ls = get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > 4 || scan_n > img_n) return fail("bad SOS component count");
        if (Ls != 6 + 2 * scan_n) return fail("bad SOS len");
        for (int i = 0; i < scan_n; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            for (; which < img_n; ++which) if (comp[which].id == id) break;
            if (which == img_n) return fail("bad SOS component");
            comp[which].hd = q >> 4;
            comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) return fail("bad huffman table index");
            order[i] = which;
        }
        spec_start = get8();
        spec_end = get8();
        const int aa = get8();
        succ_high = aa >> 4;
        succ_low = aa & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) return fail("bad SOS");
        } else {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) return fail("bad SOS");
            spec_end = 63;
        }
        return true;
    }

    bool restart_if_due()
    {
        if (--todo <= 0) {
            if (code_bits < 24) grow();
            if (!(marker >= 0xD0 && marker <= 0xD7)) return false;   // no restart marker: end of scan data
            reset_entropy();
        }
        return true;
    }

    bool decode_scan()
    {
        reset_entropy();
        int16_t block[64];
        if (!progressive) {
            if (scan_n == 1) {
                const int n = order[0];
                Component &c = comp[n];
                const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                if (!huff_dc[c.hd].valid || !huff_ac[c.ha].valid) return fail("missing huffman table");
                for (int j = 0; j < h; ++j)
                    for (int i = 0; i < w; ++i) {
                        if (!block_baseline(block, huff_dc[c.hd], huff_ac[c.ha], n, dequant[c.tq])) return false;
                        idct_block(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, block);
                        if (!restart_if_due()) return true;
                    }
            } else {
                for (int j = 0; j < mcu_y; ++j)
                    for (int i = 0; i < mcu_x; ++i) {
                        for (int k = 0; k < scan_n; ++k) {
                            const int n = order[k];
                            Component &c = comp[n];
                            if (!huff_dc[c.hd].valid || !huff_ac[c.ha].valid) return fail("missing huffman table");
                            for (int y = 0; y < c.v; ++y)
                                for (int x = 0; x < c.h; ++x) {
                                    const int x2 = (i * c.h + x) * 8, y2 = (j * c.v + y) * 8;
                                    if (!block_baseline(block, huff_dc[c.hd], huff_ac[c.ha], n, dequant[c.tq])) return false;
                                    idct_block(c.data.data() + (size_t)c.w2 * y2 + x2, c.w2, block);
                                }
                        }
                        if (!restart_if_due()) return true;
                    }
            }
            return true;
        }
        if (scan_n == 1) {
            const int n = order[0];
            Component &c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    int16_t *data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                    if (spec_start == 0) {
                        if (!huff_dc[c.hd].valid) return fail("missing huffman table");
                        if (!block_prog_dc(data, huff_dc[c.hd], n)) return false;
                    } else {
                        if (!huff_ac[c.ha].valid) return fail("missing huffman table");
                        if (!block_prog_ac(data, huff_ac[c.ha])) return false;
                    }
                    if (!restart_if_due()) return true;
                }
        } else {
            for (int j = 0; j < mcu_y; ++j)
                for (int i = 0; i < mcu_x; ++i) {
                    for (int k = 0; k < scan_n; ++k) {
                        const int n = order[k];
                        Component &c = comp[n];
                        if (!huff_dc[c.hd].valid) return fail("missing huffman table");
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) {
                                const int x2 = i * c.h + x, y2 = j * c.v + y;
                                int16_t *data = c.coeff.data() + 64 * ((size_t)x2 + (size_t)y2 * c.coeff_w);
                                if (!block_prog_dc(data, huff_dc[c.hd], n)) return false;
                            }
                    }
                    if (!restart_if_due()) return true;
                }
        }
        return true;
    }

    void finish_progressive()
    {
        for (int n = 0; n < img_n; ++n) {
            Component &c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    int16_t *data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                    for (int k = 0; k < 64; ++k) data[k] = (int16_t)(data[k] * dequant[c.tq][k]);
                    idct_block(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
                }
        }
    }

    bool decode_image()
    {
        for (int i = 0; i < 4; ++i) memset(dequant[i], 0, sizeof(dequant[i]));
        if (get8() != 0xff || get8() != 0xD8) return fail("no SOI");
        int m = next_marker();
        while (!(m == 0xC0 || m == 0xC1 || m == 0xC2)) {
            if (!process_marker(m)) return false;
            m = next_marker();
            while (m == 0xff) { if (p >= end) return fail("no SOF"); m = next_marker(); }
        }
        progressive = m == 0xC2;
        if (!process_sof()) return false;
        m = next_marker();
        while (m != 0xD9) {
            if (m == 0xDA) {
                if (!process_sos()) return false;
                if (!decode_scan()) return false;
                if (marker == 0xff) {
                    // scan data may be followed by junk: look for the next marker
                    while (p < end) {
                        if (get8() == 0xff) {
                            int x = get8();
                            while (x == 0xff) x = get8();
                            if (x != 0 && !(x >= 0xD0 && x <= 0xD7)) { marker = (uint8_t)x; break; }
                        }
                    }
                    if (marker == 0xff) break;   // ran off the end: treat as EOI
                }
            } else if (m == 0xDC) {
                const int Ld = get16();
                const int NL = get16();
                if (Ld != 4) return fail("bad DNL len");
                if (NL != img_y) return fail("bad DNL height");
            } else {
                if (!process_marker(m)) return false;
            }
            m = next_marker();
        }
        if (progressive) finish_progressive();
        return true;
    }
};

// ---- chroma upsampling ----------------------------------------------------------------------------------
typedef uint8_t *(*ResampleFn)(uint8_t *out, const uint8_t *in_near, const uint8_t *in_far, int w, int hs);

uint8_t *resample_1(uint8_t *, const uint8_t *in_near, const uint8_t *, int, int) { return const_cast<uint8_t *>(in_near); }

uint8_t *resample_v2(uint8_t *out, const uint8_t *in_near, const uint8_t *in_far, int w, int)
{
    for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * in_near[i] + in_far[i] + 2) >> 2);
    return out;
}

uint8_t *resample_h2(uint8_t *out, const uint8_t *in, const uint8_t *, int w, int)
{
    if (w == 1) { out[0] = out[1] = in[0]; return out; }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    int i;
    for (i = 1; i < w - 1; ++i) {
        const int n = 3 * in[i] + 2;
        out[i * 2 + 0] = (uint8_t)((n + in[i - 1]) >> 2);
        out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
    }
    out[i * 2 + 0] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
    out[i * 2 + 1] = in[w - 1];
    return out;
}

uint8_t *resample_hv2(uint8_t *out, const uint8_t *in_near, const uint8_t *in_far, int w, int)
{
    if (w == 1) { out[0] = out[1] = (uint8_t)((3 * in_near[0] + in_far[0] + 2) >> 2); return out; }
    int t1 = 3 * in_near[0] + in_far[0];
    out[0] = (uint8_t)((t1 + 2) >> 2);
    for (int i = 1; i < w; ++i) {
        const int t0 = t1;
        t1 = 3 * in_near[i] + in_far[i];
        out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
        out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
    }
    out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    return out;
}

uint8_t *resample_generic(uint8_t *out, const uint8_t *in_near, const uint8_t *, int w, int hs)
{
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < hs; ++j) out[i * hs + j] = in_near[i];
    return out;
}

inline uint8_t clamp8(int x) { return (unsigned)x > 255 ? (x < 0 ? 0 : 255) : (uint8_t)x; }

void ycbcr_to_rgb_row(uint8_t *out, const uint8_t *y, const uint8_t *pcb, const uint8_t *pcr, int count)
{
    auto fixed = [](double x) { return ((int)(x * 4096.0 + 0.5)) << 8; };
    const int c_r = fixed(1.40200f), c_g1 = fixed(0.71414f), c_g2 = fixed(0.34414f), c_b = fixed(1.77200f);
    for (int i = 0; i < count; ++i) {
        const int y_fixed = (y[i] << 20) + (1 << 19);
        const int cr = pcr[i] - 128, cb = pcb[i] - 128;
        int r = y_fixed + cr * c_r;
        int g = y_fixed + (cr * -c_g1) + (int)(((unsigned)(cb * -c_g2)) & 0xffff0000u);
        int b = y_fixed + cb * c_b;
        r >>= 20; g >>= 20; b >>= 20;
        out[0] = clamp8(r); out[1] = clamp8(g); out[2] = clamp8(b);
        out += 3;
    }
}

inline uint8_t blinn_8x8(uint8_t x, uint8_t y)
{
    const unsigned t = x * y + 128;
    return (uint8_t)((t + (t >> 8)) >> 8);
}

}  // namespace

bool gh_jpeg_decode(const uint8_t *data, size_t size, GhImage &out, std::string &err)
{
    err.clear();
    Decoder d;
    d.p = data;
    d.end = data + size;
    d.err = &err;
    if (!d.decode_image()) { if (err.empty()) err = "corrupt JPEG"; return false; }

    const int n = d.img_n >= 3 ? 3 : 1;                       // stbi_load(..., 0): native channel count
    const bool is_rgb = d.img_n == 3 && (d.rgb == 3 || (d.app14_transform == 0 && !d.jfif));
    const int decode_n = d.img_n;

    struct Resample { ResampleFn fn; const uint8_t *line0, *line1; int hs, vs, w_lores, ystep, ypos; std::vector<uint8_t> linebuf; };
    Resample rs[4];
    for (int k = 0; k < decode_n; ++k) {
        Resample &r = rs[k];
        r.linebuf.assign((size_t)d.img_x + 3, 0);
        r.hs = d.h_max / d.comp[k].h;
        r.vs = d.v_max / d.comp[k].v;
        r.ystep = r.vs >> 1;
        r.w_lores = (d.img_x + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = d.comp[k].data.data();
        if (r.hs == 1 && r.vs == 1) r.fn = resample_1;
        else if (r.hs == 1 && r.vs == 2) r.fn = resample_v2;
        else if (r.hs == 2 && r.vs == 1) r.fn = resample_h2;
        else if (r.hs == 2 && r.vs == 2) r.fn = resample_hv2;
        else r.fn = resample_generic;
    }

    out.width = d.img_x;
    out.height = d.img_y;
    out.channels = n;
    out.pixels.assign((size_t)n * d.img_x * d.img_y, 0);
    const uint8_t *coutput[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int j = 0; j < d.img_y; ++j) {
        uint8_t *o = out.pixels.data() + (size_t)n * d.img_x * j;
        for (int k = 0; k < decode_n; ++k) {
            Resample &r = rs[k];
            const bool y_bot = r.ystep >= (r.vs >> 1);
            coutput[k] = r.fn(r.linebuf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < d.comp[k].y) r.line1 += d.comp[k].w2;
            }
        }
        const uint8_t *y = coutput[0];
        if (n == 3) {
            if (d.img_n == 3) {
                if (is_rgb) {
                    for (int i = 0; i < d.img_x; ++i) { o[0] = y[i]; o[1] = coutput[1][i]; o[2] = coutput[2][i]; o += 3; }
                } else {
                    ycbcr_to_rgb_row(o, y, coutput[1], coutput[2], d.img_x);
                }
            } else {   // 4 components: CMYK / YCCK
                if (d.app14_transform == 0) {
                    for (int i = 0; i < d.img_x; ++i) {
                        const uint8_t m = coutput[3][i];
                        o[0] = blinn_8x8(coutput[0][i], m); o[1] = blinn_8x8(coutput[1][i], m); o[2] = blinn_8x8(coutput[2][i], m);
                        o += 3;
                    }
                } else if (d.app14_transform == 2) {
                    ycbcr_to_rgb_row(o, y, coutput[1], coutput[2], d.img_x);
                    for (int i = 0; i < d.img_x; ++i) {
                        const uint8_t m = coutput[3][i];
                        o[0] = blinn_8x8((uint8_t)(255 - o[0]), m); o[1] = blinn_8x8((uint8_t)(255 - o[1]), m); o[2] = blinn_8x8((uint8_t)(255 - o[2]), m);
                        o += 3;
                    }
                } else {
                    ycbcr_to_rgb_row(o, y, coutput[1], coutput[2], d.img_x);
                }
            }
        } else {
            memcpy(o, y, (size_t)d.img_x);
        }
    }
    return true;
}
