// image_io.cpp -- PNG (decode + encode, on zlib) and binary PNM decode for the drop-in CLIs.
// JPEG decode lives in jpeg_decode.cpp.  See image_io.h for the contract this mirrors
// (the reference's stbi_load(..., 0) / stbi_write_png calls in monolithic/src/main.c:21,41).
#include "image_io.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

bool gh_jpeg_decode(const uint8_t *data, size_t size, GhImage &out, std::string &err);   // jpeg_decode.cpp

namespace {

inline uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

// ---- PNG decode ---------------------------------------------------------------------------------

struct PngHeader {
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    int file_channels() const { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }
};

// undo the scanline filters of one (sub-)image in place; `raw` holds h rows of (1 + rowbytes)
bool unfilter(uint8_t *raw, size_t rowbytes, uint32_t h, int bpp, std::string &err)
{
    std::vector<uint8_t> zero(rowbytes, 0);
    const uint8_t *prev = zero.data();
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t *line = raw + (size_t)y * (rowbytes + 1);
        const int ft = line[0];
        uint8_t *cur = line + 1;
        switch (ft) {
        case 0: break;
        case 1: for (size_t i = bpp; i < rowbytes; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
        case 2: for (size_t i = 0; i < rowbytes; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]); break;
        case 3:
            for (size_t i = 0; i < rowbytes; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0;
                cur[i] = (uint8_t)(cur[i] + ((a + prev[i]) >> 1));
            }
            break;
        case 4:
            for (size_t i = 0; i < rowbytes; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0;
                const int c = i >= (size_t)bpp ? prev[i - bpp] : 0;
                cur[i] = (uint8_t)(cur[i] + paeth(a, prev[i], c));
            }
            break;
        default: err = "bad PNG filter type"; return false;
        }
        prev = cur;
    }
    return true;
}

// sample s (0-based, in units of `depth` bits) of an unfiltered row -> 16-bit-wide raw value
inline uint32_t sample_at(const uint8_t *row, size_t s, int depth)
{
    if (depth == 8) return row[s];
    if (depth == 16) return (uint32_t)row[2 * s] << 8 | row[2 * s + 1];
    const size_t bit = s * depth;
    return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
}

bool decode_png(const uint8_t *data, size_t size, GhImage &out, std::string &err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size < 8 || memcmp(data, sig, 8) != 0) { err = "not a PNG"; return false; }
    PngHeader hd;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false, have_trns = false;
    size_t pos = 8;
    while (pos + 8 <= size) {
        const uint32_t len = be32(data + pos);
        const uint8_t *type = data + pos + 4;
        const uint8_t *body = data + pos + 8;
        if ((size_t)len > size - pos - 8) { err = "truncated PNG chunk"; return false; }
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) { err = "bad IHDR"; return false; }
            hd.w = be32(body); hd.h = be32(body + 4);
            hd.depth = body[8]; hd.ctype = body[9]; hd.interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || hd.interlace > 1) { err = "unsupported PNG method"; return false; }
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
            have_trns = true;
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || hd.w == 0 || hd.h == 0) { err = "PNG without a valid IHDR"; return false; }
    const bool depth_ok = (hd.ctype == 0 && (hd.depth == 1 || hd.depth == 2 || hd.depth == 4 || hd.depth == 8 || hd.depth == 16)) ||
                          (hd.ctype == 3 && (hd.depth == 1 || hd.depth == 2 || hd.depth == 4 || hd.depth == 8)) ||
                          ((hd.ctype == 2 || hd.ctype == 4 || hd.ctype == 6) && (hd.depth == 8 || hd.depth == 16));
    if (!depth_ok) { err = "bad PNG colour type / bit depth"; return false; }
    if (hd.ctype == 3 && plte.size() < 3) { err = "palette PNG without PLTE"; return false; }
    if ((uint64_t)hd.w * hd.h > (1ull << 31)) { err = "PNG too large"; return false; }

    const int fch = hd.file_channels();
    const int bits_pp = fch * hd.depth;
    const int bpp = bits_pp >= 8 ? bits_pp / 8 : 1;

    // pass geometry (1 pass when not interlaced)
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1};
    static const int dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    const int npass = hd.interlace ? 7 : 1;
    size_t total = 0;
    size_t pass_off[7], pass_rb[7];
    uint32_t pass_w[7], pass_h[7];
    for (int p = 0; p < npass; ++p) {
        pass_w[p] = hd.interlace ? (hd.w - xs[p] + dx[p] - 1) / dx[p] : hd.w;
        pass_h[p] = hd.interlace ? (hd.h - ys[p] + dy[p] - 1) / dy[p] : hd.h;
        if (hd.interlace && ((int)hd.w <= xs[p] || (int)hd.h <= ys[p])) pass_w[p] = pass_h[p] = 0;
        pass_rb[p] = ((size_t)pass_w[p] * bits_pp + 7) / 8;
        pass_off[p] = total;
        if (pass_w[p] && pass_h[p]) total += (pass_rb[p] + 1) * pass_h[p];
    }

    std::vector<uint8_t> raw(total);
    {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) { err = "zlib init failed"; return false; }
        zs.next_in = idat.data();
        zs.avail_in = (uInt)idat.size();
        zs.next_out = raw.data();
        zs.avail_out = (uInt)raw.size();
        const int rc = inflate(&zs, Z_FINISH);
        const size_t got = zs.total_out;
        inflateEnd(&zs);
        if ((rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) || got < total) { err = "corrupt PNG data"; return false; }
    }

    // output channel count, stb-style
    int och;
    if (hd.ctype == 3) och = have_trns ? 4 : 3;
    else och = fch + ((have_trns && (hd.ctype == 0 || hd.ctype == 2)) ? 1 : 0);
    out.width = (int)hd.w; out.height = (int)hd.h; out.channels = och;
    out.pixels.assign((size_t)hd.w * hd.h * och, 0);

    uint32_t key[3] = {0, 0, 0};
    const bool keyed = have_trns && (hd.ctype == 0 || hd.ctype == 2);
    if (keyed) {
        if (trns.size() < (size_t)2 * fch) { err = "bad tRNS"; return false; }
        for (int c = 0; c < fch; ++c) key[c] = (uint32_t)trns[2 * c] << 8 | trns[2 * c + 1];
    }
    static const int scale_lo[5] = {0, 255, 85, 0, 17};   // grey depth 1,2,4 -> 0..255

    for (int p = 0; p < npass; ++p) {
        if (!pass_w[p] || !pass_h[p]) continue;
        uint8_t *base = raw.data() + pass_off[p];
        if (!unfilter(base, pass_rb[p], pass_h[p], bpp, err)) return false;
        for (uint32_t py = 0; py < pass_h[p]; ++py) {
            const uint8_t *row = base + (size_t)py * (pass_rb[p] + 1) + 1;
            const uint32_t oy = hd.interlace ? ys[p] + py * dy[p] : py;
            for (uint32_t px = 0; px < pass_w[p]; ++px) {
                const uint32_t ox = hd.interlace ? xs[p] + px * dx[p] : px;
                uint8_t *o = out.pixels.data() + ((size_t)oy * hd.w + ox) * och;
                if (hd.ctype == 3) {
                    const uint32_t idx = sample_at(row, px, hd.depth);
                    if ((size_t)idx * 3 + 2 < plte.size()) { o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2]; }
                    if (och == 4) o[3] = idx < trns.size() ? trns[idx] : 255;
                    continue;
                }
                bool is_key = keyed;
                for (int c = 0; c < fch; ++c) {
                    const uint32_t v = sample_at(row, (size_t)px * fch + c, hd.depth);
                    if (keyed && v != key[c]) is_key = false;
                    o[c] = hd.depth == 16 ? (uint8_t)(v >> 8) : hd.depth == 8 ? (uint8_t)v : (uint8_t)(v * scale_lo[hd.depth]);
                }
                if (keyed) o[fch] = is_key ? 0 : 255;
            }
        }
    }
    return true;
}

// ---- binary PNM (P5 / P6, maxval 255) ---------------------------------------------------------------

bool decode_pnm(const uint8_t *data, size_t size, GhImage &out, std::string &err)
{
    if (size < 3 || data[0] != 'P' || (data[1] != '5' && data[1] != '6')) { err = "not a binary PNM"; return false; }
    const int ch = data[1] == '5' ? 1 : 3;
    size_t pos = 2;
    long vals[3];
    for (int k = 0; k < 3; ++k) {
        for (;;) {
            while (pos < size && (data[pos] == ' ' || data[pos] == '\n' || data[pos] == '\r' || data[pos] == '\t')) ++pos;
            if (pos < size && data[pos] == '#') { while (pos < size && data[pos] != '\n') ++pos; continue; }
            break;
        }
        long v = 0; bool any = false;
        while (pos < size && data[pos] >= '0' && data[pos] <= '9') { v = v * 10 + (data[pos] - '0'); ++pos; any = true; if (v > (1 << 30)) break; }
        if (!any) { err = "bad PNM header"; return false; }
        vals[k] = v;
    }
    ++pos;   // single whitespace after maxval
    if (vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0) { err = "unsupported PNM (need maxval 255)"; return false; }
    const size_t need = (size_t)vals[0] * vals[1] * ch;
    if (pos + need > size) { err = "truncated PNM"; return false; }
    out.width = (int)vals[0]; out.height = (int)vals[1]; out.channels = ch;
    out.pixels.assign(data + pos, data + pos + need);
    return true;
}

void put_chunk(std::vector<uint8_t> &f, const char *type, const uint8_t *body, size_t len)
{
    const uint8_t l[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len};
    f.insert(f.end(), l, l + 4);
    const size_t start = f.size();
    f.insert(f.end(), type, type + 4);
    if (len) f.insert(f.end(), body, body + len);
    const uint32_t crc = (uint32_t)crc32(0L, f.data() + start, (uInt)(len + 4));
    const uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    f.insert(f.end(), c, c + 4);
}

}  // namespace

bool gh_image_load_memory(const uint8_t *data, size_t size, GhImage &out, std::string &err)
{
    if (size >= 8 && data[0] == 0x89 && data[1] == 'P') return decode_png(data, size, out, err);
    if (size >= 2 && data[0] == 0xFF && data[1] == 0xD8) return gh_jpeg_decode(data, size, out, err);
    if (size >= 2 && data[0] == 'P' && (data[1] == '5' || data[1] == '6')) return decode_pnm(data, size, out, err);
    err = "unknown image type";
    return false;
}

bool gh_image_load(const std::string &path, GhImage &out, std::string &err)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open file"; return false; }
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    return gh_image_load_memory(buf.data(), buf.size(), out, err);
}

// ---- PNG encoder: adaptive filtering + deflate, both parallel over bands of rows -----------------------------------------
// The reference writes its PNG with stb_image_write (monolithic/src/main.c:41), single-threaded, ~0.3 s for a 1920x1080
// frame -- far more than the kernels it wraps, and the service wrappers time the whole process.  Here the rows are cut
// into bands; every band is filtered (per row the filter with the smallest sum of |signed residual|, the libpng / stb
// heuristic) and deflated on its own thread as a raw deflate stream primed with the last 32 KiB of the previous band's
// filtered bytes (no ratio loss at the seams) and closed with a sync flush, so the pieces concatenate into one valid zlib
// stream (the pigz construction); the Adler-32 values are merged with adler32_combine.  Decoded pixels are what matters
// for parity (tests decode with PIL, with this repo's decoder and with the reference's stb decoder).
namespace {

struct PngBand {
    int y0 = 0, y1 = 0;
    std::vector<uint8_t> filtered;     // (rb + 1) bytes per row
    std::vector<uint8_t> z;            // raw deflate bytes
    uLong adler = 1;
    bool ok = true;
};

void png_filter_band(const uint8_t *pixels, size_t rb, int bpp, PngBand &b)
{
    const int rows = b.y1 - b.y0;
    b.filtered.resize((rb + 1) * (size_t)rows);
    std::vector<uint8_t> cand[5];
    for (auto &c : cand) c.resize(rb);
    const std::vector<uint8_t> zero(rb, 0);
    for (int y = b.y0; y < b.y1; ++y) {
        const uint8_t *cur = pixels + (size_t)y * rb;
        const uint8_t *prev = y ? cur - rb : zero.data();
        const size_t n0 = std::min<size_t>(bpp, rb);
        // first pixel: a = c = 0
        for (size_t i = 0; i < n0; ++i) {
            cand[0][i] = cur[i];
            cand[1][i] = cur[i];
            cand[2][i] = (uint8_t)(cur[i] - prev[i]);
            cand[3][i] = (uint8_t)(cur[i] - (prev[i] >> 1));
            cand[4][i] = (uint8_t)(cur[i] - prev[i]);                         // paeth(0, b, 0) = b
        }
        for (size_t i = n0; i < rb; ++i) cand[0][i] = cur[i];
        for (size_t i = n0; i < rb; ++i) cand[1][i] = (uint8_t)(cur[i] - cur[i - bpp]);
        for (size_t i = n0; i < rb; ++i) cand[2][i] = (uint8_t)(cur[i] - prev[i]);
        for (size_t i = n0; i < rb; ++i) cand[3][i] = (uint8_t)(cur[i] - ((cur[i - bpp] + prev[i]) >> 1));
        for (size_t i = n0; i < rb; ++i) cand[4][i] = (uint8_t)(cur[i] - paeth(cur[i - bpp], prev[i], prev[i - bpp]));
        long best_cost = -1;
        int best_ft = 0;
        for (int ft = 0; ft < 5; ++ft) {
            long cost = 0;
            const uint8_t *c = cand[ft].data();
            for (size_t i = 0; i < rb; ++i) cost += abs((int)(int8_t)c[i]);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_ft = ft; }
        }
        uint8_t *dst = b.filtered.data() + (size_t)(y - b.y0) * (rb + 1);
        dst[0] = (uint8_t)best_ft;
        memcpy(dst + 1, cand[best_ft].data(), rb);
    }
    b.adler = adler32(adler32(0L, Z_NULL, 0), b.filtered.data(), (uInt)b.filtered.size());
}

void png_deflate_band(PngBand &b, const PngBand *prevband, bool last, int level)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { b.ok = false; return; }
    if (prevband && !prevband->filtered.empty()) {
        const size_t n = std::min<size_t>(32768, prevband->filtered.size());
        deflateSetDictionary(&zs, prevband->filtered.data() + prevband->filtered.size() - n, (uInt)n);
    }
    b.z.resize(deflateBound(&zs, (uLong)b.filtered.size()) + 16);
    zs.next_in = b.filtered.data();
    zs.avail_in = (uInt)b.filtered.size();
    zs.next_out = b.z.data();
    zs.avail_out = (uInt)b.z.size();
    const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
    b.ok = last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0);
    b.z.resize(zs.total_out);
    deflateEnd(&zs);
}

// ---- the reference's own zlib stream, byte for byte ------------------------------------------------------------------------
// The reference writes its output with stb_image_write v1.16 (monolithic/src/main.c:41 -> stbi_write_png), whose built-in
// compressor is a small deterministic LZ77 + fixed-Huffman coder.  A PNG FILE equal to the reference's (the
// md5 of images/test_gray.png) needs the same parse of the filtered bytes, so this function restates that compressor's
// decisions (monolithic/include/stb_image_write.h:866-1015, public domain / MIT, Sean Barrett): one final fixed-Huffman
// block behind the header 78 5E; 3-byte hash (the shift/xor mix of :868-876) into 16384 buckets that keep the most recent
// 8..16 positions (a bucket that holds 16 drops its oldest 8 before the next insert); among the candidates inside the 32 KiB
// window the LAST one that is at least as long as the best so far wins (minimum 3, maximum 258); one-step lazy evaluation
// (a strictly longer match at the next byte turns this byte into a literal); the last 3 bytes are literals; stored blocks of
// <= 32767 bytes instead if that is smaller; Adler-32 trailer.  `quality` is stb's stbi_write_png_compression_level = 8.
std::vector<uint8_t> zlib_like_the_reference(const uint8_t *data, size_t len)
{
    constexpr int kQuality = 8, kBuckets = 16384, kSlots = 2 * kQuality;
    static const uint16_t len_base[] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 259};
    static const uint8_t len_extra[] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193,
                                         12289, 16385, 24577, 32768};
    static const uint8_t dist_extra[] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    std::vector<uint8_t> out;
    out.reserve(len / 2 + 64);
    out.push_back(0x78);
    out.push_back(0x5e);
    uint32_t acc = 0;
    int nacc = 0;
    auto put = [&](uint32_t bits, int n) {               // deflate packs from the least significant bit
        acc |= bits << nacc;
        nacc += n;
        while (nacc >= 8) { out.push_back((uint8_t)acc); acc >>= 8; nacc -= 8; }
    };
    auto put_code = [&](uint32_t code, int n) {          // Huffman codes go out most significant bit first
        uint32_t r = 0;
        for (int i = 0; i < n; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
        put(r, n);
    };
    auto put_symbol = [&](int sym) {                      // the fixed code of RFC 1951 3.2.6
        if (sym <= 143) put_code(0x30 + sym, 8);
        else if (sym <= 255) put_code(0x190 + sym - 144, 9);
        else if (sym <= 279) put_code(sym - 256, 7);
        else put_code(0xc0 + sym - 280, 8);
    };
    auto mix = [&](size_t at) {
        uint32_t h = data[at] + ((uint32_t)data[at + 1] << 8) + ((uint32_t)data[at + 2] << 16);
        h ^= h << 3; h += h >> 5; h ^= h << 4; h += h >> 17; h ^= h << 25; h += h >> 6;
        return (int)(h & (kBuckets - 1));
    };
    auto run = [&](size_t a, size_t b, size_t limit) {   // common prefix of data[a..] and data[b..], at most 258 and `limit`
        size_t n = 0;
        const size_t cap = limit < 258 ? limit : 258;
        while (n + 8 <= cap) {                            // eight bytes at a time (b + cap <= len: both reads stay inside the data)
            uint64_t x, y;
            memcpy(&x, data + a + n, 8);
            memcpy(&y, data + b + n, 8);
            if (x != y) return (int)(n + (size_t)(__builtin_ctzll(x ^ y) >> 3));
            n += 8;
        }
        while (n < cap && data[a + n] == data[b + n]) ++n;
        return (int)n;
    };
    std::vector<int64_t> slot((size_t)kBuckets * kSlots);
    std::vector<uint8_t> used(kBuckets, 0);
    put(1, 1);                                            // BFINAL
    put(1, 2);                                            // BTYPE = fixed Huffman
    size_t i = 0;
    while (i + 3 < len) {
        const int b = mix(i);
        int64_t *cand = &slot[(size_t)b * kSlots];
        int best = 3;
        int64_t where = -1;
        for (int j = 0; j < used[b]; ++j)
            if (cand[j] > (int64_t)i - 32768) {
                // (a candidate that differs at byte best-1 is shorter than the best so far: it cannot win -- skipped without counting)
                if (data[(size_t)cand[j] + best - 1] != data[i + best - 1]) continue;
                const int d = run((size_t)cand[j], i, len - i);
                if (d >= best) { best = d; where = cand[j]; }
            }
        if (used[b] == kSlots) {
            memmove(cand, cand + kQuality, sizeof(int64_t) * kQuality);
            used[b] = kQuality;
        }
        cand[used[b]++] = (int64_t)i;
        if (where >= 0) {                                 // lazy evaluation
            const int b2 = mix(i + 1);
            const int64_t *c2 = &slot[(size_t)b2 * kSlots];
            if (i + 1 + (size_t)best < len)               // a longer match at i+1 needs best+1 bytes to exist at all
                for (int j = 0; j < used[b2]; ++j)
                    if (c2[j] > (int64_t)i - 32767 && data[(size_t)c2[j] + best] == data[i + 1 + best] &&
                        run((size_t)c2[j], i + 1, len - i - 1) > best) { where = -1; break; }
        }
        if (where >= 0) {
            const int dist = (int)((int64_t)i - where);
            int j = 0;
            while (best > len_base[j + 1] - 1) ++j;
            put_symbol(j + 257);
            if (len_extra[j]) put((uint32_t)(best - len_base[j]), len_extra[j]);
            j = 0;
            while (dist > dist_base[j + 1] - 1) ++j;
            put_code((uint32_t)j, 5);
            if (dist_extra[j]) put((uint32_t)(dist - dist_base[j]), dist_extra[j]);
            i += (size_t)best;
        } else {
            put_symbol(data[i]);
            ++i;
        }
    }
    for (; i < len; ++i) put_symbol(data[i]);
    put_symbol(256);
    while (nacc) put(0, 1);
    if (out.size() > len + 2 + ((len + 32766) / 32767) * 5) {      // incompressible: stored blocks
        out.resize(2);
        for (size_t j = 0; j < len;) {
            const size_t blk = len - j > 32767 ? 32767 : len - j;
            out.push_back(len - j == blk ? 1 : 0);
            out.push_back((uint8_t)blk); out.push_back((uint8_t)(blk >> 8));
            out.push_back((uint8_t)~blk); out.push_back((uint8_t)(~blk >> 8));
            out.insert(out.end(), data + j, data + j + blk);
            j += blk;
        }
    }
    const uLong ad = adler32(adler32(0L, Z_NULL, 0), data, (uInt)len);
    for (int sh = 24; sh >= 0; sh -= 8) out.push_back((uint8_t)(ad >> sh));
    return out;
}

}  // namespace

bool gh_image_write_png(const std::string &path, int width, int height, int channels, const uint8_t *pixels, std::string &err)
{
    if (width <= 0 || height <= 0 || channels < 1 || channels > 4 || !pixels) { err = "bad image"; return false; }
    static const int ctype_of[5] = {0, 0, 4, 2, 6};
    const size_t rb = (size_t)width * channels;
    const char *lv = getenv("GH_PNG_LEVEL");
    const int level = lv ? atoi(lv) : 3;       // gray / edge frames: level 3 output is as small as level 6, at half the time
    // GRAYHIP_PNG=fast: band-parallel zlib deflate (any PNG reader decodes the same pixels, 3-10x faster to write);
    // default: the reference's own byte stream (one IDAT, stb's compressor restated above) -- files compare equal with cmp
    const char *pm = getenv("GRAYHIP_PNG");
    const bool want_reference = !(pm && strcmp(pm, "fast") == 0);
    const bool reference_bytes = want_reference && (rb + 1) * (size_t)height < (1ull << 31);
    // Beyond 2 GiB of filtered bytes the reference's writer cannot be followed (its int arithmetic overflows: stb_image_write.h's
    // stbi_zlib_compress takes `int data_len`), so the band-parallel encoder takes over: same pixels, other bytes.  Said once, on
    // request only -- stderr stays empty on success for the bench script.
    if (want_reference && !reference_bytes && getenv("GRAYHIP_VERBOSE"))
        fprintf(stderr, "grayhip: %s: %zu filtered bytes exceed the reference writer's 2 GiB range; written with the band-parallel encoder "
                        "(identical pixels, not the reference's byte stream)\n", path.c_str(), (rb + 1) * (size_t)height);

    // bands: >= 256 KiB of rows each, at most 4 per hardware thread; GH_PNG_THREADS caps the worker count
    const size_t band_bytes = 256 * 1024;
    int rows_per_band = (int)std::max<size_t>(1, (band_bytes + rb) / (rb + 1));
    int nbands = (height + rows_per_band - 1) / rows_per_band;
    const char *th = getenv("GH_PNG_THREADS");
    int nthreads = th ? atoi(th) : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    if (nthreads < 1) nthreads = 1;
    if (nbands > 4 * nthreads * 8) {        // huge frames: fewer, larger bands
        rows_per_band = (height + 4 * nthreads * 8 - 1) / (4 * nthreads * 8);
        nbands = (height + rows_per_band - 1) / rows_per_band;
    }
    if (nthreads > nbands) nthreads = nbands;
    if ((rb + 1) * (size_t)rows_per_band >= (1ull << 31)) { err = "image rows too long for the PNG encoder"; return false; }
    std::vector<PngBand> bands(nbands);
    for (int i = 0; i < nbands; ++i) { bands[i].y0 = i * rows_per_band; bands[i].y1 = std::min(height, (i + 1) * rows_per_band); }

    auto parallel_for = [&](auto &&fn) {
        std::atomic<int> next{0};
        auto worker = [&] { for (int i = next++; i < nbands; i = next++) fn(i); };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
    };
    parallel_for([&](int i) { png_filter_band(pixels, rb, channels, bands[i]); });     // same filter choice as the reference either way
    std::vector<uint8_t> z;
    if (reference_bytes) {
        std::vector<uint8_t> filtered;
        filtered.reserve((rb + 1) * (size_t)height);
        for (auto &b : bands) { filtered.insert(filtered.end(), b.filtered.begin(), b.filtered.end()); std::vector<uint8_t>().swap(b.filtered); }
        z = zlib_like_the_reference(filtered.data(), filtered.size());
    } else {
        parallel_for([&](int i) { png_deflate_band(bands[i], i ? &bands[i - 1] : nullptr, i == nbands - 1, level); });
        z.push_back(0x78);
        z.push_back(level >= 7 ? 0xDA : (level >= 6 ? 0x9C : (level >= 2 ? 0x5E : 0x01)));      // CMF/FLG: 32 KiB window, no dictionary
        uLong adler = adler32(0L, Z_NULL, 0);
        for (auto &b : bands) {
            if (!b.ok) { err = "zlib deflate failed"; return false; }
            z.insert(z.end(), b.z.begin(), b.z.end());
            adler = adler32_combine(adler, b.adler, (z_off_t)b.filtered.size());
            std::vector<uint8_t>().swap(b.z);
        }
        for (int s = 24; s >= 0; s -= 8) z.push_back((uint8_t)(adler >> s));
    }

    std::vector<uint8_t> file;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    file.insert(file.end(), sig, sig + 8);
    uint8_t ihdr[13] = {(uint8_t)(width >> 24), (uint8_t)(width >> 16), (uint8_t)(width >> 8), (uint8_t)width,
                        (uint8_t)(height >> 24), (uint8_t)(height >> 16), (uint8_t)(height >> 8), (uint8_t)height,
                        8, (uint8_t)ctype_of[channels], 0, 0, 0};
    put_chunk(file, "IHDR", ihdr, 13);
    // IDAT chunks of at most 1 GiB (chunk length is 31 bits)
    const size_t max_idat = 1u << 30;
    for (size_t off = 0; off < z.size() || off == 0; off += max_idat) {
        put_chunk(file, "IDAT", z.data() + off, std::min(max_idat, z.size() - off));
        if (z.empty()) break;
    }
    put_chunk(file, "IEND", nullptr, 0);

    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot open output file"; return false; }
    const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
    if (fclose(f) != 0 || !ok) { err = "short write"; return false; }
    return true;
}
