// image_decode.cpp -- see image_decode.h.  Plain C++17, no dependencies.
#include "image_decode.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace hostimg {

// =================================================================================================== inflate (RFC 1951)
namespace {

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t buf = 0; int cnt = 0;
    bool overrun = false;
    uint32_t bits(int n) {
        while (cnt < n) {
            uint32_t b = 0;
            if (p < end) b = *p++; else overrun = true;
            buf |= b << cnt; cnt += 8;
        }
        uint32_t v = n ? (buf & ((1u << n) - 1)) : 0;
        buf >>= n; cnt -= n;
        return v;
    }
    void align() { buf = 0; cnt = 0; }
};

struct Huff {                       // canonical code, decoded bit by bit against per-length counts (Mark Adler's "puff" scheme)
    uint16_t count[16]; uint16_t symbol[288];
    bool build(const uint8_t* lengths, int n) {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        if (count[0] == n) return true;             // no codes: legal for an unused distance tree
        int left = 1;
        for (int len = 1; len < 16; len++) { left <<= 1; left -= count[len]; if (left < 0) return false; }
        uint16_t offs[16]; offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + count[len];
        for (int i = 0; i < n; i++) if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            code |= (int)br.bits(1);
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
            if (br.overrun) return -1;
        }
        return -1;
    }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

bool inflate_codes(BitReader& br, const Huff& lit, const Huff& dist, std::vector<uint8_t>& out, std::string& err) {
    for (;;) {
        int sym = lit.decode(br);
        if (sym < 0) { err = "inflate: bad literal/length code"; return false; }
        if (sym < 256) { out.push_back((uint8_t)sym); continue; }
        if (sym == 256) return true;
        sym -= 257;
        if (sym >= 29) { err = "inflate: bad length symbol"; return false; }
        int len = kLenBase[sym] + (int)br.bits(kLenExtra[sym]);
        int ds = dist.decode(br);
        if (ds < 0 || ds >= 30) { err = "inflate: bad distance code"; return false; }
        size_t d = kDistBase[ds] + br.bits(kDistExtra[ds]);
        if (d > out.size()) { err = "inflate: distance beyond start of output"; return false; }
        size_t from = out.size() - d;
        for (int i = 0; i < len; i++) out.push_back(out[from + i]);
        if (br.overrun) { err = "inflate: truncated stream"; return false; }
    }
}

bool raw_inflate(BitReader& br, std::vector<uint8_t>& out, std::string& err) {
    static Huff fixed_lit, fixed_dist;
    static bool fixed_ready = false;
    if (!fixed_ready) {
        uint8_t l[288];
        int i = 0;
        for (; i < 144; i++) l[i] = 8;
        for (; i < 256; i++) l[i] = 9;
        for (; i < 280; i++) l[i] = 7;
        for (; i < 288; i++) l[i] = 8;
        fixed_lit.build(l, 288);
        uint8_t d[30];
        for (i = 0; i < 30; i++) d[i] = 5;
        fixed_dist.build(d, 30);
        fixed_ready = true;
    }
    int last;
    do {
        last = (int)br.bits(1);
        int type = (int)br.bits(2);
        if (br.overrun) { err = "inflate: truncated stream"; return false; }
        if (type == 0) {
            br.align();
            if (br.end - br.p < 4) { err = "inflate: truncated stored block"; return false; }
            uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((len ^ 0xffffu) != nlen) { err = "inflate: stored block length check"; return false; }
            if ((size_t)(br.end - br.p) < len) { err = "inflate: truncated stored block"; return false; }
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1) {
            if (!inflate_codes(br, fixed_lit, fixed_dist, out, err)) return false;
        } else if (type == 2) {
            int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            if (nlen > 286 || ndist > 30) { err = "inflate: bad dynamic header"; return false; }
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t lengths[320];
            memset(lengths, 0, sizeof(lengths));
            for (int i = 0; i < ncode; i++) lengths[order[i]] = (uint8_t)br.bits(3);
            Huff lencode;
            if (!lencode.build(lengths, 19)) { err = "inflate: bad code-length code"; return false; }
            uint8_t ll[320];
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = lencode.decode(br);
                if (sym < 0) { err = "inflate: bad code length symbol"; return false; }
                if (sym < 16) ll[idx++] = (uint8_t)sym;
                else {
                    int rep, val = 0;
                    if (sym == 16) { if (idx == 0) { err = "inflate: repeat with no previous length"; return false; } val = ll[idx - 1]; rep = 3 + (int)br.bits(2); }
                    else if (sym == 17) rep = 3 + (int)br.bits(3);
                    else rep = 11 + (int)br.bits(7);
                    if (idx + rep > nlen + ndist) { err = "inflate: too many code lengths"; return false; }
                    while (rep--) ll[idx++] = (uint8_t)val;
                }
                if (br.overrun) { err = "inflate: truncated stream"; return false; }
            }
            if (ll[256] == 0) { err = "inflate: no end-of-block code"; return false; }
            Huff lit, dist;
            if (!lit.build(ll, nlen) || !dist.build(ll + nlen, ndist)) { err = "inflate: over-subscribed code"; return false; }
            if (!inflate_codes(br, lit, dist, out, err)) return false;
        } else { err = "inflate: reserved block type"; return false; }
    } while (!last);
    return true;
}

uint32_t adler32(const uint8_t* p, size_t n) {
    uint32_t a = 1, b = 0;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        for (size_t i = 0; i < k; i++) { a += p[i]; b += a; }
        a %= 65521; b %= 65521;
        p += k; n -= k;
    }
    return (b << 16) | a;
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

}  // namespace

bool zlib_inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, std::string& err, size_t size_hint) {
    out.clear();
    if (size_hint) out.reserve(size_hint);
    if (n < 6) { err = "zlib: stream too short"; return false; }
    if ((src[0] & 0x0f) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) { err = "zlib: bad header"; return false; }
    BitReader br{src + 2, src + n};
    if (!raw_inflate(br, out, err)) return false;
    br.align();
    if (br.end - br.p >= 4 && be32(br.p) != adler32(out.data(), out.size())) { err = "zlib: adler32 mismatch"; return false; }
    return true;
}

bool read_file(const std::string& path, std::vector<uint8_t>& out, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if (got != out.size()) { err = "short read on " + path; return false; }
    return true;
}

// =================================================================================================== PNG
namespace {

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Undo the per-scanline filters of one (sub-)image in place; `raw` holds h rows of (1 + stride) bytes.
bool png_unfilter(uint8_t* raw, int h, size_t stride, int bpp, std::string& err) {
    for (int y = 0; y < h; y++) {
        uint8_t* row = raw + (size_t)y * (stride + 1);
        const int ft = row[0];
        uint8_t* cur = row + 1;
        const uint8_t* up = y ? row - stride : nullptr;          // previous row's pixel bytes (its filter byte precedes them)
        switch (ft) {
            case 0: break;
            case 1: for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
            case 2: if (up) for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + up[i]); break;
            case 3:
                for (size_t i = 0; i < stride; i++) {
                    int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                    cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
                }
                break;
            case 4:
                for (size_t i = 0; i < stride; i++) {
                    int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
                    cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
                }
                break;
            default: err = "png: bad filter type"; return false;
        }
    }
    return true;
}

}  // namespace

bool decode_png(const uint8_t* data, size_t n, Image8& out, std::string& err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 || memcmp(data, sig, 8)) { err = "png: bad signature"; return false; }
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat;
    uint8_t palette[256][4];
    int pal_n = 0;
    bool have_trns = false;
    uint16_t trns_key[3] = {0, 0, 0};
    for (int i = 0; i < 256; i++) { palette[i][0] = palette[i][1] = palette[i][2] = 0; palette[i][3] = 255; }
    bool seen_end = false;
    while (pos + 12 <= n && !seen_end) {
        uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if (len > n - pos - 12) { err = "png: truncated chunk"; return false; }
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) { err = "png: bad IHDR"; return false; }
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1) { err = "png: unsupported compression/filter/interlace method"; return false; }
        } else if (!memcmp(type, "PLTE", 4)) {
            pal_n = (int)(len / 3);
            if (pal_n > 256) pal_n = 256;
            for (int i = 0; i < pal_n; i++) { palette[i][0] = body[i * 3]; palette[i][1] = body[i * 3 + 1]; palette[i][2] = body[i * 3 + 2]; }
        } else if (!memcmp(type, "tRNS", 4)) {
            have_trns = true;
            if (ctype == 3) { for (uint32_t i = 0; i < len && i < 256; i++) palette[i][3] = body[i]; }
            else if (ctype == 0 && len >= 2) trns_key[0] = (uint16_t)be16(body);
            else if (ctype == 2 && len >= 6) { trns_key[0] = (uint16_t)be16(body); trns_key[1] = (uint16_t)be16(body + 2); trns_key[2] = (uint16_t)be16(body + 4); }
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) seen_end = true;
        pos += 12 + (size_t)len;
    }
    if (!w || !h || w > (1u << 15) || h > (1u << 15)) { err = "png: bad dimensions"; return false; }
    int channels;
    switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break;
        default: err = "png: bad colour type"; return false; }
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!depth_ok) { err = "png: bad bit depth"; return false; }
    const int bits_pp = depth * channels, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    std::vector<uint8_t> raw;
    if (!zlib_inflate(idat.data(), idat.size(), raw, err, ((size_t)w * bits_pp / 8 + 2) * h)) return false;

    out.width = (int)w; out.height = (int)h; out.source_bits = depth == 16 ? 16 : 8;
    out.rgba.assign((size_t)w * h * 4, 255);
    // one sample of the unfiltered row `row` at pixel x, channel c, as a 16-bit-or-less integer
    auto sample = [&](const uint8_t* row, uint32_t x, int c) -> uint32_t {
        if (depth == 8) return row[(size_t)x * channels + c];
        if (depth == 16) return be16(row + ((size_t)x * channels + c) * 2);
        uint32_t bit = x * depth;                              // depth < 8: single channel
        return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
    };
    static const uint8_t scale_lt8[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
    auto put = [&](const uint8_t* row, uint32_t x, uint32_t ox, uint32_t oy) {
        uint8_t* px = &out.rgba[((size_t)oy * w + ox) * 4];
        auto to8 = [&](uint32_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : (depth == 8 ? (uint8_t)v : (uint8_t)(v * scale_lt8[depth])); };
        if (ctype == 3) {
            uint32_t i = sample(row, x, 0);
            px[0] = palette[i & 255][0]; px[1] = palette[i & 255][1]; px[2] = palette[i & 255][2]; px[3] = palette[i & 255][3];
        } else if (ctype == 0) {
            uint32_t g = sample(row, x, 0);
            px[0] = px[1] = px[2] = to8(g);
            px[3] = (have_trns && g == trns_key[0]) ? 0 : 255;
        } else if (ctype == 4) {
            px[0] = px[1] = px[2] = to8(sample(row, x, 0)); px[3] = to8(sample(row, x, 1));
        } else if (ctype == 2) {
            uint32_t r = sample(row, x, 0), g = sample(row, x, 1), b = sample(row, x, 2);
            px[0] = to8(r); px[1] = to8(g); px[2] = to8(b);
            px[3] = (have_trns && r == trns_key[0] && g == trns_key[1] && b == trns_key[2]) ? 0 : 255;
        } else {
            px[0] = to8(sample(row, x, 0)); px[1] = to8(sample(row, x, 1)); px[2] = to8(sample(row, x, 2)); px[3] = to8(sample(row, x, 3));
        }
    };
    if (!interlace) {
        size_t stride = ((size_t)w * bits_pp + 7) / 8;
        if (raw.size() < (stride + 1) * h) { err = "png: not enough image data"; return false; }
        if (!png_unfilter(raw.data(), (int)h, stride, bpp, err)) return false;
        for (uint32_t y = 0; y < h; y++) {
            const uint8_t* row = raw.data() + (size_t)y * (stride + 1) + 1;
            for (uint32_t x = 0; x < w; x++) put(row, x, x, y);
        }
    } else {                                                    // Adam7
        static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t off = 0;
        for (int p = 0; p < 7; p++) {
            uint32_t pw = (w - xs[p] + dx[p] - 1) / dx[p], ph = (h - ys[p] + dy[p] - 1) / dy[p];
            if ((int)w <= xs[p] || (int)h <= ys[p] || !pw || !ph) continue;
            size_t stride = ((size_t)pw * bits_pp + 7) / 8;
            if (raw.size() < off + (stride + 1) * ph) { err = "png: not enough image data"; return false; }
            if (!png_unfilter(raw.data() + off, (int)ph, stride, bpp, err)) return false;
            for (uint32_t y = 0; y < ph; y++) {
                const uint8_t* row = raw.data() + off + (size_t)y * (stride + 1) + 1;
                for (uint32_t x = 0; x < pw; x++) put(row, x, xs[p] + x * dx[p], ys[p] + y * dy[p]);
            }
            off += (stride + 1) * ph;
        }
    }
    return true;
}

// =================================================================================================== JPEG (ITU T.81: baseline, extended sequential and progressive Huffman, 8 bit)
namespace {

const uint8_t kZigzag[64 + 15] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct JHuff {
    uint8_t bits[17]; uint8_t vals[256];
    int mincode[18], maxcode[18], valptr[18];
    bool present = false;
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct JComp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int w_blocks = 0, h_blocks = 0;          // allocated (MCU-padded) block grid
    int bw = 0, bh = 0;                      // blocks that cover the component's own size (non-interleaved scans)
    std::vector<int16_t> coef;               // progressive: 64 per block
    std::vector<uint8_t> plane;              // w_blocks*8 x h_blocks*8
    int dc_pred = 0;
};

struct JDec {
    const uint8_t* p; const uint8_t* end;
    uint32_t bitbuf = 0; int bitcnt = 0;
    int marker = -1;                         // marker met while filling bits
    bool nomore = false;
    uint16_t quant[4][64];
    JHuff dc[4], ac[4];
    JComp comp[4];
    int ncomp = 0, width = 0, height = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false;
    int restart_interval = 0;
    int eobrun = 0;
    int app14_transform = -1;
    bool jfif = false;

    void fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!nomore && p < end) {
                b = *p++;
                if (b == 0xff) {
                    int c = p < end ? *p++ : 0;
                    while (c == 0xff && p < end) c = *p++;
                    if (c != 0) { marker = c; nomore = true; b = 0; }
                }
            } else nomore = true;
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int getbits(int n) {
        if (!n) return 0;
        if (bitcnt < n) fill();
        int v = (int)(bitbuf >> (32 - n));
        bitbuf <<= n; bitcnt -= n;
        return v;
    }
    int getbit() { return getbits(1); }
    int decode(const JHuff& h) {
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | getbit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        return -1;
    }
    static int extend(int v, int n) { return n && v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }
    int receive_extend(int n) { return extend(getbits(n), n); }
    void reset() { bitbuf = 0; bitcnt = 0; marker = -1; nomore = false; eobrun = 0; for (int i = 0; i < 4; i++) comp[i].dc_pred = 0; }
};

inline uint8_t clamp8(long long x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// stb_image's integer IDCT (the libjpeg "islow" factorisation with 12-bit constants), dequantised input.  The arithmetic is
// 64-bit so that the coefficients of a damaged file cannot overflow it (a valid stream stays far inside 32 bits: same results).
#define JF2F(x) ((long long)(((x) * 4096 + 0.5)))
#define JFSH(x) ((x) * 4096)
#define JIDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                            \
    long long t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                           \
    p2 = s2; p3 = s6;                                                                                       \
    p1 = (p2 + p3) * JF2F(0.5411961f);                                                                      \
    t2 = p1 + p3 * JF2F(-1.847759065f);                                                                     \
    t3 = p1 + p2 * JF2F(0.765366865f);                                                                      \
    p2 = s0; p3 = s4;                                                                                       \
    t0 = JFSH(p2 + p3); t1 = JFSH(p2 - p3);                                                                 \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                                 \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                                     \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                                                 \
    p5 = (p3 + p4) * JF2F(1.175875602f);                                                                    \
    t0 = t0 * JF2F(0.298631336f); t1 = t1 * JF2F(2.053119869f); t2 = t2 * JF2F(3.072711026f); t3 = t3 * JF2F(1.501321110f); \
    p1 = p5 + p1 * JF2F(-0.899976223f); p2 = p5 + p2 * JF2F(-2.562915447f);                                 \
    p3 = p3 * JF2F(-1.961570560f); p4 = p4 * JF2F(-0.390180644f);                                           \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

void idct_block(uint8_t* out, int out_stride, const int16_t* data) {
    long long val[64], *v = val;
    const int16_t* d = data;
    for (int i = 0; i < 8; ++i, ++d, ++v) {
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
            long long dcterm = d[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
        } else {
            JIDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    v = val;
    uint8_t* o = out;
    for (int i = 0; i < 8; ++i, v += 8, o += out_stride) {
        JIDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
        o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
    }
}

bool jpeg_block_baseline(JDec& j, JComp& c, int16_t* blk, std::string& err) {
    memset(blk, 0, 64 * sizeof(int16_t));
    int t = j.decode(j.dc[c.td]);
    if (t < 0 || t > 15) { err = "jpeg: bad DC code"; return false; }
    int diff = t ? j.receive_extend(t) : 0;
    c.dc_pred += diff;
    blk[0] = (int16_t)(c.dc_pred * j.quant[c.tq][0]);
    for (int k = 1; k < 64;) {
        int rs = j.decode(j.ac[c.ta]);
        if (rs < 0) { err = "jpeg: bad AC code"; return false; }
        int r = rs >> 4, s = rs & 15;
        if (s == 0) { if (r != 15) break; k += 16; continue; }
        k += r;
        if (k > 63) { err = "jpeg: AC index overflow"; return false; }
        int z = kZigzag[k];
        blk[z] = (int16_t)(j.receive_extend(s) * j.quant[c.tq][z]);
        k++;
    }
    return true;
}

bool jpeg_block_prog_dc(JDec& j, JComp& c, int16_t* blk, int ah, int al, std::string& err) {
    if (ah == 0) {
        int t = j.decode(j.dc[c.td]);
        if (t < 0 || t > 15) { err = "jpeg: bad DC code"; return false; }
        int diff = t ? j.receive_extend(t) : 0;
        c.dc_pred += diff;
        blk[0] = (int16_t)(c.dc_pred * (1 << al));
    } else if (j.getbit()) blk[0] += (int16_t)(1 << al);
    return true;
}

bool jpeg_block_prog_ac(JDec& j, JComp& c, int16_t* blk, int ss, int se, int ah, int al, std::string& err) {
    const JHuff& h = j.ac[c.ta];
    if (ah == 0) {
        if (j.eobrun) { j.eobrun--; return true; }
        for (int k = ss; k <= se;) {
            int rs = j.decode(h);
            if (rs < 0) { err = "jpeg: bad AC code"; return false; }
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) { j.eobrun = (1 << r); if (r) j.eobrun += j.getbits(r); j.eobrun--; break; }
                k += 16;
            } else {
                k += r;
                if (k > 63) { err = "jpeg: AC index overflow"; return false; }
                blk[kZigzag[k]] = (int16_t)(j.receive_extend(s) * (1 << al));
                k++;
            }
        }
    } else {                                   // refinement (T.81 G.1.2.3)
        const int16_t bit = (int16_t)(1 << al);
        auto refine = [&](int16_t& v) { if (v != 0 && j.getbit() && (v & bit) == 0) v += v > 0 ? bit : (int16_t)-bit; };
        if (j.eobrun) {
            j.eobrun--;
            for (int k = ss; k <= se; k++) refine(blk[kZigzag[k]]);
            return true;
        }
        int k = ss;
        while (k <= se) {
            int rs = j.decode(h);
            if (rs < 0) { err = "jpeg: bad AC code"; return false; }
            int r = rs >> 4, s = rs & 15;
            int newval = 0;
            if (s == 0) {
                if (r < 15) { j.eobrun = (1 << r) - 1; if (r) j.eobrun += j.getbits(r); r = 64; }   // end of band for this block: only refine what is left
            } else {
                if (s != 1) { err = "jpeg: bad refinement code"; return false; }
                newval = j.getbit() ? bit : -bit;
            }
            while (k <= se) {
                int16_t& v = blk[kZigzag[k++]];
                if (v != 0) refine(v);
                else {
                    if (r == 0) { v = (int16_t)newval; break; }
                    r--;
                }
            }
        }
    }
    return true;
}

}  // namespace

bool decode_jpeg(const uint8_t* data, size_t n, Image8& out, std::string& err) {
    if (n < 4 || data[0] != 0xff || data[1] != 0xd8) { err = "jpeg: bad signature"; return false; }
    JDec* jp = new JDec();
    struct Guard { JDec* p; ~Guard() { delete p; } } guard{jp};
    JDec& j = *jp;
    memset(j.quant, 0, sizeof(j.quant));
    size_t pos = 2;
    bool got_sof = false, done = false;
    auto alloc = [&]() {
        for (int i = 0; i < j.ncomp; i++) {
            JComp& c = j.comp[i];
            c.w_blocks = j.mcux * c.h; c.h_blocks = j.mcuy * c.v;
            c.bw = ((j.width * c.h + j.hmax - 1) / j.hmax + 7) / 8;
            c.bh = ((j.height * c.v + j.vmax - 1) / j.vmax + 7) / 8;
            c.plane.assign((size_t)c.w_blocks * 8 * c.h_blocks * 8, 0);
            if (j.progressive) c.coef.assign((size_t)c.w_blocks * c.h_blocks * 64, 0);
        }
    };
    while (!done && pos + 4 <= n) {
        if (data[pos] != 0xff) { pos++; continue; }
        int m = data[pos + 1];
        if (m == 0xff) { pos++; continue; }
        pos += 2;
        if (m == 0xd9) break;                                       // EOI
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;        // standalone
        if (pos + 2 > n) break;
        size_t len = be16(data + pos);
        if (len < 2 || pos + len > n) { err = "jpeg: truncated segment"; return false; }
        const uint8_t* s = data + pos + 2;
        const size_t sl = len - 2;
        switch (m) {
            case 0xdb: {                                            // DQT
                size_t k = 0;
                while (k < sl) {
                    int pq = s[k] >> 4, tq = s[k] & 15;
                    k++;
                    if (tq > 3) { err = "jpeg: bad DQT"; return false; }
                    for (int i = 0; i < 64; i++) {
                        if (k + (pq ? 2 : 1) > sl) { err = "jpeg: bad DQT"; return false; }
                        j.quant[tq][kZigzag[i]] = pq ? (uint16_t)be16(s + k) : s[k];
                        k += pq ? 2 : 1;
                    }
                }
            } break;
            case 0xc4: {                                            // DHT
                size_t k = 0;
                while (k + 17 <= sl) {
                    int tc = s[k] >> 4, th = s[k] & 15;
                    if (tc > 1 || th > 3) { err = "jpeg: bad DHT"; return false; }
                    JHuff& h = tc ? j.ac[th] : j.dc[th];
                    int total = 0;
                    h.bits[0] = 0;
                    for (int i = 1; i <= 16; i++) { h.bits[i] = s[k + i]; total += h.bits[i]; }
                    k += 17;
                    if (total > 256 || k + total > sl) { err = "jpeg: bad DHT"; return false; }
                    memcpy(h.vals, s + k, total);
                    k += total;
                    h.build();
                }
            } break;
            case 0xc0: case 0xc1: case 0xc2: {                      // SOF0/1/2
                if (sl < 6 || s[0] != 8) { err = "jpeg: only 8-bit precision is supported"; return false; }
                j.progressive = m == 0xc2;
                j.height = (int)be16(s + 1); j.width = (int)be16(s + 3); j.ncomp = s[5];
                if (!j.width || !j.height || (j.ncomp != 1 && j.ncomp != 3)) { err = "jpeg: unsupported component count or size"; return false; }
                if (sl < (size_t)6 + 3 * j.ncomp) { err = "jpeg: bad SOF"; return false; }
                j.hmax = j.vmax = 1;
                for (int i = 0; i < j.ncomp; i++) {
                    JComp& c = j.comp[i];
                    c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i];
                    if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) { err = "jpeg: bad sampling factors"; return false; }
                    if (c.h > j.hmax) j.hmax = c.h;
                    if (c.v > j.vmax) j.vmax = c.v;
                }
                for (int i = 0; i < j.ncomp; i++) if (j.hmax % j.comp[i].h || j.vmax % j.comp[i].v) { err = "jpeg: non-integer sampling ratio"; return false; }
                j.mcux = (j.width + 8 * j.hmax - 1) / (8 * j.hmax); j.mcuy = (j.height + 8 * j.vmax - 1) / (8 * j.vmax);
                alloc();
                got_sof = true;
            } break;
            case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                err = "jpeg: unsupported coding process (lossless / hierarchical / arithmetic)"; return false;
            case 0xdd: if (sl >= 2) j.restart_interval = (int)be16(s); break;
            case 0xee: if (sl >= 12 && !memcmp(s, "Adobe", 5)) j.app14_transform = s[11]; break;
            case 0xe0: if (sl >= 5 && !memcmp(s, "JFIF", 5)) j.jfif = true; break;
            case 0xda: {                                            // SOS + entropy-coded data
                if (!got_sof) { err = "jpeg: SOS before SOF"; return false; }
                int ns = s[0];
                if (ns < 1 || ns > j.ncomp || sl < (size_t)1 + 2 * ns + 3) { err = "jpeg: bad SOS"; return false; }
                int order[4];
                for (int i = 0; i < ns; i++) {
                    int cid = s[1 + 2 * i], which = -1;
                    for (int c = 0; c < j.ncomp; c++) if (j.comp[c].id == cid) which = c;
                    if (which < 0) { err = "jpeg: SOS names an unknown component"; return false; }
                    order[i] = which;
                    j.comp[which].td = s[2 + 2 * i] >> 4; j.comp[which].ta = s[2 + 2 * i] & 15;
                    if (j.comp[which].td > 3 || j.comp[which].ta > 3) { err = "jpeg: bad table selector"; return false; }
                }
                int ss = s[1 + 2 * ns], se = s[2 + 2 * ns], ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
                if (!j.progressive) { ss = 0; se = 63; ah = al = 0; }
                else if (ss > 63 || se > 63 || ss > se || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) { err = "jpeg: bad spectral selection"; return false; }
                j.p = data + pos + len; j.end = data + n;
                j.reset();
                int todo = j.restart_interval ? j.restart_interval : 0x7fffffff;
                int16_t tmp[64];
                auto restart_check = [&]() -> bool {                 // between MCUs
                    if (--todo > 0) return true;
                    if (j.bitcnt < 24) j.fill();
                    if (j.marker >= 0xd0 && j.marker <= 0xd7) { j.reset(); todo = j.restart_interval ? j.restart_interval : 0x7fffffff; return true; }
                    return j.restart_interval == 0;
                };
                auto do_block = [&](JComp& c, int bx, int by) -> bool {
                    if (!j.progressive) {
                        if (!jpeg_block_baseline(j, c, tmp, err)) return false;
                        idct_block(&c.plane[((size_t)by * 8 * c.w_blocks + bx) * 8], c.w_blocks * 8, tmp);
                        return true;
                    }
                    int16_t* blk = &c.coef[((size_t)by * c.w_blocks + bx) * 64];
                    return ss == 0 ? jpeg_block_prog_dc(j, c, blk, ah, al, err) : jpeg_block_prog_ac(j, c, blk, ss, se, ah, al, err);
                };
                bool ok = true;
                if (ns == 1) {                                      // non-interleaved: the component's own block raster
                    JComp& c = j.comp[order[0]];
                    for (int by = 0; by < c.bh && ok; by++)
                        for (int bx = 0; bx < c.bw && ok; bx++) {
                            if (!do_block(c, bx, by)) return false;
                            if (!restart_check()) ok = false;
                        }
                } else {
                    for (int my = 0; my < j.mcuy && ok; my++)
                        for (int mx = 0; mx < j.mcux && ok; mx++) {
                            for (int i = 0; i < ns; i++) {
                                JComp& c = j.comp[order[i]];
                                for (int v = 0; v < c.v; v++)
                                    for (int h = 0; h < c.h; h++)
                                        if (!do_block(c, mx * c.h + h, my * c.v + v)) return false;
                            }
                            if (!restart_check()) ok = false;
                        }
                }
                // resume marker parsing after the entropy-coded segment
                size_t q = (size_t)(j.p - data);
                if (j.marker >= 0) {                                // p is just past the marker byte
                    q -= 2;
                } else {
                    while (q + 1 < n && !(data[q] == 0xff && data[q + 1] != 0 && !(data[q + 1] >= 0xd0 && data[q + 1] <= 0xd7))) q++;
                }
                pos = q;
                len = 0;
                if (!j.progressive) done = true;                    // baseline: one frame, one (set of) scan(s); keep going only if more SOS follow
                if (!j.progressive) {
                    // a baseline file may still split components over several scans: continue parsing until EOI
                    done = false;
                }
            } break;
            default: break;
        }
        pos += len;
    }
    if (!got_sof) { err = "jpeg: no frame header"; return false; }
    if (j.progressive) {
        int16_t tmp[64];
        for (int i = 0; i < j.ncomp; i++) {
            JComp& c = j.comp[i];
            for (int by = 0; by < c.h_blocks; by++)
                for (int bx = 0; bx < c.w_blocks; bx++) {
                    const int16_t* blk = &c.coef[((size_t)by * c.w_blocks + bx) * 64];
                    for (int k = 0; k < 64; k++) tmp[k] = (int16_t)(blk[k] * j.quant[c.tq][k]);
                    idct_block(&c.plane[((size_t)by * 8 * c.w_blocks + bx) * 8], c.w_blocks * 8, tmp);
                }
        }
    }
    // ---- upsample (stb_image's filters) and colour-convert
    const int W = j.width, H = j.height;
    out.width = W; out.height = H; out.source_bits = 8;
    out.rgba.assign((size_t)W * H * 4, 255);
    std::vector<uint8_t> line[3];
    for (int i = 0; i < j.ncomp; i++) line[i].assign((size_t)W + 3 + 8 * 4, 0);
    struct Res { int hs, vs, ystep, ypos, w_lores; const uint8_t *line0, *line1; } rs[3];
    for (int i = 0; i < j.ncomp; i++) {
        JComp& c = j.comp[i];
        rs[i].hs = j.hmax / c.h; rs[i].vs = j.vmax / c.v; rs[i].ystep = rs[i].vs >> 1; rs[i].ypos = 0;
        rs[i].w_lores = (W + rs[i].hs - 1) / rs[i].hs;
        rs[i].line0 = rs[i].line1 = c.plane.data();
    }
    auto div4 = [](int x) { return (uint8_t)(x >> 2); };
    auto div16 = [](int x) { return (uint8_t)(x >> 4); };
    for (int y = 0; y < H; y++) {
        const uint8_t* src[3] = {nullptr, nullptr, nullptr};
        for (int i = 0; i < j.ncomp; i++) {
            JComp& c = j.comp[i];
            Res& r = rs[i];
            const bool y_bot = r.ystep >= (r.vs >> 1);
            const uint8_t* in_near = y_bot ? r.line1 : r.line0;
            const uint8_t* in_far = y_bot ? r.line0 : r.line1;
            uint8_t* o = line[i].data();
            const int w = r.w_lores;
            if (r.hs == 1 && r.vs == 1) src[i] = in_near;
            else if (r.hs == 1 && r.vs == 2) { for (int x = 0; x < w; x++) o[x] = div4(3 * in_near[x] + in_far[x] + 2); src[i] = o; }
            else if (r.hs == 2 && r.vs == 1) {
                if (w == 1) o[0] = o[1] = in_near[0];
                else {
                    o[0] = in_near[0];
                    o[1] = div4(in_near[0] * 3 + in_near[1] + 2);
                    int x;
                    for (x = 1; x < w - 1; ++x) { int nn = 3 * in_near[x] + 2; o[x * 2 + 0] = div4(nn + in_near[x - 1]); o[x * 2 + 1] = div4(nn + in_near[x + 1]); }
                    o[x * 2 + 0] = div4(in_near[w - 2] * 3 + in_near[w - 1] + 2);
                    o[x * 2 + 1] = in_near[w - 1];
                }
                src[i] = o;
            } else if (r.hs == 2 && r.vs == 2) {
                if (w == 1) o[0] = o[1] = div4(3 * in_near[0] + in_far[0] + 2);
                else {
                    int t1 = 3 * in_near[0] + in_far[0], t0;
                    o[0] = div4(t1 + 2);
                    for (int x = 1; x < w; ++x) {
                        t0 = t1; t1 = 3 * in_near[x] + in_far[x];
                        o[x * 2 - 1] = div16(3 * t0 + t1 + 8);
                        o[x * 2] = div16(3 * t1 + t0 + 8);
                    }
                    o[w * 2 - 1] = div4(t1 + 2);
                }
                src[i] = o;
            } else { for (int x = 0; x < w; x++) for (int k = 0; k < r.hs; k++) o[x * r.hs + k] = in_near[x]; src[i] = o; }
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < (j.height * c.v + j.vmax - 1) / j.vmax) r.line1 += (size_t)c.w_blocks * 8;
            }
        }
        uint8_t* px = &out.rgba[(size_t)y * W * 4];
        if (j.ncomp == 1) { for (int x = 0; x < W; x++) { px[x * 4] = px[x * 4 + 1] = px[x * 4 + 2] = src[0][x]; } }
        else {
            const bool is_rgb = (j.comp[0].id == 'R' && j.comp[1].id == 'G' && j.comp[2].id == 'B') || (j.app14_transform == 0 && !j.jfif);
            for (int x = 0; x < W; x++) {
                if (is_rgb) { px[x * 4] = src[0][x]; px[x * 4 + 1] = src[1][x]; px[x * 4 + 2] = src[2][x]; continue; }
#define JFLOAT2FIXED(v) (((int)((v) * 4096.0f + 0.5f)) << 8)
                int y_fixed = (src[0][x] << 20) + (1 << 19);
                int cr = src[2][x] - 128, cb = src[1][x] - 128;
                int r = y_fixed + cr * JFLOAT2FIXED(1.40200f);
                int g = y_fixed + (cr * -JFLOAT2FIXED(0.71414f)) + ((cb * -JFLOAT2FIXED(0.34414f)) & 0xffff0000);
                int b = y_fixed + cb * JFLOAT2FIXED(1.77200f);
                r >>= 20; g >>= 20; b >>= 20;
                px[x * 4] = clamp8(r); px[x * 4 + 1] = clamp8(g); px[x * 4 + 2] = clamp8(b);
            }
        }
    }
    return true;
}

bool decode_image8(const uint8_t* data, size_t n, Image8& out, std::string& err) {
    if (n >= 8 && data[0] == 0x89 && data[1] == 'P') return decode_png(data, n, out, err);
    if (n >= 3 && data[0] == 0xff && data[1] == 0xd8) return decode_jpeg(data, n, out, err);
    err = "image: neither PNG nor JPEG";
    return false;
}

// =================================================================================================== Radiance .hdr (RGBE)
bool decode_hdr(const uint8_t* data, size_t n, ImageF& out, std::string& err) {
    size_t pos = 0;
    auto getline = [&](std::string& s) -> bool {
        s.clear();
        if (pos >= n) return false;
        while (pos < n && data[pos] != '\n') s += (char)data[pos++];
        if (pos < n) pos++;
        return true;
    };
    std::string ln;
    if (!getline(ln) || (ln != "#?RADIANCE" && ln != "#?RGBE")) { err = "hdr: bad signature"; return false; }
    bool fmt = false;
    for (;;) {
        if (!getline(ln)) { err = "hdr: truncated header"; return false; }
        if (ln.empty()) break;
        if (ln == "FORMAT=32-bit_rle_rgbe") fmt = true;
    }
    if (!fmt) { err = "hdr: unsupported format"; return false; }
    if (!getline(ln)) { err = "hdr: missing resolution"; return false; }
    int w = 0, h = 0;
    if (sscanf(ln.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24)) { err = "hdr: unsupported data layout"; return false; }
    out.width = w; out.height = h; out.half_source = false;
    out.rgb.assign((size_t)w * h * 3, 0.f);
    auto convert = [](float* o, const uint8_t* in) {              // stbi__hdr_convert, 3 components
        if (in[3] != 0) {
            float f1 = ldexpf(1.0f, (int)in[3] - (128 + 8));
            o[0] = in[0] * f1; o[1] = in[1] * f1; o[2] = in[2] * f1;
        } else o[0] = o[1] = o[2] = 0;
    };
    std::vector<uint8_t> scan((size_t)w * 4);
    bool flat = w < 8 || w >= 32768;
    for (int y = 0; y < h && !flat; y++) {
        if (pos + 4 > n) { err = "hdr: truncated data"; return false; }
        int c1 = data[pos], c2 = data[pos + 1], len = data[pos + 2];
        if (c1 != 2 || c2 != 2 || (len & 0x80)) {
            if (y != 0) { err = "hdr: mixed flat / RLE scanlines"; return false; }
            flat = true;                                            // not run-length encoded: the whole image is flat RGBE
            break;
        }
        len = (len << 8) | data[pos + 3];
        pos += 4;
        if (len != w) { err = "hdr: scanline width mismatch"; return false; }
        for (int k = 0; k < 4; k++) {
            int i = 0;
            while (i < w) {
                if (pos >= n) { err = "hdr: truncated data"; return false; }
                int count = data[pos++];
                if (count > 128) {
                    count -= 128;
                    if (count == 0 || count > w - i || pos >= n) { err = "hdr: bad run"; return false; }
                    uint8_t v = data[pos++];
                    for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = v;
                } else {
                    if (count == 0 || count > w - i || pos + count > n) { err = "hdr: bad literal run"; return false; }
                    for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = data[pos++];
                }
            }
        }
        for (int x = 0; x < w; x++) convert(&out.rgb[((size_t)y * w + x) * 3], &scan[(size_t)x * 4]);
    }
    if (flat) {
        if (pos + (size_t)w * h * 4 > n) { err = "hdr: truncated data"; return false; }
        for (size_t i = 0; i < (size_t)w * h; i++) convert(&out.rgb[i * 3], data + pos + i * 4);
    }
    return true;
}

// =================================================================================================== OpenEXR (single-part scan-line images)
float half_to_float(uint16_t h) {
    uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023, bits;
    if (e == 0) {
        if (m == 0) bits = s;
        else { int sh = 0; while (!(m & 1024)) { m <<= 1; sh++; } m &= 1023; bits = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13); }
    } else if (e == 31) bits = s | 0x7f800000u | (m << 13);
    else bits = s | ((e + 127 - 15) << 23) | (m << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

bool decode_exr(const uint8_t* data, size_t n, ImageF& out, std::string& err, bool single_channel) {
    auto le32 = [&](size_t p) -> uint32_t { return (uint32_t)data[p] | ((uint32_t)data[p + 1] << 8) | ((uint32_t)data[p + 2] << 16) | ((uint32_t)data[p + 3] << 24); };
    auto le64 = [&](size_t p) -> uint64_t { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); };
    if (n < 8 || le32(0) != 20000630u) { err = "exr: bad magic"; return false; }
    uint32_t ver = le32(4);
    if ((ver & 0xff) != 2 || (ver & 0x200) || (ver & 0x800) || (ver & 0x1000)) { err = "exr: unsupported version / tiled / deep / multipart"; return false; }   // EnvironmentMap.cpp:162-165
    size_t pos = 8;
    struct Chan { std::string name; int type; int xs, ys; };
    std::vector<Chan> chans;
    int compression = -1, dw[4] = {0, 0, -1, -1};
    bool have_dw = false, increasing_y = true;
    auto cstr = [&](std::string& s) -> bool { s.clear(); while (pos < n && data[pos]) s += (char)data[pos++]; if (pos >= n) return false; pos++; return true; };
    for (;;) {
        std::string name, type;
        if (!cstr(name)) { err = "exr: truncated header"; return false; }
        if (name.empty()) break;
        if (!cstr(type) || pos + 4 > n) { err = "exr: truncated header"; return false; }
        uint32_t size = le32(pos);
        pos += 4;
        if (size > n - pos) { err = "exr: truncated attribute"; return false; }
        const size_t a = pos;
        if (name == "channels" && type == "chlist") {
            size_t save = pos;
            for (;;) {
                std::string cn;
                if (!cstr(cn)) { err = "exr: bad channel list"; return false; }
                if (cn.empty()) break;
                if (pos + 16 > n) { err = "exr: bad channel list"; return false; }
                chans.push_back({cn, (int)le32(pos), (int)le32(pos + 8), (int)le32(pos + 12)});
                pos += 16;
            }
            pos = save;
        } else if (name == "compression" && size >= 1) compression = data[a];
        else if (name == "dataWindow" && size >= 16) { for (int i = 0; i < 4; i++) dw[i] = (int)le32(a + 4 * i); have_dw = true; }
        else if (name == "lineOrder" && size >= 1) increasing_y = data[a] != 1;
        pos = a + size;
    }
    if (!have_dw || chans.empty() || compression < 0) { err = "exr: missing required attributes"; return false; }
    if (compression > 9) { err = "exr: unknown compression type"; return false; }
    for (auto& c : chans) if (c.xs != 1 || c.ys != 1) { err = "exr: sub-sampled channels are unsupported"; return false; }
    const long long w64 = (long long)dw[2] - dw[0] + 1, h64 = (long long)dw[3] - dw[1] + 1;
    if (w64 <= 0 || h64 <= 0 || w64 > (1 << 16) || h64 > (1 << 16)) { err = "exr: bad data window"; return false; }
    const int w = (int)w64, h = (int)h64;
    if (chans[0].type != 1 && chans[0].type != 2) { err = "exr: unsupported pixel type"; return false; }              // EnvironmentMap.cpp:177-185
    int ci[3] = {-1, -1, -1};
    for (size_t i = 0; i < chans.size(); i++) { if (chans[i].name == "R") ci[0] = (int)i; else if (chans[i].name == "G") ci[1] = (int)i; else if (chans[i].name == "B") ci[2] = (int)i; }
    if (single_channel) {                                           // LoadLookupTables (GpuResources.cpp:92-93): exactly one HALF channel, any name
        if (chans.size() != 1 || chans[0].type != 1) { err = "exr: lookup tables must have exactly one HALF channel"; return false; }
        ci[0] = ci[1] = ci[2] = 0;
    }
    if (ci[0] < 0 || ci[1] < 0 || ci[2] < 0) { err = "exr: missing R, G or B channel"; return false; }              // :199-203
    std::vector<size_t> coff(chans.size());
    size_t bytes_per_px_row = 0;                                    // bytes of one pixel column across all channels
    for (size_t i = 0; i < chans.size(); i++) { coff[i] = bytes_per_px_row; bytes_per_px_row += chans[i].type == 1 ? 2 : 4; }
    // scan lines per block by compression type: NONE, RLE, ZIPS 1; ZIP, PXR24 16; PIZ, B44, B44A, DWAA 32; DWAB 256
    static const int kLines[10] = {1, 1, 1, 16, 32, 16, 32, 32, 32, 256};
    const int lines_per_block = kLines[compression];
    const int nblocks = (h + lines_per_block - 1) / lines_per_block;
    if (pos + (size_t)nblocks * 8 > n) { err = "exr: truncated offset table"; return false; }
    out.width = w; out.height = h; out.half_source = chans[0].type == 1;
    out.rgb.assign((size_t)w * h * 3, 0.f);
    std::vector<uint8_t> buf, tmp;
    for (int b = 0; b < nblocks; b++) {
        uint64_t off = le64(pos + (size_t)b * 8);
        if (off > n || n - off < 8) { err = "exr: bad block offset"; return false; }                 // (off + 8 could wrap)
        const long long y64 = (long long)(int)le32((size_t)off) - dw[1];
        uint32_t sz = le32((size_t)off + 4);
        if (off + 8 + sz > n || y64 < 0 || y64 >= h) { err = "exr: bad block"; return false; }
        const int y0 = (int)y64;
        const int lines = (y0 + lines_per_block <= h) ? lines_per_block : h - y0;
        const size_t expect = (size_t)lines * w * bytes_per_px_row;
        const uint8_t* src = data + off + 8;
        if (compression == 0 || sz == expect) {                     // blocks that do not shrink are stored raw
            if (sz < expect) { err = "exr: short raw block"; return false; }
            buf.assign(src, src + expect);
        } else {
            if (compression > 3) { err = "exr: unsupported compression (NONE, RLE, ZIPS, ZIP, or any type whose blocks are stored raw)"; return false; }
            if (compression == 1) {                               // RLE
                tmp.clear();
                size_t i = 0;
                while (i < sz) {
                    int8_t c = (int8_t)src[i++];
                    if (c < 0) { size_t cnt = (size_t)(-c); if (i + cnt > sz) { err = "exr: bad RLE run"; return false; } tmp.insert(tmp.end(), src + i, src + i + cnt); i += cnt; }
                    else { if (i >= sz) { err = "exr: bad RLE run"; return false; } tmp.insert(tmp.end(), (size_t)c + 1, src[i]); i++; }
                }
            } else if (!zlib_inflate(src, sz, tmp, err, expect)) return false;
            if (tmp.size() != expect) { err = "exr: decompressed block has the wrong size"; return false; }
            for (size_t i = 1; i < tmp.size(); i++) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);            // predictor
            buf.resize(expect);
            const size_t half = (expect + 1) / 2;                   // de-interleave
            for (size_t i = 0, a = 0, c = half; i < expect;) { buf[i++] = tmp[a++]; if (i < expect) buf[i++] = tmp[c++]; }
        }
        for (int l = 0; l < lines; l++) {
            const uint8_t* row = buf.data() + (size_t)l * w * bytes_per_px_row;
            // within a scan line channels are stored one after another (alphabetical order), each w samples
            size_t choff = 0;
            std::vector<size_t> start(chans.size());
            for (size_t i = 0; i < chans.size(); i++) { start[i] = choff; choff += (size_t)w * (chans[i].type == 1 ? 2 : 4); }
            for (int k = 0; k < 3; k++) {
                const Chan& c = chans[ci[k]];
                const uint8_t* s = row + start[ci[k]];
                float* o = &out.rgb[((size_t)(y0 + l) * w) * 3 + k];
                for (int x = 0; x < w; x++) {
                    float v;
                    if (c.type == 1) v = half_to_float((uint16_t)(s[x * 2] | (s[x * 2 + 1] << 8)));
                    else if (c.type == 2) { uint32_t u = (uint32_t)s[x * 4] | ((uint32_t)s[x * 4 + 1] << 8) | ((uint32_t)s[x * 4 + 2] << 16) | ((uint32_t)s[x * 4 + 3] << 24); memcpy(&v, &u, 4); }
                    else { uint32_t u = (uint32_t)s[x * 4] | ((uint32_t)s[x * 4 + 1] << 8) | ((uint32_t)s[x * 4 + 2] << 16) | ((uint32_t)s[x * 4 + 3] << 24); v = (float)u; }
                    o[(size_t)x * 3] = v;
                }
            }
        }
    }
    (void)increasing_y;          // blocks carry their own y coordinate, so either line order lands in the right rows
    return true;
}

// =================================================================================================== writers
namespace {

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }
void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& body) {
    put_be32(out, (uint32_t)body.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    put_be32(out, crc32_update(0xffffffffu, out.data() + start, out.size() - start) ^ 0xffffffffu);
}
// deflate with a greedy hash-chain LZ77 matcher and the fixed Huffman code (RFC 1951 3.2.6): small, dependency-free, and good
// enough for rendered images (typically 2-4x on 8-bit output)
struct BitWriter {
    std::vector<uint8_t>& out; uint32_t acc = 0; int n = 0;
    explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
    void bits(uint32_t v, int c) { acc |= v << n; n += c; while (n >= 8) { out.push_back((uint8_t)acc); acc >>= 8; n -= 8; } }
    void huff(uint32_t code, int len) { uint32_t r = 0; for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i); bits(r, len); }   // codes go MSB first
    void flush() { if (n) { out.push_back((uint8_t)acc); acc = 0; n = 0; } }
};
void fixed_literal(BitWriter& w, int s) {
    if (s < 144) w.huff(0x30 + s, 8); else if (s < 256) w.huff(0x190 + (s - 144), 9); else if (s < 280) w.huff(s - 256, 7); else w.huff(0xc0 + (s - 280), 8);
}
void zlib_deflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
    out.push_back(0x78); out.push_back(0x9c);
    BitWriter w(out);
    w.bits(1, 1); w.bits(1, 2);                                  // final block, fixed Huffman
    const int kHashBits = 15, kWindow = 32768, kMaxChain = 32;
    std::vector<int32_t> head((size_t)1 << kHashBits, -1), prev(n ? n : 1, -1);
    auto hash = [&](size_t i) { return ((uint32_t)src[i] * 2654435761u ^ (uint32_t)src[i + 1] * 40503u ^ (uint32_t)src[i + 2] * 2246822519u) >> (32 - kHashBits); };
    size_t i = 0;
    while (i < n) {
        int best_len = 0, best_dist = 0;
        if (i + 3 <= n) {
            uint32_t h = hash(i);
            int32_t cand = head[h];
            int chain = 0;
            while (cand >= 0 && (int)(i - (size_t)cand) <= kWindow && chain++ < kMaxChain) {
                int l = 0;
                const size_t maxl = n - i < 258 ? n - i : 258;
                while ((size_t)l < maxl && src[(size_t)cand + l] == src[i + l]) l++;
                if (l > best_len) { best_len = l; best_dist = (int)(i - (size_t)cand); if (l == 258) break; }
                cand = prev[(size_t)cand];
            }
        }
        const size_t step = best_len >= 3 ? (size_t)best_len : 1;
        if (best_len >= 3) {
            int ls = 28;
            while (kLenBase[ls] > best_len) ls--;
            fixed_literal(w, 257 + ls);
            w.bits((uint32_t)(best_len - kLenBase[ls]), kLenExtra[ls]);
            int ds = 29;
            while (kDistBase[ds] > best_dist) ds--;
            w.huff((uint32_t)ds, 5);
            w.bits((uint32_t)(best_dist - kDistBase[ds]), kDistExtra[ds]);
        } else fixed_literal(w, src[i]);
        for (size_t k = 0; k < step; k++, i++)
            if (i + 3 <= n) { uint32_t h = hash(i); prev[i] = head[h]; head[h] = (int32_t)i; }
    }
    fixed_literal(w, 256);
    w.flush();
    put_be32(out, adler32(src, n));
}
bool write_all(const std::string& path, const std::vector<uint8_t>& bytes, std::string& err) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create " + path; return false; }
    size_t put = fwrite(bytes.data(), 1, bytes.size(), f);
    fclose(f);
    if (put != bytes.size()) { err = "short write on " + path; return false; }
    return true;
}
uint16_t float_to_half(float f) {                               // round to nearest even
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, mant = x & 0x7fffffu;
    int exp = (int)((x >> 23) & 0xff) - 127 + 15;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0u));
    if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        mant |= 0x800000u;
        int shift = 14 - exp;
        uint32_t h = mant >> shift, rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)exp << 10) | (mant >> 13), rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
}

}  // namespace

bool encode_png(const uint8_t* rgba, int w, int h, int channels, std::vector<uint8_t>& out, std::string& err) {
    if (!rgba || w <= 0 || h <= 0 || (channels != 3 && channels != 4)) { err = "png: bad arguments"; return false; }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.assign(sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(channels == 4 ? 6 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr);
    // filter type 1 (Sub) on every row: cheap and effective on smooth images
    const size_t stride = (size_t)w * channels;
    std::vector<uint8_t> raw((stride + 1) * h);
    for (int y = 0; y < h; y++) {
        uint8_t* row = &raw[(size_t)y * (stride + 1)];
        row[0] = 1;
        const uint8_t* s = rgba + (size_t)y * w * 4;
        for (int x = 0; x < w; x++)
            for (int c = 0; c < channels; c++) {
                uint8_t cur = s[x * 4 + c], left = x ? s[(x - 1) * 4 + c] : 0;
                row[1 + (size_t)x * channels + c] = (uint8_t)(cur - left);
            }
    }
    std::vector<uint8_t> z;
    zlib_deflate(raw.data(), raw.size(), z);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    return true;
}
bool write_png(const std::string& path, const uint8_t* rgba, int w, int h, int channels, std::string& err) {
    std::vector<uint8_t> bytes;
    return encode_png(rgba, w, h, channels, bytes, err) && write_all(path, bytes, err);
}
// Portable float map "PF": little-endian RGB32F, bottom row first
bool write_pfm(const std::string& path, const float* rgb, int w, int h, std::string& err) {
    if (!rgb || w <= 0 || h <= 0) { err = "pfm: bad arguments"; return false; }
    char head[64];
    int n = snprintf(head, sizeof(head), "PF\n%d %d\n-1.0\n", w, h);
    std::vector<uint8_t> bytes(head, head + n);
    for (int y = h - 1; y >= 0; y--) { const uint8_t* row = (const uint8_t*)(rgb + (size_t)y * w * 3); bytes.insert(bytes.end(), row, row + (size_t)w * 12); }
    return write_all(path, bytes, err);
}
// OpenEXR, single part, scan lines, no compression, channels B G R (HALF or FLOAT)
bool write_exr(const std::string& path, const float* rgb, int w, int h, bool half, std::string& err) {
    if (!rgb || w <= 0 || h <= 0) { err = "exr: bad arguments"; return false; }
    std::vector<uint8_t> o;
    auto le32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); };
    auto le64 = [&](uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); };
    auto str = [&](const char* s) { o.insert(o.end(), s, s + strlen(s) + 1); };
    auto f32 = [&](float f) { uint32_t u; memcpy(&u, &f, 4); le32(u); };
    le32(20000630u); le32(2u);
    str("channels"); str("chlist"); le32(3 * 18 + 1);
    for (const char* c : {"B", "G", "R"}) { str(c); le32(half ? 1u : 2u); o.push_back(0); o.push_back(0); o.push_back(0); o.push_back(0); le32(1); le32(1); }
    o.push_back(0);
    str("compression"); str("compression"); le32(1); o.push_back(0);
    str("dataWindow"); str("box2i"); le32(16); le32(0); le32(0); le32((uint32_t)(w - 1)); le32((uint32_t)(h - 1));
    str("displayWindow"); str("box2i"); le32(16); le32(0); le32(0); le32((uint32_t)(w - 1)); le32((uint32_t)(h - 1));
    str("lineOrder"); str("lineOrder"); le32(1); o.push_back(0);
    str("pixelAspectRatio"); str("float"); le32(4); f32(1.0f);
    str("screenWindowCenter"); str("v2f"); le32(8); f32(0.0f); f32(0.0f);
    str("screenWindowWidth"); str("float"); le32(4); f32(1.0f);
    o.push_back(0);
    const size_t px = half ? 2 : 4, line = (size_t)w * 3 * px, table = o.size();
    for (int y = 0; y < h; y++) le64((uint64_t)(table + (size_t)h * 8 + (size_t)y * (8 + line)));
    for (int y = 0; y < h; y++) {
        le32((uint32_t)y); le32((uint32_t)line);
        for (int c = 2; c >= 0; c--)                                // B, G, R
            for (int x = 0; x < w; x++) {
                float v = rgb[((size_t)y * w + x) * 3 + c];
                if (half) { uint16_t hv = float_to_half(v); o.push_back((uint8_t)hv); o.push_back((uint8_t)(hv >> 8)); }
                else f32(v);
            }
    }
    return write_all(path, o, err);
}

}  // namespace hostimg
