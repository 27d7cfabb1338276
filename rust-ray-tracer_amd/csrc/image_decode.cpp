// image_decode.cpp -- build-owned texture decode: baseline/extended-sequential Huffman JPEG (8-bit, unsubsampled
// or greyscale) and non-interlaced 8-bit PNG.  Stands where the reference calls the `image` crate
// (`ImageReader::open(..).decode()`, src/file_management/utils.rs:345-350; image 0.25.9 -> zune-jpeg 0.5.8 / png 0.18.0,
// Cargo.lock).  Those crates are not in the reference tree, JPEG decoders are not bit-identical to one another and no
// reference test pins decoded texels, so parity at this boundary is UNPINNED: the hot path's input is
// defined as the decoded RGB8 array, and the same bytes feed the GPU path and the oracle.
//
// The JPEG arithmetic follows the published IJG definitions so that it can be checked against an independent
// libjpeg build (tests compare with PIL/libjpeg-turbo): the LL&M 13-bit "slow integer" inverse DCT and the
// 16.16 fixed-point YCbCr->RGB conversion.
#include <zlib.h>

#include <cstdio>
#include <cstring>

#include "model.hpp"
#include "parallel.hpp"

namespace rrt {
namespace {

[[noreturn]] void fail(int status, const std::string& what) { throw Error{status, what}; }

std::vector<uint8_t> slurp(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) fail(RRT_ERR_IO, "Cannot read texture file: " + path);   // utils.rs:346-347
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    return buf;
}

// ===================================================================== JPEG
struct Huff {
    bool present = false;
    uint8_t vals[256];
    int32_t mincode[17], maxcode[18], valptr[17];
    static constexpr int kLook = 9;
    uint16_t look[1 << kLook];       // the next 9 bits -> (code length << 8 | symbol) for codes of <= 9 bits, 0 = longer code (bit-by-bit path)
    void build(const uint8_t counts[16], const uint8_t* symbols, int nsym) {
        std::memcpy(vals, symbols, nsym);
        std::memset(look, 0, sizeof look);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k; mincode[len] = code;
            if (len <= kLook)
                for (int j = 0; j < counts[len - 1] && k + j < nsym; j++) {
                    const int c = (code + j) << (kLook - len);
                    if (c + (1 << (kLook - len)) > (1 << kLook)) break;   // (an over-subscribed table of a hostile file: leave it to the slow path, which rejects it)
                    for (int f = 0; f < (1 << (kLook - len)); f++) look[c + f] = (uint16_t)((len << 8) | symbols[k + j]);
                }
            code += counts[len - 1]; k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

// Bits of the entropy-coded segment, most significant first.  A 0xFF00 pair is the byte 0xFF; any other 0xFF xx is a marker: it is left for the
// caller and zeros are fed from there on (as libjpeg does for a truncated scan).
struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint64_t acc = 0; int nbits = 0;      // the low `nbits` bits of acc are the unread bits
    bool hit_marker = false;
    void fill() {
        while (nbits <= 56) {
            uint8_t b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    const uint8_t b2 = p < end ? *p : 0;
                    if (b2 == 0) p++;                              // stuffed zero
                    else { hit_marker = true; p--; b = 0; }        // a marker: feed zeros, leave it for the caller
                }
            }
            acc = (acc << 8) | b; nbits += 8;
        }
    }
    int peek(int n) { if (nbits < n) fill(); return (int)((acc >> (nbits - n)) & ((1u << n) - 1u)); }
    void skip(int n) { nbits -= n; }
    int bit() { const int v = peek(1); skip(1); return v; }
    int bits(int n) { if (n == 0) return 0; const int v = peek(n); skip(n); return v; }
    void reset_at(const uint8_t* q) { p = q; acc = 0; nbits = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader& br, const Huff& h) {
    const uint16_t e = h.look[br.peek(Huff::kLook)];
    if (e) { br.skip(e >> 8); return e & 0xFF; }
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    fail(RRT_ERR_PARSE, "Cannot decode texture file: bad Huffman code");
}

inline int extend(int v, int nbits) { return v < (1 << (nbits - 1)) ? v - (1 << nbits) + 1 : v; }

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// IJG "islow" inverse DCT: Loeffler-Ligtenberg-Moschytz, CONST_BITS = 13, PASS1_BITS = 2.
inline int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }
inline uint8_t clamp255(int32_t x) { return x < 0 ? 0 : x > 255 ? 255 : (uint8_t)x; }

void idct_islow(const int32_t in[64], uint8_t* out, size_t stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int32_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                      F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    int32_t ws[64];
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < 8; i++) {
            int32_t s[8];
            if (pass == 0) for (int k = 0; k < 8; k++) s[k] = in[8 * k + i];      // columns
            else           for (int k = 0; k < 8; k++) s[k] = ws[8 * i + k];      // rows
            int32_t z2 = s[2], z3 = s[6];
            int32_t z1 = (z2 + z3) * F0_541;
            int32_t tmp2 = z1 + z3 * (-F1_847);
            int32_t tmp3 = z1 + z2 * F0_765;
            int32_t tmp0 = (s[0] + s[4]) * (1 << CB);
            int32_t tmp1 = (s[0] - s[4]) * (1 << CB);
            int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = s[7]; tmp1 = s[5]; tmp2 = s[3]; tmp3 = s[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; int32_t z4 = tmp1 + tmp3;
            int32_t z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            int32_t r[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
            if (pass == 0) for (int k = 0; k < 8; k++) ws[8 * k + i] = descale(r[k], CB - P1);
            else           for (int k = 0; k < 8; k++) out[stride * i + k] = clamp255(descale(r[k], CB + P1 + 3) + 128);
        }
    }
}

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int pred = 0; };

void decode_jpeg(const std::vector<uint8_t>& buf, std::vector<uint8_t>& out, uint32_t& W, uint32_t& H, uint32_t& channels) {
    const uint8_t* p = buf.data(); const uint8_t* end = p + buf.size();
    auto need = [&](size_t n) { if ((size_t)(end - p) < n) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated JPEG"); };
    need(2);
    if (p[0] != 0xFF || p[1] != 0xD8) fail(RRT_ERR_PARSE, "Cannot decode texture file: not a JPEG");
    p += 2;
    uint16_t qt[4][64] = {}; bool qt_present[4] = {};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int restart_interval = 0; bool have_sof = false; int adobe_transform = -1;
    for (;;) {
        need(2);
        if (p[0] != 0xFF) fail(RRT_ERR_PARSE, "Cannot decode texture file: marker expected");
        while (p < end && p[0] == 0xFF && p + 1 < end && p[1] == 0xFF) p++;   // fill bytes
        uint8_t m = p[1]; p += 2;
        if (m == 0xD9) fail(RRT_ERR_PARSE, "Cannot decode texture file: no scan");
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        need(2);
        size_t L = ((size_t)p[0] << 8) | p[1];
        if (L < 2) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad segment");
        need(L);
        const uint8_t* s = p + 2; const uint8_t* se = p + L;
        if (m == 0xDB) {                                    // DQT
            while (s < se) {
                int pq = s[0] >> 4, tq = s[0] & 15; s++;
                if (tq > 3 || (size_t)(se - s) < (size_t)(pq ? 128 : 64)) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad DQT");
                for (int k = 0; k < 64; k++) { qt[tq][kZigzag[k]] = pq ? (uint16_t)((s[0] << 8) | s[1]) : s[0]; s += pq ? 2 : 1; }
                qt_present[tq] = true;
            }
        } else if (m == 0xC4) {                             // DHT
            while (s < se) {
                if (se - s < 17) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad DHT");
                int tc = s[0] >> 4, th = s[0] & 15; s++;
                const uint8_t* counts = s; s += 16;
                int n = 0; for (int k = 0; k < 16; k++) n += counts[k];
                if (th > 3 || tc > 1 || n > 256 || se - s < n) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad DHT");
                (tc ? ac : dc)[th].build(counts, s, n); s += n;
            }
        } else if (m == 0xC0 || m == 0xC1) {                // SOF0 / SOF1 (Huffman, sequential)
            if (L < 8 || s[0] != 8) fail(RRT_ERR_UNSUPPORTED, "JPEG: only 8-bit precision");
            H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4];
            int nc = s[5];
            if ((nc != 1 && nc != 3) || L < (size_t)(8 + 3 * nc) || W == 0 || H == 0) fail(RRT_ERR_UNSUPPORTED, "JPEG: unsupported component count");
            comps.resize(nc);
            for (int k = 0; k < nc; k++) { comps[k].id = s[6 + 3 * k]; comps[k].h = s[7 + 3 * k] >> 4; comps[k].v = s[7 + 3 * k] & 15; comps[k].tq = s[8 + 3 * k] & 3; }
            for (auto& c : comps) if (c.h != 1 || c.v != 1) fail(RRT_ERR_UNSUPPORTED, "JPEG: chroma subsampling is not supported");
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            fail(RRT_ERR_UNSUPPORTED, "JPEG: only baseline/sequential Huffman");
        } else if (m == 0xDD) {                             // DRI
            if (L >= 4) restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xEE) {                             // APP14 Adobe
            if (L >= 14 && !std::memcmp(s, "Adobe", 5)) adobe_transform = s[11];
        } else if (m == 0xDA) {                             // SOS
            if (!have_sof) fail(RRT_ERR_PARSE, "Cannot decode texture file: SOS before SOF");
            int ns = s[0];
            if (ns != (int)comps.size() || L < (size_t)(6 + 2 * ns)) fail(RRT_ERR_UNSUPPORTED, "JPEG: non-interleaved scans are not supported");
            for (int k = 0; k < ns; k++) {
                int cid = s[1 + 2 * k]; bool found = false;
                for (auto& c : comps) if (c.id == cid) { c.td = s[2 + 2 * k] >> 4; c.ta = s[2 + 2 * k] & 15; found = true; }
                if (!found) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad SOS");
            }
            p += L;
            break;
        }
        p += L;
    }
    for (auto& c : comps) if (!qt_present[c.tq] || c.td > 3 || c.ta > 3 || !dc[c.td].present || !ac[c.ta].present) fail(RRT_ERR_PARSE, "Cannot decode texture file: missing table");

    const uint32_t bw = (W + 7) / 8, bh = (H + 7) / 8;
    const size_t stride = (size_t)bw * 8;
    // ---- phase 1, sequential by nature: the entropy-coded segment -> quantised coefficients (natural order) of every block of every component
    const size_t n_blocks = (size_t)bw * bh, nc = comps.size();
    std::vector<int16_t> coef(n_blocks * nc * 64, 0);
    BitReader br{p, end};
    int to_restart = restart_interval;
    for (uint32_t by = 0; by < bh; by++) {
        for (uint32_t bx = 0; bx < bw; bx++) {
            if (restart_interval && to_restart == 0) {
                // byte-align, expect RSTn (the reader never passes a marker: it is at or ahead of br.p)
                const uint8_t* q = br.p;
                while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
                if (q + 1 >= end) fail(RRT_ERR_PARSE, "Cannot decode texture file: missing RST");
                br.reset_at(q + 2);
                for (auto& c : comps) c.pred = 0;
                to_restart = restart_interval;
            }
            for (size_t ci = 0; ci < nc; ci++) {
                Component& c = comps[ci];
                int16_t* blk = &coef[(((size_t)by * bw + bx) * nc + ci) * 64];
                int t = decode_symbol(br, dc[c.td]);
                if (t > 11) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad DC category");          // 8-bit JPEG: DC differences have at most 11 bits
                int diff = t ? extend(br.bits(t), t) : 0;
                c.pred += diff;
                if (c.pred < -32768 || c.pred > 32767) fail(RRT_ERR_PARSE, "Cannot decode texture file: DC predictor out of range");
                blk[0] = (int16_t)c.pred;
                for (int k = 1; k < 64;) {
                    int rs = decode_symbol(br, ac[c.ta]);
                    int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                    k += r;
                    if (k > 63) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad AC run");
                    if (sz > 10) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad AC size");                // 8-bit JPEG: AC coefficients have at most 10 bits
                    blk[kZigzag[k]] = (int16_t)extend(br.bits(sz), sz);
                    k++;
                }
            }
            if (restart_interval) to_restart--;
        }
    }

    // ---- phase 2, every row of blocks on its own: dequantise, inverse DCT, colour conversion (a 1024 x 1024 texture: 3 x 16 384 blocks -- the larger
    // part of the decode, and the part the host's cores can share)
    const bool grey = nc == 1;
    channels = grey ? 1 : 3; out.resize((size_t)W * H * channels);
    const bool ycc = adobe_transform < 0 ? !(!grey && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') : adobe_transform != 0;
    // IJG jdcolor.c: 16.16 fixed point, FIX(x) = (int)(x*65536 + 0.5)
    constexpr int32_t ONE_HALF = 1 << 15, F1_402 = 91881, F1_772 = 116130, F0_714 = 46802, F0_344 = 22554;
    parallel_ranges(bh, 4, [&](size_t rb, size_t re, size_t) {
        std::vector<uint8_t> rows(nc * 8 * stride);                      // the 8 pixel rows of one row of blocks, per component
        for (size_t by = rb; by < re; by++) {
            for (uint32_t bx = 0; bx < bw; bx++)
                for (size_t ci = 0; ci < nc; ci++) {
                    const int16_t* q = &coef[((by * bw + bx) * nc + ci) * 64];
                    const uint16_t* t = qt[comps[ci].tq];
                    int32_t blk[64];
                    for (int k = 0; k < 64; k++) { const int32_t v = (int32_t)q[k] * t[k]; blk[k] = v < -(1 << 15) ? -(1 << 15) : v > (1 << 15) ? (1 << 15) : v; }   // hostile tables: keep the IDCT inside int32 (a valid file never gets near)
                    idct_islow(blk, rows.data() + ci * 8 * stride + (size_t)bx * 8, stride);
                }
            for (uint32_t r = 0; r < 8; r++) {
                const size_t y = by * 8 + r;
                if (y >= H) break;
                const uint8_t* Y = rows.data() + r * stride;
                if (grey) { std::memcpy(out.data() + y * W, Y, W); continue; }
                const uint8_t* Cb = rows.data() + (8 + r) * stride; const uint8_t* Cr = rows.data() + (16 + r) * stride;
                uint8_t* o = out.data() + y * W * 3;
                for (uint32_t x = 0; x < W; x++) {
                    if (!ycc) { o[3 * x] = Y[x]; o[3 * x + 1] = Cb[x]; o[3 * x + 2] = Cr[x]; continue; }
                    int32_t yy = Y[x], cb = Cb[x] - 128, cr = Cr[x] - 128;
                    int32_t rr = yy + ((F1_402 * cr + ONE_HALF) >> 16);
                    int32_t g = yy + ((-F0_344 * cb + ONE_HALF - F0_714 * cr) >> 16);
                    int32_t b = yy + ((F1_772 * cb + ONE_HALF) >> 16);
                    o[3 * x] = clamp255(rr); o[3 * x + 1] = clamp255(g); o[3 * x + 2] = clamp255(b);
                }
            }
        }
    });
}

// ===================================================================== PNG
constexpr uint32_t kMaxImageDim = 65535;   // JPEG's own limit; the `image` crate enforces limits of its own on the reference side
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

void decode_png(const std::vector<uint8_t>& buf, std::vector<uint8_t>& out, uint32_t& W, uint32_t& H, uint32_t& channels) {
    size_t i = 8;
    std::vector<uint8_t> idat, palette;
    int depth = 0, ctype = -1, interlace = 0; bool trns = false;
    while (i + 12 <= buf.size()) {
        uint32_t len = be32(&buf[i]); const uint8_t* type = &buf[i + 4];
        if (i + 12 + (size_t)len > buf.size()) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated PNG");
        const uint8_t* d = &buf[i + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) { W = be32(d); H = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (!std::memcmp(type, "PLTE", 4)) palette.assign(d, d + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns = true;
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        i += 12 + (size_t)len;
    }
    if (ctype < 0 || W == 0 || H == 0) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad PNG header");
    if (W > kMaxImageDim || H > kMaxImageDim) fail(RRT_ERR_PARSE, "Cannot decode texture file: PNG dimensions beyond the supported 65535 x 65535");   // (row+1)*H below must not wrap
    if (depth != 8 || interlace != 0 || trns) fail(RRT_ERR_UNSUPPORTED, "PNG: only 8-bit, non-interlaced, no tRNS");
    int src_ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!src_ch) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad PNG colour type");
    const size_t row = (size_t)W * src_ch;
    std::vector<uint8_t> raw((row + 1) * H);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) fail(RRT_ERR_PARSE, "Cannot decode texture file: PNG inflate failed");
    std::vector<uint8_t> img(row * H);
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t ft = raw[(row + 1) * y]; const uint8_t* s = &raw[(row + 1) * y + 1];
        uint8_t* o = &img[row * y]; const uint8_t* up = y ? &img[row * (y - 1)] : nullptr;
        for (size_t x = 0; x < row; x++) {
            int a = x >= (size_t)src_ch ? o[x - src_ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)src_ch) ? up[x - src_ch] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: fail(RRT_ERR_PARSE, "Cannot decode texture file: bad PNG filter");
            }
            o[x] = (uint8_t)(s[x] + pred);
        }
    }
    if (ctype == 3) {                                    // indexed -> Rgb8, as the `image` crate expands palettes
        channels = 3; out.resize((size_t)W * H * 3);
        for (size_t k = 0; k < (size_t)W * H; k++) {
            size_t pi = (size_t)img[k] * 3;
            if (pi + 3 > palette.size()) fail(RRT_ERR_PARSE, "Cannot decode texture file: palette index out of range");
            out[3 * k] = palette[pi]; out[3 * k + 1] = palette[pi + 1]; out[3 * k + 2] = palette[pi + 2];
        }
    } else { channels = (uint32_t)src_ch; out = std::move(img); }
}

}  // namespace

void decode_image_file(const std::string& path, std::vector<uint8_t>& bytes, uint32_t& width, uint32_t& height, uint32_t& channels) {
    std::vector<uint8_t> buf = slurp(path);
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (buf.size() >= 8 && !std::memcmp(buf.data(), png_sig, 8)) decode_png(buf, bytes, width, height, channels);
    else if (buf.size() >= 2 && buf[0] == 0xFF && buf[1] == 0xD8) decode_jpeg(buf, bytes, width, height, channels);
    else fail(RRT_ERR_UNSUPPORTED, "Cannot decode texture file (not JPEG/PNG): " + path);
}

}  // namespace rrt
