// image_decode.cpp -- build-owned texture decode: Huffman JPEG (8-bit; sequential or progressive; 4:4:4, 4:2:2, 4:2:0 or greyscale; one scan
// or many; restart markers), PNG (8-bit samples or 1-8-bit palette, interlaced or not), uncompressed BMP and 24-bit TGA.  Stands where the reference calls the `image` crate
// (`ImageReader::open(..).decode()`, src/file_management/utils.rs:345-350; image 0.25.9 -> zune-jpeg 0.5.8 / png 0.18.0,
// Cargo.lock).  Those crates are not in the reference tree, JPEG decoders are not bit-identical to one another and no
// reference test pins decoded texels, so parity at this boundary is UNPINNED: the hot path's input is
// defined as the decoded RGB8 array, and the same bytes feed the GPU path and the oracle.
//
// The JPEG arithmetic follows the published IJG definitions so that it can be checked against an independent
// libjpeg build (tests compare with PIL/libjpeg-turbo): the LL&M 13-bit "slow integer" inverse DCT and the
// 16.16 fixed-point YCbCr->RGB conversion.
#include <zlib.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <memory>

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

// -1: no code of <= 16 bits matches (a malformed stream -- or a speculative decoder that has not found the code boundaries yet, see entropy_parallel)
template <class Reader> inline int decode_symbol(Reader& br, const Huff& h) {
    const uint16_t e = h.look[br.peek(Huff::kLook)];
    if (e) { br.skip(e >> 8); return e & 0xFF; }
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    return -1;
}

inline int extend(int v, int nbits) { return v < (1 << (nbits - 1)) ? v - (1 << nbits) + 1 : v; }

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// One block of the scan: the DC difference and the AC coefficients (natural order) into blk[1..63]; blk[0] is left to the caller, who owns the
// predictor.  Returns 0, or which rule of an 8-bit baseline stream was broken.
enum BlockError { kBlockOk = 0, kBadCode, kBadDcCategory, kBadAcRun, kBadAcSize };
inline const char* block_error_text(int e) {
    switch (e) {
        case kBadCode: return "Cannot decode texture file: bad Huffman code";
        case kBadDcCategory: return "Cannot decode texture file: bad DC category";      // 8-bit JPEG: DC differences have at most 11 bits
        case kBadAcRun: return "Cannot decode texture file: bad AC run";
        default: return "Cannot decode texture file: bad AC size";                        // 8-bit JPEG: AC coefficients have at most 10 bits
    }
}
template <class Reader> inline int decode_block(Reader& br, const Huff& dct, const Huff& act, int& dc_diff, int16_t* blk) {
    const int t = decode_symbol(br, dct);
    if (t < 0) return kBadCode;
    if (t > 11) return kBadDcCategory;
    dc_diff = t ? extend(br.bits(t), t) : 0;
    for (int k = 1; k < 64;) {
        const int rs = decode_symbol(br, act);
        if (rs < 0) return kBadCode;
        const int r = rs >> 4, sz = rs & 15;
        if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
        k += r;
        if (k > 63) return kBadAcRun;
        if (sz > 10) return kBadAcSize;
        blk[kZigzag[k]] = (int16_t)extend(br.bits(sz), sz);
        k++;
    }
    return kBlockOk;
}

// IJG "islow" inverse DCT: Loeffler-Ligtenberg-Moschytz, CONST_BITS = 13, PASS1_BITS = 2.
inline uint8_t clamp255(int32_t x) { return x < 0 ? 0 : x > 255 ? 255 : (uint8_t)x; }

// 32-bit arithmetic that wraps: on the coefficients of a real picture nothing here comes near 2^31, but a hostile file can drive the sums past it, and
// signed overflow is undefined where libjpeg's INT32 code merely produces garbage.  Same bits as int32_t arithmetic whenever that is defined.
struct Wrap32 {
    uint32_t u;
    Wrap32() = default;
    Wrap32(int32_t v) : u((uint32_t)v) {}
    int32_t s() const { return (int32_t)u; }
    friend Wrap32 operator+(Wrap32 a, Wrap32 b) { Wrap32 r; r.u = a.u + b.u; return r; }
    friend Wrap32 operator-(Wrap32 a, Wrap32 b) { Wrap32 r; r.u = a.u - b.u; return r; }
    friend Wrap32 operator*(Wrap32 a, Wrap32 b) { Wrap32 r; r.u = a.u * b.u; return r; }
    Wrap32& operator+=(Wrap32 b) { u += b.u; return *this; }
    Wrap32& operator*=(Wrap32 b) { u *= b.u; return *this; }
};
inline int32_t descale(Wrap32 x, int n) { return (Wrap32(1 << (n - 1)) + x).s() >> n; }

void idct_islow(const int32_t in[64], uint8_t* out, size_t stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int32_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                      F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    int32_t ws[64];
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < 8; i++) {
            Wrap32 s[8];
            if (pass == 0) for (int k = 0; k < 8; k++) s[k] = in[8 * k + i];      // columns
            else           for (int k = 0; k < 8; k++) s[k] = ws[8 * i + k];      // rows
            Wrap32 z2 = s[2], z3 = s[6];
            Wrap32 z1 = (z2 + z3) * F0_541;
            Wrap32 tmp2 = z1 + z3 * (-F1_847);
            Wrap32 tmp3 = z1 + z2 * F0_765;
            Wrap32 tmp0 = (s[0] + s[4]) * (1 << CB);
            Wrap32 tmp1 = (s[0] - s[4]) * (1 << CB);
            Wrap32 tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = s[7]; tmp1 = s[5]; tmp2 = s[3]; tmp3 = s[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; Wrap32 z4 = tmp1 + tmp3;
            Wrap32 z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            Wrap32 r[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
            if (pass == 0) for (int k = 0; k < 8; k++) ws[8 * k + i] = descale(r[k], CB - P1);
            else           for (int k = 0; k < 8; k++) out[stride * i + k] = clamp255((Wrap32(descale(r[k], CB + P1 + 3)) + 128).s());
        }
    }
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int pred = 0;
    uint32_t bw = 0, bh = 0;          // block grid, padded to whole MCUs of the frame
    uint32_t w = 0, hgt = 0;          // samples that belong to the image: ceil(W h / hmax), ceil(H v / vmax)
    std::vector<int16_t> coef;        // planar coefficients (multi-scan files only): bw x bh x 64
};

// ---- phase 1 on several threads.  A Huffman stream has no index: where block 20 000 starts is known only after decoding the 19 999 before it.  But a
// decoder dropped into the middle of the stream -- wrong bit, wrong block of the MCU, wrong coefficient -- falls into step with the true decode after a
// few blocks (the codes are short and the end-of-block code keeps re-aligning it), and from the first block START the two share (same bit position, same
// place in the MCU) they decode the same thing for ever: the parse has no other state.  So: the scan (stuffed zeros removed) is cut into T byte ranges;
// decoder i starts at the first bit of range i as if an MCU began there, notes (bit position, place in the MCU) of every block it starts, and keeps its
// blocks in a buffer of its own; when all have crossed their range, decoder i runs on into range i + 1 until it starts a block exactly where decoder
// i + 1 started one.  Decoder 0 is right from its first bit, so by induction everything from each meeting point on is right; what a decoder produced
// before it was met is dropped.  DC values are differences against a predictor the late starters do not know: they count from 0 and the true value at
// the meeting point gives the offset.  Anything unexpected -- no meeting inside the next range, a malformed block in a kept stretch, too few blocks, a
// DC value out of range -- and the function returns false: the caller decodes serially, as before, and reports what it finds.  Same coefficients as the
// serial decode or no result; the first 1024 x 1024 texture of the teapot scene took 9 ms of one core, the six of them were most of the loader's time.
struct CleanReader {
    const uint8_t* base; size_t n_bytes;      // the scan without stuffing; zeros are fed past the end (as the serial reader does at a marker)
    size_t pos = 0; uint64_t acc = 0; int nbits = 0;
    void fill() {
        if (pos + 4 <= n_bytes) { acc = (acc << 32) | ((uint32_t)base[pos] << 24 | (uint32_t)base[pos + 1] << 16 | (uint32_t)base[pos + 2] << 8 | base[pos + 3]); pos += 4; nbits += 32; }
        else while (nbits <= 32) { acc = (acc << 8) | (pos < n_bytes ? base[pos] : 0); pos++; nbits += 8; }
    }
    int peek(int n) { if (nbits < n) fill(); return (int)((acc >> (nbits - n)) & ((1u << n) - 1u)); }
    void skip(int n) { nbits -= n; }
    int bit() { const int v = peek(1); skip(1); return v; }
    int bits(int n) { if (n == 0) return 0; const int v = peek(n); skip(n); return v; }
    uint64_t bitpos() const { return (uint64_t)pos * 8 - (uint64_t)nbits; }
};

// The blocks of one MCU of an interleaved scan, in stream order: slot -> component (a 4:2:0 MCU is Y Y Y Y Cb Cr; at most 10 blocks, T.81 B.2.3).
struct McuLayout { int n_slots = 0; int comp_of[10] = {}; const Huff* dct[10] = {}; const Huff* act[10] = {}; };

struct ScanPart {
    std::vector<int16_t> coef;        // 64 per block, AC filled in; [0] written by the stitch
    std::vector<int32_t> dc;          // per block: DC with the part's predictors starting from 0
    std::vector<uint64_t> start;      // per block: bit position << 4 | slot in the MCU
    std::vector<uint32_t> bad;        // blocks that broke a rule (expected before the part has fallen into step, fatal after)
    CleanReader br{nullptr, 0}; int slot = 0; int32_t pred[4] = {0, 0, 0, 0};
    size_t n_main = 0;                // blocks that start inside the part's own range
    bool met = false; size_t met_at = 0, met_next = 0; int32_t pred_at_meeting[4] = {0, 0, 0, 0};
    void one_block(const McuLayout& L) {
        const size_t b = start.size();
        start.push_back((br.bitpos() << 4) | (uint64_t)slot);
        coef.resize((b + 1) * 64, 0);
        int diff = 0;
        const int e = decode_block(br, *L.dct[slot], *L.act[slot], diff, &coef[b * 64]);
        if (e) { bad.push_back((uint32_t)b); dc.push_back(0); slot = 0; return; }      // (out of step: try again from here as the first block of an MCU)
        const int c = L.comp_of[slot];
        pred[c] += diff; dc.push_back(pred[c]);
        slot = slot + 1 == L.n_slots ? 0 : slot + 1;
    }
};

bool entropy_parallel(const uint8_t* scan, const uint8_t* end, const McuLayout& L, int nc, size_t n_total /* blocks of the scan */,
                      std::vector<ScanPart>& parts, std::vector<const int16_t*>& block_ptr) {
    const bool trace = std::getenv("RRT_SETUP_TRACE") != nullptr;
    auto lap = [&, last = std::chrono::steady_clock::now()](const char* what) mutable {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[jpeg entropy] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count()); last = now;
    };
    // the scan up to its first marker, stuffing removed
    std::vector<uint8_t> clean; clean.reserve((size_t)(end - scan));
    for (const uint8_t* q = scan; q < end;) {
        const uint8_t* f = (const uint8_t*)std::memchr(q, 0xFF, (size_t)(end - q));
        if (!f) { clean.insert(clean.end(), q, end); break; }
        clean.insert(clean.end(), q, f);
        if (f + 1 < end && f[1] == 0) { clean.push_back(0xFF); q = f + 2; } else break;   // a marker (or a lone 0xFF at the end of the file): the scan ends here
    }
    lap("stuffing removed");
    const size_t nb = clean.size();
    size_t part_bytes = 24u << 10;                                                      // below this a part is not worth its thread
    if (const char* e = std::getenv("RRT_JPEG_PART_BYTES")) part_bytes = std::max<size_t>(8, (size_t)std::atoll(e));   // (tests: the stitching on small files)
    const size_t T = std::min<size_t>({(size_t)host_threads(), (size_t)24, nb / part_bytes});
    if (T < 2 || n_total >= (1ull << 32) || nb >= (1ull << 40)) return false;
    const int ns = L.n_slots;
    parts.assign(T, ScanPart{});
    auto part_end_bit = [&](size_t i) { return (uint64_t)(i + 1 == T ? nb : nb * (i + 1) / T) * 8; };
    parallel_ranges(T, 1, [&](size_t b, size_t e, size_t) {
        for (size_t i = b; i < e; i++) {
            ScanPart& P = parts[i];
            P.br = CleanReader{clean.data(), nb}; P.br.pos = nb * i / T;
            const size_t guess = n_total / T + n_total / (4 * T) + 64;
            P.coef.reserve(guess * 64); P.dc.reserve(guess); P.start.reserve(guess);
            const uint64_t stop = part_end_bit(i);
            while (P.br.bitpos() < stop && P.start.size() < n_total + 8) P.one_block(L);
            P.n_main = P.start.size();
        }
    });
    lap("parts decoded");
    parallel_ranges(T - 1, 1, [&](size_t b, size_t e, size_t) {
        for (size_t i = b; i < e; i++) {
            ScanPart& P = parts[i]; const ScanPart& N = parts[i + 1];
            size_t r = 0;
            for (;;) {
                const uint64_t key = (P.br.bitpos() << 4) | (uint64_t)P.slot;
                while (r < N.n_main && (N.start[r] >> 4) < (key >> 4)) r++;
                if (r >= N.n_main) break;                                            // crossed the whole next range without meeting its decoder
                if (N.start[r] == key) { P.met = true; P.met_at = P.start.size(); P.met_next = r; std::memcpy(P.pred_at_meeting, P.pred, sizeof P.pred); break; }
                if (P.start.size() >= n_total + 8) break;
                P.one_block(L);
            }
        }
    });
    lap("run-on until met");
    // stitch: global number of each part's block 0, first kept block, predictor offsets
    std::vector<int64_t> first_global(T, 0); std::vector<size_t> keep_from(T, 0), keep_to(T, 0);
    std::vector<std::array<int64_t, 4>> offset(T, std::array<int64_t, 4>{0, 0, 0, 0});
    for (size_t i = 0; i + 1 < T; i++) {
        const ScanPart& P = parts[i]; const ScanPart& N = parts[i + 1];
        if (!P.met || P.met_at < keep_from[i]) return false;
        keep_to[i] = P.met_at; keep_from[i + 1] = P.met_next;
        first_global[i + 1] = first_global[i] + (int64_t)P.met_at - (int64_t)P.met_next;
        const int64_t g = first_global[i] + (int64_t)P.met_at;
        if (g < 0 || (int)(g % ns) != (int)(N.start[P.met_next] & 15u)) return false;
        for (int c = 0; c < nc; c++) {
            int64_t local = 0;                                                       // the next part's own predictor of component c when it started the meeting block
            for (size_t k = P.met_next; k-- > 0;) if (L.comp_of[N.start[k] & 15u] == c && !std::binary_search(N.bad.begin(), N.bad.end(), (uint32_t)k)) { local = N.dc[k]; break; }
            offset[i + 1][c] = (int64_t)P.pred_at_meeting[c] + offset[i][c] - local;
        }
    }
    keep_to[T - 1] = parts[T - 1].start.size();
    if (first_global[T - 1] + (int64_t)keep_to[T - 1] < (int64_t)n_total) return false;   // a truncated scan: the serial decode feeds zeros and reports nothing -- let it
    block_ptr.assign(n_total, nullptr);
    std::vector<char> ok(T, 1);
    parallel_ranges(T, 1, [&](size_t b, size_t e, size_t) {
        for (size_t i = b; i < e; i++) {
            ScanPart& P = parts[i];
            for (size_t k = keep_from[i]; k < keep_to[i]; k++) {
                const int64_t g = first_global[i] + (int64_t)k;
                if (g >= (int64_t)n_total) break;
                const int s = (int)(P.start[k] & 15u);
                const int64_t v = (int64_t)P.dc[k] + offset[i][L.comp_of[s]];
                if (g < 0 || (int)(g % ns) != s || v < -32768 || v > 32767) { ok[i] = 0; break; }
                P.coef[k * 64] = (int16_t)v; block_ptr[(size_t)g] = &P.coef[k * 64];
            }
            for (uint32_t k : P.bad) if (k >= keep_from[i] && k < keep_to[i] && first_global[i] + (int64_t)k < (int64_t)n_total) ok[i] = 0;
        }
    });
    for (size_t i = 0; i < T; i++) if (!ok[i]) return false;
    for (size_t g = 0; g < n_total; g++) if (!block_ptr[g]) return false;
    if (trace) {
        size_t extra = 0, dropped = 0;
        for (size_t i = 0; i < T; i++) { extra += parts[i].start.size() - parts[i].n_main; dropped += keep_from[i]; }
        lap("stitched");
        fprintf(stderr, "[jpeg entropy] %zu parts, %zu blocks: %zu decoded twice (run-on), %zu dropped (out of step)\n", T, n_total, extra, dropped);
    }
    return true;
}

// ---- scans of a progressive frame (T.81 annex G; the procedures of IJG's jdphuff.c), one block at a time into the component's coefficient plane
struct ProgressiveState { uint32_t eobrun = 0; };
inline void store_coef(int16_t* dst, int32_t v) {
    if (v < -32768 || v > 32767) fail(RRT_ERR_PARSE, "Cannot decode texture file: coefficient out of range");
    *dst = (int16_t)v;
}
void prog_dc_first(BitReader& br, const Huff& dct, Component& c, int Al, int16_t* blk) {
    const int t = decode_symbol(br, dct);
    if (t < 0) fail(RRT_ERR_PARSE, block_error_text(kBadCode));
    if (t > 11) fail(RRT_ERR_PARSE, block_error_text(kBadDcCategory));
    c.pred += t ? extend(br.bits(t), t) : 0;
    if (c.pred < -32768 || c.pred > 32767) fail(RRT_ERR_PARSE, "Cannot decode texture file: DC predictor out of range");
    store_coef(blk, c.pred * (1 << Al));
}
void prog_dc_refine(BitReader& br, int Al, int16_t* blk) { if (br.bit()) blk[0] = (int16_t)(blk[0] | (1 << Al)); }
void prog_ac_first(BitReader& br, const Huff& act, ProgressiveState& st, int Ss, int Se, int Al, int16_t* blk) {
    if (st.eobrun) { st.eobrun--; return; }
    for (int k = Ss; k <= Se; k++) {
        const int rs = decode_symbol(br, act);
        if (rs < 0) fail(RRT_ERR_PARSE, block_error_text(kBadCode));
        const int r = rs >> 4, s = rs & 15;
        if (s) {
            k += r;
            if (k > Se) fail(RRT_ERR_PARSE, block_error_text(kBadAcRun));
            if (s > 14) fail(RRT_ERR_PARSE, block_error_text(kBadAcSize));
            store_coef(&blk[kZigzag[k]], extend(br.bits(s), s) * (1 << Al));
        } else if (r == 15) k += 15;
        else { st.eobrun = 1u << r; if (r) st.eobrun += (uint32_t)br.bits(r); st.eobrun--; break; }
    }
}
void prog_ac_refine(BitReader& br, const Huff& act, ProgressiveState& st, int Ss, int Se, int Al, int16_t* blk) {
    const int p1 = 1 << Al, m1 = -(1 << Al);
    auto correct = [&](int16_t* q) {                                    // one more bit of a coefficient that is already non-zero
        if (br.bit() && (*q & p1) == 0) *q = (int16_t)(*q + (*q >= 0 ? p1 : m1));
    };
    int k = Ss;
    if (st.eobrun == 0) {
        for (; k <= Se; k++) {
            const int rs = decode_symbol(br, act);
            if (rs < 0) fail(RRT_ERR_PARSE, block_error_text(kBadCode));
            int r = rs >> 4, s = rs & 15;
            if (s) {
                if (s != 1) fail(RRT_ERR_PARSE, block_error_text(kBadAcSize));
                s = br.bit() ? p1 : m1;
            } else if (r != 15) { st.eobrun = 1u << r; if (r) st.eobrun += (uint32_t)br.bits(r); break; }
            // pass the coefficients that are already non-zero (each takes a correction bit) and r zero ones
            do {
                int16_t* q = &blk[kZigzag[k]];
                if (*q != 0) correct(q);
                else if (--r < 0) break;
                k++;
            } while (k <= Se);
            if (s) { if (k > Se) fail(RRT_ERR_PARSE, block_error_text(kBadAcRun)); blk[kZigzag[k]] = (int16_t)s; }
        }
    }
    if (st.eobrun) {
        for (; k <= Se; k++) { int16_t* q = &blk[kZigzag[k]]; if (*q != 0) correct(q); }
        st.eobrun--;
    }
}

// libjpeg's "fancy" (triangle-filter) chroma upsampling, jdsample.c: h2v1_fancy_upsample / h2v2_fancy_upsample, sample for sample; w > 2 (narrower
// components are replicated, as there).  `out` takes 2 w samples.
void upsample_h2v1(const uint8_t* in, uint32_t w, uint8_t* out) {
    int v = *in++;
    *out++ = (uint8_t)v; *out++ = (uint8_t)((v * 3 + in[0] + 2) >> 2);
    for (uint32_t n = w - 2; n > 0; n--) { v = (*in++) * 3; *out++ = (uint8_t)((v + in[-2] + 1) >> 2); *out++ = (uint8_t)((v + in[0] + 2) >> 2); }
    v = *in;
    *out++ = (uint8_t)((v * 3 + in[-1] + 1) >> 2); *out++ = (uint8_t)v;
}
void upsample_h2v2(const uint8_t* near, const uint8_t* far, uint32_t w, uint8_t* out) {
    int cur = near[0] * 3 + far[0], next = near[1] * 3 + far[1], last;
    near += 2; far += 2;
    *out++ = (uint8_t)((cur * 4 + 8) >> 4); *out++ = (uint8_t)((cur * 3 + next + 7) >> 4);
    last = cur; cur = next;
    for (uint32_t n = w - 2; n > 0; n--) {
        next = (*near++) * 3 + (*far++);
        *out++ = (uint8_t)((cur * 3 + last + 8) >> 4); *out++ = (uint8_t)((cur * 3 + next + 7) >> 4);
        last = cur; cur = next;
    }
    *out++ = (uint8_t)((cur * 3 + last + 8) >> 4); *out++ = (uint8_t)((cur * 4 + 7) >> 4);
}

void decode_jpeg(const std::vector<uint8_t>& buf, std::vector<uint8_t>& out, uint32_t& W, uint32_t& H, uint32_t& channels) {
    const uint8_t* p = buf.data(); const uint8_t* end = p + buf.size();
    auto need = [&](size_t n) { if ((size_t)(end - p) < n) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated JPEG"); };
    need(2);
    if (p[0] != 0xFF || p[1] != 0xD8) fail(RRT_ERR_PARSE, "Cannot decode texture file: not a JPEG");
    p += 2;
    uint16_t qt[4][64] = {}; bool qt_present[4] = {};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int restart_interval = 0; bool have_sof = false, progressive = false; int adobe_transform = -1;
    uint32_t mcus_x = 0, mcus_y = 0; int hmax = 1, vmax = 1;
    struct Scan { int ns = 0; int ci[4] = {}; int Ss = 0, Se = 63, Ah = 0, Al = 0; };
    // Markers up to and including the next SOS.  Returns false at EOI (or at the end of the data: a file cut after a scan still shows what it holds).
    auto next_scan = [&](Scan& sc, bool first) -> bool {
        for (;;) {
            if ((size_t)(end - p) < 2) { if (first) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated JPEG"); return false; }
            if (p[0] != 0xFF) { if (first) fail(RRT_ERR_PARSE, "Cannot decode texture file: marker expected"); p++; continue; }   // (between scans: bytes the entropy decoder left over)
            while (p + 1 < end && p[1] == 0xFF) p++;            // fill bytes
            if (p + 1 >= end) { if (first) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated JPEG"); return false; }
            const uint8_t m = p[1]; p += 2;
            if (m == 0xD9) { if (first) fail(RRT_ERR_PARSE, "Cannot decode texture file: no scan"); return false; }
            if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if ((size_t)(end - p) < 2) { if (first) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated JPEG"); return false; }
            const size_t L = ((size_t)p[0] << 8) | p[1];
            if (L < 2) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad segment");
            if ((size_t)(end - p) < L) { if (first) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated JPEG"); return false; }
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
            } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {   // SOF0 / SOF1 (sequential) / SOF2 (progressive), Huffman
                if (have_sof) fail(RRT_ERR_PARSE, "Cannot decode texture file: second frame header");
                if (L < 8 || s[0] != 8) fail(RRT_ERR_UNSUPPORTED, "JPEG: only 8-bit precision");
                H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4];
                const int nc = s[5];
                if ((nc != 1 && nc != 3) || L < (size_t)(8 + 3 * nc) || W == 0 || H == 0) fail(RRT_ERR_UNSUPPORTED, "JPEG: unsupported component count");
                comps.resize(nc);
                for (int k = 0; k < nc; k++) { comps[k].id = s[6 + 3 * k]; comps[k].h = s[7 + 3 * k] >> 4; comps[k].v = s[7 + 3 * k] & 15; comps[k].tq = s[8 + 3 * k] & 3; }
                if (nc == 1) comps[0].h = comps[0].v = 1;        // one component: never interleaved, its sampling factors mean nothing (T.81 A.2.2)
                for (auto& c : comps) { if (c.h < 1 || c.v < 1) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad sampling factor"); hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
                // what libjpeg upsamples with its triangle filters (and PIL pins here): full resolution, 2:1 horizontally, 2:1 both ways -- 4:4:4, 4:2:2, 4:2:0
                for (auto& c : comps) {
                    const bool full = c.h == hmax && c.v == vmax, h2v1 = c.h * 2 == hmax && c.v == vmax, h2v2 = c.h * 2 == hmax && c.v * 2 == vmax;
                    if (!(full || h2v1 || h2v2) || hmax > 2 || vmax > 2) fail(RRT_ERR_UNSUPPORTED, "JPEG: chroma subsampling other than 4:4:4, 4:2:2 and 4:2:0 is not supported");
                }
                if (comps[0].h != hmax || comps[0].v != vmax) fail(RRT_ERR_UNSUPPORTED, "JPEG: a subsampled first component is not supported");
                mcus_x = (W + 8 * hmax - 1) / (8 * hmax); mcus_y = (H + 8 * vmax - 1) / (8 * vmax);
                for (auto& c : comps) { c.bw = mcus_x * c.h; c.bh = mcus_y * c.v; c.w = (W * c.h + hmax - 1) / hmax; c.hgt = (H * c.v + vmax - 1) / vmax; }
                progressive = m == 0xC2; have_sof = true;
            } else if (m >= 0xC3 && m <= 0xCF && m != 0xC8 && m != 0xCC) {
                fail(RRT_ERR_UNSUPPORTED, "JPEG: only Huffman-coded sequential and progressive DCT");
            } else if (m == 0xDD) {                             // DRI
                if (L >= 4) restart_interval = (s[0] << 8) | s[1];
            } else if (m == 0xEE) {                             // APP14 Adobe
                if (L >= 14 && !std::memcmp(s, "Adobe", 5)) adobe_transform = s[11];
            } else if (m == 0xDA) {                             // SOS
                if (!have_sof) fail(RRT_ERR_PARSE, "Cannot decode texture file: SOS before SOF");
                sc = Scan{};
                sc.ns = s[0];
                if (sc.ns < 1 || sc.ns > (int)comps.size() || L < (size_t)(6 + 2 * sc.ns)) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad SOS");
                for (int k = 0; k < sc.ns; k++) {
                    const int cid = s[1 + 2 * k]; int found = -1;
                    for (size_t c = 0; c < comps.size(); c++) if (comps[c].id == cid) { comps[c].td = s[2 + 2 * k] >> 4; comps[c].ta = s[2 + 2 * k] & 15; found = (int)c; }
                    if (found < 0 || comps[found].td > 3 || comps[found].ta > 3) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad SOS");
                    for (int j = 0; j < k; j++) if (sc.ci[j] == found) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad SOS");
                    sc.ci[k] = found;
                }
                sc.Ss = s[1 + 2 * sc.ns]; sc.Se = s[2 + 2 * sc.ns]; sc.Ah = s[3 + 2 * sc.ns] >> 4; sc.Al = s[3 + 2 * sc.ns] & 15;
                p += L;
                return true;
            }
            p += L;
        }
    };
    Scan sc;
    next_scan(sc, true);
    const size_t nc = comps.size();
    for (auto& c : comps) if (!qt_present[c.tq]) fail(RRT_ERR_PARSE, "Cannot decode texture file: missing table");
    const bool trace = std::getenv("RRT_SETUP_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();

    // ---- phase 1: the entropy-coded data -> quantised coefficients (natural order) of every block of every component
    std::vector<int16_t> coef; std::vector<ScanPart> parts; std::vector<const int16_t*> block_ptr;
    McuLayout layout; int slot_base[4] = {};
    for (size_t c = 0; c < nc; c++) { slot_base[c] = layout.n_slots; for (int k = 0; k < comps[c].h * comps[c].v; k++) layout.comp_of[layout.n_slots++] = (int)c; }
    const size_t n_stream = (size_t)mcus_x * mcus_y * layout.n_slots;
    const bool one_scan = !progressive && sc.ns == (int)nc;            // the usual file: one interleaved scan holds everything
    bool in_parallel = false;
    auto restart_if_due = [&](BitReader& br, int& to_restart, ProgressiveState* ps) {
        if (!restart_interval || to_restart != 0) return;
        const uint8_t* q = br.p;                                        // byte-align, expect RSTn (the reader never passes a marker: it is at or ahead of br.p)
        while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
        if (q + 1 >= end) fail(RRT_ERR_PARSE, "Cannot decode texture file: missing RST");
        br.reset_at(q + 2);
        for (auto& c : comps) c.pred = 0;
        if (ps) ps->eobrun = 0;
        to_restart = restart_interval;
    };
    if (one_scan) {
        for (auto& c : comps) if (!dc[c.td].present || !ac[c.ta].present) fail(RRT_ERR_PARSE, "Cannot decode texture file: missing table");
        for (int s = 0; s < layout.n_slots; s++) { layout.dct[s] = &dc[comps[layout.comp_of[s]].td]; layout.act[s] = &ac[comps[layout.comp_of[s]].ta]; }
        // on several threads when the scan is long and has no restart markers (entropy_parallel above); one block after the other otherwise, and
        // whenever that attempt declines
        in_parallel = !restart_interval && !std::getenv("RRT_JPEG_SERIAL") && entropy_parallel(p, end, layout, (int)nc, n_stream, parts, block_ptr);
        if (!in_parallel) {
            parts.clear();
            coef.assign(n_stream * 64, 0);
            block_ptr.resize(n_stream);
            for (size_t g = 0; g < n_stream; g++) block_ptr[g] = &coef[g * 64];
            BitReader br{p, end};
            int to_restart = restart_interval;
            size_t g = 0;
            for (size_t m = 0; m < (size_t)mcus_x * mcus_y; m++) {
                restart_if_due(br, to_restart, nullptr);
                for (int s = 0; s < layout.n_slots; s++, g++) {
                    Component& c = comps[layout.comp_of[s]];
                    int diff = 0;
                    if (const int e = decode_block(br, *layout.dct[s], *layout.act[s], diff, &coef[g * 64])) fail(RRT_ERR_PARSE, block_error_text(e));
                    c.pred += diff;
                    if (c.pred < -32768 || c.pred > 32767) fail(RRT_ERR_PARSE, "Cannot decode texture file: DC predictor out of range");
                    coef[g * 64] = (int16_t)c.pred;
                }
                if (restart_interval) to_restart--;
            }
        }
    } else {
        // several scans (progressive, or a sequential file with a scan per component): coefficient planes, one scan after the other, until EOI
        for (auto& c : comps) c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        for (bool more = true; more; more = next_scan(sc, false)) {
            if (progressive) {
                if (sc.Ss > sc.Se || sc.Se > 63 || sc.Al > 13 || (sc.Ss == 0 && sc.Se != 0) || (sc.Ss > 0 && sc.ns != 1)) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad progressive scan");
            } else { sc.Ss = 0; sc.Se = 63; sc.Ah = sc.Al = 0; }
            for (int k = 0; k < sc.ns; k++) {
                const Component& c = comps[sc.ci[k]];
                const bool need_dc = sc.Ss == 0 && sc.Ah == 0, need_ac = sc.Se > 0;
                if ((need_dc && !dc[c.td].present) || (need_ac && !ac[c.ta].present)) fail(RRT_ERR_PARSE, "Cannot decode texture file: missing table");
            }
            for (auto& c : comps) c.pred = 0;
            BitReader br{p, end};
            ProgressiveState ps;
            int to_restart = restart_interval;
            auto block = [&](Component& c, uint32_t bx, uint32_t by) {
                int16_t* blk = &c.coef[((size_t)by * c.bw + bx) * 64];
                if (!progressive) {
                    int diff = 0;
                    if (const int e = decode_block(br, dc[c.td], ac[c.ta], diff, blk)) fail(RRT_ERR_PARSE, block_error_text(e));
                    c.pred += diff;
                    if (c.pred < -32768 || c.pred > 32767) fail(RRT_ERR_PARSE, "Cannot decode texture file: DC predictor out of range");
                    blk[0] = (int16_t)c.pred;
                } else if (sc.Ss == 0) { if (sc.Ah == 0) prog_dc_first(br, dc[c.td], c, sc.Al, blk); else prog_dc_refine(br, sc.Al, blk); }
                else if (sc.Ah == 0) prog_ac_first(br, ac[c.ta], ps, sc.Ss, sc.Se, sc.Al, blk);
                else prog_ac_refine(br, ac[c.ta], ps, sc.Ss, sc.Se, sc.Al, blk);
            };
            if (sc.ns == 1) {                                           // not interleaved: the component's own blocks, only those that hold image samples (T.81 A.2.2)
                Component& c = comps[sc.ci[0]];
                const uint32_t nbx = (c.w + 7) / 8, nby = (c.hgt + 7) / 8;
                for (uint32_t by = 0; by < nby; by++)
                    for (uint32_t bx = 0; bx < nbx; bx++) { restart_if_due(br, to_restart, &ps); block(c, bx, by); if (restart_interval) to_restart--; }
            } else {
                for (uint32_t my = 0; my < mcus_y; my++)
                    for (uint32_t mx = 0; mx < mcus_x; mx++) {
                        restart_if_due(br, to_restart, &ps);
                        for (int k = 0; k < sc.ns; k++) {
                            Component& c = comps[sc.ci[k]];
                            for (int sy = 0; sy < c.v; sy++) for (int sx = 0; sx < c.h; sx++) block(c, mx * c.h + sx, my * c.v + sy);
                        }
                        if (restart_interval) to_restart--;
                    }
            }
            p = br.p;                                                    // the reader never passes a marker: the next one is at or after this
        }
    }
    auto block_of = [&](size_t ci, uint32_t bx, uint32_t by) -> const int16_t* {
        const Component& c = comps[ci];
        if (!one_scan) return &c.coef[((size_t)by * c.bw + bx) * 64];
        const size_t m = (size_t)(by / c.v) * mcus_x + bx / c.h;
        return block_ptr[m * layout.n_slots + slot_base[ci] + (by % c.v) * c.h + bx % c.h];
    };

    // ---- phase 2: dequantise + inverse DCT + (upsampling +) colour conversion; subsampled files via a sample plane per component (padded to whole blocks)
    const auto t_entropy = std::chrono::steady_clock::now();
    const bool grey = nc == 1;
    channels = grey ? 1 : 3; out.resize((size_t)W * H * channels);
    const bool ycc = adobe_transform < 0 ? !(!grey && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') : adobe_transform != 0;
    // IJG jdcolor.c: 16.16 fixed point, FIX(x) = (int)(x*65536 + 0.5)
    constexpr int32_t ONE_HALF = 1 << 15, F1_402 = 91881, F1_772 = 116130, F0_714 = 46802, F0_344 = 22554;
    auto convert_row = [&](const uint8_t* Y, const uint8_t* Cb, const uint8_t* Cr, uint8_t* o) {
        for (uint32_t x = 0; x < W; x++) {
            if (!ycc) { o[3 * x] = Y[x]; o[3 * x + 1] = Cb[x]; o[3 * x + 2] = Cr[x]; continue; }
            int32_t yy = Y[x], cb = Cb[x] - 128, cr = Cr[x] - 128;
            int32_t rr = yy + ((F1_402 * cr + ONE_HALF) >> 16);
            int32_t g = yy + ((-F0_344 * cb + ONE_HALF - F0_714 * cr) >> 16);
            int32_t b = yy + ((F1_772 * cb + ONE_HALF) >> 16);
            o[3 * x] = clamp255(rr); o[3 * x + 1] = clamp255(g); o[3 * x + 2] = clamp255(b);
        }
    };
    auto idct_block = [&](size_t ci, uint32_t bx, uint32_t by, uint8_t* dst, size_t stride) {
        const int16_t* q = block_of(ci, bx, by);
        const uint16_t* t = qt[comps[ci].tq];
        int32_t blk[64];
        for (int k = 0; k < 64; k++) { const int32_t v = (int32_t)q[k] * t[k]; blk[k] = v < -(1 << 15) ? -(1 << 15) : v > (1 << 15) ? (1 << 15) : v; }   // hostile tables: keep the IDCT inside int32 (a valid file never gets near)
        idct_islow(blk, dst, stride);
    };
    bool full_res = true;
    for (auto& c : comps) full_res = full_res && c.h == hmax && c.v == vmax;
    if (full_res) {
        // every component at full resolution (the scene's textures): a row of blocks at a time through a small buffer of the worker's own, converted at
        // once -- whole sample planes would be 3 MB of freshly mapped pages per texture, faulted in by a hundred threads at the same moment
        const uint32_t bw = comps[0].bw, bh = comps[0].bh;
        const size_t stride = (size_t)bw * 8;
        parallel_ranges(bh, 4, [&](size_t rb, size_t re, size_t) {
            std::vector<uint8_t> rows(nc * 8 * stride);
            for (size_t by = rb; by < re; by++) {
                for (uint32_t bx = 0; bx < bw; bx++)
                    for (size_t ci = 0; ci < nc; ci++) idct_block(ci, bx, (uint32_t)by, rows.data() + ci * 8 * stride + (size_t)bx * 8, stride);
                for (uint32_t r = 0; r < 8; r++) {
                    const size_t y = by * 8 + r;
                    if (y >= H) break;
                    const uint8_t* Y = rows.data() + r * stride;
                    if (grey) std::memcpy(out.data() + y * W, Y, W);
                    else convert_row(Y, rows.data() + (8 + r) * stride, rows.data() + (16 + r) * stride, out.data() + y * W * 3);
                }
            }
        });
    } else {
    std::vector<std::unique_ptr<uint8_t[]>> plane(nc);
    std::vector<size_t> row_first(nc + 1, 0);                           // block rows of all components, end to end
    for (size_t c = 0; c < nc; c++) { plane[c].reset(new uint8_t[(size_t)comps[c].bw * 8 * comps[c].bh * 8 + 16]); row_first[c + 1] = row_first[c] + comps[c].bh; }
    parallel_ranges(row_first[nc], 2, [&](size_t rb, size_t re, size_t) {
        for (size_t r = rb; r < re; r++) {
            size_t ci = 0; while (r >= row_first[ci + 1]) ci++;
            const Component& c = comps[ci];
            const uint32_t by = (uint32_t)(r - row_first[ci]);
            const size_t stride = (size_t)c.bw * 8;
            for (uint32_t bx = 0; bx < c.bw; bx++) idct_block(ci, bx, by, plane[ci].get() + (size_t)by * 8 * stride + (size_t)bx * 8, stride);
        }
    });
    // ... then chroma upsampling (libjpeg's triangle filters; the rows above the first and below the last are those rows again, jdmainct.c) and the
    // colour conversion, every pixel row on its own
    parallel_ranges(H, 16, [&](size_t yb, size_t ye, size_t) {
        std::vector<uint8_t> tmp[3];
        for (size_t c = 1; c < nc; c++) tmp[c].resize((size_t)comps[c].w * 2 + 8);
        for (size_t y = yb; y < ye; y++) {
            const uint8_t* row[3] = {nullptr, nullptr, nullptr};
            for (size_t ci = 0; ci < nc; ci++) {
                const Component& c = comps[ci];
                const size_t stride = (size_t)c.bw * 8;
                const uint8_t* base = plane[ci].get();
                if (c.h == hmax && c.v == vmax) row[ci] = base + y * stride;
                else if (c.w <= 2) {                                     // jdsample.c uses the triangle filters for components wider than two samples only: else replication
                    const uint8_t* in = base + (c.v == vmax ? y : y / 2) * stride;
                    for (uint32_t x = 0; x < 2 * c.w; x++) tmp[ci][x] = in[x / 2];
                    row[ci] = tmp[ci].data();
                }
                else if (c.v == vmax) { upsample_h2v1(base + y * stride, c.w, tmp[ci].data()); row[ci] = tmp[ci].data(); }
                else {
                    const size_t r = y / 2;
                    const size_t far = (y & 1) ? std::min<size_t>(r + 1, c.hgt - 1) : (r ? r - 1 : 0);
                    upsample_h2v2(base + r * stride, base + far * stride, c.w, tmp[ci].data()); row[ci] = tmp[ci].data();
                }
            }
            convert_row(row[0], row[1], row[2], out.data() + y * W * 3);
        }
    });
    }
    if (trace) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "[jpeg] %u x %u, %zu bytes%s: entropy decode %.2f ms (%zu threads), dequantise + IDCT + upsampling + colour %.2f ms\n", W, H, buf.size(), progressive ? ", progressive" : "",
                std::chrono::duration<double, std::milli>(t_entropy - t_begin).count(), in_parallel ? parts.size() : (size_t)1, std::chrono::duration<double, std::milli>(t_end - t_entropy).count());
    }
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
    if (trns || depth == 16) fail(RRT_ERR_UNSUPPORTED, "PNG: 16-bit samples and tRNS transparency are not supported");   // (neither decodes to 3 bytes per pixel in the `image` crate)
    const int src_ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!src_ch) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad PNG colour type");
    const bool depth_ok = depth == 8 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4));
    if (!depth_ok || interlace > 1) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad PNG header");
    // one sub-image per pass (Adam7: seven, each a lattice of the picture; else the picture itself), every one filtered on its own (PNG spec 8.2, 9)
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass kAdam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass kWhole = {0, 0, 1, 1};
    const int n_pass = interlace ? 7 : 1;
    const size_t bpp = std::max<size_t>(1, (size_t)src_ch * depth / 8);                       // filter distance in bytes
    auto pass_dims = [&](const Pass& P, uint32_t& pw, uint32_t& ph) { pw = W > P.x0 ? (W - P.x0 + P.dx - 1) / P.dx : 0; ph = H > P.y0 ? (H - P.y0 + P.dy - 1) / P.dy : 0; };
    size_t raw_total = 0;
    for (int k = 0; k < n_pass; k++) {
        uint32_t pw, ph; pass_dims(interlace ? kAdam7[k] : kWhole, pw, ph);
        if (pw && ph) raw_total += (((size_t)pw * src_ch * depth + 7) / 8 + 1) * ph;
    }
    std::vector<uint8_t> raw(raw_total);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) fail(RRT_ERR_PARSE, "Cannot decode texture file: PNG inflate failed");
    std::vector<uint8_t> img((size_t)W * H * src_ch);                                         // one byte per sample
    const int maxval = (1 << depth) - 1;
    size_t at = 0;
    std::vector<uint8_t> prev, cur;
    for (int k = 0; k < n_pass; k++) {
        const Pass& P = interlace ? kAdam7[k] : kWhole;
        uint32_t pw, ph; pass_dims(P, pw, ph);
        if (!pw || !ph) continue;
        const size_t row = ((size_t)pw * src_ch * depth + 7) / 8;
        prev.assign(row, 0); cur.assign(row, 0);
        for (uint32_t y = 0; y < ph; y++) {
            const uint8_t ft = raw[at]; const uint8_t* sline = &raw[at + 1]; at += row + 1;
            for (size_t x = 0; x < row; x++) {
                const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
                int pred = 0;
                switch (ft) {
                    case 0: pred = 0; break;
                    case 1: pred = a; break;
                    case 2: pred = b; break;
                    case 3: pred = (a + b) >> 1; break;
                    case 4: { int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                    default: fail(RRT_ERR_PARSE, "Cannot decode texture file: bad PNG filter");
                }
                cur[x] = (uint8_t)(sline[x] + pred);
            }
            uint8_t* orow = &img[((size_t)(P.y0 + (size_t)y * P.dy) * W) * src_ch];
            for (uint32_t x = 0; x < pw; x++) {
                uint8_t* o = orow + ((size_t)P.x0 + (size_t)x * P.dx) * src_ch;
                if (depth == 8) for (int q = 0; q < src_ch; q++) o[q] = cur[(size_t)x * src_ch + q];
                else {                                                                       // 1, 2, 4 bits: one channel, most significant bits first
                    const int v = (cur[((size_t)x * depth) >> 3] >> (8 - depth - (((size_t)x * depth) & 7))) & maxval;
                    o[0] = ctype == 3 ? (uint8_t)v : (uint8_t)(v * 255 / maxval);             // palette indices as they are, grey levels scaled to 8 bits
                }
            }
            prev.swap(cur);
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

// ===================================================================== BMP, TGA (the uncompressed formats an .mtl is likely to name besides JPEG and PNG)
uint32_t le16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
uint32_t le32(const uint8_t* p) { return le16(p) | (le16(p + 2) << 16); }

// Windows bitmap: BITMAPINFOHEADER family, uncompressed 24-bit BGR or 1/4/8-bit palette (-> RGB8, as the `image` crate expands them); rows bottom-up
// unless the height is negative.  32-bit and compressed bitmaps do not decode to 3 bytes per pixel: refused.
void decode_bmp(const std::vector<uint8_t>& buf, std::vector<uint8_t>& out, uint32_t& W, uint32_t& H, uint32_t& channels) {
    if (buf.size() < 54) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated BMP");
    const uint32_t data_off = le32(&buf[10]), hdr = le32(&buf[14]);
    if (hdr < 40 || 14 + (size_t)hdr > buf.size()) fail(RRT_ERR_UNSUPPORTED, "BMP: unsupported header");
    const int32_t w = (int32_t)le32(&buf[18]), h = (int32_t)le32(&buf[22]);
    const uint32_t planes = le16(&buf[26]), bpp = le16(&buf[28]), comp = le32(&buf[30]); uint32_t n_pal = le32(&buf[46]);
    if (planes != 1 || w <= 0 || h == 0 || h == INT32_MIN) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad BMP header");
    W = (uint32_t)w; H = (uint32_t)(h < 0 ? -h : h);
    if (W > kMaxImageDim || H > kMaxImageDim) fail(RRT_ERR_PARSE, "Cannot decode texture file: BMP dimensions beyond the supported 65535 x 65535");
    if (comp != 0 || !(bpp == 24 || bpp == 8 || bpp == 4 || bpp == 1)) fail(RRT_ERR_UNSUPPORTED, "BMP: only uncompressed 24-bit or palette bitmaps");
    const size_t row = (((size_t)W * bpp + 31) / 32) * 4;
    if ((size_t)data_off > buf.size() || row * H > buf.size() - data_off) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated BMP");
    const uint8_t* pal = &buf[14 + hdr];
    if (bpp <= 8) { if (n_pal == 0) n_pal = 1u << bpp; if (n_pal > 256 || 14 + (size_t)hdr + 4 * (size_t)n_pal > buf.size()) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad BMP palette"); }
    channels = 3; out.resize((size_t)W * H * 3);
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t* s = &buf[data_off + row * (h < 0 ? y : H - 1 - y)];
        uint8_t* o = &out[(size_t)y * W * 3];
        for (uint32_t x = 0; x < W; x++) {
            if (bpp == 24) { o[3 * x] = s[3 * x + 2]; o[3 * x + 1] = s[3 * x + 1]; o[3 * x + 2] = s[3 * x]; continue; }
            const uint32_t idx = bpp == 8 ? s[x] : bpp == 4 ? (s[x >> 1] >> ((x & 1) ? 0 : 4)) & 15u : (s[x >> 3] >> (7 - (x & 7))) & 1u;
            if (idx >= n_pal) fail(RRT_ERR_PARSE, "Cannot decode texture file: palette index out of range");
            o[3 * x] = pal[4 * idx + 2]; o[3 * x + 1] = pal[4 * idx + 1]; o[3 * x + 2] = pal[4 * idx];
        }
    }
}

// Truevision TGA: true-colour 24-bit, plain (type 2) or run-length coded (type 10); rows bottom-up unless the descriptor says top-down.  Has no signature:
// recognised by the file name, as the `image` crate does for ImageReader::open.
void decode_tga(const std::vector<uint8_t>& buf, std::vector<uint8_t>& out, uint32_t& W, uint32_t& H, uint32_t& channels) {
    if (buf.size() < 18) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated TGA");
    const uint32_t id_len = buf[0], cmap_type = buf[1], type = buf[2], cmap_len = le16(&buf[5]), cmap_bits = buf[7], bpp = buf[16], desc = buf[17];
    W = le16(&buf[12]); H = le16(&buf[14]);
    if (W == 0 || H == 0) fail(RRT_ERR_PARSE, "Cannot decode texture file: bad TGA header");
    if (!(type == 2 || type == 10) || bpp != 24 || (desc & 0x10)) fail(RRT_ERR_UNSUPPORTED, "TGA: only 24-bit true-colour images (plain or run-length coded)");
    size_t at = 18 + (size_t)id_len + (cmap_type ? (size_t)cmap_len * ((cmap_bits + 7) / 8) : 0);
    channels = 3; out.resize((size_t)W * H * 3);
    const size_t n = (size_t)W * H;
    const bool top_down = desc & 0x20;
    auto put = [&](size_t k, const uint8_t* bgr) {
        const size_t y = k / W, x = k % W;
        uint8_t* o = &out[((top_down ? y : H - 1 - y) * W + x) * 3];
        o[0] = bgr[2]; o[1] = bgr[1]; o[2] = bgr[0];
    };
    if (type == 2) {
        if (at > buf.size() || n * 3 > buf.size() - at) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated TGA");
        for (size_t k = 0; k < n; k++) put(k, &buf[at + 3 * k]);
    } else {
        for (size_t k = 0; k < n;) {
            if (at >= buf.size()) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated TGA");
            const uint32_t c = buf[at++], run = (c & 127u) + 1;
            if (k + run > n) fail(RRT_ERR_PARSE, "Cannot decode texture file: TGA run past the end of the image");
            if (c & 128u) { if (buf.size() - at < 3) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated TGA"); for (uint32_t r = 0; r < run; r++) put(k + r, &buf[at]); at += 3; }
            else { if (buf.size() - at < 3 * (size_t)run) fail(RRT_ERR_PARSE, "Cannot decode texture file: truncated TGA"); for (uint32_t r = 0; r < run; r++) put(k + r, &buf[at + 3 * r]); at += 3 * (size_t)run; }
            k += run;
        }
    }
}

void decode_image_file(const std::string& path, std::vector<uint8_t>& bytes, uint32_t& width, uint32_t& height, uint32_t& channels) {
    std::vector<uint8_t> buf = slurp(path);
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (buf.size() >= 8 && !std::memcmp(buf.data(), png_sig, 8)) decode_png(buf, bytes, width, height, channels);
    else if (buf.size() >= 2 && buf[0] == 0xFF && buf[1] == 0xD8) decode_jpeg(buf, bytes, width, height, channels);
    else if (buf.size() >= 2 && buf[0] == 'B' && buf[1] == 'M') decode_bmp(buf, bytes, width, height, channels);
    else if (path.size() >= 4 && (path.compare(path.size() - 4, 4, ".tga") == 0 || path.compare(path.size() - 4, 4, ".TGA") == 0)) decode_tga(buf, bytes, width, height, channels);
    else fail(RRT_ERR_UNSUPPORTED, "Cannot decode texture file (not JPEG, PNG, BMP or TGA): " + path);
}

}  // namespace rrt
