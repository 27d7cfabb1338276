// obj_loader.cpp -- the .obj/.mtl reader feeding the hot path (src/file_management/utils.rs:22-379 of the reference).
// Same directive set, same defaults and the same failure points; a reference `expect`/`unwrap`/`assert!` panic
// becomes rrt::Error{RRT_ERR_PARSE|RRT_ERR_IO} carried back over the C ABI as a status code.
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <memory>
#include <unordered_map>
#include <optional>
#include <sstream>
#include <string_view>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "model.hpp"
#include "parallel.hpp"

namespace rrt {
namespace {

using sv = std::string_view;

[[noreturn]] void fail(int status, const std::string& what) { throw Error{status, what}; }

// str::split_whitespace (ASCII subset)
struct Tokens {
    sv rest;
    std::optional<sv> next() {
        size_t i = 0;
        while (i < rest.size() && (rest[i] == ' ' || rest[i] == '\t' || rest[i] == '\r' || rest[i] == '\n' || rest[i] == '\f' || rest[i] == '\v')) i++;
        if (i == rest.size()) { rest = {}; return std::nullopt; }
        size_t j = i;
        while (j < rest.size() && !(rest[j] == ' ' || rest[j] == '\t' || rest[j] == '\r' || rest[j] == '\n' || rest[j] == '\f' || rest[j] == '\v')) j++;
        sv tok = rest.substr(i, j - i);
        rest = rest.substr(j);
        return tok;
    }
};

// <f64 as FromStr>: decimal/exponent forms, optional sign, inf/infinity/nan; correctly rounded like from_chars.
double parse_f64(sv s) {
    sv body = s;
    if (!body.empty() && body[0] == '+') body.remove_prefix(1);   // from_chars rejects '+', Rust accepts it
    if (body.empty() || body[0] == '+') fail(RRT_ERR_PARSE, "Could not parse value: '" + std::string(s) + "'");
    double v = 0;
    auto r = std::from_chars(body.data(), body.data() + body.size(), v, std::chars_format::general);
    if (r.ec == std::errc::result_out_of_range) {   // Rust saturates to +-inf / 0 instead of failing
        std::string tmp(body);
        v = std::strtod(tmp.c_str(), nullptr);
    } else if (r.ec != std::errc() || r.ptr != body.data() + body.size()) {
        fail(RRT_ERR_PARSE, "Could not parse value: '" + std::string(s) + "'");   // utils.rs:222
    }
    return v;
}

// <usize as FromStr>
uint64_t parse_usize(sv s) {
    sv body = s;
    if (!body.empty() && body[0] == '+') body.remove_prefix(1);
    uint64_t v = 0;
    auto r = std::from_chars(body.data(), body.data() + body.size(), v, 10);
    if (body.empty() || r.ec != std::errc() || r.ptr != body.data() + body.size())
        fail(RRT_ERR_PARSE, "Could not parse value: '" + std::string(s) + "'");       // utils.rs:222
    return v;
}

Vec3 get_vertex(Tokens& t) {   // utils.rs:228-234
    auto x = t.next(), y = t.next(), z = t.next();
    if (!x || !y) fail(RRT_ERR_PARSE, "Cannot parse vertex");
    Vec3 v; v.x = parse_f64(*x); v.y = parse_f64(*y); v.z = z ? parse_f64(*z) : 0.0;
    return v;
}

Vec3 get_color_coefficient(Tokens& t) {   // utils.rs:370-379
    Vec3 c = get_vertex(t);
    if (!(c.x <= 1.0 && c.y <= 1.0 && c.z <= 1.0 && c.x >= 0.0 && c.y >= 0.0 && c.z >= 0.0))
        fail(RRT_ERR_PARSE, "All lighting intensity coefficients must be between 0.0 and 1.0");
    return c;
}

std::string read_file(const std::string& path) {   // fs::read_to_string, main.rs:28 / utils.rs:171
    std::ifstream f(path, std::ios::binary);
    if (!f) fail(RRT_ERR_IO, "Could not read file: " + path);
    std::ostringstream ss; ss << f.rdbuf();
    return ss.str();
}

// str::lines(): split on '\n', drop one trailing '\r'
template <class F> void for_each_line(sv text, F&& fn) {
    size_t i = 0;
    while (i < text.size()) {
        size_t j = text.find('\n', i);
        if (j == sv::npos) j = text.size();
        sv line(text.data() + i, j - i);
        if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
        fn(line);
        i = j + 1;
    }
}

// Texture decode started before the .obj text is tokenised.  The reference decodes a texture where its "map_Ka"/"bump" line stands, which is inside the
// "mtllib" directive, which stands among the first lines of the file -- but the directive is only executed (in file order, with everything before
// it) after the tokeniser has been over the whole text.  The six JPEGs of the teapot's materials take as long as that, so: the head of the text is
// searched for an "mtllib" line, the files named by "map_Ka"/"bump" lines of that .mtl are decoded on the host pool meanwhile, and when the directive
// is executed for real, a texture whose file is already decoded is taken from here.  Only successful decodes are kept (same bytes as a decode on the
// spot); everything that can fail fails where and when it did before.
struct TexturePrefetch {
    struct Image { std::vector<uint8_t> bytes; uint32_t w = 0, h = 0, ch = 0; };
    std::mutex mu;
    std::unordered_map<std::string, Image> images;                        // by path
    std::unique_ptr<AsyncTask> task;
    static std::shared_ptr<TexturePrefetch> start(const std::string& dir, sv head);
    void wait() { if (task) task->wait(); }                               // (decodes here and now if no worker has started on it)
    bool take(const std::string& path, Image& out) {                     // after wait()
        std::lock_guard<std::mutex> g(mu);
        auto it = images.find(path);
        if (it == images.end()) return false;
        out = std::move(it->second); images.erase(it);
        return true;
    }
};

struct Loader {
    std::string dir;
    Model& m;
    std::unordered_map<std::string, uint32_t> material_by_name;   // MaterialMap.materials, material.rs:25-28

    // get_texture_from_file_name, utils.rs:345-368.  The reference decodes a texture where its "map_Ka"/"bump" line stands; here the line only
    // reserves the texture's index and the files of one .mtl are decoded together afterwards, one worker per file (six 1024x1024 JPEGs are
    // most of the teapot's set-up time).  A decode failure is still reported in line order: before any later failure of the .mtl text.
    std::vector<std::pair<uint32_t, std::string>> pending_textures;
    std::shared_ptr<TexturePrefetch> prefetch;
    uint32_t load_texture(const std::string& name) {
        m.textures.emplace_back();
        pending_textures.emplace_back((uint32_t)m.textures.size() - 1, name);
        return (uint32_t)m.textures.size() - 1;
    }
    void decode_pending_textures() {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::pair<uint32_t, std::string>> work;
        work.swap(pending_textures);
        if (prefetch) prefetch->wait();
        parallel_ranges(work.size(), 1, [&](size_t b, size_t e, size_t) {
            for (size_t i = b; i < e; i++) {
                std::vector<uint8_t> bytes; uint32_t w = 0, h = 0, ch = 0;
                TexturePrefetch::Image ready;
                if (prefetch && prefetch->take(dir + work[i].second, ready)) { bytes = std::move(ready.bytes); w = ready.w; h = ready.h; ch = ready.ch; }
                else decode_image_file(dir + work[i].second, bytes, w, h, ch);
                // utils.rs:353 walks `as_bytes().chunks(3)` whatever the colour type; only a 3-byte-per-pixel buffer gives
                // width*height colours, anything else indexes out of bounds later (raytracer.rs:55).  Refuse it here.
                if (ch != 3) fail(RRT_ERR_UNSUPPORTED, "texture '" + work[i].second + "' is not 3 bytes per pixel");
                Texture& t = m.textures[work[i].first];
                t.rgb = std::move(bytes); t.width = w; t.height = h;
            }
        });
        m.texture_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }

    void parse_mtl(const std::string& text) {   // parse_mtl_file_lines, utils.rs:22-137
        std::unordered_map<std::string, uint32_t> name_texture_map;   // local to one call, utils.rs:33
        std::optional<std::string> name;
        std::optional<Vec3> ka, kd, ks; std::optional<double> ns, kr; std::optional<uint32_t> tex, bump;

        auto on_line = [&](sv line) {
            Tokens t{line};
            auto type = t.next();
            if (!type) return;
            if (*type == "newmtl" || *type == "END") {                 // utils.rs:50-82
                if (name) {
                    if (!tex) fail(RRT_ERR_PARSE, "material '" + *name + "' has no map_Ka texture");   // utils.rs:61 unwrap
                    rrt_material mat{};
                    Vec3 z{};
                    Vec3 a = ka.value_or(z), d = kd.value_or(z), s = ks.value_or(z);
                    mat.ka = {a.x, a.y, a.z}; mat.kd = {d.x, d.y, d.z}; mat.ks = {s.x, s.y, s.z};
                    mat.ns = ns.value_or(240.0);                        // utils.rs:60
                    mat.kr = kr.value_or(0.0);                          // utils.rs:63
                    mat.tex = (int32_t)*tex; mat.bump = bump ? (int32_t)*bump : -1;
                    m.materials.push_back(mat);
                    material_by_name[*name] = (uint32_t)m.materials.size() - 1;   // HashMap::insert replaces
                    ka.reset(); kd.reset(); ks.reset(); ns.reset(); tex.reset(); bump.reset(); kr.reset();   // utils.rs:70-77
                }
                auto nn = t.next();                                     // utils.rs:80-81
                name = nn ? std::optional<std::string>(std::string(*nn)) : std::nullopt;
            } else if (*type == "map_Ka" || *type == "bump") {         // utils.rs:83-110
                auto nm = t.next();
                if (!nm) fail(RRT_ERR_PARSE, "Expected a texture name");
                std::string key(*nm);
                uint32_t id;
                auto it = name_texture_map.find(key);
                if (it != name_texture_map.end()) id = it->second;
                else { id = load_texture(key); name_texture_map[key] = id; }
                if (*type == "map_Ka") tex = id; else bump = id;
            } else if (*type == "Ka") { ka = get_color_coefficient(t);  // utils.rs:111-122
            } else if (*type == "Kd") { kd = get_color_coefficient(t);
            } else if (*type == "Ks") { ks = get_color_coefficient(t);
            } else if (*type == "Ns") {                                 // utils.rs:123-127
                auto x = t.next();
                if (!x) fail(RRT_ERR_PARSE, "Expected a valid Ns float value");
                ns = parse_f64(*x);
            } else if (*type == "Kr") {                                 // utils.rs:128-132
                auto x = t.next();
                double r = x ? parse_f64(*x) : 0.0;
                if (r < 0.0) r = 0.0;                                   // f64::clamp keeps NaN
                if (r > 1.0) r = 1.0;
                kr = r;
            }
        };
        try {
            for_each_line(text, on_line);
            on_line("END");                                             // utils.rs:44-45
        } catch (const Error&) {
            decode_pending_textures();                                  // a texture named before the failing line fails first, as in the reference
            throw;
        }
        decode_pending_textures();
    }

    // get_vertex_attributes, utils.rs:236-251: "i", "i/t" or "i/t/n"
    static void vertex_attributes(sv tok, uint64_t& vi, std::optional<uint64_t>& ti, std::optional<uint64_t>& ni) {
        size_t p1 = tok.find('/');
        vi = parse_usize(tok.substr(0, p1));
        ti.reset(); ni.reset();
        if (p1 == sv::npos) return;
        size_t p2 = tok.find('/', p1 + 1);
        ti = parse_usize(tok.substr(p1 + 1, p2 == sv::npos ? sv::npos : p2 - p1 - 1));
        if (p2 == sv::npos) return;
        size_t p3 = tok.find('/', p2 + 1);
        ni = parse_usize(tok.substr(p2 + 1, p3 == sv::npos ? sv::npos : p3 - p2 - 1));
    }

    // parse_obj_file_lines, utils.rs:139-213, in two phases so that a 250 MB soup parses on every host core:
    //  (1) the text is cut at line ends into one chunk per worker; each chunk is tokenised on its own: "v"/"vt"/"vn" values, "f" index
    //      triples with the number of v/vt/vn lines the CHUNK had seen before them, and the rare directives ("mtllib", "usemtl") and
    //      parse failures as `specials` in line order;
    //  (2) the chunks are walked in file order (directives take effect exactly where they stand: utils.rs:168-187), the faces are resolved
    //      in parallel against the v/vt/vn counts of THEIR line (a face sees only what the reference had parsed by then: utils.rs:272-329),
    //      and the earliest failure in file order is the one reported, as the reference's first panic would be.
    struct FaceRec { uint64_t vi[3], ti[3], ni[3]; uint32_t nv, nvt, nvn; uint8_t has_t, has_n, bad_k; uint32_t bad_msg; };
    struct Special { enum Kind { kMtllib, kUsemtl, kError } kind; size_t face_pos; std::string arg; int status; };
    struct Chunk { std::vector<Vec3> v, vt, vn; std::vector<FaceRec> faces; std::vector<Special> specials; std::vector<std::string> msgs; };

    static void parse_chunk(sv text, Chunk& C) {
        size_t i = 0;
        while (i < text.size()) {
            size_t j = text.find('\n', i);
            if (j == sv::npos) j = text.size();
            sv line = text.substr(i, j - i);
            if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
            i = j + 1;
            Tokens t{line};
            auto type = t.next();
            if (!type) continue;
            try {
                if (*type == "v") C.v.push_back(get_vertex(t));                 // utils.rs:188-191
                else if (*type == "vt") C.vt.push_back(get_vertex(t));          // utils.rs:199-202
                else if (*type == "vn") C.vn.push_back(get_vertex(t));          // utils.rs:203-206
                else if (*type == "f") {                                        // utils.rs:192-198, 253-343
                    FaceRec f{}; f.bad_k = 3; f.nv = (uint32_t)C.v.size(); f.nvt = (uint32_t)C.vt.size(); f.nvn = (uint32_t)C.vn.size();
                    sv toks[3];
                    for (int k = 0; k < 3; k++) {
                        auto tok = t.next();
                        if (!tok) fail(RRT_ERR_PARSE, "No data for vertex " + std::to_string(k + 1));
                        toks[k] = *tok;
                    }
                    for (int k = 0; k < 3 && f.bad_k == 3; k++) {
                        try {
                            std::optional<uint64_t> ti, ni;
                            vertex_attributes(toks[k], f.vi[k], ti, ni);
                            if (ti) { f.ti[k] = *ti; f.has_t |= (uint8_t)(1u << k); }
                            if (ni) { f.ni[k] = *ni; f.has_n |= (uint8_t)(1u << k); }
                        } catch (const Error& e) {                               // reported only if no earlier vertex of this face fails its range check first
                            f.bad_k = (uint8_t)k; f.bad_msg = (uint32_t)C.msgs.size(); C.msgs.push_back(e.detail);
                        }
                    }
                    C.faces.push_back(f);
                } else if (*type == "mtllib" || *type == "usemtl") {
                    auto nm = t.next();
                    if (!nm) fail(RRT_ERR_PARSE, *type == "mtllib" ? "Invalid .mtl file name" : "Invalid material name");
                    C.specials.push_back(Special{*type == "mtllib" ? Special::kMtllib : Special::kUsemtl, C.faces.size(), std::string(*nm), 0});
                }
            } catch (const Error& e) {
                C.specials.push_back(Special{Special::kError, C.faces.size(), e.detail, e.status});
                return;                                                         // nothing after the reference's first panic matters
            }
        }
    }

    void parse_obj(sv text) {
        // (1)
        const size_t want = std::max<size_t>(1, std::min<size_t>(host_threads(), text.size() / (512u << 10)));   // (host pool: a range costs microseconds, a megabyte of text ~1 ms)
        std::vector<size_t> cut{0};
        for (size_t p = 1; p < want; p++) {
            size_t at = text.find('\n', std::max(cut.back(), text.size() * p / want));
            if (at == sv::npos || at + 1 >= text.size()) break;
            if (at + 1 > cut.back()) cut.push_back(at + 1);
        }
        cut.push_back(text.size());
        const size_t n_chunks = cut.size() - 1;
        std::vector<Chunk> chunks(n_chunks);
        const bool trace = std::getenv("RRT_LOADER_TRACE") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(now() - a).count(); };
        auto tA = now();
        parallel_ranges(n_chunks, 1, [&](size_t b, size_t e, size_t) { for (size_t c = b; c < e; c++) parse_chunk(text.substr(cut[c], cut[c + 1] - cut[c]), chunks[c]); });
        if (trace) fprintf(stderr, "[loader] %zu chunks tokenised in %.1f ms\n", n_chunks, since(tA));
        tA = now();
        // (2) walk in file order: directives, the material of every face, the first failing directive / parse failure
        std::vector<size_t> off_v(n_chunks + 1, 0), off_vt(n_chunks + 1, 0), off_vn(n_chunks + 1, 0), off_f(n_chunks + 1, 0);
        for (size_t c = 0; c < n_chunks; c++) {
            off_v[c + 1] = off_v[c] + chunks[c].v.size(); off_vt[c + 1] = off_vt[c] + chunks[c].vt.size(); off_vn[c + 1] = off_vn[c] + chunks[c].vn.size();
            off_f[c + 1] = off_f[c] + chunks[c].faces.size();
        }
        std::vector<uint32_t> face_mat(off_f[n_chunks], 0);
        std::optional<uint32_t> current_material;
        std::optional<Error> stop; size_t stop_face = off_f[n_chunks];         // faces [0, stop_face) precede the first failing directive
        auto assign = [&](size_t c, size_t from, size_t to) {                   // faces [from, to) of chunk c take the current material
            if (from >= to) return true;
            if (!current_material) { stop = Error{RRT_ERR_PARSE, "face before any usemtl"}; stop_face = off_f[c] + from; return false; }   // utils.rs:193 unwrap on None
            std::fill(face_mat.begin() + off_f[c] + from, face_mat.begin() + off_f[c] + to, *current_material);
            return true;
        };
        for (size_t c = 0; c < n_chunks && !stop; c++) {
            size_t done = 0;
            for (const Special& sp : chunks[c].specials) {
                if (!assign(c, done, sp.face_pos)) break;
                done = sp.face_pos;
                try {
                    if (sp.kind == Special::kError) fail(sp.status, sp.arg);
                    if (sp.kind == Special::kMtllib) parse_mtl(read_file(dir + sp.arg));                 // utils.rs:168-175
                    else {                                                                               // utils.rs:176-187
                        auto it = material_by_name.find(sp.arg);
                        if (it == material_by_name.end()) fail(RRT_ERR_PARSE, "Material not found, is it in your mtl file?");
                        current_material = it->second;
                    }
                } catch (const Error& e) { stop = e; stop_face = off_f[c] + sp.face_pos; break; }
            }
            if (!stop) assign(c, done, chunks[c].faces.size());
        }
        if (trace) fprintf(stderr, "[loader] walk %.1f ms\n", since(tA));
        tA = now();
        // all "v"/"vt"/"vn" values stay where the tokeniser put them: value number i (file order) is element i - off[c] of chunk c's array, c found by
        // binary search over the chunk offsets (<= 128 chunks).  (Round 2 concatenated them first: 216 MB of copies, 30 ms of the 1 M soup's 75 ms parse.)
        auto chunk_of = [&](const std::vector<size_t>& off, uint64_t i) { return (size_t)(std::upper_bound(off.begin(), off.end(), (size_t)i) - off.begin()) - 1; };
        // the faces before the first failing directive, each against the counts of its own line (get_triangle, utils.rs:253-343)
        m.triangles.resize_uninit(stop_face);
        std::vector<size_t> chunk_of_begin(n_chunks);
        parallel_ranges(n_chunks, 1, [&](size_t b, size_t e, size_t) {
            for (size_t c = b; c < e; c++) {
                const Chunk& C = chunks[c];
                for (size_t k = 0; k < C.faces.size() && off_f[c] + k < stop_face; k++) {
                    const FaceRec& f = C.faces[k];
                    const uint64_t nv = off_v[c] + f.nv, nvt = off_vt[c] + f.nvt, nvn = off_vn[c] + f.nvn;
                    Triangle& tri = m.triangles[off_f[c] + k];
                    tri.mat = face_mat[off_f[c] + k]; tri._pad = 0;
                    Vec3* P[3] = {&tri.v1, &tri.v2, &tri.v3}; Vec3* T[3] = {&tri.t1, &tri.t2, &tri.t3}; Vec3* N[3] = {&tri.n1, &tri.n2, &tri.n3};
                    for (int q = 0; q < 3; q++) {
                        if (f.bad_k == q) throw FaceError{off_f[c] + k, Error{RRT_ERR_PARSE, C.msgs[f.bad_msg]}};
                        if (f.vi[q] == 0 || f.vi[q] - 1 >= nv) throw FaceError{off_f[c] + k, Error{RRT_ERR_PARSE, "No vertex with this index"}};   // utils.rs:272-283
                        { const uint64_t i = f.vi[q] - 1; const size_t cc = chunk_of(off_v, i); *P[q] = chunks[cc].v[i - off_v[cc]]; }
                        const uint64_t ti = f.ti[q] - 1, ni = f.ni[q] - 1;                                   // release-mode wrap for index 0 -> out of range -> default
                        if (((f.has_t >> q) & 1u) && ti < nvt) { const size_t cc = chunk_of(off_vt, ti); *T[q] = chunks[cc].vt[ti - off_vt[cc]]; } else *T[q] = Vec3{};   // utils.rs:285-329
                        if (((f.has_n >> q) & 1u) && ni < nvn) { const size_t cc = chunk_of(off_vn, ni); *N[q] = chunks[cc].vn[ni - off_vn[cc]]; } else *N[q] = Vec3{};
                    }
                }
            }
        });
        if (trace) fprintf(stderr, "[loader] faces %.1f ms\n", since(tA));
        if (stop) throw *stop;
    }
    struct FaceError { size_t face; Error err; };
};

std::shared_ptr<TexturePrefetch> TexturePrefetch::start(const std::string& dir, sv head) {
    std::string mtl;
    for_each_line(head.substr(0, std::min<size_t>(head.size(), 1u << 16)), [&](sv line) {
        if (!mtl.empty()) return;
        Tokens t{line};
        auto type = t.next();
        if (type && *type == "mtllib") if (auto nm = t.next()) mtl = std::string(*nm);
    });
    if (mtl.empty() || host_threads() < 2) return nullptr;
    auto self = std::make_shared<TexturePrefetch>();
    TexturePrefetch* raw = self.get();                                    // (the task is owned by *self: it ends before self does)
    self->task.reset(new AsyncTask([raw, dir, mtl] {
        try {
            const std::string text = read_file(dir + mtl);
            std::vector<std::string> names;
            for_each_line(text, [&](sv line) {
                Tokens t{line};
                auto type = t.next();
                if (!type || !(*type == "map_Ka" || *type == "bump")) return;
                if (auto nm = t.next()) if (std::find(names.begin(), names.end(), std::string(*nm)) == names.end()) names.emplace_back(*nm);
            });
            std::vector<Image> got(names.size());
            parallel_ranges(names.size(), 1, [&](size_t b, size_t e, size_t) {
                for (size_t i = b; i < e; i++) {
                    try { decode_image_file(dir + names[i], got[i].bytes, got[i].w, got[i].h, got[i].ch); }
                    catch (...) { got[i] = Image{}; }                     // not kept: the directive decodes it again and reports what is wrong with it
                }
            });
            std::lock_guard<std::mutex> g(raw->mu);
            for (size_t i = 0; i < names.size(); i++) if (!got[i].bytes.empty()) raw->images.emplace(dir + names[i], std::move(got[i]));
        } catch (...) {}
    }));
    return self;
}

}  // namespace

void load_obj(const std::string& obj_path, const Box& root, Model& out) {
    out.root = root;                       // (out is a freshly constructed Model: rrt_model_load_obj)
    size_t slash = obj_path.find_last_of('/');
    Loader L{slash == std::string::npos ? std::string() : obj_path.substr(0, slash + 1), out, {}, {}, nullptr};
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    // fs::read_to_string (main.rs:28) as a read-only mapping: the workers fault the pages in while they parse
    struct Mapping {
        const char* p = nullptr; size_t n = 0; int fd = -1;
        ~Mapping() { if (p && n) munmap(const_cast<char*>(p), n); if (fd >= 0) close(fd); }
    } map;
    map.fd = open(obj_path.c_str(), O_RDONLY);
    struct stat st{};
    if (map.fd < 0 || fstat(map.fd, &st) != 0 || !S_ISREG(st.st_mode)) fail(RRT_ERR_IO, "Could not read file: " + obj_path);
    map.n = (size_t)st.st_size;
    if (map.n) {
        void* q = mmap(nullptr, map.n, PROT_READ, MAP_PRIVATE, map.fd, 0);
        if (q == MAP_FAILED) { map.n = 0; fail(RRT_ERR_IO, "Could not read file: " + obj_path); }
        map.p = static_cast<const char*>(q);
    }
    const auto t1 = clk::now();
    L.prefetch = TexturePrefetch::start(L.dir, sv(map.p ? map.p : "", map.n));
    try { L.parse_obj(sv(map.p ? map.p : "", map.n)); }
    catch (const Loader::FaceError& fe) { throw fe.err; }
    const auto t2 = clk::now();
    // The reference pushes each triangle into the octree as it is parsed (utils.rs:196).  Inserting them afterwards in the same order builds the same
    // tree, and here that happens where the tree is needed: on the GPU inside rrt_raytracer_create (scene_build.hip), or on the host when a getter
    // or a RRT_FLAG_HOST_SETUP raytracer asks for it (host_tree, octree.cpp).
    out.read_ms = ms(t0, t1); out.parse_ms = ms(t1, t2) - out.texture_ms;
}

}  // namespace rrt
