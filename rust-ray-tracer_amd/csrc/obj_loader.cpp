// obj_loader.cpp -- the .obj/.mtl reader feeding the hot path (src/file_management/utils.rs:22-379 of the reference).
// Same directive set, same defaults and the same failure points; a reference `expect`/`unwrap`/`assert!` panic
// becomes rrt::Error{RRT_ERR_PARSE|RRT_ERR_IO} carried back over the C ABI as a status code.
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <optional>
#include <sstream>
#include <string_view>

#include "model.hpp"

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
template <class F> void for_each_line(const std::string& text, F&& fn) {
    size_t i = 0;
    while (i < text.size()) {
        size_t j = text.find('\n', i);
        if (j == std::string::npos) j = text.size();
        sv line(text.data() + i, j - i);
        if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
        fn(line);
        i = j + 1;
    }
}

struct Loader {
    std::string dir;
    Model& m;
    std::unordered_map<std::string, uint32_t> material_by_name;   // MaterialMap.materials, material.rs:25-28
    std::vector<Vec3> v, vt, vn;                                  // SceneData.vertices / vertex_texture_coords / vertex_normal_coords

    uint32_t load_texture(const std::string& name) {   // get_texture_from_file_name, utils.rs:345-368
        std::vector<uint8_t> bytes; uint32_t w = 0, h = 0, ch = 0;
        const auto t0 = std::chrono::steady_clock::now();
        decode_image_file(dir + name, bytes, w, h, ch);
        m.texture_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        // utils.rs:353 walks `as_bytes().chunks(3)` whatever the colour type; only a 3-byte-per-pixel buffer gives
        // width*height colours, anything else indexes out of bounds later (raytracer.rs:55).  Refuse it here.
        if (ch != 3) fail(RRT_ERR_UNSUPPORTED, "texture '" + name + "' is not 3 bytes per pixel");
        Texture t; t.rgb = std::move(bytes); t.width = w; t.height = h;
        m.textures.push_back(std::move(t));
        return (uint32_t)m.textures.size() - 1;
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
        for_each_line(text, on_line);
        on_line("END");                                                 // utils.rs:44-45
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

    static Vec3 lookup_or_default(const std::vector<Vec3>& arr, const std::optional<uint64_t>& idx) {   // utils.rs:285-329
        if (!idx) return Vec3{};
        uint64_t i = *idx - 1;                       // release-mode wrap for index 0 -> out of range -> default
        return i < arr.size() ? arr[i] : Vec3{};
    }

    Triangle get_triangle(Tokens& t, uint32_t mat) {   // utils.rs:253-343
        Triangle tri; tri.mat = mat;
        Vec3* P[3] = {&tri.v1, &tri.v2, &tri.v3};
        Vec3* T[3] = {&tri.t1, &tri.t2, &tri.t3};
        Vec3* N[3] = {&tri.n1, &tri.n2, &tri.n3};
        sv toks[3];
        for (int k = 0; k < 3; k++) {
            auto tok = t.next();
            if (!tok) fail(RRT_ERR_PARSE, "No data for vertex " + std::to_string(k + 1));
            toks[k] = *tok;
        }
        for (int k = 0; k < 3; k++) {
            uint64_t vi; std::optional<uint64_t> ti, ni;
            vertex_attributes(toks[k], vi, ti, ni);
            if (vi == 0 || vi - 1 >= v.size()) fail(RRT_ERR_PARSE, "No vertex with this index");   // utils.rs:272-283
            *P[k] = v[vi - 1];
            *T[k] = lookup_or_default(vt, ti);
            *N[k] = lookup_or_default(vn, ni);
        }
        return tri;
    }

    void parse_obj(const std::string& text) {   // parse_obj_file_lines, utils.rs:139-213
        std::optional<uint32_t> current_material;
        for_each_line(text, [&](sv line) {
            Tokens t{line};
            auto type = t.next();
            if (!type) return;
            if (*type == "mtllib") {                                    // utils.rs:168-175
                auto nm = t.next();
                if (!nm) fail(RRT_ERR_PARSE, "Invalid .mtl file name");
                parse_mtl(read_file(dir + std::string(*nm)));
            } else if (*type == "usemtl") {                             // utils.rs:176-187
                auto nm = t.next();
                if (!nm) fail(RRT_ERR_PARSE, "Invalid material name");
                auto it = material_by_name.find(std::string(*nm));
                if (it == material_by_name.end()) fail(RRT_ERR_PARSE, "Material not found, is it in your mtl file?");
                current_material = it->second;
            } else if (*type == "v") { v.push_back(get_vertex(t));      // utils.rs:188-191
            } else if (*type == "vt") { vt.push_back(get_vertex(t));    // utils.rs:199-202
            } else if (*type == "vn") { vn.push_back(get_vertex(t));    // utils.rs:203-206
            } else if (*type == "f") {                                  // utils.rs:192-198
                if (!current_material) fail(RRT_ERR_PARSE, "face before any usemtl");
                m.triangles.push_back(get_triangle(t, *current_material));
            }
        });
    }
};

}  // namespace

void load_obj(const std::string& obj_path, const Box& root, Model& out) {
    out = Model{};
    out.root = root;
    size_t slash = obj_path.find_last_of('/');
    Loader L{slash == std::string::npos ? std::string() : obj_path.substr(0, slash + 1), out, {}, {}, {}, {}};
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    const std::string text = read_file(obj_path);
    const auto t1 = clk::now();
    L.parse_obj(text);
    const auto t2 = clk::now();
    // the reference pushes each triangle into the octree as it is parsed (utils.rs:196); inserting them afterwards
    // in the same order builds the same tree
    build_octree(out.triangles, out.root, out.tree);
    const auto t3 = clk::now();
    out.read_ms = ms(t0, t1); out.parse_ms = ms(t1, t2) - out.texture_ms; out.octree_ms = ms(t2, t3);
}

}  // namespace rrt
