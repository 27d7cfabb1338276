// api.cpp -- the extern "C" surface declared in include/rrt.h: scene upload (once, to HBM) and kernel launches.
// No CPU rendering path exists in this library; every compute entry point launches the HIP kernels of render.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "device_scene.hpp"
#include "model.hpp"
#include "parallel.hpp"
#include "scene_build.hpp"

struct rrt_model { rrt::Model m; };

struct rrt_raytracer {
    int device = 0;
    rrt::DevScene scene{};
    rrt_options opt{};
    std::vector<void*> allocs;       // every hipMalloc of this raytracer
    void* arena = nullptr; size_t arena_bytes = 0, arena_used = 0;   // the scene's buffers (one allocation)
    uint64_t scene_bytes = 0;
    double filter_pad = 0;           // the pad the set-up gave the index's boxes (clusters.cpp / scene_build.hip: kPadFraction of the scene magnitude)
    bool all_inside_root = false;    // no triangle of the tree pokes out of the root box (then a child's subtree box lies inside its octant box: render.hip's certain-hit test)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    rrt_stats stats{};
    bool stats_pending = false;
    bool launched = false;           // some launch has been recorded in `stats`
    int walk = 0;                    // traversal variant used by this raytracer's frame launches: 0 lane filter, 1 bundle filter, 2 ray walk (see rrt.h)
    int walk_rays = -1;              // ... and by its per-ray entry points (rrt_get_ray_colours / rrt_intersect_rays): -1 = not measured yet (tune_rays_variant)
    bool variant_forced = false;
    void* host_fb = nullptr;         // device framebuffer kept between rrt_render calls (host-buffer entry point)
    size_t host_fb_bytes = 0;
    uint32_t n_suspects = 0;         // triangles whose plane contains the origin (exactness guard, clusters.cpp)
    double index_ms = 0, upload_ms = 0, hip_init_ms = 0;  // set-up stages of rrt_raytracer_create
    double octree_ms = 0, create_ms = 0;                  // GPU set-up: octree build on the device; wall time of the whole rrt_raytracer_create
    bool gpu_setup = false;                               // scene built on the device (default) or on the host (RRT_FLAG_HOST_SETUP)
    rrt::GpuScene gs{};                                   // GPU set-up: the device-side octree and the sizes of the scene buffers
    rrt_model_info tree_info{};                           // GPU set-up: what rrt_model_get_info reports, from the device-built tree
    struct Buf { const void* p = nullptr; size_t bytes = 0; } bufs[16];   // rrt_raytracer_get_buffer
    hipStream_t own_stream = nullptr;   // rrt_render's stream: the device's shared set-up stream (scene_build.hip: setup_stream; not owned)
    uint32_t tuned_w = 0, tuned_h = 0, tuned_world = 0;   // frame size the variant below belongs to
    uint32_t size_frames = 0;        // frames rendered at that size so far
    bool size_measured = false;      // ... and whether the variants have been timed on it (second frame of a size)
};

namespace rrt {
namespace {
thread_local std::string g_detail;
}
void set_error_detail(const std::string& s) { g_detail = s; }
}  // namespace rrt

namespace {

using namespace rrt;

struct HipFail { hipError_t e; const char* what; };
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw HipFail{_e, #expr}; } while (0)

template <class F> int guarded(F&& f) {
    try { return f(); }
    catch (const Error& e) { set_error_detail(e.detail); return e.status; }
    catch (const HipFail& h) {
        set_error_detail(std::string(h.what) + ": " + hipGetErrorString(h.e));
        (void)hipGetLastError();
        return (h.e == hipErrorOutOfMemory) ? RRT_ERR_OOM : (h.e == hipErrorNoDevice || h.e == hipErrorInvalidDevice) ? RRT_ERR_NO_DEVICE : RRT_ERR_HIP;
    }
    catch (const std::bad_alloc&) { set_error_detail("host allocation failed"); return RRT_ERR_OOM; }
    catch (const std::exception& e) { set_error_detail(e.what()); return RRT_ERR_INVALID_ARG; }
    catch (...) { set_error_detail("unknown failure"); return RRT_ERR_INVALID_ARG; }
}

// The HIP runtime, the process's first queue and this library's code object come up lazily, at the first HIP call that needs them (160-240 ms on a fresh
// process: runtime start 60-95, first queue 80-140, page-locked ring 13; RRT_SETUP_TRACE prints them).  A model is
// always loaded before a raytracer is created, so the loaders start that work on a helper thread and rrt_raytracer_create finds it done.
struct DeviceWarmer {
    std::thread th; std::once_flag once; std::mutex mu; bool joined = false;
    // HIP's current device is per thread, and the device a raytracer will use is only known at rrt_raytracer_create.  The helper warms the device
    // named by RRT_WARM_DEVICE, else LOCAL_RANK (one process per GPU under torchrun), else device 0 when it is the only one visible; with several
    // devices visible and no hint it brings up nothing device-specific (it would put a context and an allocation on GPU 0 for every rank).
    static int hinted_device(int n_dev) {
        for (const char* name : {"RRT_WARM_DEVICE", "LOCAL_RANK"})
            if (const char* e = std::getenv(name)) { char* end = nullptr; const long v = std::strtol(e, &end, 10); if (end != e && v >= 0 && v < n_dev) return (int)v; }
        return n_dev == 1 ? 0 : -1;
    }
    void start() {
        std::call_once(once, [this] {
            th = std::thread([] {
                const bool trace = std::getenv("RRT_SETUP_TRACE") != nullptr;
                auto lap = [trace, last = std::chrono::steady_clock::now()](const char* what) mutable {
                    if (!trace) return;
                    const auto now = std::chrono::steady_clock::now();
                    fprintf(stderr, "[warm-up]    %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count()); last = now;
                };
                int n = 0;
                if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { (void)hipGetLastError(); return; }
                lap("hipGetDeviceCount (runtime start)");
                const int dev = hinted_device(n);
                if (dev >= 0 && hipSetDevice(dev) == hipSuccess && hipFree(nullptr) == hipSuccess) {
                    lap("hipSetDevice + hipFree(0) (context)");
                    void* p = nullptr; char probe[256] = {};
                    if (hipMalloc(&p, 1 << 20) == hipSuccess) {           // the first allocation and the first host-to-device copy of a process set up the
                        lap("first hipMalloc");
                        (void)hipMemcpy(p, probe, sizeof probe, hipMemcpyHostToDevice);   // the process's first queue (80-140 ms, whoever causes it: without this copy the first stream pays it)
                        lap("first hipMemcpy (pageable)");
                        (void)hipFree(p);
                    }
                    lap("first hipFree");
                    preload_kernels();
                    lap("code object (preload_kernels)");
                    staged_upload_warm();                                 // the pinned staging ring of the set-up uploads (scene_build.hip)
                    lap("pinned ring + set-up streams");
                }
                (void)hipGetLastError();
            });
        });
    }
    void join() { std::lock_guard<std::mutex> lk(mu); if (!joined && th.joinable()) th.join(); joined = true; }
    ~DeviceWarmer() { join(); }
};
DeviceWarmer g_warmer;

Box default_root(const double* root) {
    Box b;
    if (root) { b.lo[0] = root[0]; b.hi[0] = root[1]; b.lo[1] = root[2]; b.hi[1] = root[3]; b.lo[2] = root[4]; b.hi[2] = root[5]; }
    else for (int k = 0; k < 3; k++) { b.lo[k] = -20.0; b.hi[k] = 20.0; }   // utils.rs:145
    return b;
}

void validate_model(const Model& m) {
    for (auto& t : m.textures) if (t.width == 0 || t.height == 0 || t.rgb.size() != (size_t)3 * t.width * t.height) throw Error{RRT_ERR_INVALID_ARG, "texture with bad dimensions"};
    for (auto& mat : m.materials) {
        if (mat.tex < 0 || (size_t)mat.tex >= m.textures.size()) throw Error{RRT_ERR_INVALID_ARG, "material texture index out of range"};
        if (mat.bump >= (int32_t)m.textures.size()) throw Error{RRT_ERR_INVALID_ARG, "material bump index out of range"};
        if (mat.bump >= 0) {
            // The bump texel is addressed with the COLOUR texture's (x, y) and the bump map's width (raytracer.rs:127-128).  Where that index can
            // leave the bump map the reference panics on the first such hit; on the GPU it would be a wild read, so the scene is refused up front.
            const Texture& t = m.textures[mat.tex]; const Texture& b = m.textures[mat.bump];
            if ((uint64_t)b.width * (t.height - 1) + (t.width - 1) >= (uint64_t)b.width * b.height)
                throw Error{RRT_ERR_INVALID_ARG, "bump map too small for the texture whose texel indices address it (raytracer.rs:127-128 would index out of bounds)"};
        }
    }
    for (auto& t : m.triangles) if (t.mat >= m.materials.size()) throw Error{RRT_ERR_INVALID_ARG, "triangle material index out of range"};
}

// Scene buffers are carved out of ONE device allocation (a hipMalloc per buffer costs milliseconds each: 15 of them were most of the teapot's
// upload time); a buffer that does not fit the arena's estimate gets an allocation of its own.
template <class T> T* upload(rrt_raytracer* rt, const T* host, size_t count) {
    void* d = nullptr;
    const size_t bytes = sizeof(T) * (count ? count : 1), padded = (bytes + 255) & ~(size_t)255;
    if (rt->arena && rt->arena_used + padded <= rt->arena_bytes) { d = static_cast<char*>(rt->arena) + rt->arena_used; rt->arena_used += padded; }
    else { HIP_TRY(hipMalloc(&d, bytes)); rt->allocs.push_back(d); }
    if (count) HIP_TRY(hipMemcpyAsync(d, host, sizeof(T) * count, hipMemcpyHostToDevice, nullptr));   // (host buffers outlive the hipDeviceSynchronize that ends the upload)
    rt->scene_bytes += sizeof(T) * count;
    return static_cast<T*>(d);
}

struct DeviceGuard {
    int prev = 0;
    explicit DeviceGuard(int dev) { HIP_TRY(hipGetDevice(&prev)); if (prev != dev) HIP_TRY(hipSetDevice(dev)); cur = dev; }
    ~DeviceGuard() { if (prev != cur) (void)hipSetDevice(prev); }
    int cur = 0;
};

FrameParams frame_params(const rrt_raytracer* rt, uint32_t width, uint32_t height, uint32_t rank, uint32_t world, bool tiled) {
    FrameParams f{};
    f.width = width; f.height = height;
    f.x_scale = rt->opt.vp_w / (double)width;      // engine.rs:189
    f.y_scale = rt->opt.vp_h / (double)height;     // engine.rs:190
    f.z_value = rt->opt.vp_d;                      // engine.rs:191
    f.tiles_x = (width + 7) / 8; f.tiles_y = (height + 7) / 8;
    f.rank = rank; f.world = world; f.tiled_output = tiled ? 1u : 0u;
    f.tile_begin = 0; f.tile_end = f.tiles_x * f.tiles_y; f.row_begin = 0; f.row_end = height;
    // XCD-aware block order (render.hip): worth 2-4 % where the scene is far larger than an XCD's L2 (100 k / 1 M-triangle soups), costs 4 % on the teapot
    // (profiles/r03_xcd_chunk_sweep.txt; chunks as 64 x 64-pixel squares instead of 512 x 8 strips: 1 % slower again): on for scenes of 50 000 triangle slots and more.
    static const int forced = [] { const char* e = std::getenv("RRT_XCD_CHUNK"); return e ? std::atoi(e) : -1; }();
    f.xcd_chunk = forced >= 0 ? (uint32_t)forced : (rt->scene.n_slots >= 50000u ? 256u : 0u);
    return f;
}

void check_frame(const rrt_raytracer* rt, uint32_t width, uint32_t height) {
    if (!rt) throw Error{RRT_ERR_INVALID_ARG, "null raytracer"};
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7FFFFFFFull) throw Error{RRT_ERR_INVALID_ARG, "bad frame size"};
}

// All traversal variants produce identical pixels; which is faster depends on how coherent the rays of a wave are (scene, camera, frame size).
// The reference renders ONE frame per run, so the first frame of a size costs nothing extra: it runs the variant a measured rule picks (node-coherent
// walk; bundle filter when the frame has more than ~1200 primary rays per triangle, lane filter below).  A caller that comes back for a SECOND
// frame of the same size is rendering repeatedly, and that frame is first rendered with every variant (each twice: the first run warms caches) on the
// caller's buffer and stream, timed with HIP events; the fastest is kept for that size.  This synchronises the stream once per size.
void tune_variant(rrt_raytracer* rt, const FrameParams& f, uint32_t* d_out, void* stream) {
    if (rt->variant_forced) return;
    if (!(rt->tuned_w == f.width && rt->tuned_h == f.height && rt->tuned_world == f.world)) {
        rt->tuned_w = f.width; rt->tuned_h = f.height; rt->tuned_world = f.world;
        rt->size_frames = 0; rt->size_measured = false;
        // First frame of a size (for a host that renders one frame per run, as the reference does, this IS the choice): the bundle filter pays once the
        // frame holds enough rays per triangle for a 4x4-pixel wave to stay inside few nodes and long lists -- measured over 3 models x 5 frame sizes
        // and the 100 k soup (profiles/r03_variant_sweep.json, re-measured with the final kernels: bundle wins at >= 1234 primary rays per triangle, by 4-12 %;
        // lane filter wins at <= 719, by 12-180 %; nothing measured in between).
        const double rays_per_triangle = 4.0 * (double)f.width * (double)f.height / (double)(rt->scene.n_slots ? rt->scene.n_slots : 1u);
        rt->walk = rays_per_triangle > 1200.0 ? 1 : 0;
    }
    if (rt->size_measured) return;
    if (++rt->size_frames < 2) return;
    rt->size_measured = true;
    constexpr int kVariants = 3;
    float ms[kVariants] = {0, 0, 0};
    if (f.world == 1) {
        for (int variant = 0; variant < kVariants; variant++)
            for (int rep = 0; rep < 2; rep++) {
                HIP_TRY(hipEventRecord(rt->ev0, (hipStream_t)stream));
                HIP_TRY((hipError_t)launch_render(rt->scene, f, d_out, stream, variant));
                HIP_TRY(hipEventRecord(rt->ev1, (hipStream_t)stream));
                HIP_TRY(hipEventSynchronize(rt->ev1));
                HIP_TRY(hipEventElapsedTime(&ms[variant], rt->ev0, rt->ev1));
            }
    } else {
        // One rank's share of a frame is a SHORT launch (a few waves per wave slot): alone it is bound by the latency of its last waves, not by
        // throughput, and a multi-GPU host keeps several frames in flight on separate streams precisely to hide that (bench.py).
        // So the variants are compared the way they will run: three launches at once on three streams, wall time per variant.
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        hipStream_t st[3] = {nullptr, nullptr, nullptr};
        struct Cleanup { hipStream_t* s; ~Cleanup() { for (int i = 0; i < 3; i++) if (s[i]) (void)hipStreamDestroy(s[i]); } } cl{st};
        for (auto& q : st) HIP_TRY(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
        for (int variant = 0; variant < kVariants; variant++)
            for (int rep = 0; rep < 2; rep++) {                           // rep 0 warms up
                HIP_TRY(hipEventRecord(rt->ev0, st[0]));
                for (int round = 0; round < 2; round++)
                    for (auto q : st) HIP_TRY((hipError_t)launch_render(rt->scene, f, d_out, q, variant));   // same pixels from every launch: the overlapping writes agree
                for (int i = 1; i < 3; i++) HIP_TRY(hipStreamSynchronize(st[i]));
                HIP_TRY(hipEventRecord(rt->ev1, st[0]));
                HIP_TRY(hipEventSynchronize(rt->ev1));
                HIP_TRY(hipEventElapsedTime(&ms[variant], rt->ev0, rt->ev1));
            }
    }
    rt->walk = 0;
    for (int variant = 1; variant < kVariants; variant++) if (ms[variant] < ms[rt->walk]) rt->walk = variant;
}

// The per-ray entry points take whatever rays the caller has: a coherent pixel grid or rays in all directions, and the three traversal variants are
// up to 5x apart on those (scattered rays: the ray walk; tools/random_rays_probe.py).  The first batch of at least kTuneMinRays rays is therefore
// used to measure them on its first kTuneSample rays (each twice, the first run warms caches; same outputs from every variant), and the fastest is
// kept for later calls.  A forced variant (RRT_FLAG_*_FILTER / RAY_WALK / NO_CULL) is used as is; smaller batches run the frame variant.
constexpr uint32_t kTuneMinRays = 16384, kTuneSample = 65536;
template <class Launch> int rays_variant(rrt_raytracer* rt, uint32_t n, Launch&& launch) {
    if (rt->variant_forced) return rt->walk;
    if (rt->walk_rays >= 0) return rt->walk_rays;
    if (n < kTuneMinRays) return rt->walk;
    const uint32_t m = n < kTuneSample ? n : kTuneSample;
    float best = 0; int best_v = 0;
    for (int variant = 0; variant < 3; variant++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            HIP_TRY(hipEventRecord(rt->ev0, nullptr));
            HIP_TRY((hipError_t)launch(m, variant));
            HIP_TRY(hipEventRecord(rt->ev1, nullptr));
            HIP_TRY(hipEventSynchronize(rt->ev1));
            HIP_TRY(hipEventElapsedTime(&ms, rt->ev0, rt->ev1));
        }
        if (variant == 0 || ms < best) { best = ms; best_v = variant; }
    }
    rt->walk_rays = best_v;
    return best_v;
}
// (kernel time of a per-ray launch into rrt_stats, like a frame's)
void record_rays(rrt_raytracer* rt, uint32_t n, int variant) {
    rt->stats.width = n; rt->stats.height = 1; rt->stats.rays_primary = n;
    rt->stats.scene_bytes = rt->scene_bytes; rt->stats.filter_variant = (uint32_t)variant; rt->stats.origin_plane_triangles = rt->n_suspects;
    rt->stats_pending = true; rt->launched = true;
}

void record_launch(rrt_raytracer* rt, uint32_t width, uint32_t height, uint32_t rank, uint32_t world) {
    rt->stats.width = width; rt->stats.height = height;
    const uint64_t wt = 2ull * (width / 2), ht = height >= 2 ? (uint64_t)(2 * (height / 2) - 1) : 0;   // traced pixels: see render_kernel
    rt->stats.rays_primary = world == 1 ? 4ull * wt * ht : 0;   // per-rank share is not tracked
    (void)rank;
    rt->stats.scene_bytes = rt->scene_bytes;
    rt->stats.filter_variant = (uint32_t)rt->walk; rt->stats.origin_plane_triangles = rt->n_suspects;
    rt->stats_pending = true; rt->launched = true;
}

}  // namespace

extern "C" {

const char* rrt_strerror(int status) {
    switch (status) {
        case RRT_OK: return "ok";
        case RRT_ERR_INVALID_ARG: return "invalid argument";
        case RRT_ERR_HIP: return "HIP runtime error";
        case RRT_ERR_OOM: return "out of memory";
        case RRT_ERR_IO: return "could not read file";
        case RRT_ERR_PARSE: return "parse error";
        case RRT_ERR_DEPTH: return "octree too deep";
        case RRT_ERR_NO_DEVICE: return "no usable HIP device";
        case RRT_ERR_UNSUPPORTED: return "unsupported input";
        default: return "unknown status";
    }
}
const char* rrt_last_error_detail(void) { return rrt::g_detail.c_str(); }
const char* rrt_build_info(void) { return "librrt_hip: offload-arch=gfx950, f64, -ffp-contract=off, wave64 node-coherent octree walk"; }
void rrt_free(void* p) { std::free(p); }

int rrt_device_count(int* count) {
    return guarded([&]() -> int {
        if (!count) throw Error{RRT_ERR_INVALID_ARG, "null count"};
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
        *count = n;
        return RRT_OK;
    });
}

int rrt_model_load_obj(const char* obj_path, const double* root, rrt_model** out) {
    return guarded([&]() -> int {
        if (!obj_path || !out) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        g_warmer.start();
        auto m = std::make_unique<rrt_model>();
        load_obj(obj_path, default_root(root), m->m);
        validate_model(m->m);
        *out = m.release();
        return RRT_OK;
    });
}

int rrt_model_from_arrays(uint32_t n_tris, const double* pos, const double* uv, const double* nrm, const uint32_t* mat,
                          uint32_t n_mats, const rrt_material* mats, uint32_t n_tex, const rrt_texture* tex,
                          const double* root, rrt_model** out) {
    return guarded([&]() -> int {
        if (!out || (n_tris && (!pos || !uv || !nrm || !mat)) || (n_mats && !mats) || (n_tex && !tex)) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        g_warmer.start();
        auto m = std::make_unique<rrt_model>();
        Model& M = m->m;
        M.root = default_root(root);
        M.materials.assign(mats, mats + n_mats);
        M.textures.resize(n_tex);
        for (uint32_t i = 0; i < n_tex; i++) if (!tex[i].rgb) throw Error{RRT_ERR_INVALID_ARG, "null texture data"};
        parallel_ranges(n_tex, 1, [&](size_t lo, size_t hi, size_t) {
            for (size_t i = lo; i < hi; i++) {
                M.textures[i].width = tex[i].width; M.textures[i].height = tex[i].height;
                M.textures[i].rgb.assign(tex[i].rgb, tex[i].rgb + (size_t)3 * tex[i].width * tex[i].height);
            }
        });
        M.triangles.resize_uninit(n_tris);
        parallel_ranges(n_tris, 1 << 14, [&](size_t lo, size_t hi, size_t) {
            auto rd = [](const double* p) { Vec3 v; v.x = p[0]; v.y = p[1]; v.z = p[2]; return v; };
            for (size_t i = lo; i < hi; i++) {
                Triangle& t = M.triangles[i];
                t.v1 = rd(pos + 9 * i); t.v2 = rd(pos + 9 * i + 3); t.v3 = rd(pos + 9 * i + 6);
                t.t1 = rd(uv + 9 * i);  t.t2 = rd(uv + 9 * i + 3);  t.t3 = rd(uv + 9 * i + 6);
                t.n1 = rd(nrm + 9 * i); t.n2 = rd(nrm + 9 * i + 3); t.n3 = rd(nrm + 9 * i + 6);
                t.mat = mat[i]; t._pad = 0;
            }
        });
        // (the octree is built where it is needed: on the GPU in rrt_raytracer_create, or by host_tree() for the getters below)
        validate_model(M);
        *out = m.release();
        return RRT_OK;
    });
}

void rrt_model_destroy(rrt_model* m) { delete m; }

int rrt_model_get_info(const rrt_model* m, rrt_model_info* out) {
    return guarded([&]() -> int {
        if (!m || !out) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        const FlatOctree& T = host_tree(m->m);
        std::memset(out, 0, sizeof *out);
        out->n_tris = (uint32_t)m->m.triangles.size();
        out->n_tris_in_tree = (uint32_t)T.own_idx.size();
        out->n_nodes = (uint32_t)T.box.size(); out->max_depth = T.max_depth;
        out->n_mats = (uint32_t)m->m.materials.size(); out->n_tex = (uint32_t)m->m.textures.size();
        out->root_own_count = T.own_off[1] - T.own_off[0];
        for (size_t i = 0; i + 1 < T.own_off.size(); i++) out->max_own_count = std::max(out->max_own_count, T.own_off[i + 1] - T.own_off[i]);
        return RRT_OK;
    });
}

int rrt_model_get_triangles(const rrt_model* m, double* pos, double* uv, double* nrm, uint32_t* mat) {
    return guarded([&]() -> int {
        if (!m) throw Error{RRT_ERR_INVALID_ARG, "null model"};
        auto wr = [](double* p, const Vec3& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; };
        for (size_t i = 0; i < m->m.triangles.size(); i++) {
            const Triangle& t = m->m.triangles[i];
            if (pos) { wr(pos + 9 * i, t.v1); wr(pos + 9 * i + 3, t.v2); wr(pos + 9 * i + 6, t.v3); }
            if (uv)  { wr(uv + 9 * i, t.t1);  wr(uv + 9 * i + 3, t.t2);  wr(uv + 9 * i + 6, t.t3); }
            if (nrm) { wr(nrm + 9 * i, t.n1); wr(nrm + 9 * i + 3, t.n2); wr(nrm + 9 * i + 6, t.n3); }
            if (mat) mat[i] = t.mat;
        }
        return RRT_OK;
    });
}

int rrt_model_get_materials(const rrt_model* m, rrt_material* out) {
    return guarded([&]() -> int {
        if (!m || !out) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        std::copy(m->m.materials.begin(), m->m.materials.end(), out);
        return RRT_OK;
    });
}

int rrt_model_get_texture(const rrt_model* m, uint32_t index, rrt_texture* out) {
    return guarded([&]() -> int {
        if (!m || !out || index >= m->m.textures.size()) throw Error{RRT_ERR_INVALID_ARG, "bad texture index"};
        const Texture& t = m->m.textures[index];
        out->rgb = t.rgb.data(); out->width = t.width; out->height = t.height;
        return RRT_OK;
    });
}

int rrt_model_get_octree(const rrt_model* m, double* aabb, uint32_t* first_child, uint32_t* tri_count, uint32_t* own_off, uint32_t* own_idx) {
    return guarded([&]() -> int {
        if (!m) throw Error{RRT_ERR_INVALID_ARG, "null model"};
        const FlatOctree& T = host_tree(m->m);
        const size_t n = T.box.size();
        if (aabb) for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) { aabb[6 * i + k] = T.box[i].lo[k]; aabb[6 * i + 3 + k] = T.box[i].hi[k]; }
        if (first_child) std::copy(T.first_child.begin(), T.first_child.end(), first_child);
        if (tri_count) std::copy(T.tri_count.begin(), T.tri_count.end(), tri_count);
        if (own_off) std::copy(T.own_off.begin(), T.own_off.end(), own_off);
        if (own_idx) std::copy(T.own_idx.begin(), T.own_idx.end(), own_idx);
        return RRT_OK;
    });
}

int rrt_decode_image_file(const char* path, uint8_t** rgb, uint32_t* width, uint32_t* height) {
    return guarded([&]() -> int {
        if (!path || !rgb || !width || !height) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        std::vector<uint8_t> bytes; uint32_t w = 0, h = 0, ch = 0;
        decode_image_file(path, bytes, w, h, ch);
        if (ch != 3) throw Error{RRT_ERR_UNSUPPORTED, "image is not 3 bytes per pixel"};
        uint8_t* p = static_cast<uint8_t*>(std::malloc(bytes.size() ? bytes.size() : 1));
        if (!p) throw std::bad_alloc();
        std::memcpy(p, bytes.data(), bytes.size());
        *rgb = p; *width = w; *height = h;
        return RRT_OK;
    });
}

// ------------------------------------------------------------------------------------------------ raytracer
namespace {

// ids of rrt_raytracer_get_buffer (rrt.h: RRT_BUF_*)
enum { kBufNodes = 0, kBufGeom, kBufAttr, kBufSupers, kBufCboxes, kBufChildBoxes, kBufTboxes, kBufSuspects, kBufOctBox, kBufOctFirstChild, kBufOctTriCount, kBufOctOwnOff, kBufOctOwnIdx, kBufSlotTri, kBufSlotPos, kBufCount };

// what a set-up needs of a scene besides its triangles: materials and RGB8 textures (borrowed views)
struct SceneTables { const rrt_material* mats; uint32_t n_mats; std::vector<rrt_texture> tex; };
SceneTables tables_of(const Model& M) {
    SceneTables T{M.materials.data(), (uint32_t)M.materials.size(), {}};
    for (auto& t : M.textures) T.tex.push_back(rrt_texture{t.rgb.data(), t.width, t.height});
    return T;
}
void validate_tables(const SceneTables& T) {
    for (auto& t : T.tex) if (!t.rgb || t.width == 0 || t.height == 0) throw Error{RRT_ERR_INVALID_ARG, "texture with bad dimensions"};
    for (uint32_t i = 0; i < T.n_mats; i++) {
        const rrt_material& mat = T.mats[i];
        if (mat.tex < 0 || (size_t)mat.tex >= T.tex.size()) throw Error{RRT_ERR_INVALID_ARG, "material texture index out of range"};
        if (mat.bump >= (int32_t)T.tex.size()) throw Error{RRT_ERR_INVALID_ARG, "material bump index out of range"};
        if (mat.bump >= 0) {   // see validate_model: the bump texel is addressed with the colour texture's (x, y) and the bump map's width (raytracer.rs:127-128)
            const rrt_texture& t = T.tex[mat.tex]; const rrt_texture& b = T.tex[mat.bump];
            if ((uint64_t)b.width * (t.height - 1) + (t.width - 1) >= (uint64_t)b.width * b.height)
                throw Error{RRT_ERR_INVALID_ARG, "bump map too small for the texture whose texel indices address it (raytracer.rs:127-128 would index out of bounds)"};
        }
    }
}

void upload_materials_and_textures(rrt_raytracer* rt, const SceneTables& T, hipStream_t st, std::vector<DevTexture>& texs, std::vector<DevMaterial>& mats) {
    texs.resize(T.tex.size());
    for (size_t i = 0; i < texs.size(); i++) {
        void* d = nullptr;
        const size_t bytes = (size_t)3 * T.tex[i].width * T.tex[i].height;
        const size_t padded = (bytes + 255) & ~(size_t)255;
        if (rt->arena && rt->arena_used + padded <= rt->arena_bytes) { d = static_cast<char*>(rt->arena) + rt->arena_used; rt->arena_used += padded; }
        else { HIP_TRY(hipMalloc(&d, bytes ? bytes : 1)); rt->allocs.push_back(d); }
        try { staged_upload(d, T.tex[i].rgb, bytes, st); } catch (const HipBuildFail& f) { throw HipFail{(hipError_t)f.hip_error, f.what}; }
        rt->scene_bytes += bytes;
        texs[i].rgb = static_cast<const uint8_t*>(d); texs[i].width = T.tex[i].width; texs[i].height = T.tex[i].height;
    }
    mats.resize(T.n_mats);
    for (size_t i = 0; i < mats.size(); i++) {
        const rrt_material& s = T.mats[i]; DevMaterial& d = mats[i];
        d.ka[0] = s.ka.x; d.ka[1] = s.ka.y; d.ka[2] = s.ka.z; d.kd[0] = s.kd.x; d.kd[1] = s.kd.y; d.kd[2] = s.kd.z;
        d.ks[0] = s.ks.x; d.ks[1] = s.ks.y; d.ks[2] = s.ks.z; d.ns = s.ns; d.kr = s.kr; d.tex = s.tex; d.bump = s.bump;
        d.tex_desc = texs[s.tex]; d.bump_desc = s.bump >= 0 ? texs[s.bump] : DevTexture{nullptr, 0, 0};
    }
}

// ---- set-up on the HOST (round-2 path, RRT_FLAG_HOST_SETUP): octree.cpp + clusters.cpp + the fill loops below, then one upload.  Kept as the
// second implementation the GPU set-up is checked against byte for byte (tests/test_gpu_build.py), and for A/B timing.
void setup_on_host(rrt_raytracer* rt, const Model& M, rrt_vec3 origin, const rrt_options& o, uint32_t& max_depth) {
    using clk = std::chrono::steady_clock;
    const FlatOctree& T = host_tree(M);
    rt->octree_ms = M.octree_ms;
    max_depth = T.max_depth;
    const size_t n_nodes = T.box.size(), n_slots = T.own_idx.size();   // n_slots: every triangle in the tree appears in exactly one own list
    const auto t_index0 = clk::now();
    ClusterSet CS;
    build_clusters(M, !(o.flags & RRT_FLAG_NO_CULL), CS);
    const size_t n_slots_c = CS.slot_tri.size();
    // (plain arrays: a std::vector would zero 250 MB on one thread before the workers fill it)
    std::unique_ptr<DevNode[]> nodes(new DevNode[n_nodes ? n_nodes : 1]);
    parallel_ranges(n_nodes, 1 << 14, [&](size_t nb, size_t ne, size_t) {
    for (size_t i = nb; i < ne; i++) {
        DevNode& d = nodes[i];
        for (int k = 0; k < 3; k++) {
            d.lo[k] = T.box[i].lo[k]; d.hi[k] = T.box[i].hi[k];
            // the split plane: child TFR (index 6, octree.rs:216-225) has lo == mid on every axis; for a leaf recompute it as subdivide would
            d.mid[k] = T.first_child[i] ? T.box[T.first_child[i] + 6].lo[k] : d.lo[k] + (d.hi[k] - d.lo[k]) / 2.0;
        }
        d.first_child = T.first_child[i]; d.sup_begin = CS.node_sup_begin[i]; d.sup_count = CS.node_sup_count[i];
        d.s0_begin = d.sup_count ? CS.supers[d.sup_begin].tri_begin : 0;
        d.flags = (T.tri_count[i] ? 0x100u : 0u) | ((d.sup_count ? CS.supers[d.sup_begin].tri_count : 0u) << 24);
        d.leaf_base = CS.node_leaf_slot[i] != kPadSlot ? CS.node_leaf_slot[i] : 0;   // a leaf has no children: the field holds its own dense slot instead
        if (d.first_child) for (uint32_t k = 8; k-- > 0;) {
            if (T.tri_count[d.first_child + k]) d.flags |= 1u << k;
            if (CS.node_leaf_slot[d.first_child + k] != kPadSlot) { d.flags |= 1u << (9 + k); d.leaf_base = CS.node_leaf_slot[d.first_child + k]; }   // ends at the first one
        }
    }
    });
    std::unique_ptr<DevTriGeom[]> geom(new DevTriGeom[n_slots_c ? n_slots_c : 1]); std::unique_ptr<DevTriAttr[]> attr(new DevTriAttr[n_slots_c ? n_slots_c : 1]);
    parallel_ranges(n_slots_c, 1 << 14, [&](size_t sb, size_t se, size_t) {
    for (size_t s = sb; s < se; s++) {
        if (CS.slot_tri[s] == kPadSlot) { std::memset(&geom[s], 0, sizeof(DevTriGeom)); std::memset(&attr[s], 0, sizeof(DevTriAttr)); attr[s].orig = kPadSlot; continue; }
        const Triangle& t = M.triangles[CS.slot_tri[s]];
        DevTriGeom& g = geom[s];
        g.v1[0] = t.v1.x; g.v1[1] = t.v1.y; g.v1[2] = t.v1.z;
        g.e1[0] = t.v2.x - t.v1.x; g.e1[1] = t.v2.y - t.v1.y; g.e1[2] = t.v2.z - t.v1.z;   // ray.rs:60
        g.e2[0] = t.v3.x - t.v1.x; g.e2[1] = t.v3.y - t.v1.y; g.e2[2] = t.v3.z - t.v1.z;   // ray.rs:61
        g.pos = CS.slot_pos[s]; g._pad = 0;
        DevTriAttr& a = attr[s];
        a.uv[0] = t.t1.x; a.uv[1] = t.t1.y; a.uv[2] = t.t2.x; a.uv[3] = t.t2.y; a.uv[4] = t.t3.x; a.uv[5] = t.t3.y;
        a.nrm[0] = t.n1.x; a.nrm[1] = t.n1.y; a.nrm[2] = t.n1.z; a.nrm[3] = t.n2.x; a.nrm[4] = t.n2.y; a.nrm[5] = t.n2.z;
        a.nrm[6] = t.n3.x; a.nrm[7] = t.n3.y; a.nrm[8] = t.n3.z;
        a.mat = t.mat; a.orig = CS.slot_tri[s];
    }
    });
    const auto t_index1 = clk::now();
    {
        size_t need = (size_t)1 << 20;
        for (auto& t : M.textures) need += t.rgb.size() + 256;
        need += n_nodes * sizeof(DevNode) + n_slots_c * (sizeof(DevTriGeom) + sizeof(DevTriAttr)) + 4096;
        need += (CS.supers.size() + CS.cboxes.size() + CS.child_boxes.size() + CS.tboxes.size()) * 32 + 4096;
        need += M.materials.size() * sizeof(DevMaterial) + M.textures.size() * sizeof(DevTexture) + (RRT_MAX_SUSPECTS + 1) * sizeof(DevSuspect);
        HIP_TRY(hipMalloc(&rt->arena, need));
        rt->allocs.push_back(rt->arena);
        rt->arena_bytes = need;
    }
    std::vector<DevTexture> texs; std::vector<DevMaterial> mats;
    upload_materials_and_textures(rt, tables_of(M), nullptr, texs, mats);
    DevScene& S = rt->scene;
    auto keep = [&](int id, const void* p, size_t bytes) { rt->bufs[id].p = p; rt->bufs[id].bytes = bytes; };
    S.nodes = upload(rt, nodes.get(), n_nodes);                              keep(kBufNodes, S.nodes, n_nodes * sizeof(DevNode));
    S.geom = upload(rt, geom.get(), n_slots_c);                              keep(kBufGeom, S.geom, n_slots_c * sizeof(DevTriGeom));
    S.supers = upload(rt, CS.supers.data(), CS.supers.size());               keep(kBufSupers, S.supers, CS.supers.size() * sizeof(DevSuper));
    S.cboxes = upload(rt, CS.cboxes.data(), CS.cboxes.size());               keep(kBufCboxes, S.cboxes, CS.cboxes.size() * sizeof(DevClusterBox));
    S.child_boxes = upload(rt, CS.child_boxes.data(), CS.child_boxes.size()); keep(kBufChildBoxes, S.child_boxes, CS.child_boxes.size() * sizeof(DevClusterBox));
    S.tboxes = upload(rt, CS.tboxes.data(), CS.tboxes.size());               keep(kBufTboxes, S.tboxes, CS.tboxes.size() * sizeof(DevClusterBox));
    S.has_groups = CS.has_groups ? 1u : 0u;
    S.bounds_plain = 1u;
    for (size_t i = 0; i < n_nodes; i++) { const DevNode& d = nodes[i];
        for (int k = 0; k < 3; k++)
            for (double v : {d.lo[k], d.mid[k], d.hi[k]})
                if (!(v == 0.0 || (std::fabs(v) > 0x1p-200 && std::fabs(v) < 0x1p200))) S.bounds_plain = 0u;
    }
    S.cull_limit = (float)(CS.scene_magnitude * 4.0);
    std::vector<DevSuspect> sus;                                      // (lives until the hipDeviceSynchronize below: uploads are asynchronous)
    {   // exactness guard of the index for rays from `origin` (clusters.cpp, find_origin_suspects)
        const double org[3] = {origin.x, origin.y, origin.z};
        if (!(o.flags & RRT_FLAG_NO_CULL)) find_origin_suspects(M, org, CS.pad, sus);
        rt->n_suspects = (uint32_t)sus.size();
        {   // does any triangle of the tree poke out of the root box?  (NaN coordinates count as poking out)
            std::atomic<int> out_of_root{0};
            const Triangle* tr = M.triangles.data();
            parallel_ranges(M.triangles.size(), 1 << 14, [&](size_t b, size_t e, size_t) {
                for (size_t i = b; i < e && !out_of_root.load(std::memory_order_relaxed); i++) {
                    const Vec3* v[3] = {&tr[i].v1, &tr[i].v2, &tr[i].v3};
                    double lo[3], hi[3];
                    for (int a = 0; a < 3; a++) {
                        const double c[3] = {a == 0 ? v[0]->x : a == 1 ? v[0]->y : v[0]->z, a == 0 ? v[1]->x : a == 1 ? v[1]->y : v[1]->z, a == 0 ? v[2]->x : a == 1 ? v[2]->y : v[2]->z};
                        lo[a] = std::fmin(c[0], std::fmin(c[1], c[2])); hi[a] = std::fmax(c[0], std::fmax(c[1], c[2]));
                    }
                    bool touch = true, inside = true;
                    for (int a = 0; a < 3; a++) { if (hi[a] < M.root.lo[a] || lo[a] > M.root.hi[a]) touch = false; if (!(lo[a] >= M.root.lo[a] && hi[a] <= M.root.hi[a])) inside = false; }
                    if (touch && !inside) out_of_root.store(1, std::memory_order_relaxed);
                }
            });
            rt->all_inside_root = out_of_root.load() == 0; rt->filter_pad = CS.pad;
        }
        if (sus.size() > RRT_MAX_SUSPECTS) sus.resize(1);              // beyond the cap every ray from the origin runs unfiltered; the list is not read
        S.suspects = upload(rt, sus.data(), sus.size());               keep(kBufSuspects, S.suspects, sus.size() * sizeof(DevSuspect));
    }
    S.attr = upload(rt, attr.get(), n_slots_c);                              keep(kBufAttr, S.attr, n_slots_c * sizeof(DevTriAttr));
    S.mats = upload(rt, mats.data(), mats.size());
    S.tex = upload(rt, texs.data(), texs.size());
    S.n_nodes = (uint32_t)n_nodes; S.n_slots = (uint32_t)n_slots; S.n_mats = (uint32_t)mats.size(); S.n_tex = (uint32_t)texs.size();
    S.fc_mask = CS.inline_leaves ? 0x00FFFFFFu : 0xFFFFFFFFu;
    HIP_TRY(hipDeviceSynchronize());
    rt->index_ms = std::chrono::duration<double, std::milli>(t_index1 - t_index0).count();
    rt->upload_ms = std::chrono::duration<double, std::milli>(clk::now() - t_index1).count();
}

// ---- set-up on the GPU (default): the triangle array goes up through pinned staging, then octree, index and records are built there
// (scene_build.hip).  Nothing of the tree ever exists on the host unless a getter asks for it.
void setup_on_gpu(rrt_raytracer* rt, const TriSource& src, uint32_t n_tris, const Box& root, const SceneTables& T, rrt_vec3 origin, const rrt_options& o, uint32_t& max_depth) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    const bool trace = std::getenv("RRT_SETUP_TRACE") != nullptr;
    auto lap = [&, last = t0](const char* what) mutable { if (trace) { const auto n = clk::now(); fprintf(stderr, "[create]     %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - last).count()); last = n; } };
    hipStream_t st = nullptr;
    try { st = (hipStream_t)setup_stream(); } catch (const HipBuildFail& f) { throw HipFail{(hipError_t)f.hip_error, f.what}; }
    {   // textures + small tables: one allocation
        size_t need = (size_t)1 << 16;
        for (auto& t : T.tex) need += (size_t)3 * t.width * t.height + 256;
        need += T.n_mats * sizeof(DevMaterial) + T.tex.size() * sizeof(DevTexture) + 1024;
        HIP_TRY(hipMalloc(&rt->arena, need));
        rt->allocs.push_back(rt->arena);
        rt->arena_bytes = need;
    }
    lap("stream, table arena");
    std::vector<DevTexture> texs; std::vector<DevMaterial> mats;
    GpuScene& G = rt->gs;
    const double org[3] = {origin.x, origin.y, origin.z};
    // The textures go up beside the build: their copies into page-locked memory are host work (19 MB: ~1.2 ms for the teapot's six), the build is GPU work
    // and waits -- started once the triangles have been through the ring, on a stream of their own, and joined before anything else of *rt is touched.
    hipStream_t st_tex = nullptr;
    try { st_tex = (hipStream_t)upload_stream(); } catch (const HipBuildFail& f) { throw HipFail{(hipError_t)f.hip_error, f.what}; }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::unique_ptr<AsyncTask> tex_task;                                  // (declared after what it refers to: it is waited for first when the frame unwinds)
    auto start_textures = [&] {
        tex_task.reset(new AsyncTask([&, dev] {
            HIP_TRY(hipSetDevice(dev));                                   // (HIP's current device is per thread, and a pool worker keeps its last one)
            upload_materials_and_textures(rt, T, st_tex, texs, mats);
        }));
    };
    try { gpu_build_scene(src, n_tris, root, !(o.flags & RRT_FLAG_NO_CULL), org, st, G, start_textures); }
    catch (const HipBuildFail& f) { tex_task.reset(); if (G.scene_alloc) { (void)hipFree(G.scene_alloc); G.scene_alloc = nullptr; } throw HipFail{(hipError_t)f.hip_error, f.what}; }
    catch (...) { tex_task.reset(); if (G.scene_alloc) { (void)hipFree(G.scene_alloc); G.scene_alloc = nullptr; } throw; }
    lap("gpu_build_scene");
    try { if (!tex_task) start_textures(); tex_task->wait(); }
    catch (...) { if (G.scene_alloc) { (void)hipFree(G.scene_alloc); G.scene_alloc = nullptr; } throw; }
    rt->allocs.push_back(G.scene_alloc);
    lap("texture upload joined");
    max_depth = G.max_depth;
    DevScene& S = rt->scene;
    S.nodes = G.nodes; S.geom = G.geom; S.attr = G.attr; S.supers = G.supers; S.cboxes = G.cboxes; S.child_boxes = G.child_boxes; S.tboxes = G.tboxes; S.suspects = G.suspects;
    auto keep = [&](int id, const void* p, size_t bytes) { rt->bufs[id].p = p; rt->bufs[id].bytes = bytes; };
    keep(kBufNodes, G.nodes, (size_t)G.n_nodes * sizeof(DevNode)); keep(kBufGeom, G.geom, (size_t)G.n_slots_total * sizeof(DevTriGeom)); keep(kBufAttr, G.attr, (size_t)G.n_slots_total * sizeof(DevTriAttr));
    keep(kBufSupers, G.supers, (size_t)G.n_sup_records * sizeof(DevSuper)); keep(kBufCboxes, G.cboxes, ((size_t)G.n_clusters + 8) * sizeof(DevClusterBox));
    keep(kBufChildBoxes, G.child_boxes, ((size_t)(G.n_nodes > 1 ? G.n_nodes - 1 : 0) + 8) * sizeof(DevClusterBox)); keep(kBufTboxes, G.tboxes, ((size_t)G.n_list_slots + 8) * sizeof(DevClusterBox));
    keep(kBufSuspects, G.suspects, (size_t)(G.n_suspects > RRT_MAX_SUSPECTS ? 0 : G.n_suspects) * sizeof(DevSuspect));
    keep(kBufOctBox, G.oct_box, (size_t)G.n_nodes * 48); keep(kBufOctFirstChild, G.oct_first_child, (size_t)G.n_nodes * 4); keep(kBufOctTriCount, G.oct_tri_count, (size_t)G.n_nodes * 4);
    keep(kBufOctOwnOff, G.oct_own_off, ((size_t)G.n_nodes + 1) * 4); keep(kBufOctOwnIdx, G.oct_own_idx, (size_t)G.n_in_tree * 4);
    keep(kBufSlotTri, G.slot_tri, (size_t)G.n_slots_total * 4); keep(kBufSlotPos, G.slot_pos, (size_t)G.n_slots_total * 4);
    rt->scene_bytes += (size_t)G.n_nodes * sizeof(DevNode) + (size_t)G.n_slots_total * (sizeof(DevTriGeom) + sizeof(DevTriAttr))
                     + ((size_t)G.n_sup_records + G.n_clusters + 8 + (G.n_nodes > 1 ? G.n_nodes - 1 : 0) + 8 + G.n_list_slots + 8) * 32;
    S.has_groups = G.has_groups; S.bounds_plain = G.bounds_plain;
    S.cull_limit = (float)(G.scene_magnitude * 4.0);
    rt->n_suspects = G.n_suspects;
    rt->all_inside_root = G.all_inside_root != 0; rt->filter_pad = G.pad;
    // small tables go through the same stream
    {
        void* d_m = static_cast<char*>(rt->arena) + rt->arena_used; rt->arena_used += (mats.size() * sizeof(DevMaterial) + 255) & ~(size_t)255;
        void* d_t = static_cast<char*>(rt->arena) + rt->arena_used; rt->arena_used += (texs.size() * sizeof(DevTexture) + 255) & ~(size_t)255;
        if (rt->arena_used > rt->arena_bytes) throw Error{RRT_ERR_OOM, "internal: table arena too small"};
        if (!mats.empty()) HIP_TRY(hipMemcpyAsync(d_m, mats.data(), mats.size() * sizeof(DevMaterial), hipMemcpyHostToDevice, st));
        if (!texs.empty()) HIP_TRY(hipMemcpyAsync(d_t, texs.data(), texs.size() * sizeof(DevTexture), hipMemcpyHostToDevice, st));
        S.mats = static_cast<const DevMaterial*>(d_m); S.tex = static_cast<const DevTexture*>(d_t);
    }
    S.n_nodes = G.n_nodes; S.n_slots = G.n_in_tree; S.n_mats = (uint32_t)mats.size(); S.n_tex = (uint32_t)texs.size();
    S.fc_mask = G.inline_leaves ? 0x00FFFFFFu : 0xFFFFFFFFu;
    HIP_TRY(hipStreamSynchronize(st_tex));
    HIP_TRY(hipStreamSynchronize(st));
    lap("final synchronise");
    rt->octree_ms = G.ms_octree; rt->index_ms = G.ms_index;
    rt->upload_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count() - G.ms_octree - G.ms_index;   // uploads, allocations, synchronisation
    rrt_model_info& I = rt->tree_info;
    I.n_tris = n_tris; I.n_tris_in_tree = G.n_in_tree; I.n_nodes = G.n_nodes; I.max_depth = G.max_depth;
    I.n_mats = T.n_mats; I.n_tex = (uint32_t)T.tex.size(); I.max_own_count = 0; I.root_own_count = 0;
}

}  // namespace

namespace {
// rrt_raytracer_create and rrt_raytracer_create_from_arrays: everything but where the triangles come from
int create_raytracer(const rrt_light* lights, uint32_t n_lights, rrt_vec3 origin, const rrt_options* opt, int device, rrt_raytracer** out,
                     const std::function<void(rrt_raytracer*, const rrt_options&, uint32_t&)>& setup) {
    if (!out || (n_lights && !lights)) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
    if (n_lights > RRT_MAX_LIGHTS) throw Error{RRT_ERR_INVALID_ARG, "too many lights (max 16)"};
    for (uint32_t i = 0; i < n_lights; i++) if (lights[i].kind > 2) throw Error{RRT_ERR_INVALID_ARG, "bad light kind"};
    rrt_options o;
    if (opt) o = *opt; else { o.surface_offset = 0.0001; o.max_reflection_depth = 5; o.flags = 0; o.vp_w = o.vp_h = o.vp_d = 1.0; }
    if (o.max_reflection_depth > RRT_MAX_REFLECT) throw Error{RRT_ERR_INVALID_ARG, "max_reflection_depth > 8"};
    const auto t_create0 = std::chrono::steady_clock::now();
    g_warmer.join();
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) { (void)hipGetLastError(); throw Error{RRT_ERR_NO_DEVICE, "no HIP device visible"}; }
    if (device < 0 || device >= n_dev) throw Error{RRT_ERR_NO_DEVICE, "device index out of range"};
    const auto t_init0 = std::chrono::steady_clock::now();
    DeviceGuard guard(device);
    HIP_TRY(hipFree(nullptr));                                          // brings the HIP context of this device up (a one-off of the process: ~90 ms on a fresh one)
    const double hip_init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_init0).count();

    std::unique_ptr<rrt_raytracer, void (*)(rrt_raytracer*)> rt(new rrt_raytracer, rrt_raytracer_destroy);
    rt->hip_init_ms = hip_init_ms;
    rt->device = device; rt->opt = o;
    uint32_t max_depth = 1;
    setup(rt.get(), o, max_depth);

    DevScene& S = rt->scene;
    S.cull_enabled = (o.flags & RRT_FLAG_NO_CULL) ? 0u : 1u;
    S.cull_half_over_limit = S.cull_limit > 0.0f ? 0.5f / S.cull_limit : 0.0f;
    S.inner_shrink = (S.cull_enabled && (rt->all_inside_root || std::getenv("RRT_FORCE_CERTAIN_HIT") /* developer: shows what the flag guards against */) && !std::getenv("RRT_NO_CERTAIN_HIT")) ? (float)(2.0 * rt->filter_pad) : 0.0f;   // 2 x the pad the boxes were built with; render.hip, single-candidate child test
    S.n_suspects = rt->n_suspects;
    S.n_lights = n_lights; S.max_reflection_depth = o.max_reflection_depth; S.stack_levels = max_depth > 1 ? max_depth - 1 : 1;   // (stack_levels: only internal nodes push a frame; the deepest level holds leaves)
    S.origin[0] = origin.x; S.origin[1] = origin.y; S.origin[2] = origin.z;
    S.surface_offset = o.surface_offset;
    for (uint32_t i = 0; i < n_lights; i++) {
        S.lights[i].kind = lights[i].kind; S.lights[i]._pad = 0; S.lights[i].intensity = lights[i].intensity;
        S.lights[i].v[0] = lights[i].v.x; S.lights[i].v[1] = lights[i].v.y; S.lights[i].v[2] = lights[i].v.z;
    }
#ifdef RRT_PROFILE
    { void* pb = nullptr; HIP_TRY(hipMalloc(&pb, 32 * sizeof(unsigned long long))); HIP_TRY(hipMemset(pb, 0, 32 * sizeof(unsigned long long)));
      rt->allocs.push_back(pb); S.prof = static_cast<unsigned long long*>(pb); }
#endif
    HIP_TRY(hipEventCreate(&rt->ev0)); HIP_TRY(hipEventCreate(&rt->ev1));
    // Own-list filter variant: forced by a flag, else a measured rule on the first frame of each frame size and measured on the second (tune_variant)
    rt->variant_forced = (o.flags & (RRT_FLAG_BUNDLE_FILTER | RRT_FLAG_LANE_FILTER | RRT_FLAG_RAY_WALK | RRT_FLAG_NO_CULL)) != 0;
    rt->walk = (o.flags & RRT_FLAG_NO_CULL) ? 0 : (o.flags & RRT_FLAG_BUNDLE_FILTER) ? 1 : (o.flags & RRT_FLAG_RAY_WALK) ? 2 : 0;
    rt->create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create0).count();
    *out = rt.release();
    return RRT_OK;
}
}  // namespace

int rrt_raytracer_create(const rrt_model* m, const rrt_light* lights, uint32_t n_lights, rrt_vec3 origin,
                         const rrt_options* opt, int device, rrt_raytracer** out) {
    return guarded([&]() -> int {
        if (!m) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        const Model& M = m->m;
        return create_raytracer(lights, n_lights, origin, opt, device, out, [&](rrt_raytracer* rt, const rrt_options& o, uint32_t& max_depth) {
            rt->gpu_setup = !(o.flags & RRT_FLAG_HOST_SETUP);
            if (rt->gpu_setup) { TriSource src; src.tris = M.triangles.data(); setup_on_gpu(rt, src, (uint32_t)M.triangles.size(), M.root, tables_of(M), origin, o, max_depth); }
            else setup_on_host(rt, M, origin, o, max_depth);
        });
    });
}

// RayTracer straight from the host's own arrays: rrt_model_from_arrays + rrt_raytracer_create without the model -- the arrays are uploaded from where
// they lie (through the pinned staging ring) and packed into triangle records on the device, so the library keeps no host copy of the scene and the
// loaders' copy (15 ms of the 1 M soup's 54 ms first frame) is not made.  Same scene in HBM, same frames.
int rrt_raytracer_create_from_arrays(uint32_t n_tris, const double* pos, const double* uv, const double* nrm, const uint32_t* mat,
                                     uint32_t n_mats, const rrt_material* mats, uint32_t n_tex, const rrt_texture* tex, const double* root,
                                     const rrt_light* lights, uint32_t n_lights, rrt_vec3 origin, const rrt_options* opt, int device, rrt_raytracer** out) {
    return guarded([&]() -> int {
        if ((n_tris && (!pos || !uv || !nrm || !mat)) || (n_mats && !mats) || (n_tex && !tex)) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        if (opt && (opt->flags & RRT_FLAG_HOST_SETUP)) throw Error{RRT_ERR_UNSUPPORTED, "RRT_FLAG_HOST_SETUP needs a model (rrt_model_from_arrays + rrt_raytracer_create)"};
        SceneTables T{mats, n_mats, std::vector<rrt_texture>(tex, tex + n_tex)};
        validate_tables(T);
        std::atomic<bool> bad{false};
        parallel_ranges(n_tris, 1 << 16, [&](size_t lo, size_t hi, size_t) { for (size_t i = lo; i < hi; i++) if (mat[i] >= n_mats) bad = true; });
        if (bad) throw Error{RRT_ERR_INVALID_ARG, "triangle material index out of range"};
        g_warmer.start();
        const Box box = default_root(root);
        return create_raytracer(lights, n_lights, origin, opt, device, out, [&](rrt_raytracer* rt, const rrt_options& o, uint32_t& max_depth) {
            rt->gpu_setup = true;
            TriSource src; src.pos = pos; src.uv = uv; src.nrm = nrm; src.mat = mat;
            setup_on_gpu(rt, src, n_tris, box, T, origin, o, max_depth);
        });
    });
}

// The octree as the GPU set-up built it (scene_build.hip), in the reference's node numbering: same layout as rrt_model_get_octree.
int rrt_raytracer_get_octree(const rrt_raytracer* rt, rrt_model_info* info, double* aabb, uint32_t* first_child, uint32_t* tri_count, uint32_t* own_off, uint32_t* own_idx) {
    return guarded([&]() -> int {
        if (!rt) throw Error{RRT_ERR_INVALID_ARG, "null raytracer"};
        if (!rt->gpu_setup) throw Error{RRT_ERR_UNSUPPORTED, "this raytracer was set up on the host (RRT_FLAG_HOST_SETUP): ask the model (rrt_model_get_octree)"};
        DeviceGuard guard(rt->device);
        const GpuScene& G = rt->gs;
        if (aabb) HIP_TRY(hipMemcpy(aabb, G.oct_box, (size_t)G.n_nodes * 48, hipMemcpyDeviceToHost));
        if (first_child) HIP_TRY(hipMemcpy(first_child, G.oct_first_child, (size_t)G.n_nodes * 4, hipMemcpyDeviceToHost));
        if (tri_count) HIP_TRY(hipMemcpy(tri_count, G.oct_tri_count, (size_t)G.n_nodes * 4, hipMemcpyDeviceToHost));
        if (own_off) HIP_TRY(hipMemcpy(own_off, G.oct_own_off, ((size_t)G.n_nodes + 1) * 4, hipMemcpyDeviceToHost));
        if (own_idx && G.n_in_tree) HIP_TRY(hipMemcpy(own_idx, G.oct_own_idx, (size_t)G.n_in_tree * 4, hipMemcpyDeviceToHost));
        if (info) {
            *info = rt->tree_info;
            std::vector<uint32_t> off((size_t)G.n_nodes + 1);
            HIP_TRY(hipMemcpy(off.data(), G.oct_own_off, off.size() * 4, hipMemcpyDeviceToHost));
            info->root_own_count = off[1] - off[0];
            for (size_t i = 0; i + 1 < off.size(); i++) info->max_own_count = std::max(info->max_own_count, off[i + 1] - off[i]);
        }
        return RRT_OK;
    });
}

// Developer / test introspection: the bytes of one of the scene buffers in HBM (RRT_BUF_*).  out may be NULL to ask for the size only.
int rrt_raytracer_get_buffer(const rrt_raytracer* rt, uint32_t which, void* out, size_t capacity, size_t* bytes) {
    return guarded([&]() -> int {
        if (!rt || which >= (uint32_t)kBufCount) throw Error{RRT_ERR_INVALID_ARG, "bad buffer id"};
        const auto& b = rt->bufs[which];
        if (!b.p && b.bytes) throw Error{RRT_ERR_UNSUPPORTED, "buffer not kept by this set-up path"};
        if (bytes) *bytes = b.bytes;
        if (out) {
            if (capacity < b.bytes) throw Error{RRT_ERR_INVALID_ARG, "buffer too small"};
            DeviceGuard guard(rt->device);
            if (b.bytes) HIP_TRY(hipMemcpy(out, b.p, b.bytes, hipMemcpyDeviceToHost));
        }
        return RRT_OK;
    });
}

void rrt_raytracer_destroy(rrt_raytracer* rt) {
    if (!rt) return;
    int prev = 0;
    if (hipGetDevice(&prev) == hipSuccess) {
        (void)hipSetDevice(rt->device);
        for (void* p : rt->allocs) (void)hipFree(p);
        if (rt->host_fb) (void)hipFree(rt->host_fb);
        if (rt->ev0) (void)hipEventDestroy(rt->ev0);
        if (rt->ev1) (void)hipEventDestroy(rt->ev1);
        (void)hipSetDevice(prev);
    }
    delete rt;
}

uint32_t rrt_tiles_per_rank(uint32_t width, uint32_t height, uint32_t world) {
    if (world == 0) return 0;
    const uint32_t n = ((width + 7) / 8) * ((height + 7) / 8);
    return (n + world - 1) / world;
}

int rrt_render_tiles_device(rrt_raytracer* rt, uint32_t width, uint32_t height, uint32_t rank, uint32_t world, void* d_tiles, void* stream) {
    return guarded([&]() -> int {
        check_frame(rt, width, height);
        if (!d_tiles || world == 0 || rank >= world) throw Error{RRT_ERR_INVALID_ARG, "bad rank/world/buffer"};
        DeviceGuard guard(rt->device);
        const FrameParams f = frame_params(rt, width, height, rank, world, true);
        tune_variant(rt, f, static_cast<uint32_t*>(d_tiles), stream);
        HIP_TRY(hipEventRecord(rt->ev0, (hipStream_t)stream));
        HIP_TRY((hipError_t)launch_render(rt->scene, f, static_cast<uint32_t*>(d_tiles), stream, rt->walk));
        HIP_TRY(hipEventRecord(rt->ev1, (hipStream_t)stream));
        record_launch(rt, width, height, rank, world);
        return RRT_OK;
    });
}

int rrt_render_device(rrt_raytracer* rt, uint32_t width, uint32_t height, void* d_fb, void* stream) {
    return guarded([&]() -> int {
        check_frame(rt, width, height);
        if (!d_fb) throw Error{RRT_ERR_INVALID_ARG, "null framebuffer"};
        DeviceGuard guard(rt->device);
        const FrameParams f = frame_params(rt, width, height, 0, 1, false);
        tune_variant(rt, f, static_cast<uint32_t*>(d_fb), stream);
        HIP_TRY(hipEventRecord(rt->ev0, (hipStream_t)stream));
        HIP_TRY((hipError_t)launch_render(rt->scene, f, static_cast<uint32_t*>(d_fb), stream, rt->walk));
        HIP_TRY(hipEventRecord(rt->ev1, (hipStream_t)stream));
        record_launch(rt, width, height, 0, 1);
        return RRT_OK;
    });
}

int rrt_detile_device(rrt_raytracer* rt, uint32_t width, uint32_t height, uint32_t world, const void* d_gathered, void* d_fb, void* stream) {
    return guarded([&]() -> int {
        check_frame(rt, width, height);
        if (!d_gathered || !d_fb || world == 0) throw Error{RRT_ERR_INVALID_ARG, "bad argument"};
        DeviceGuard guard(rt->device);
        HIP_TRY((hipError_t)launch_detile(width, height, world, static_cast<const uint32_t*>(d_gathered), static_cast<uint32_t*>(d_fb), stream));
        return RRT_OK;
    });
}

// Page-locks a caller-owned framebuffer (e.g. the Rust host's Canvas.buffer, engine.rs:127) so that rrt_render can DMA the frame straight
// into it.  Optional: rrt_render works on pageable memory too, through a pinned staging buffer and one extra host copy.
int rrt_host_buffer_register(void* ptr, size_t bytes) {
    return guarded([&]() -> int {
        if (!ptr || !bytes) throw Error{RRT_ERR_INVALID_ARG, "null buffer"};
        HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
        return RRT_OK;
    });
}
int rrt_host_buffer_unregister(void* ptr) {
    return guarded([&]() -> int {
        if (!ptr) throw Error{RRT_ERR_INVALID_ARG, "null buffer"};
        HIP_TRY(hipHostUnregister(ptr));
        return RRT_OK;
    });
}

int rrt_render(rrt_raytracer* rt, uint32_t width, uint32_t height, uint32_t* out_fb) {
    return guarded([&]() -> int {
        check_frame(rt, width, height);
        if (!out_fb) throw Error{RRT_ERR_INVALID_ARG, "null framebuffer"};
        DeviceGuard guard(rt->device);
        const size_t bytes = sizeof(uint32_t) * (size_t)width * height;
        if (rt->host_fb_bytes < bytes) {                                   // the device-side frame is kept and reused from call to call
            if (rt->host_fb) { (void)hipFree(rt->host_fb); rt->host_fb = nullptr; rt->host_fb_bytes = 0; }
            HIP_TRY(hipMalloc(&rt->host_fb, bytes));
            rt->host_fb_bytes = bytes;
        }
        if (!rt->own_stream) { try { rt->own_stream = (hipStream_t)setup_stream(); } catch (const HipBuildFail& f) { throw HipFail{(hipError_t)f.hip_error, f.what}; } }
        const int rc = rrt_render_device(rt, width, height, rt->host_fb, rt->own_stream);
        if (rc != RRT_OK) return rc;
        // Is the caller's framebuffer page-locked (rrt_host_buffer_register, hipHostMalloc, ...)?  Then one asynchronous DMA into it.
        hipPointerAttribute_t attr{};
        const bool pinned = hipPointerGetAttributes(&attr, out_fb) == hipSuccess && attr.type == hipMemoryTypeHost;
        if (!pinned) (void)hipGetLastError();
        if (pinned) {
            HIP_TRY(hipMemcpyAsync(out_fb, rt->host_fb, bytes, hipMemcpyDeviceToHost, rt->own_stream));
            HIP_TRY(hipStreamSynchronize(rt->own_stream));                 // blocking: the frame is in out_fb on return
            return RRT_OK;
        }
        // Pageable framebuffer: through the device's pinned staging ring, chunk DMAs running ahead of the copies out (scene_build.hip)
        try { staged_download(out_fb, rt->host_fb, bytes, rt->own_stream); } catch (const HipBuildFail& f) { throw HipFail{(hipError_t)f.hip_error, f.what}; }
        return RRT_OK;
    });
}

// Scene::draw_scene as the reference paces it (engine.rs:196-253): the scene rows y in [-H/2, H/2) in chunks of `chunk_rows` (50 there), each chunk
// traced, put into the canvas (put_pixel, engine.rs:146-158: scene row y -> canvas row H - (y + H/2), i.e. bottom-up), then canvas.update().
int rrt_render_progressive(rrt_raytracer* rt, uint32_t width, uint32_t height, uint32_t* out_fb, uint32_t chunk_rows, rrt_update_fn on_update, void* user) {
    return guarded([&]() -> int {
        check_frame(rt, width, height);
        if (!out_fb) throw Error{RRT_ERR_INVALID_ARG, "null framebuffer"};
        if (chunk_rows == 0) chunk_rows = 50;                              // engine.rs:195
        DeviceGuard guard(rt->device);
        const size_t bytes = sizeof(uint32_t) * (size_t)width * height;
        if (rt->host_fb_bytes < bytes) {
            if (rt->host_fb) { (void)hipFree(rt->host_fb); rt->host_fb = nullptr; rt->host_fb_bytes = 0; }
            HIP_TRY(hipMalloc(&rt->host_fb, bytes));
            rt->host_fb_bytes = bytes;
        }
        uint32_t* d_fb = static_cast<uint32_t*>(rt->host_fb);
        FrameParams f = frame_params(rt, width, height, 0, 1, false);
        tune_variant(rt, f, d_fb, nullptr);                               // (first frame of a new size: picks the filter variant on the full frame)
        HIP_TRY(hipMemsetAsync(d_fb, 0, bytes, nullptr));                  // Canvas::new, engine.rs:135
        std::memset(out_fb, 0, bytes);
        const int64_t H = height, half = H / 2;
        HIP_TRY(hipEventRecord(rt->ev0, nullptr));
        for (int64_t cs = -half; cs < half; cs += chunk_rows) {            // engine.rs:198-199
            const int64_t ce = std::min<int64_t>(cs + chunk_rows, half);
            // canvas rows of the scene rows [cs, ce): H - (y + H/2); the row that lands on H (y = -H/2) is rejected by put_pixel (engine.rs:152-155)
            const int64_t r_lo = H - (ce - 1 + half), r_hi = std::min<int64_t>(H - (cs + half), H - 1);   // inclusive
            if (r_lo <= r_hi) {
                f.row_begin = (uint32_t)r_lo; f.row_end = (uint32_t)r_hi + 1;
                f.tile_begin = (f.row_begin / 8) * f.tiles_x; f.tile_end = ((f.row_end + 7) / 8) * f.tiles_x;
                HIP_TRY((hipError_t)launch_render(rt->scene, f, d_fb, nullptr, rt->walk));
                HIP_TRY(hipMemcpy(out_fb + (size_t)f.row_begin * width, d_fb + (size_t)f.row_begin * width,
                                  sizeof(uint32_t) * (size_t)width * (f.row_end - f.row_begin), hipMemcpyDeviceToHost));
            }
            if (on_update) on_update(user, out_fb, width, height, r_lo <= r_hi ? (uint32_t)r_lo : 0u, r_lo <= r_hi ? (uint32_t)(r_hi - r_lo + 1) : 0u);   // canvas.update(), engine.rs:253
        }
        HIP_TRY(hipEventRecord(rt->ev1, nullptr));
        record_launch(rt, width, height, 0, 1);
        return RRT_OK;
    });
}

int rrt_get_ray_colours(rrt_raytracer* rt, uint32_t n, const double* origins, const double* dirs, uint32_t* colours) {
    return guarded([&]() -> int {
        if (!rt || (n && (!origins || !dirs || !colours))) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        if (n == 0) return (int)RRT_OK;
        DeviceGuard guard(rt->device);
        double *d_o = nullptr, *d_d = nullptr; uint32_t* d_c = nullptr;
        struct Cleanup { void** p[3]; ~Cleanup() { for (auto q : p) if (*q) (void)hipFree(*q); } } cl{{(void**)&d_o, (void**)&d_d, (void**)&d_c}};
        HIP_TRY(hipMalloc((void**)&d_o, sizeof(double) * 3 * (size_t)n)); HIP_TRY(hipMalloc((void**)&d_d, sizeof(double) * 3 * (size_t)n)); HIP_TRY(hipMalloc((void**)&d_c, sizeof(uint32_t) * (size_t)n));
        HIP_TRY(hipMemcpy(d_o, origins, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_d, dirs, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice));
        const int variant = rays_variant(rt, n, [&](uint32_t m, int v) { return launch_ray_colours(rt->scene, m, d_o, d_d, d_c, nullptr, v); });
        HIP_TRY(hipEventRecord(rt->ev0, nullptr));
        HIP_TRY((hipError_t)launch_ray_colours(rt->scene, n, d_o, d_d, d_c, nullptr, variant));
        HIP_TRY(hipEventRecord(rt->ev1, nullptr));
        record_rays(rt, n, variant);
        HIP_TRY(hipMemcpy(colours, d_c, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
        return (int)RRT_OK;
    });
}

int rrt_intersect_rays(rrt_raytracer* rt, uint32_t n, const double* origins, const double* dirs, const double* max_t,
                       uint8_t* hit, double* t, double* u, double* v, uint32_t* tri) {
    return guarded([&]() -> int {
        if (!rt || (n && (!origins || !dirs || !hit || !t || !u || !v || !tri))) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        if (n == 0) return (int)RRT_OK;
        DeviceGuard guard(rt->device);
        void* bufs[9] = {};
        struct Cleanup { void** b; ~Cleanup() { for (int i = 0; i < 9; i++) if (b[i]) (void)hipFree(b[i]); } } cl{bufs};
        const size_t N = n;
        const size_t sizes[9] = {24 * N, 24 * N, 8 * N, N, 8 * N, 8 * N, 8 * N, 4 * N, 0};
        for (int i = 0; i < 8; i++) HIP_TRY(hipMalloc(&bufs[i], sizes[i]));
        HIP_TRY(hipMemcpy(bufs[0], origins, 24 * N, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(bufs[1], dirs, 24 * N, hipMemcpyHostToDevice));
        if (max_t) HIP_TRY(hipMemcpy(bufs[2], max_t, 8 * N, hipMemcpyHostToDevice));
        auto launch = [&](uint32_t m, int v) {
            return launch_intersect(rt->scene, m, (const double*)bufs[0], (const double*)bufs[1], max_t ? (const double*)bufs[2] : nullptr,
                                    (uint8_t*)bufs[3], (double*)bufs[4], (double*)bufs[5], (double*)bufs[6], (uint32_t*)bufs[7], nullptr, v);
        };
        const int variant = rays_variant(rt, n, launch);
        HIP_TRY(hipEventRecord(rt->ev0, nullptr));
        HIP_TRY((hipError_t)launch(n, variant));
        HIP_TRY(hipEventRecord(rt->ev1, nullptr));
        record_rays(rt, n, variant);
        HIP_TRY(hipMemcpy(hit, bufs[3], N, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(t, bufs[4], 8 * N, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(u, bufs[5], 8 * N, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(v, bufs[6], 8 * N, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(tri, bufs[7], 4 * N, hipMemcpyDeviceToHost));
        return (int)RRT_OK;
    });
}

// ------------------------------------------------------------------------------------------------ N GPUs of one node
// The frame's 8x8-pixel tiles are dealt round-robin to the ranks (tile k -> rank k % world), every rank traces its tiles into a compact tile-major
// buffer, ONE gather collects the buffers on rank 0 -- grouped ncclSend / ncclRecv, every peer on its own xGMI link (a ring would make the 7 hops) --
// and rank 0 de-tiles into the row-major frame.  No exchange inside the frame.  Two hosts of the same code:
//   rrt_multi_create   one process drives all GPUs (the Rust host: `hipSetDevice` loop + ncclCommInitAll);
//   rrt_dist_create    one process per GPU (bench.py under torch.distributed.run): ncclCommInitRank with an id the caller broadcasts.
// Frames are enqueued into a ring of slots, each with its own stream per GPU, tile buffer and gather buffer, so that the tracing of a frame overlaps
// the gather and de-tiling of the one before (a rank's share of a frame is a short launch: several in flight keep the GPU full).
// RCCL is bound at first use with dlopen: librrt_hip.so itself has no link-time dependency on it (single-GPU hosts never load it), and a process
// that already holds an RCCL (PyTorch) shares that one.
}  // extern "C" (reopened below)

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl& rccl() {
    static Rccl R;
    if (R.lib) return R;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);                 // an RCCL this process already holds (PyTorch's)
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) throw Error{RRT_ERR_UNSUPPORTED, std::string("RCCL is not available: ") + (dlerror() ? dlerror() : "dlopen failed")};
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) throw Error{RRT_ERR_UNSUPPORTED, std::string("RCCL lacks ") + n}; return p; };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId"); R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll"); R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart"); R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
    R.Send = (decltype(R.Send))sym("ncclSend"); R.Recv = (decltype(R.Recv))sym("ncclRecv"); R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    R.lib = h;
    return R;
}
#define NCCL_TRY(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) throw Error{RRT_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(_r)}; } while (0)

constexpr uint32_t kMaxSlots = 8;

struct Member {                      // one rank that lives in this process
    rrt_raytracer* rt = nullptr;
    int rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream[kMaxSlots] = {};
    uint32_t* tiles[kMaxSlots] = {};      // this rank's tile-major buffer per slot (rank 0: a view into its gather buffer, or its own with loopback)
    uint32_t* gathered[kMaxSlots] = {};   // rank 0 only: [world][tiles_per_rank][64]
    hipEvent_t done[kMaxSlots] = {};      // rank 0 only: the slot's frame is de-tiled
    hipEvent_t traced[kMaxSlots] = {};    // rank 0 only: its own tiles are traced (start of the wait for the peers)
    uint32_t* fb = nullptr;               // rank 0 only: frame kept for the host-framebuffer entry point
    size_t fb_bytes = 0;
};

}  // namespace

struct rrt_multi {
    std::vector<Member> members;     // ranks of this process (all of them for rrt_multi_create, one for rrt_dist_create)
    uint32_t world = 1, depth = 1, next_slot = 0, last_slot = 0;
    uint32_t w = 0, h = 0, tpr = 0;  // buffers are sized for this frame size
    bool loopback = false;           // rank 0 sends its own tiles to itself through RCCL too (single-GPU test of the transport)
    bool owns_raytracers = false;
};

namespace {

void multi_free_buffers(rrt_multi* g) {
    for (Member& m : g->members) {
        DeviceGuard guard(m.rt->device);
        for (uint32_t s = 0; s < kMaxSlots; s++) {
            if (m.gathered[s]) { (void)hipFree(m.gathered[s]); if (m.rank == 0 && !g->loopback) m.tiles[s] = nullptr; m.gathered[s] = nullptr; }
            if (m.tiles[s]) { (void)hipFree(m.tiles[s]); m.tiles[s] = nullptr; }
        }
    }
    g->w = g->h = g->tpr = 0;
}

void multi_size_buffers(rrt_multi* g, uint32_t w, uint32_t h) {
    if (g->w == w && g->h == h) return;
    for (Member& m : g->members) { DeviceGuard guard(m.rt->device); for (uint32_t s = 0; s < g->depth; s++) HIP_TRY(hipStreamSynchronize(m.stream[s])); }
    multi_free_buffers(g);
    const uint32_t tpr = rrt_tiles_per_rank(w, h, g->world);
    const size_t chunk = (size_t)tpr * 64 * sizeof(uint32_t);
    for (Member& m : g->members) {
        DeviceGuard guard(m.rt->device);
        for (uint32_t s = 0; s < g->depth; s++) {
            if (m.rank == 0) {
                HIP_TRY(hipMalloc((void**)&m.gathered[s], chunk * g->world));
                if (g->loopback) HIP_TRY(hipMalloc((void**)&m.tiles[s], chunk)); else m.tiles[s] = m.gathered[s];   // rank 0 traces straight into chunk 0
            } else {
                HIP_TRY(hipMalloc((void**)&m.tiles[s], chunk));
            }
        }
    }
    g->w = w; g->h = h; g->tpr = tpr;
}

// trace -> gather -> de-tile of one frame, enqueued on the next slot's streams; d_fb lives on rank 0's device (may be null on processes without rank 0)
void multi_enqueue(rrt_multi* g, uint32_t w, uint32_t h, void* d_fb) {
    multi_size_buffers(g, w, h);
    const uint32_t s = g->next_slot; g->next_slot = (g->next_slot + 1) % g->depth;
    const size_t count = (size_t)g->tpr * 64;
    for (Member& m : g->members) {                                        // every stream is in order: a slot's previous frame has left its buffers by now
        const int rc = rrt_render_tiles_device(m.rt, w, h, (uint32_t)m.rank, g->world, m.tiles[s], m.stream[s]);
        if (rc != RRT_OK) throw Error{rc, std::string("rank ") + std::to_string(m.rank) + ": " + rrt_last_error_detail()};
        if (m.rank == 0) { DeviceGuard guard(m.rt->device); HIP_TRY(hipEventRecord(m.traced[s], m.stream[s])); }
    }
    g->last_slot = s;
    if (g->world > 1 || g->loopback) {
        Rccl& R = rccl();
        NCCL_TRY(R.GroupStart());
        for (Member& m : g->members) {
            DeviceGuard guard(m.rt->device);
            if (m.rank == 0) {
                for (uint32_t r = g->loopback ? 0u : 1u; r < g->world; r++) NCCL_TRY(R.Recv(m.gathered[s] + (size_t)r * count, count, ncclUint32, (int)r, m.comm, m.stream[s]));
                if (g->loopback) NCCL_TRY(R.Send(m.tiles[s], count, ncclUint32, 0, m.comm, m.stream[s]));
            } else {
                NCCL_TRY(R.Send(m.tiles[s], count, ncclUint32, 0, m.comm, m.stream[s]));
            }
        }
        NCCL_TRY(R.GroupEnd());
    }
    for (Member& m : g->members) {
        if (m.rank != 0) continue;
        if (!d_fb) throw Error{RRT_ERR_INVALID_ARG, "rank 0 needs a framebuffer"};
        DeviceGuard guard(m.rt->device);
        HIP_TRY((hipError_t)launch_detile(w, h, g->world, m.gathered[s], static_cast<uint32_t*>(d_fb), m.stream[s]));
        HIP_TRY(hipEventRecord(m.done[s], m.stream[s]));
    }
}

void multi_sync(rrt_multi* g) {
    for (Member& m : g->members) { DeviceGuard guard(m.rt->device); for (uint32_t s = 0; s < g->depth; s++) HIP_TRY(hipStreamSynchronize(m.stream[s])); }
}

void multi_init_member(rrt_multi* g, Member& m) {
    DeviceGuard guard(m.rt->device);
    for (uint32_t s = 0; s < g->depth; s++) {
        HIP_TRY(hipStreamCreateWithFlags(&m.stream[s], hipStreamNonBlocking));
        if (m.rank == 0) { HIP_TRY(hipEventCreate(&m.done[s])); HIP_TRY(hipEventCreate(&m.traced[s])); }
    }
}

}  // namespace

extern "C" {

int rrt_multi_create(rrt_raytracer* const* rts, uint32_t n, uint32_t frames_in_flight, uint32_t flags, rrt_multi** out) {
    return guarded([&]() -> int {
        if (!rts || !out || n == 0) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        if (frames_in_flight == 0) frames_in_flight = 1;
        if (frames_in_flight > kMaxSlots) throw Error{RRT_ERR_INVALID_ARG, "frames_in_flight > 8"};
        std::vector<int> devs(n);
        for (uint32_t i = 0; i < n; i++) {
            if (!rts[i]) throw Error{RRT_ERR_INVALID_ARG, "null raytracer"};
            devs[i] = rts[i]->device;
            for (uint32_t j = 0; j < i; j++) if (devs[j] == devs[i]) throw Error{RRT_ERR_INVALID_ARG, "two raytracers on one device (RCCL wants one rank per GPU)"};
        }
        std::unique_ptr<rrt_multi, void (*)(rrt_multi*)> g(new rrt_multi, rrt_multi_destroy);
        g->world = n; g->depth = frames_in_flight; g->loopback = (flags & RRT_MULTI_LOOPBACK) != 0;
        g->members.resize(n);
        for (uint32_t i = 0; i < n; i++) { g->members[i].rt = rts[i]; g->members[i].rank = (int)i; }
        if (n > 1 || g->loopback) {
            std::vector<ncclComm_t> comms(n);
            NCCL_TRY(rccl().CommInitAll(comms.data(), (int)n, devs.data()));
            for (uint32_t i = 0; i < n; i++) g->members[i].comm = comms[i];
        }
        for (Member& m : g->members) multi_init_member(g.get(), m);
        *out = g.release();
        return RRT_OK;
    });
}

int rrt_dist_unique_id(void* out128) {
    return guarded([&]() -> int {
        if (!out128) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        ncclUniqueId id;
        NCCL_TRY(rccl().GetUniqueId(&id));
        std::memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
        return RRT_OK;
    });
}

int rrt_dist_create(rrt_raytracer* rt, uint32_t rank, uint32_t world, const void* unique_id128, uint32_t frames_in_flight, rrt_multi** out) {
    return guarded([&]() -> int {
        if (!rt || !out || world == 0 || rank >= world || (world > 1 && !unique_id128)) throw Error{RRT_ERR_INVALID_ARG, "bad argument"};
        if (frames_in_flight == 0) frames_in_flight = 1;
        if (frames_in_flight > kMaxSlots) throw Error{RRT_ERR_INVALID_ARG, "frames_in_flight > 8"};
        std::unique_ptr<rrt_multi, void (*)(rrt_multi*)> g(new rrt_multi, rrt_multi_destroy);
        g->world = world; g->depth = frames_in_flight;
        g->members.resize(1);
        g->members[0].rt = rt; g->members[0].rank = (int)rank;
        if (world > 1) {
            DeviceGuard guard(rt->device);
            ncclUniqueId id;
            std::memcpy(id.internal, unique_id128, NCCL_UNIQUE_ID_BYTES);
            NCCL_TRY(rccl().CommInitRank(&g->members[0].comm, (int)world, id, (int)rank));
        }
        multi_init_member(g.get(), g->members[0]);
        *out = g.release();
        return RRT_OK;
    });
}

void rrt_multi_destroy(rrt_multi* g) {
    if (!g) return;
    try {
        for (Member& m : g->members) { if (!m.rt) continue; DeviceGuard guard(m.rt->device); for (uint32_t s = 0; s < kMaxSlots; s++) if (m.stream[s]) (void)hipStreamSynchronize(m.stream[s]); }
        multi_free_buffers(g);
        for (Member& m : g->members) {
            if (!m.rt) continue;
            DeviceGuard guard(m.rt->device);
            if (m.comm) (void)rccl().CommDestroy(m.comm);
            for (uint32_t s = 0; s < kMaxSlots; s++) { if (m.stream[s]) (void)hipStreamDestroy(m.stream[s]); if (m.done[s]) (void)hipEventDestroy(m.done[s]); if (m.traced[s]) (void)hipEventDestroy(m.traced[s]); }
            if (m.fb) (void)hipFree(m.fb);
        }
    } catch (...) {}
    delete g;
}

int rrt_multi_enqueue(rrt_multi* g, uint32_t width, uint32_t height, void* d_fb) {
    return guarded([&]() -> int {
        if (!g) throw Error{RRT_ERR_INVALID_ARG, "null handle"};
        check_frame(g->members[0].rt, width, height);
        multi_enqueue(g, width, height, d_fb);
        return RRT_OK;
    });
}

int rrt_multi_sync(rrt_multi* g) {
    return guarded([&]() -> int {
        if (!g) throw Error{RRT_ERR_INVALID_ARG, "null handle"};
        multi_sync(g);
        return RRT_OK;
    });
}

int rrt_multi_last_gather_ms(rrt_multi* g, double* out_ms) {
    return guarded([&]() -> int {
        if (!g || !out_ms) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        *out_ms = -1.0;
        for (Member& m : g->members) {
            if (m.rank != 0) continue;
            DeviceGuard guard(m.rt->device);
            HIP_TRY(hipEventSynchronize(m.done[g->last_slot]));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, m.traced[g->last_slot], m.done[g->last_slot]));
            *out_ms = ms;
        }
        return RRT_OK;
    });
}

int rrt_render_multi(rrt_multi* g, uint32_t width, uint32_t height, uint32_t* out_fb) {
    return guarded([&]() -> int {
        if (!g) throw Error{RRT_ERR_INVALID_ARG, "null handle"};
        check_frame(g->members[0].rt, width, height);
        Member* root = nullptr;
        for (Member& m : g->members) if (m.rank == 0) root = &m;
        const size_t bytes = sizeof(uint32_t) * (size_t)width * height;
        if (root) {
            if (!out_fb) throw Error{RRT_ERR_INVALID_ARG, "null framebuffer"};
            DeviceGuard guard(root->rt->device);
            if (root->fb_bytes < bytes) { if (root->fb) (void)hipFree(root->fb); root->fb = nullptr; root->fb_bytes = 0; HIP_TRY(hipMalloc((void**)&root->fb, bytes)); root->fb_bytes = bytes; }
        }
        const uint32_t slot = g->next_slot;
        multi_enqueue(g, width, height, root ? root->fb : nullptr);
        if (root) {
            DeviceGuard guard(root->rt->device);
            HIP_TRY(hipMemcpyAsync(out_fb, root->fb, bytes, hipMemcpyDeviceToHost, root->stream[slot]));   // pinned (rrt_host_buffer_register) or pageable destination
        }
        multi_sync(g);                                                    // blocking: the frame is in out_fb on return
        return RRT_OK;
    });
}

#ifdef RRT_PROFILE
// developer build only: read and clear the 16 work counters
int rrt_prof_counters(rrt_raytracer* rt, unsigned long long* out16) {
    return guarded([&]() -> int {
        DeviceGuard guard(rt->device);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out16, rt->scene.prof, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(rt->scene.prof, 0, 24 * sizeof(unsigned long long)));
        return RRT_OK;
    });
}
// developer build `make band`: read and clear the four (alpha, delta)-band pair counters (render.hip: band_count)
int rrt_prof_band_counters(rrt_raytracer* rt, unsigned long long* out4) {
    return guarded([&]() -> int {
        DeviceGuard guard(rt->device);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out4, rt->scene.prof + 24, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(rt->scene.prof + 24, 0, 4 * sizeof(unsigned long long)));
        return RRT_OK;
    });
}
#endif

int rrt_get_setup_times(const rrt_model* m, const rrt_raytracer* rt, rrt_setup_times* out) {
    return guarded([&]() -> int {
        if (!out) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        std::memset(out, 0, sizeof *out);
        if (m) { out->read_ms = m->m.read_ms; out->parse_ms = m->m.parse_ms; out->texture_ms = m->m.texture_ms; out->octree_ms = m->m.octree_ms; }
        if (rt) {
            out->index_ms = rt->index_ms; out->upload_ms = rt->upload_ms; out->hip_init_ms = rt->hip_init_ms; out->create_ms = rt->create_ms;
            out->gpu_setup = rt->gpu_setup ? 1.0 : 0.0;
            if (rt->gpu_setup || !m) out->octree_ms = rt->octree_ms;       // the tree this raytracer traces was built on its device
        }
        return RRT_OK;
    });
}

int rrt_last_stats(const rrt_raytracer* rt_c, rrt_stats* out) {
    return guarded([&]() -> int {
        rrt_raytracer* rt = const_cast<rrt_raytracer*>(rt_c);
        if (!rt || !out) throw Error{RRT_ERR_INVALID_ARG, "null argument"};
        if (rt->stats_pending) {
            DeviceGuard guard(rt->device);
            HIP_TRY(hipEventSynchronize(rt->ev1));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, rt->ev0, rt->ev1));
            rt->stats.kernel_ms = ms;
            rt->stats_pending = false;
        }
        if (!rt->launched) rt->stats.filter_variant = (uint32_t)rt->walk;   // (before the first launch: the forced variant, or 0)
        rt->stats.origin_plane_triangles = rt->n_suspects; rt->stats.scene_bytes = rt->scene_bytes;
        {   // the exactness band of the index (clusters.cpp: find_origin_suspects has the per-pair formulas)
            const double mag = (double)rt->scene.cull_limit / 4.0, pad = mag / 32768.0, eps = 0x1p-53;
            rt->stats.filter_pad = rt->scene.cull_enabled ? pad : 0.0;
            rt->stats.filter_alpha_unit = (rt->scene.cull_enabled && pad > 0) ? 8.0 * 64.0 * eps * mag / pad : 0.0;
            rt->stats.filter_delta_unit = (rt->scene.cull_enabled && pad > 0) ? 2.0 * (rt->stats.filter_alpha_unit * mag + 64.0 * eps * mag) : 0.0;
        }
        *out = rt->stats;
        return RRT_OK;
    });
}

}  // extern "C"
