// rrt_host.hpp -- C++ host mirror of the reference's Rust host types over the C ABI (include/rrt.h).
//
// The reference is compiled Rust and no Rust toolchain exists in this pipeline, so the host side above the C ABI is written in C++ with the
// reference's names, argument meaning and ownership (errors: the reference panics, this throws rrt::host::Error):
//   Vector3d                       src/scene/engine.rs:9-14
//   Light::{Ambient,Point,Directional}   src/scene/entities.rs:5-9
//   SceneData  <- parse_obj_file_lines   src/file_management/utils.rs:139, src/scene/scenedata.rs:5-13
//   RayTracer{scene_data, lights, origin} + get_ray_colour      src/scene/raytracer.rs:22-31
//   Canvas{width,height,buffer} + Scene::new / draw_scene       src/scene/engine.rs:123-167, 171-255
// Header-only; link with -lrrt_hip.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <functional>
#include <vector>

#include "../../../include/rrt.h"

namespace rrt::host {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& what) : std::runtime_error(what + ": " + rrt_strerror(st) + " (" + rrt_last_error_detail() + ")"), status(st) {}
};
inline void check(int st, const char* what) { if (st != RRT_OK) throw Error(st, what); }

struct Vector3d { double x, y, z; };   // engine.rs:9-14

struct Light {                         // entities.rs:5-9
    rrt_light raw;
    static Light Ambient(double intensity) { return Light{{0u, 0u, intensity, {0, 0, 0}}}; }
    static Light Point(double intensity, Vector3d position) { return Light{{1u, 0u, intensity, {position.x, position.y, position.z}}}; }
    static Light Directional(double intensity, Vector3d direction) { return Light{{2u, 0u, intensity, {direction.x, direction.y, direction.z}}}; }
};

class SceneData {                      // scenedata.rs:5-13 (triangles, materials, textures, octree -- all inside the rrt_model)
public:
    explicit SceneData(rrt_model* m) : m_(m) {}
    SceneData(SceneData&& o) noexcept : m_(std::exchange(o.m_, nullptr)) {}
    SceneData(const SceneData&) = delete;
    ~SceneData() { rrt_model_destroy(m_); }
    const rrt_model* model() const { return m_; }
    rrt_model_info info() const { rrt_model_info i; check(rrt_model_get_info(m_, &i), "rrt_model_get_info"); return i; }
private:
    rrt_model* m_;
};

// fs::read_to_string + parse_obj_file_lines (main.rs:28-30, utils.rs:139-213); root = Octree::new(-20,20,...) (utils.rs:145)
inline SceneData parse_obj_file(const std::string& path) {
    rrt_model* m = nullptr;
    check(rrt_model_load_obj(path.c_str(), nullptr, &m), "parse_obj_file");
    return SceneData(m);
}

class RayTracer {                      // raytracer.rs:22-26; uploads the scene to one MI355X
public:
    RayTracer(const SceneData& scene_data, const std::vector<Light>& lights, Vector3d origin, int device = 0, const rrt_options* opt = nullptr) {
        std::vector<rrt_light> raw;
        for (const Light& l : lights) raw.push_back(l.raw);
        check(rrt_raytracer_create(scene_data.model(), raw.data(), (uint32_t)raw.size(), rrt_vec3{origin.x, origin.y, origin.z}, opt, device, &rt_), "RayTracer");
    }
    RayTracer(RayTracer&& o) noexcept : rt_(std::exchange(o.rt_, nullptr)) {}
    RayTracer(const RayTracer&) = delete;
    ~RayTracer() { rrt_raytracer_destroy(rt_); }
    // raytracer.rs:29 -> 0x00RRGGBB
    uint32_t get_ray_colour(Vector3d origin, Vector3d direction) const {
        const double o[3] = {origin.x, origin.y, origin.z}, d[3] = {direction.x, direction.y, direction.z};
        uint32_t c = 0;
        check(rrt_get_ray_colours(rt_, 1, o, d, &c), "get_ray_colour");
        return c;
    }
    rrt_raytracer* handle() const { return rt_; }
    rrt_stats last_stats() const { rrt_stats s; check(rrt_last_stats(rt_, &s), "rrt_last_stats"); return s; }
private:
    rrt_raytracer* rt_ = nullptr;
};

struct Canvas {                        // engine.rs:123-167 without the minifb window
    size_t width, height;
    std::vector<uint32_t> buffer;      // engine.rs:127,135: width*height, zero-initialised
    std::function<void(const Canvas&)> on_update;   // stands in for window.update_with_buffer (engine.rs:162-166); empty = no display
    size_t updates = 0;
    Canvas(size_t w, size_t h) : width(w), height(h), buffer(w * h, 0u) {}
    void update() { ++updates; if (on_update) on_update(*this); }      // engine.rs:160-167
};

class Scene {                          // engine.rs:171-255
public:
    Canvas canvas;
    Scene(size_t width, size_t height) : canvas(width, height) {}       // Scene::new, engine.rs:177
    // Scene::draw_scene (engine.rs:186): one HIP launch instead of the rayon row loop; the reference consumes `rt`, here it is borrowed
    void draw_scene(const RayTracer& rt) {
        check(rrt_render(rt.handle(), (uint32_t)canvas.width, (uint32_t)canvas.height, canvas.buffer.data()), "draw_scene");
        canvas.update();
    }
    // the reference's pacing (engine.rs:196-253): 50 scene rows per chunk, bottom of the canvas first, canvas.update() after every chunk
    void draw_scene_progressive(const RayTracer& rt) {
        check(rrt_render_progressive(rt.handle(), (uint32_t)canvas.width, (uint32_t)canvas.height, canvas.buffer.data(), 0,
                                     [](void* user, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t) { static_cast<Canvas*>(user)->update(); }, &canvas),
              "draw_scene_progressive");
    }
};

// the constants `main` hard-codes (main.rs:32-67)
inline std::vector<Light> default_lights() {
    return {Light::Ambient(0.5), Light::Point(0.4, {-7.0, 1.0, -15.0}), Light::Point(0.5, {0.0, 1.0, -41.0}), Light::Directional(0.4, {-5.0, 0.0, 20.0})};
}
inline Vector3d default_origin() { return {0.0, 2.0, -10.0}; }

}  // namespace rrt::host
