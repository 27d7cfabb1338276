// render_cli.cpp -- the reference's `main` (src/main.rs:17-83) over the C++ host mirror: argv[1] = .obj, hard-coded lights and camera,
// Scene::new(W,H).draw_scene(rt), "It took ... to draw the scene"; the minifb window loop (main.rs:80-82) is replaced by writing a binary PPM.
//   render_cli <file.obj> [out.ppm] [width height] [--progressive]     (--progressive: the reference's 50-row chunks with an update after each, engine.rs:196-253)
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "rrt_host.hpp"

int main(int argc, char** argv) {
    using namespace rrt::host;
    if (argc < 2) { std::fprintf(stderr, "First argument needs to be the name of a file with vertex and triangle data\n"); return 2; }   // main.rs:22-24
    const char* out = argc > 2 ? argv[2] : "out.ppm";
    const size_t W = argc > 4 ? (size_t)std::atoi(argv[3]) : 800, H = argc > 4 ? (size_t)std::atoi(argv[4]) : 800;                       // main.rs:14-15
    try {
        std::printf("using model file: %s\n", argv[1]);                                                                                   // main.rs:26
        SceneData scene_data = parse_obj_file(argv[1]);
        RayTracer rt(scene_data, default_lights(), default_origin());
        Scene scene(W, H);
        scene.draw_scene(rt);                                                                                                             // first frame: uploads code, tunes the filter
        const auto t0 = std::chrono::steady_clock::now();
        bool progressive = false;
        for (int i = 2; i < argc; i++) if (std::string(argv[i]) == "--progressive") progressive = true;
        if (progressive) { scene.canvas.updates = 0; scene.draw_scene_progressive(rt); std::printf("canvas updates: %zu\n", scene.canvas.updates); }
        else scene.draw_scene(rt);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("It took: %.2fms to draw the scene (kernel %.3f ms)\n", ms, rt.last_stats().kernel_ms);                               // main.rs:76
        std::FILE* f = std::fopen(out, "wb");
        if (!f) { std::fprintf(stderr, "cannot write %s\n", out); return 1; }
        std::fprintf(f, "P6\n%zu %zu\n255\n", W, H);
        for (uint32_t c : scene.canvas.buffer) { const unsigned char px[3] = {(unsigned char)(c >> 16), (unsigned char)(c >> 8), (unsigned char)c}; std::fwrite(px, 1, 3, f); }
        std::fclose(f);
        std::printf("draw finished\n");                                                                                                   // main.rs:78
    } catch (const Error& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
