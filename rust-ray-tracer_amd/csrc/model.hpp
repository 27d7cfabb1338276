// model.hpp -- host-side scene IR of the MI355X hot path (SceneData, src/scene/scenedata.rs:5-13 of the reference).
// Plain f64 everywhere, as the reference (src/scene/engine.rs:9-14).
#pragma once
#include <sys/mman.h>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rrt.h"

namespace rrt {

struct Vec3 { double x = 0, y = 0, z = 0; };

// Triangle, src/scene/entities.rs:72-84 (Arc<Material> -> material index)
struct Triangle {
    Vec3 v1, v2, v3;
    Vec3 t1, t2, t3;   // only .x/.y are read (raytracer.rs:45-50)
    Vec3 n1, n2, n3;
    uint32_t mat = 0;
    uint32_t _pad = 0;
};
static_assert(sizeof(Triangle) == 224, "Triangle: 27 doubles + material index + padding (uploaded as is, scene_build.hip)");

// SceneData.triangles.  A plain array whose resize() does NOT initialise: at a million triangles a value-initialising std::vector::resize is a
// 224 MB single-threaded memset before the (parallel) writers even start.  Every writer assigns every field of every element.
class TriArray {
    Triangle* p_ = nullptr; size_t n_ = 0;
public:
    TriArray() = default;
    TriArray(const TriArray&) = delete; TriArray& operator=(const TriArray&) = delete;
    ~TriArray() { std::free(p_); }
    void resize_uninit(size_t n) {
        std::free(p_); p_ = nullptr; n_ = 0;
        if (n) {
            // Large arrays on transparent huge pages where the kernel offers them: the parallel writers fault every page in, and 224 MB is 55 000
            // small pages (taken one at a time under the address-space lock) or 107 large ones.
            const size_t bytes = n * sizeof(Triangle), huge = (size_t)2 << 20;
            if (bytes >= 4 * huge) {
                void* q = nullptr;
                if (posix_memalign(&q, huge, (bytes + huge - 1) / huge * huge) != 0) throw std::bad_alloc();
                (void)madvise(q, (bytes + huge - 1) / huge * huge, MADV_HUGEPAGE);
                p_ = static_cast<Triangle*>(q);
            } else { p_ = static_cast<Triangle*>(std::malloc(bytes)); if (!p_) throw std::bad_alloc(); }
        }
        n_ = n;
    }
    size_t size() const { return n_; }
    Triangle* data() { return p_; } const Triangle* data() const { return p_; }
    Triangle& operator[](size_t i) { return p_[i]; } const Triangle& operator[](size_t i) const { return p_[i]; }
    const Triangle* begin() const { return p_; } const Triangle* end() const { return p_ + n_; }
};

// Texture, entities.rs:86-91.  rgb.size() == 3*width*height
struct Texture { std::vector<uint8_t> rgb; uint32_t width = 0, height = 0; };

struct Box { double lo[3], hi[3]; };   // Aabb, src/collision/aabb.rs:4-8

// The octree of src/collision/octree.rs flattened for upload: children of a node are always the 8 consecutive
// node ids appended by one subdivide() call (octree.rs:226-238), so one `first_child` suffices (0 = leaf).
struct FlatOctree {
    std::vector<Box> box;                 // node AABB
    std::vector<uint32_t> first_child;    // 0 = leaf
    std::vector<uint32_t> tri_count;      // OctantNode.triangle_count (octree.rs:75) -- NOT the own-list length
    std::vector<uint32_t> own_off;        // CSR over nodes, size n_nodes+1
    std::vector<uint32_t> own_idx;        // OctantNode.triangles, insertion order (first-wins tie-break, ray.rs:124)
    uint32_t max_depth = 0;               // root = 1
};

struct Model {
    TriArray triangles;                   // SceneData.triangles, push order
    std::vector<rrt_material> materials;
    std::vector<Texture> textures;
    Box root{};
    // The host copy of the octree is built on demand (host_tree(): rrt_model_get_info / rrt_model_get_octree, and raytracers created with
    // RRT_FLAG_HOST_SETUP).  The default set-up builds the tree on the GPU inside rrt_raytracer_create (scene_build.hip) and never needs it.
    mutable FlatOctree tree;
    mutable std::mutex tree_mu;
    mutable bool tree_ready = false;
    double read_ms = 0, parse_ms = 0, texture_ms = 0;   // wall time of the set-up stages (reported by rrt_get_setup_times)
    mutable double octree_ms = 0;                        // host octree build, when it ran
};
// builds m.tree on the host if it is not there yet (octree.cpp); throws Error{RRT_ERR_DEPTH} for a tree deeper than RRT_MAX_OCTREE_DEPTH
const FlatOctree& host_tree(const Model& m);

// octree.cpp
void build_octree(const Triangle* tris, size_t n_tris, const Box& root, FlatOctree& out);

// obj_loader.cpp -- throws rrt::Error
struct Error { int status; std::string detail; };
void load_obj(const std::string& obj_path, const Box& root, Model& out);

// image_decode.cpp -- native-layout bytes as `DynamicImage::as_bytes()` would hand out (utils.rs:353);
// channels = 1 (Luma8), 3 (Rgb8) or 4 (Rgba8).  Throws rrt::Error.
void decode_image_file(const std::string& path, std::vector<uint8_t>& bytes, uint32_t& width, uint32_t& height, uint32_t& channels);

void set_error_detail(const std::string& s);

}  // namespace rrt
