// model.hpp -- host-side scene IR of the MI355X hot path (SceneData, src/scene/scenedata.rs:5-13 of the reference).
// Plain f64 everywhere, as the reference (src/scene/engine.rs:9-14).
#pragma once
#include <cstdint>
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
    std::vector<Triangle> triangles;      // SceneData.triangles, push order
    std::vector<rrt_material> materials;
    std::vector<Texture> textures;
    Box root{};
    FlatOctree tree;
    double read_ms = 0, parse_ms = 0, texture_ms = 0, octree_ms = 0;   // wall time of the set-up stages (reported by rrt_get_setup_times)
};

// octree.cpp
void build_octree(const std::vector<Triangle>& tris, const Box& root, FlatOctree& out);

// obj_loader.cpp -- throws rrt::Error
struct Error { int status; std::string detail; };
void load_obj(const std::string& obj_path, const Box& root, Model& out);

// image_decode.cpp -- native-layout bytes as `DynamicImage::as_bytes()` would hand out (utils.rs:353);
// channels = 1 (Luma8), 3 (Rgb8) or 4 (Rgba8).  Throws rrt::Error.
void decode_image_file(const std::string& path, std::vector<uint8_t>& bytes, uint32_t& width, uint32_t& height, uint32_t& channels);

void set_error_detail(const std::string& s);

}  // namespace rrt
