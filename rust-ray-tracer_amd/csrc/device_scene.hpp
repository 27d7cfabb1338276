// device_scene.hpp -- layout of the scene in HBM and the host-side launch interface of the HIP kernels.
//
// Everything is uploaded once per raytracer (rrt_raytracer_create).  Layout choices:
//  * geometry is stored in OWN-LIST SLOT ORDER: slot s = position in the concatenation of every node's
//    `triangles` Vec (octree.rs:9) in node-id order, so a node's triangles are one contiguous run and the
//    traversal needs no tri_index indirection (ray.rs:119-120).  orig[s] maps back to push order.
//  * TriGeom keeps v1 and the two edges e1 = v2-v1, e2 = v3-v1.  The reference recomputes the edges on every
//    test (ray.rs:60-61); a f64 subtraction gives the same bits whenever it is done, so they are precomputed.
//  * one node is one 64-byte record: box + first_child + own run + occupancy flags of its 8 children, read with
//    wave-uniform (scalar) loads.
#pragma once
#include <cstdint>

namespace rrt {

struct DevNode {                 // 96 B
    double lo[3], hi[3];         // Aabb, aabb.rs:4-8
    double mid[3];               // the split planes lo + (hi-lo)/2 (octree.rs:136-138): with lo/hi they ARE the 8 children's boxes
    uint32_t first_child;        // 0 = leaf; children are first_child..first_child+7 (octree.rs:226-238)
    uint32_t sup_begin;          // first super-cluster of this node's own triangle list (clusters.cpp)
    uint32_t sup_count;
    uint32_t flags;              // bit k (0..7): child k has triangle_count > 0; bit 8: this node has triangle_count > 0 (ray.rs:112);
                                 // bit 9+k: child k is a single-triangle leaf whose triangle is tested here, at its parent (clusters.cpp);
                                 // bits 24..30: number of slots of the first super-cluster (<= 64; the only one when sup_count == 1)
    uint32_t s0_begin;           // first slot of the first super-cluster
    uint32_t leaf_base;          // dense slot of the triangle of the first such leaf child; the j-th such child (in child order) has slot leaf_base + j.
                                 // In the record of such a leaf itself: its own dense slot.
};
static_assert(sizeof(DevNode) == 96, "DevNode must be 96 bytes");

struct DevTriGeom { double v1[3], e1[3], e2[3]; uint32_t pos, _pad; };   // 80 B; pos = position in the node's own list (tie-break, ray.rs:124)
static_assert(sizeof(DevTriGeom) == 80, "DevTriGeom must be 80 bytes");

// own-list index (clusters.cpp): padded f32 boxes, rounded outward
// A super-cluster owns the slots [tri_begin, tri_begin + tri_count), tri_begin a multiple of 8, tri_count <= 64; its cluster c is the
// 8 slots from tri_begin + 8c, and the box of the cluster that starts at slot s is cboxes[s / 8].  Slots between the end of a
// super-cluster and the next multiple of 8 are padding (all-zero geometry, never hit).
// Box records.  While clusters.cpp builds the index they hold (lo, hi); build_clusters ends by rewriting every record IN PLACE to the device
// form (centre, half-extent) with [c - h, c + h] a superset of [lo, hi] (boxes_to_centre_half): the kernels' slab tests take the near and far
// plane of an axis as one packed FMA  c*inv + n -/+ h*|inv|  instead of two FMAs and a min/max pair.  An empty box is (0, -FLT_MAX): never
// hit; a box of the no-cull index is (0, FLT_MAX): always hit.
struct DevSuper { union { float lo[3]; float c[3]; }; union { float hi[3]; float h[3]; }; uint32_t tri_begin, tri_count; };     // 32 B
struct DevClusterBox { union { float lo[3]; float c[3]; }; union { float hi[3]; float h[3]; }; uint32_t _pad[2]; };             // 32 B
static_assert(sizeof(DevClusterBox) == 32 && sizeof(DevSuper) == 32, "cluster records must be 32 bytes");

struct DevTriAttr {                                                // 128 B: one cache line per shaded hit
    double uv[6];                // t1.x,t1.y, t2.x,t2.y, t3.x,t3.y (raytracer.rs:45-50 read x,y only)
    double nrm[9];               // n1, n2, n3
    uint32_t mat, orig;          // material index; triangle index in push order
};
static_assert(sizeof(DevTriAttr) == 128, "DevTriAttr must be 128 bytes");

struct DevTexture { const uint8_t* rgb; uint32_t width, height; };               // entities.rs:86-91
// material.rs:11-22, with the descriptors of its texture and bump map inline: a hit reads material -> texel, not material -> texture table -> texel
struct DevMaterial { double ka[3], kd[3], ks[3], ns, kr; int32_t tex, bump; DevTexture tex_desc, bump_desc; };
struct DevLight { uint32_t kind, _pad; double intensity; double v[3]; };         // entities.rs:5-9

// Exactness guard of the own-list index: a triangle whose plane contains the raytracer's origin to within rounding-noise
// distance.  A ray from that origin that is also parallel to the plane to within `alpha` lies IN the plane, the reference's Moller-Trumbore result
// for the pair is rounding noise, and the box filters -- which assume an accepted pair lies inside the padded box -- are switched off for that ray.
struct DevSuspect { double n[3]; double alpha2; };   // unit normal of the plane; squared sine threshold (>= 1: every direction)
#define RRT_MAX_SUSPECTS 64

#define RRT_MAX_LIGHTS 16
#define RRT_MAX_REFLECT 8

struct DevScene {                // passed to kernels by value (kernarg segment -> SGPRs)
    const DevNode* nodes;
    const DevTriGeom* geom;
    const DevSuper* supers;
    const DevClusterBox* cboxes;
    const DevClusterBox* child_boxes;
    const DevClusterBox* tboxes;          // one padded box per slot (per triangle)
    const DevTriAttr* attr;
    const DevMaterial* mats;
    const DevTexture* tex;
    uint32_t n_nodes, n_slots, n_mats, n_tex;
    uint32_t n_lights, max_reflection_depth, stack_levels;
    uint32_t fc_mask;            // 0x00FFFFFF when stack frames carry a leaf-hit mask above first_child (scenes below 2^24 nodes), else 0xFFFFFFFF
    double origin[3];
    double surface_offset;
    float cull_limit;            // rays with |origin| or |direction| components beyond this (or non-finite) skip the box culling
    float cull_half_over_limit;  // 0.5 / cull_limit (the fp32 filter's parameter scale, render.hip make_ray32)
    uint32_t cull_enabled;
    uint32_t has_groups;         // some own list carries group records (clusters.cpp): launches use the kernel instantiation that handles them
    uint32_t n_suspects;         // triangles whose plane passes through `origin` (see DevSuspect); more than RRT_MAX_SUSPECTS: every ray from `origin` runs unfiltered
    const DevSuspect* suspects;
    float inner_shrink;          // 2 x the filter's pad when every triangle of the tree lies inside the root box (a child's subtree box then lies inside its octant box), else 0: render.hip, RRT_CERTAIN_HIT
    uint32_t bounds_plain;       // every node plane (lo, mid, hi) is 0 or has magnitude in [2^-200, 2^200]: the walk may share the reciprocal of a ray's direction across its slab quotients (render.hip, RayRcp)
    DevLight lights[RRT_MAX_LIGHTS];
#ifdef RRT_PROFILE
    unsigned long long* prof;    // developer build only (make prof): 16 wave-level work counters + 8 s_memtime region timers, see tools/profile_counters.py
#endif
};

struct FrameParams {
    uint32_t width, height;      // canvas size
    double x_scale, y_scale, z_value;   // engine.rs:189-191
    uint32_t tiles_x, tiles_y;   // 8x8-pixel tiles covering the canvas
    uint32_t rank, world;        // tile k belongs to rank k % world
    uint32_t tiled_output;       // 0: out is the row-major framebuffer; 1: out is this rank's tile-major buffer
    uint32_t xcd_chunk;          // blocks per chunk of the XCD-aware block order (render.hip: render_kernel), 0 = plain order
    // A launch may cover a band of the frame only (progressive display, engine.rs:196-253): the tiles [tile_begin, tile_end) in row-major tile
    // order, and of those only the canvas rows [row_begin, row_end).  The whole frame: tile_begin = 0, tile_end = tiles_x*tiles_y, rows [0, height).
    uint32_t tile_begin, tile_end, row_begin, row_end;
};

// kernel launches (render.hip).  All return hipError_t cast to int; stream is a hipStream_t.
// walk: 0 = node-coherent walk with the lane filter, 1 = node-coherent walk with the bundle filter, 2 = ray walk (render.hip)
int launch_render(const DevScene& s, const FrameParams& f, uint32_t* d_out, void* stream, int walk);
int launch_detile(uint32_t width, uint32_t height, uint32_t world, const uint32_t* d_gathered, uint32_t* d_fb, void* stream);
int launch_ray_colours(const DevScene& s, uint32_t n, const double* d_origins, const double* d_dirs, uint32_t* d_colours, void* stream, int walk);
int launch_intersect(const DevScene& s, uint32_t n, const double* d_origins, const double* d_dirs, const double* d_max_t,
                     uint8_t* d_hit, double* d_t, double* d_u, double* d_v, uint32_t* d_tri, void* stream, int walk);
uint32_t stack_bytes_per_wave(uint32_t levels);
void preload_kernels();

}  // namespace rrt

#include <vector>
namespace rrt {
struct Model;
struct ClusterSet {
    std::vector<DevSuper> supers;
    std::vector<DevClusterBox> tboxes;                 // per slot, + 8 spare records
    std::vector<DevClusterBox> child_boxes;            // tight padded bounds of the subtree of node c at [c - 1]; the 8 children of a node are consecutive
    std::vector<DevClusterBox> cboxes;                 // one per 8 slots, + 8 spare records so a 4x64-byte burst never leaves the buffer
    std::vector<uint32_t> slot_tri, slot_pos;          // per device slot: triangle index in push order (kPadSlot for padding), position in its node's own list
    std::vector<uint32_t> node_sup_begin, node_sup_count;
    std::vector<uint32_t> node_leaf_slot;              // slot of the triangle of a single-triangle leaf that is tested at its parent (kPadSlot otherwise)
    bool has_groups = false;                           // some list got group records
    bool inline_leaves = false;                        // single-triangle leaves are tested at their parents (node_leaf_slot, DevNode::leaf_base)
    uint32_t n_list_slots = 0;                         // slots [0, n_list_slots) belong to own lists (cboxes/tboxes cover these); leaf slots follow
    double scene_magnitude = 0;
    double pad = 0;                                    // absolute padding of every index box
};
constexpr uint32_t kPadSlot = 0xFFFFFFFFu;
void build_clusters(const Model& m, bool enable_cull, ClusterSet& out);
// exactness guard: the triangles (of the tree) whose plane contains `origin` to within the distance at which a ray from `origin` can be
// coplanar-to-rounding with them
void find_origin_suspects(const Model& m, const double origin[3], double pad, std::vector<DevSuspect>& out);

}  // namespace rrt
