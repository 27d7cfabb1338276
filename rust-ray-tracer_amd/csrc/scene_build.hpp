// scene_build.hpp -- the once-per-scene set-up ON THE GPU: octree build (src/collision/octree.rs:41-241, order-exact),
// own-list index (clusters.cpp restated level-parallel) and the device records (DevNode / DevTriGeom / DevTriAttr), all from the uploaded
// triangle array.  scene_build.hip implements it; api.cpp drives it from rrt_raytracer_create.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

#include "device_scene.hpp"
#include "model.hpp"

namespace rrt {

// Everything the trace kernels read, as built on the device, plus the flattened octree in the reference's node numbering (kept on the device
// until the raytracer is destroyed so that rrt_raytracer_get_octree can hand it out).
struct GpuScene {
    // final buffers: ONE allocation (`scene_alloc`), carved up
    void* scene_alloc = nullptr; size_t scene_alloc_bytes = 0;
    DevNode* nodes = nullptr; DevTriGeom* geom = nullptr; DevTriAttr* attr = nullptr;
    DevSuper* supers = nullptr; DevClusterBox* cboxes = nullptr; DevClusterBox* child_boxes = nullptr; DevClusterBox* tboxes = nullptr;
    DevSuspect* suspects = nullptr;
    // flattened octree, final ids (octree.rs numbering): same allocation
    double* oct_box = nullptr;          // [n_nodes][6] lo xyz, hi xyz
    uint32_t* oct_first_child = nullptr, *oct_tri_count = nullptr, *oct_own_off = nullptr /* n_nodes + 1 */, *oct_own_idx = nullptr /* n_in_tree */;
    uint32_t* slot_tri = nullptr, *slot_pos = nullptr;   // per device slot (tests)
    uint32_t n_nodes = 0, n_in_tree = 0, n_list_slots = 0, n_slots_total = 0, n_sup_records = 0, n_clusters = 0, max_depth = 0, max_own = 0;
    uint32_t has_groups = 0, inline_leaves = 0, bounds_plain = 1, n_suspects = 0, all_inside_root = 0;   // all_inside_root: no triangle of the tree pokes out of the root box
    double scene_magnitude = 0, pad = 0;
    double ms_upload = 0, ms_octree = 0, ms_index = 0;   // GPU time of the three stages (HIP events on the build stream)
};

// Builds the scene on the current HIP device from `tris` (HOST array, uploaded here through pinned staging).  enable_cull = !RRT_FLAG_NO_CULL.
// Throws rrt::Error (RRT_ERR_DEPTH when the octree is deeper than RRT_MAX_OCTREE_DEPTH) or HipBuildFail.  The caller owns out.scene_alloc (hipFree).
struct HipBuildFail { int hip_error; const char* what; };
// The triangles either as the model's array (Triangle records, SceneData.triangles) or as the caller's own arrays (rrt_raytracer_create_from_arrays:
// pos / uv / nrm [n][3][3] doubles, mat [n]) -- those are uploaded as they are and packed into Triangle records on the device.
struct TriSource { const Triangle* tris = nullptr; const double* pos = nullptr; const double* uv = nullptr; const double* nrm = nullptr; const uint32_t* mat = nullptr; };
// `after_upload` (may be empty) is called once the triangles have been handed to the staging ring, before the octree build: the caller's other uploads
// (textures) can start there, beside the build, without competing with the triangles for the ring.
void gpu_build_scene(const TriSource& src, uint32_t n_tris, const Box& root, bool enable_cull, const double origin[3], void* stream, GpuScene& out,
                     const std::function<void()>& after_upload = {});

// Pinned-staging upload of a host buffer (pageable or not) to device memory on `stream`: worker threads fill a ring of page-locked chunks while
// the DMA engine drains it.  Returns after the last chunk has been ENQUEUED and copied out of `src` (src may be freed; dst is ready after a stream sync).
void staged_upload(void* dst, const void* src, size_t bytes, void* stream);
void staged_upload_warm();    // allocates the current device's ring and set-up stream (called from the warm-up thread so that the first upload does not pay for it)
// Device memory -> pageable host memory through the same ring, blocking (chunk DMAs run ahead of the copies out of the ring).
void staged_download(void* dst, const void* src_dev, size_t bytes, void* stream);
// The current device's shared non-blocking stream (hipStream_t) for set-up work and blocking host-framebuffer renders: creating a stream costs
// milliseconds, the reference's whole frame takes less.  Owned by the library; never destroyed.
void* setup_stream();
void* upload_stream();    // a second one, for uploads that run beside work on setup_stream() (textures beside the scene build)

}  // namespace rrt
