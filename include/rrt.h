/*
 * rrt.h -- C ABI of the MI355X-native per-pixel hot path of conor722/rust-ray-tracer.
 *
 * The reference has no FFI; its natural seam is `Scene::draw_scene(&mut self, rt: RayTracer)`
 * (src/scene/engine.rs:186) called from `main` (src/main.rs:68-74), one level above
 * `RayTracer::get_ray_colour` (src/scene/raytracer.rs:29).  This header is what a Rust
 * `extern "C"` block for that seam binds (see INTEGRATION.md for the binding a maintainer adds).
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every function returns RRT_OK (0) or a
 * negative rrt_status; nothing aborts or unwinds across the boundary (the reference panics instead:
 * main.rs:24,28; utils.rs:61,85,170,184,193,222,275,347-349).  All input pointers are borrowed for the
 * duration of the call only; the library copies what it keeps.  One caller thread per handle.
 * Host-side set-up work (parsing, texture decode, staging copies) runs on a process-wide pool of worker threads that the library creates on demand and keeps
 * (RRT_HOST_THREADS caps a stage's share of it; with LOCAL_WORLD_SIZE / WORLD_SIZE set, the hardware threads are divided among the ranks of the node); the
 * workers live until the process ends, so the library must not be unloaded (dlclose) while the process runs.  Tracing itself uses no host threads; rrt_render into a pageable buffer copies the frame out of the staging ring on the pool.
 *
 * There is NO CPU fallback in this library: every compute entry point runs hand-written HIP kernels on
 * gfx950 and fails with RRT_ERR_NO_DEVICE / RRT_ERR_HIP when no GPU is usable.
 */
#ifndef RRT_H
#define RRT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RRT_OK = 0,
    RRT_ERR_INVALID_ARG = -1,
    RRT_ERR_HIP = -2,          /* a HIP runtime call failed; rrt_last_error_detail() has the text */
    RRT_ERR_OOM = -3,
    RRT_ERR_IO = -4,           /* "Could not read file" (main.rs:28, utils.rs:171, 346-347) */
    RRT_ERR_PARSE = -5,        /* any `expect`/`unwrap` failure of utils.rs (bad number, missing vertex, ...) */
    RRT_ERR_DEPTH = -6,        /* octree deeper than RRT_MAX_OCTREE_DEPTH (e.g. duplicate triangles, octree.rs:79-92) */
    RRT_ERR_NO_DEVICE = -7,
    RRT_ERR_UNSUPPORTED = -8   /* e.g. a texture format the build-owned decoder does not read */
} rrt_status;

#define RRT_MAX_OCTREE_DEPTH 40

/* == Vector3d, src/scene/engine.rs:9-14 */
typedef struct { double x, y, z; } rrt_vec3;

/* == Light, src/scene/entities.rs:5-9.  kind: 0 Ambient{intensity}; 1 Point{intensity, position=v};
 * 2 Directional{intensity, direction=v} */
typedef struct { uint32_t kind; uint32_t _pad; double intensity; rrt_vec3 v; } rrt_light;

/* == Material, src/scene/material.rs:11-22 (name dropped).  tex/bump index rrt textures; bump = -1 => None */
typedef struct { rrt_vec3 ka, kd, ks; double ns, kr; int32_t tex, bump; } rrt_material;

/* == Texture, src/scene/entities.rs:86-91: RGB8, row-major, index = width*y + x (raytracer.rs:55) */
typedef struct { const uint8_t *rgb; uint32_t width, height; } rrt_texture;

/* RRT_FLAG_NO_CULL: walk every node's triangle list in full, in list order, exactly as ray.rs:119-129 does (no cluster boxes).
 * Default (0): the lists are indexed by padded cluster boxes that skip triangles a ray cannot reach.  Results are identical for every ray that does
 * not lie IN a triangle's plane to within rounding noise (where the reference's own Moller-Trumbore answer is noise): DESIGN.md section 4 gives the
 * bound.  Rays from the raytracer's origin -- every primary ray -- are guarded against that case too (rrt_stats.origin_plane_triangles); secondary
 * rays are not.  Tests compare the two modes bit for bit on every config and on 10^6 constructed near-coplanar rays. */
#define RRT_FLAG_NO_CULL 1u
/* The index boxes are tested either by every ray against one box at a time (LANE filter) or by 64 boxes at a time against the wave's ray
 * bundle (BUNDLE filter; faster on coherent rays, slower on scattered ones).  Both give the same pixels.  By default the FIRST frame of a frame
 * size runs the variant a measured rule picks -- bundle filter above ~1200 primary rays per triangle of the scene, lane filter below
 * (profiles/r03_variant_sweep.json: right on 17 of 18 scene x size points, 5 % off on the other) -- so a host that renders one frame per run, as the
 * reference does, pays nothing for the choice; a SECOND frame of the same size is first rendered with every variant (a one-off stream
 * synchronisation), the fastest being kept for that size; these flags force one. */
#define RRT_FLAG_LANE_FILTER 2u
#define RRT_FLAG_BUNDLE_FILTER 4u
/* Both of the above walk the octree node-coherently (one node per wave step, records in scalar registers): right for rays that share nodes.
 * RAY_WALK lets every ray of a wave visit its own node in every step (records through vector memory): right for scattered rays (large soups,
 * mirror bounces).  Same pixels again; the default measures all three on the second frame of a size. */
#define RRT_FLAG_RAY_WALK 8u
/* Set-up.  By default rrt_raytracer_create builds everything the kernels read ON THE GPU from the uploaded triangle array: the octree exactly as
 * Octree::push_triangle builds it one triangle at a time (octree.rs:41-241; level-parallel and order-exact, csrc/scene_build.hip), this build's index
 * over the own lists, and the device records.  HOST_SETUP does the same on the host's cores (csrc/octree.cpp, clusters.cpp) and uploads the result:
 * the two paths produce the same bytes in HBM (tests/test_gpu_build.py); the flag exists for that check and for A/B timing. */
#define RRT_FLAG_HOST_SETUP 16u

/* Render constants that the reference hard-codes; NULL => these defaults. */
typedef struct {
    double surface_offset;            /* 1e-4, raytracer.rs:17 */
    uint32_t max_reflection_depth;    /* 5,    raytracer.rs:20 (<= 8 supported) */
    uint32_t flags;                   /* 0, or RRT_FLAG_* */
    double vp_w, vp_h, vp_d;          /* 1,1,1 Viewport::default, engine.rs:113-119 */
} rrt_options;

typedef struct {
    uint32_t n_tris, n_tris_in_tree;  /* triangles outside the root AABB are silently dropped, octree.rs:71-73 */
    uint32_t n_nodes, max_depth;      /* max_depth counts the root as 1 */
    uint32_t n_mats, n_tex;
    uint32_t root_own_count, max_own_count;
} rrt_model_info;

typedef struct {
    double   kernel_ms;               /* HIP-event time of the last render's trace kernel on its stream */
    uint32_t width, height;
    uint64_t rays_primary;            /* 4 * pixels actually traced */
    uint64_t scene_bytes;             /* bytes resident in HBM for this raytracer (geometry+octree+textures) */
    uint32_t filter_variant;          /* 0 = LANE filter, 1 = BUNDLE filter, 2 = RAY walk (forced, or measured on the first frame of this size) */
    uint32_t origin_plane_triangles;  /* triangles whose plane contains the raytracer's origin to rounding distance: rays from the origin that lie
                                       * in such a plane run with the index filters off (exactness guard); 0 for ordinary scenes */
    /* The index's exactness band.  filter_pad = absolute padding of every index box (2^-15 of the scene magnitude).  A (ray,
     * triangle) pair can only be treated differently from the reference-order walk if the ray's direction is within alpha of the triangle's plane AND
     * its origin within delta of it, alpha = filter_alpha_unit * R / sin(phi), delta = filter_delta_unit * R^2 / sin(phi)^2 (leading term), with
     * R = |origin - v1| + the longer edge in units of the scene magnitude and phi the angle between the edges: the `unit` values are those of a
     * right-angled triangle at the scene's magnitude (about 2e-9 and 1e-7 of it).  profiles/r03_band_counts.json: no pair of the five BASELINE frames
     * that the reference accepts lies inside the band. */
    double filter_pad, filter_alpha_unit, filter_delta_unit;
} rrt_stats;

/* ------------------------------------------------------------------ model = SceneData (scenedata.rs:5-13), host side */
typedef struct rrt_model rrt_model;

/* parse_obj_file_lines (utils.rs:139-213) on the file at obj_path (read as main.rs:28); "mtllib"/texture names
 * resolve relative to the .obj's directory (the reference resolves against the cwd; it is run from its root).
 * root = {min_x,max_x,min_y,max_y,min_z,max_z}, NULL => Octree::new(-20,20,-20,20,-20,20) (utils.rs:145).
 * The reference pushes every triangle into the octree as it parses (utils.rs:192-198).  Here the tree is built from the finished triangle list, in
 * the same order, WHERE IT IS NEEDED: on the GPU by rrt_raytracer_create, or on the host the first time rrt_model_get_info / rrt_model_get_octree
 * (or a RRT_FLAG_HOST_SETUP raytracer) asks for it.  RRT_ERR_DEPTH is therefore reported by those calls, not by the two loaders. */
int rrt_model_load_obj(const char *obj_path, const double *root, rrt_model **out);

/* For a host that already parsed the scene (the Rust host holds SceneData.triangles in push order):
 * pos/uv/nrm are [n][3][3] doubles (v1,v2,v3; uv z ignored), mat[n] indexes mats.  Textures are copied. */
int rrt_model_from_arrays(uint32_t n_tris, const double *pos, const double *uv, const double *nrm, const uint32_t *mat,
                          uint32_t n_mats, const rrt_material *mats, uint32_t n_tex, const rrt_texture *tex,
                          const double *root, rrt_model **out);
void rrt_model_destroy(rrt_model *m);

int rrt_model_get_info(const rrt_model *m, rrt_model_info *out);
int rrt_model_get_triangles(const rrt_model *m, double *pos, double *uv, double *nrm, uint32_t *mat);   /* any may be NULL */
int rrt_model_get_materials(const rrt_model *m, rrt_material *out);
int rrt_model_get_texture(const rrt_model *m, uint32_t index, rrt_texture *out);                        /* borrowed view */
/* The octree exactly as octree.rs:41-241 builds it, flattened: aabb[n_nodes][6] = min xyz,max xyz;
 * first_child (0 = leaf, else 8 consecutive ids, octree.rs:226-238); tri_count (octree.rs:75);
 * own_off[n_nodes+1], own_idx[] = each node's `triangles` Vec in insertion order. */
int rrt_model_get_octree(const rrt_model *m, double *aabb, uint32_t *first_child, uint32_t *tri_count,
                         uint32_t *own_off, uint32_t *own_idx);

/* Build-owned JPEG / PNG decode to RGB8 (stands where `image::ImageReader::open().decode()` does, utils.rs:345-368).  JPEG: 8-bit Huffman,
 * sequential or progressive, 4:4:4 / 4:2:2 / 4:2:0, any number of scans, restart markers -- bit-equal to libjpeg (islow IDCT, fancy upsampling);
 * PNG: 8-bit RGB or palette of 1-8 bits, Adam7-interlaced or not; BMP: uncompressed 24-bit or palette; TGA (by file name): 24-bit, plain or run-length coded.
 * Anything else, and anything that does not decode to 3 bytes per pixel (greyscale, alpha -- the reference walks
 * `as_bytes().chunks(3)` whatever the colour type), is RRT_ERR_UNSUPPORTED.  *rgb is malloc'd; free with rrt_free. */
int rrt_decode_image_file(const char *path, uint8_t **rgb, uint32_t *width, uint32_t *height);
void rrt_free(void *p);

/* ------------------------------------------------------------------ raytracer = RayTracer{scene_data,lights,origin} (raytracer.rs:22-26) on one GPU */
typedef struct rrt_raytracer rrt_raytracer;

/* Uploads the scene once to HBM of HIP device `device`.  opt may be NULL. */
int rrt_raytracer_create(const rrt_model *m, const rrt_light *lights, uint32_t n_lights, rrt_vec3 origin,
                         const rrt_options *opt, int device, rrt_raytracer **out);
/* The same raytracer straight from the host's own arrays (arguments as rrt_model_from_arrays, then as rrt_raytracer_create): what a host that has
 * parsed the scene itself -- the Rust host holds SceneData.triangles -- calls when it needs no rrt_model.  The arrays are uploaded from where they lie
 * and packed into triangle records on the device; the library keeps no host copy of the scene.  Same scene in HBM, same frames; the loaders' copy
 * (15 ms of a million triangles' 54 ms first frame) is not made.  RRT_FLAG_HOST_SETUP is refused here (RRT_ERR_UNSUPPORTED). */
int rrt_raytracer_create_from_arrays(uint32_t n_tris, const double *pos, const double *uv, const double *nrm, const uint32_t *mat,
                                     uint32_t n_mats, const rrt_material *mats, uint32_t n_tex, const rrt_texture *tex, const double *root,
                                     const rrt_light *lights, uint32_t n_lights, rrt_vec3 origin, const rrt_options *opt, int device,
                                     rrt_raytracer **out);
void rrt_raytracer_destroy(rrt_raytracer *rt);

/* Scene::draw_scene (engine.rs:186-255) + Canvas::put_pixel (engine.rs:146-158): fills out_fb[width*height]
 * (host memory), 0x00RRGGBB (entities.rs:32-36), row 0 = top; pixels the reference never writes (row 0, and for
 * odd sizes row 1 / the last column) are 0 as in Canvas::new (engine.rs:135).  Blocking. */
int rrt_render(rrt_raytracer *rt, uint32_t width, uint32_t height, uint32_t *out_fb);

/* Optional: page-lock a caller-owned framebuffer (the Rust host's Canvas.buffer: Vec<u32>, engine.rs:127) so that rrt_render copies the
 * frame into it with one asynchronous DMA.  Without it rrt_render stages through pinned memory of its own plus one host copy
 * (pipelined in row chunks).  Unregister before the buffer is freed or reallocated. */
int rrt_host_buffer_register(void *ptr, size_t bytes);
int rrt_host_buffer_unregister(void *ptr);

/* Same, framebuffer in device memory of rt's device; enqueued on `stream` (hipStream_t, NULL = default), not synchronised. */
int rrt_render_device(rrt_raytracer *rt, uint32_t width, uint32_t height, void *d_fb, void *stream);

/* Screen-tile partition for N GPUs (one process per GPU): the frame is cut into 8x8-pixel tiles, tile k (row-major)
 * belongs to rank k % world.  Renders this rank's tiles into d_tiles[rrt_tiles_per_rank][64] (tile-major, device).
 * After a gather (or all-gather) of the per-rank buffers (RCCL, done by the caller), rrt_detile_device turns
 * d_gathered[world][tiles_per_rank][64] into the row-major framebuffer d_fb[width*height]. */
uint32_t rrt_tiles_per_rank(uint32_t width, uint32_t height, uint32_t world);
int rrt_render_tiles_device(rrt_raytracer *rt, uint32_t width, uint32_t height, uint32_t rank, uint32_t world,
                            void *d_tiles, void *stream);
int rrt_detile_device(rrt_raytracer *rt, uint32_t width, uint32_t height, uint32_t world, const void *d_gathered,
                      void *d_fb, void *stream);

/* ------------------------------------------------------------------ the N GPUs of one node
 * The gather is inside the library: grouped RCCL point-to-point sends to rank 0 (every peer on its own xGMI link) and de-tiling on rank 0's GPU.
 * rrt_multi_create: ONE process drives all GPUs -- the Rust host creates one raytracer per device (rrt_raytracer_create(..., device = i, ...)) and
 *   hands them over; rts[0]'s device receives the frame.  Uses ncclCommInitAll; RCCL is loaded on first use (dlopen), not at link time.
 * rrt_dist_create: ONE process per GPU (MPI / torch.distributed style): rank 0 calls rrt_dist_unique_id, the caller ships the 128 bytes to every rank
 *   by its own means, every rank calls rrt_dist_create (collective: ncclCommInitRank).
 * frames_in_flight (1..8) sizes a ring of slots, each with its own streams and buffers: rrt_multi_enqueue returns as soon as trace -> gather ->
 * de-tile of the frame are enqueued, frame i + 1 is traced while frame i is gathered.  rrt_render_multi is the blocking host-framebuffer form of
 * Scene::draw_scene (engine.rs:186): out_fb as in rrt_render; processes that do not hold rank 0 pass NULL.
 * RRT_MULTI_LOOPBACK (rrt_multi_create with n = 1): rank 0's own tiles travel through ncclSend/ncclRecv too -- a one-GPU test of the transport. */
typedef struct rrt_multi rrt_multi;
#define RRT_MULTI_LOOPBACK 1u
int rrt_multi_create(rrt_raytracer *const *rts, uint32_t n, uint32_t frames_in_flight, uint32_t flags, rrt_multi **out);
int rrt_dist_unique_id(void *out128);
int rrt_dist_create(rrt_raytracer *rt, uint32_t rank, uint32_t world, const void *unique_id128, uint32_t frames_in_flight, rrt_multi **out);
void rrt_multi_destroy(rrt_multi *g);
int rrt_multi_enqueue(rrt_multi *g, uint32_t width, uint32_t height, void *d_fb /* device memory of rank 0's GPU; NULL elsewhere */);
int rrt_multi_sync(rrt_multi *g);
int rrt_render_multi(rrt_multi *g, uint32_t width, uint32_t height, uint32_t *out_fb);
/* HIP-event time, on rank 0's stream, from "rank 0's own tiles are traced" to "frame de-tiled" of the last enqueued frame: the wait for the slowest
 * peer + the gather + the de-tiling (-1 on processes without rank 0). */
int rrt_multi_last_gather_ms(rrt_multi *g, double *out_ms);

/* Scene::draw_scene with the reference's progressive display (engine.rs:196-253): the scene rows are traced in chunks of chunk_rows (0 = the
 * reference's 50) from y = -H/2 upward, i.e. from the BOTTOM of the canvas to the top (put_pixel, engine.rs:146-158); after each chunk its rows are
 * in out_fb (zero-initialised like Canvas::new, engine.rs:135) and on_update -- the stand-in for canvas.update(), engine.rs:253 -- is called on the
 * calling thread with the canvas rows [first_row, first_row + n_rows) that the chunk wrote (n_rows = 0 for a chunk that wrote none).  The finished
 * frame equals rrt_render's.  on_update may be NULL. */
typedef void (*rrt_update_fn)(void *user, const uint32_t *fb, uint32_t width, uint32_t height, uint32_t first_row, uint32_t n_rows);
int rrt_render_progressive(rrt_raytracer *rt, uint32_t width, uint32_t height, uint32_t *out_fb, uint32_t chunk_rows,
                           rrt_update_fn on_update, void *user);

/* Batched RayTracer::get_ray_colour (raytracer.rs:29): n rays, origins/dirs [n][3] host doubles -> colours[n] 0x00RRGGBB.
 * The two per-ray entry points take whatever rays the caller has, so unless a variant is forced they pick theirs by measurement too: the first batch of
 * at least 16384 rays is timed with all three variants on its first 65536 rays and the fastest is kept for later calls (smaller batches before that
 * run the frame variant).  rrt_last_stats after a per-ray call: kernel_ms and filter_variant of that launch, width = n, height = 1. */
int rrt_get_ray_colours(rrt_raytracer *rt, uint32_t n, const double *origins, const double *dirs, uint32_t *colours);

/* Batched Ray::intersect_with_octant_with_max_t(octree, 0, max_t) (ray.rs:104-168): hit[n] 0/1, t,u,v [n], tri[n] = index in
 * push order.  max_t may be NULL (= +inf, ray.rs:96-102). */
int rrt_intersect_rays(rrt_raytracer *rt, uint32_t n, const double *origins, const double *dirs, const double *max_t,
                       uint8_t *hit, double *t, double *u, double *v, uint32_t *tri);

/* The octree of a raytracer whose set-up ran on the GPU (the default), read back from its device: same layout as rrt_model_get_octree; info (may be
 * NULL) as rrt_model_get_info.  Any pointer may be NULL.  RRT_ERR_UNSUPPORTED for a RRT_FLAG_HOST_SETUP raytracer (ask the model). */
int rrt_raytracer_get_octree(const rrt_raytracer *rt, rrt_model_info *info, double *aabb, uint32_t *first_child, uint32_t *tri_count,
                             uint32_t *own_off, uint32_t *own_idx);
/* Test / developer introspection: the bytes of one scene buffer in HBM.  out == NULL asks for the size only. */
enum { RRT_BUF_NODES = 0, RRT_BUF_GEOM, RRT_BUF_ATTR, RRT_BUF_SUPERS, RRT_BUF_CBOXES, RRT_BUF_CHILD_BOXES, RRT_BUF_TBOXES, RRT_BUF_SUSPECTS,
       RRT_BUF_OCT_BOX, RRT_BUF_OCT_FIRST_CHILD, RRT_BUF_OCT_TRI_COUNT, RRT_BUF_OCT_OWN_OFF, RRT_BUF_OCT_OWN_IDX, RRT_BUF_SLOT_TRI, RRT_BUF_SLOT_POS };
int rrt_raytracer_get_buffer(const rrt_raytracer *rt, uint32_t which, void *out, size_t capacity, size_t *bytes);

int rrt_last_stats(const rrt_raytracer *rt, rrt_stats *out);
/* Time of the set-up stages that run once per scene (the reference does all of them inside parse_obj_file_lines, utils.rs:139-213, before its one
 * frame): model side = file read, .obj/.mtl parse, texture decode; raytracer side = octree build (octree.rs:41-241), own-list index build, upload to
 * HBM.  With the default GPU set-up octree_ms and index_ms are HIP-event times on the build stream and upload_ms is the rest of
 * rrt_raytracer_create's wall time (pinned-staging uploads of triangles and textures, allocations, synchronisation); with RRT_FLAG_HOST_SETUP they
 * are host wall times (octree_ms: the model's host build).  hip_init_ms = bringing the device's HIP context up at the start of
 * rrt_raytracer_create (a one-off of the process, near 0 for every later raytracer); create_ms = wall time of the whole rrt_raytracer_create;
 * gpu_setup = 1.0 / 0.0.  Either handle may be NULL (its fields stay 0). */
typedef struct { double read_ms, parse_ms, texture_ms, octree_ms, index_ms, upload_ms, hip_init_ms, create_ms, gpu_setup; } rrt_setup_times;
int rrt_get_setup_times(const rrt_model *m, const rrt_raytracer *rt, rrt_setup_times *out);
int rrt_device_count(int *count);
const char *rrt_strerror(int status);
const char *rrt_last_error_detail(void);   /* thread-local text of the last failure */
const char *rrt_build_info(void);          /* offload arch, fp-contract mode */

#ifdef __cplusplus
}
#endif
#endif
