// scene_build.hip -- the once-per-scene set-up on the GPU: what the reference does inside its parser, one triangle at a time
// (src/file_management/utils.rs:192-198 -> Octree::push_triangle, src/collision/octree.rs:41-241), restated LEVEL-PARALLEL and order-exact, followed
// by this build's own-list index (clusters.cpp) and the device records the trace kernels read.  Every array it produces is byte-identical to what the
// host path (octree.cpp + clusters.cpp + the fill loops of api.cpp, RRT_FLAG_HOST_SETUP) produces for finite input; tests/test_gpu_build.py checks that.
//
// Why the insertion of octree.rs:54-108 is level-parallel.  At a node the arrivals, in push order, are a1 < a2 < ...:
//   * a1 finds an empty leaf and stays (octree.rs:77-78);
//   * a2 finds a leaf that holds a triangle: the node subdivides (octree.rs:79-80) -- so a node subdivides iff it has >= 2 arrivals, and the
//     subdivision is triggered by a2;
//   * a2 and every later arrival is classified against the 8 child boxes, which depend on the node's box only (octree.rs:121-241), and descends iff
//     it touches exactly one of them (octree.rs:88-104) -- independently of every other triangle;
//   * triangle_count = number of arrivals (octree.rs:75); the node's `triangles` Vec = the arrivals that stayed, in push order.
// So one level is: per node the two smallest arriving indices and the arrival count (atomics, wave-aggregated: results do not depend on the order
// they land in), subdivide the nodes with >= 2 arrivals, classify.  One insertion triggers at most one subdivision (the newcomer then lands in a
// fresh, empty child), so the reference's node ids follow from the subdivision events sorted by their trigger triangle: event number i (a prefix
// sum over a per-triangle "is a trigger" flag) creates the ids 1 + 8 i .. 8 + 8 i (octree.rs:226-238).  Own lists = a stable sort of the triangles by
// final node id (rocPRIM radix sort; the classification, commit and index kernels below are hand-written).
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "parallel.hpp"
#include "scene_build.hpp"

namespace rrt {
namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kBlock = 256;
constexpr uint32_t kClusterTris = 8, kSuperTris = 64, kGroupSupers = 8, kGroupThreshold = 24;   // clusters.cpp
constexpr double kPadFraction = 1.0 / 32768.0;

#define HB_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw HipBuildFail{(int)_e, #expr}; } while (0)

static_assert(sizeof(Triangle) == 224, "Triangle is uploaded as is: 27 doubles + the material index");

inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// ------------------------------------------------------------------------------------------------ octree build
struct Oct {
    const Triangle* tris; uint32_t n;
    const double* pos;                  // the caller's position array [n][3][3] when the triangle records are not packed yet (rrt_raytracer_create_from_arrays), else null
    double* tbox;                       // [n][6] Aabb::from_triangle (aabb.rs:25-47)
    uint32_t* cur;                      // [n] node (temporary id) the triangle arrives at on the current level, kNone once it has settled / was dropped
    uint32_t* own;                      // [n] node (temporary id) whose `triangles` Vec holds it, kNone = outside the root (octree.rs:71-73)
    uint32_t* trig;                     // [n] 1 = this triangle's insertion triggered a subdivision
    double* nbox;                       // [cap][6]
    uint32_t *first, *second, *cnt, *child_base;   // [cap] two smallest arrivals, arrival count (= triangle_count), first child (temporary id, 0 = leaf)
    uint32_t* block_parent;             // [cap / 8] parent of the k-th block of 8 children (temporary ids 1 + 8k .. 8 + 8k)
    uint32_t* ctr;                      // [0] blocks so far, [1] arrivals at the next level
    double rlo[3], rhi[3];
};

// Arrivals are recorded per node as (smallest index, count).  The upper levels have few nodes (the root: one; then 8, 64, 512): a million atomics
// on a handful of addresses serialise in the L2 (measured: 1.8 ms per kernel on the 1 M soup's levels 1-3), so every workgroup first merges its
// arrivals in LDS (ds_min / ds_add) and sends one atomic per node it touched; levels with more than kLdsNodes nodes go to memory directly (merged
// within the wave where neighbouring triangles share a node).  min and + commute: the result does not depend on the order anything lands in.
constexpr int kOctBlock = 1024;                 // threads per workgroup of the three per-triangle octree kernels
constexpr uint32_t kLdsNodes = 1024;            // a level of at most this many nodes is merged in LDS
constexpr unsigned kOctMaxGrid = 128;           // workgroups (grid-stride over the triangles): few, so that few partial results reach memory

__device__ __forceinline__ void arrive_direct(uint32_t* mn, uint32_t* count, uint32_t node, uint32_t t, bool active) {
    const int lane = threadIdx.x & 63;          // lanes hold increasing triangle indices: the lowest lane of a group has its minimum
    bool pending = active;
    for (int it = 0; it < 4; ++it) {
        const unsigned long long pm = __ballot(pending);
        if (!pm) break;
        const int leader = __ffsll((long long)pm) - 1;
        const uint32_t n0 = (uint32_t)__shfl((int)node, leader);
        const bool mine = pending && node == n0;
        const unsigned long long same = __ballot(mine);
        if (mine) {
            if (lane == leader) { atomicMin(&mn[n0], t); if (count) atomicAdd(&count[n0], (uint32_t)__popcll(same)); }
            pending = false;
        }
    }
    if (pending) { atomicMin(&mn[node], t); if (count) atomicAdd(&count[node], 1u); }
}
struct LdsMerge {                               // one level's nodes [base, base + count) merged per workgroup
    uint32_t* s_min; uint32_t* s_cnt; uint32_t base, count; bool on;
    __device__ __forceinline__ void begin(uint32_t* smin, uint32_t* scnt, uint32_t base_, uint32_t count_) {
        s_min = smin; s_cnt = scnt; base = base_; count = count_; on = count_ <= kLdsNodes;
        if (on) for (uint32_t k = threadIdx.x; k < count; k += blockDim.x) { s_min[k] = kNone; s_cnt[k] = 0u; }
        __syncthreads();
    }
    __device__ __forceinline__ void add(uint32_t* gmin, uint32_t* gcnt, uint32_t node, uint32_t t, bool active) {
        if (on) { if (active) { atomicMin(&s_min[node - base], t); atomicAdd(&s_cnt[node - base], 1u); } }
        else arrive_direct(gmin, gcnt, node, t, active);
    }
    __device__ __forceinline__ void flush(uint32_t* gmin, uint32_t* gcnt) {
        __syncthreads();
        if (on) for (uint32_t k = threadIdx.x; k < count; k += blockDim.x) if (s_cnt[k]) { atomicMin(&gmin[base + k], s_min[k]); if (gcnt) atomicAdd(&gcnt[base + k], s_cnt[k]); }
    }
};
__device__ __forceinline__ void count_block(uint32_t* ctr, uint32_t* s_total, uint32_t mine) {   // (s_total zeroed and synchronised by the caller)
    if (mine) atomicAdd(s_total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && *s_total) atomicAdd(ctr, *s_total);
}

__global__ void __launch_bounds__(kOctBlock) k_oct_init(Oct S) {
    __shared__ uint32_t s_min[1], s_cnt[1], s_total;
    if (threadIdx.x == 0) s_total = 0u;
    LdsMerge M; M.begin(s_min, s_cnt, 0u, 1u);
    uint32_t n_act = 0;
    for (uint32_t t = blockIdx.x * kOctBlock + threadIdx.x; t < S.n; t += gridDim.x * kOctBlock) {
        double v[9];
        if (S.pos) { for (int k = 0; k < 9; k++) v[k] = S.pos[9 * (size_t)t + k]; }
        else { const Triangle& T = S.tris[t]; v[0] = T.v1.x; v[1] = T.v1.y; v[2] = T.v1.z; v[3] = T.v2.x; v[4] = T.v2.y; v[5] = T.v2.z; v[6] = T.v3.x; v[7] = T.v3.y; v[8] = T.v3.z; }
        // Aabb::from_triangle, aabb.rs:25-47 (f64::min/max == fmin/fmax)
        const double lo[3] = {fmin(v[0], fmin(v[3], v[6])), fmin(v[1], fmin(v[4], v[7])), fmin(v[2], fmin(v[5], v[8]))};
        const double hi[3] = {fmax(v[0], fmax(v[3], v[6])), fmax(v[1], fmax(v[4], v[7])), fmax(v[2], fmax(v[5], v[8]))};
        bool touch = true, inside = true;                        // Aabb::intersects, aabb.rs:49-60 (inclusive)
        for (int k = 0; k < 3; k++) { S.tbox[6 * (size_t)t + k] = lo[k]; S.tbox[6 * (size_t)t + 3 + k] = hi[k]; if (hi[k] < S.rlo[k] || lo[k] > S.rhi[k]) touch = false; if (!(lo[k] >= S.rlo[k] && hi[k] <= S.rhi[k])) inside = false; }
        if (touch && !inside) S.ctr[2] = 1u;                      // a triangle of the tree pokes out of the root box (any number of writers, one value)
        S.cur[t] = touch ? 0u : kNone; S.own[t] = kNone; S.trig[t] = 0u;
        if (touch) { atomicMin(&s_min[0], t); n_act++; }
    }
    if (n_act) atomicAdd(&s_cnt[0], n_act);
    M.flush(S.first, S.cnt);
    count_block(&S.ctr[1], &s_total, n_act);
}

__global__ void __launch_bounds__(kOctBlock) k_oct_second(Oct S, uint32_t lb, uint32_t le) {   // second-smallest arrival of every node of the level [lb, le)
    __shared__ uint32_t s_min[kLdsNodes], s_cnt[kLdsNodes];
    LdsMerge M; M.begin(s_min, s_cnt, lb, le - lb);
    const uint32_t n_round = (S.n + 63u) & ~63u;                 // whole waves stay in the loop (arrive_direct uses wave ballots)
    for (uint32_t t = blockIdx.x * kOctBlock + threadIdx.x; t < n_round; t += gridDim.x * kOctBlock) {
        const uint32_t m = t < S.n ? S.cur[t] : kNone;
        const bool cand = m != kNone && S.first[m] != t;
        M.add(S.second, nullptr, cand ? m : lb, t, cand);
    }
    M.flush(S.second, nullptr);
}

// children in the reference's order BBL,BFL,BFR,BBR,TBL,TFL,TFR,TBR (octree.rs:216-225): (x-half, y-half, z-half), 0 = [lo,mid], 1 = [mid,hi]
__constant__ int c_half[8][3] = {{0, 0, 0}, {0, 0, 1}, {1, 0, 1}, {1, 0, 0}, {0, 1, 0}, {0, 1, 1}, {1, 1, 1}, {1, 1, 0}};
__constant__ uint32_t c_child_of[8] = {0, 1, 4, 5, 3, 2, 7, 6};   // (x-half << 2 | y-half << 1 | z-half) -> child number

__global__ void __launch_bounds__(kBlock) k_oct_split(Oct S, uint32_t lb, uint32_t le) {   // Octree::subdivide, octree.rs:121-241, for every node of the level with >= 2 arrivals
    const uint32_t m = lb + blockIdx.x * kBlock + threadIdx.x;
    if (m == lb) S.ctr[1] = 0u;
    if (m >= le || S.cnt[m] < 2u) return;
    const uint32_t k = atomicAdd(&S.ctr[0], 1u), base = 1u + 8u * k;
    S.child_base[m] = base; S.block_parent[k] = m; S.trig[S.second[m]] = 1u;
    double lo[3], hi[3], mid[3];
    for (int a = 0; a < 3; a++) { lo[a] = S.nbox[6 * (size_t)m + a]; hi[a] = S.nbox[6 * (size_t)m + 3 + a]; mid[a] = lo[a] + (hi[a] - lo[a]) / 2.0; }   // octree.rs:136-146
    for (int c = 0; c < 8; c++) {
        const size_t q = base + c;
        for (int a = 0; a < 3; a++) { S.nbox[6 * q + a] = c_half[c][a] ? mid[a] : lo[a]; S.nbox[6 * q + 3 + a] = c_half[c][a] ? hi[a] : mid[a]; }
        S.first[q] = kNone; S.second[q] = kNone; S.cnt[q] = 0u; S.child_base[q] = 0u;
    }
}

__global__ void __launch_bounds__(kOctBlock) k_oct_descend(Oct S, uint32_t next_begin, uint32_t blocks_before) {   // push_at_octant, octree.rs:77-104, for every arrival of the level
    __shared__ uint32_t s_min[kLdsNodes], s_cnt[kLdsNodes], s_total;
    if (threadIdx.x == 0) s_total = 0u;
    LdsMerge M; M.begin(s_min, s_cnt, next_begin, 8u * (S.ctr[0] - blocks_before));   // the level below: the children k_oct_split has just created
    const uint32_t n_round = (S.n + 63u) & ~63u;
    uint32_t n_go = 0;
    for (uint32_t t = blockIdx.x * kOctBlock + threadIdx.x; t < n_round; t += gridDim.x * kOctBlock) {
        const uint32_t m = t < S.n ? S.cur[t] : kNone;
        bool go = false; uint32_t child = next_begin;
        if (m != kNone) {
            if (S.first[m] == t) { S.own[t] = m; S.cur[t] = kNone; }                    // the first arrival found an empty leaf (octree.rs:77-78)
            else {
                // The reference tests the newcomer's box against each child box (aabb.rs:49-60).  It touches the node's own box (checked at the root, true by
                // induction below), and the children are that box cut at its mid planes, so per axis "touches the lower half" is exactly !(lo > mid) and
                // "touches the upper half" !(hi < mid): 6 comparisons, the count of touched children is the product (octree.cpp has the same form).
                const double* tb = &S.tbox[6 * (size_t)t]; const double* nb = &S.nbox[6 * (size_t)m];
                bool l[3], h[3];
                for (int a = 0; a < 3; a++) { const double mid = nb[a] + (nb[3 + a] - nb[a]) / 2.0; l[a] = !(tb[a] > mid); h[a] = !(tb[3 + a] < mid); }
                const int n_touch = ((int)l[0] + (int)h[0]) * ((int)l[1] + (int)h[1]) * ((int)l[2] + (int)h[2]);
                if (n_touch == 1) { child = S.child_base[m] + c_child_of[((uint32_t)h[0] << 2) | ((uint32_t)h[1] << 1) | (uint32_t)h[2]]; S.cur[t] = child; go = true; }
                else { S.own[t] = m; S.cur[t] = kNone; }                                // octree.rs:90-92,102-104
            }
        }
        M.add(S.first, S.cnt, child, t, go);
        n_go += go ? 1u : 0u;
    }
    M.flush(S.first, S.cnt);
    count_block(&S.ctr[1], &s_total, n_go);
}

// ---- temporary ids -> the reference's ids
struct Remap {
    uint32_t n_nodes, n_blocks, n;
    const uint32_t *second, *cnt, *child_base, *block_parent, *rank, *own;
    const double* nbox;
    uint32_t* newblock;                 // [n_blocks]
    uint32_t* tmp2final;                // [n_nodes]
    double* fbox; uint32_t *ffc, *ftc;  // final arrays
    uint32_t *key, *val, *own_count;    // per triangle: final node (n_nodes = dropped), index; per final node (+1): own-list length
};
__global__ void __launch_bounds__(kBlock) k_newblock(Remap R) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k < R.n_blocks) R.newblock[k] = R.rank[R.second[R.block_parent[k]]];
}
__device__ __forceinline__ uint32_t final_id(const uint32_t* newblock, uint32_t m) { return m == 0u ? 0u : 1u + 8u * newblock[(m - 1u) >> 3] + ((m - 1u) & 7u); }
__global__ void __launch_bounds__(kBlock) k_remap_nodes(Remap R) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= R.n_nodes) return;
    const uint32_t F = final_id(R.newblock, m);
    R.tmp2final[m] = F;
    for (int a = 0; a < 6; a++) R.fbox[6 * (size_t)F + a] = R.nbox[6 * (size_t)m + a];
    R.ffc[F] = R.child_base[m] ? 1u + 8u * R.newblock[(R.child_base[m] - 1u) >> 3] : 0u;
    R.ftc[F] = R.cnt[m];
}
__global__ void __launch_bounds__(kBlock) k_tri_keys(Remap R) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= R.n) return;
    const uint32_t o = R.own[t], F = o == kNone ? R.n_nodes : final_id(R.newblock, o);
    R.key[t] = F; R.val[t] = t;
    atomicAdd(&R.own_count[F], 1u);
}

// ------------------------------------------------------------------------------------------------ own-list index (clusters.cpp on the device)
__device__ __forceinline__ float round_down_f(double x) { float f = (float)x; if ((double)f > x) f = nextafterf(f, -FLT_MAX); return isfinite(f) ? f : -FLT_MAX; }
__device__ __forceinline__ float round_up_f(double x) { float f = (float)x; if ((double)f < x) f = nextafterf(f, FLT_MAX); return isfinite(f) ? f : FLT_MAX; }
__device__ __forceinline__ void to_centre_half(float* lo, float* hi) {   // clusters.cpp: to_centre_half
    for (int k = 0; k < 3; k++) {
        const double L = lo[k], H = hi[k];
        float c, h;
        if (L > H) { c = 0.0f; h = -FLT_MAX; }
        else if (L <= -FLT_MAX || H >= FLT_MAX) { c = 0.0f; h = FLT_MAX; }
        else {
            c = (float)((L + H) * 0.5);
            h = round_up_f(fmax((double)c - L, H - (double)c));
            h = nextafterf(h, FLT_MAX);
            if (!isfinite(c) || !isfinite(h)) { c = 0.0f; h = FLT_MAX; }
        }
        lo[k] = c; hi[k] = h;
    }
}

struct Idx {
    uint32_t n_nodes, n_in, enable_cull, inline_leaves;
    double pad;
    const uint32_t *ffc, *ftc, *own_off, *own_idx, *skey;          // octree (final ids); skey[i] = node of own_idx[i]
    const double* tbox;
    const Triangle* tris;
    uint32_t *a_slots, *a_sup, *a_leaf;                            // per node: slots (multiple of 8), super-cluster records, "is an inline leaf"
    uint32_t *slot_base, *sup_base, *leaf_rank;                    // their exclusive prefix sums (n_nodes + 1)
    uint32_t* flags;                                               // [0] has_groups, [1] max own count, [2] bounds_plain violated, [3] suspects found
    uint32_t *perm_a, *perm_b; double *cen_a, *cen_b;              // per own-list entry: position in the node's list, centroid; double-buffered
    uint32_t *slot_tri, *slot_pos, *cluster_node;
    float* cl_lohi;                                                // [clusters][6] padded f32 (lo, hi) before the centre/half form
    unsigned long long* nb;                                        // [n_nodes][6] own / subtree bounds as order-preserving integers (3 lo, 3 hi)
    uint32_t n_list_slots, n_slots_total;
    DevSuper* supers; DevClusterBox *cboxes, *tboxes, *child_boxes;
    DevNode* nodes; DevTriGeom* geom; DevTriAttr* attr;
};

__device__ __forceinline__ unsigned long long ord_f64(double x) { const unsigned long long b = (unsigned long long)__double_as_longlong(x); return (b >> 63) ? ~b : (b | 0x8000000000000000ull); }
__device__ __forceinline__ double unord_f64(unsigned long long k) { return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k)); }

__global__ void __launch_bounds__(kBlock) k_idx_sizes(Idx X) {
    const uint32_t F = blockIdx.x * kBlock + threadIdx.x;
    if (F > X.n_nodes) return;
    if (F == X.n_nodes) { X.a_slots[F] = 0; X.a_sup[F] = 0; X.a_leaf[F] = 0; return; }
    const uint32_t n = X.own_off[F + 1] - X.own_off[F];
    const uint32_t n_sup = (n + kSuperTris - 1) / kSuperTris;
    const bool grouped = X.enable_cull && n_sup > kGroupThreshold;
    X.a_slots[F] = (n + 7u) & ~7u;
    X.a_sup[F] = n_sup + (grouped ? (n_sup + kGroupSupers - 1) / kGroupSupers : 0u);
    X.a_leaf[F] = (X.inline_leaves && F >= 1u && X.ffc[F] == 0u && n == 1u) ? 1u : 0u;
    if (grouped) X.flags[0] = 1u;
    if (n > 8u) atomicMax(&X.flags[1], n);
    for (int a = 0; a < 3; a++) { X.nb[6 * (size_t)F + a] = ord_f64(DBL_MAX); X.nb[6 * (size_t)F + 3 + a] = ord_f64(-DBL_MAX); }
}

__global__ void __launch_bounds__(kBlock) k_idx_items(Idx X) {   // position in the list, centroid of the triangle's box (clusters.cpp: tri_box)
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= X.n_in) return;
    const uint32_t F = X.skey[i];
    X.perm_a[i] = i - X.own_off[F];
    const double* tb = &X.tbox[6 * (size_t)X.own_idx[i]];
    for (int a = 0; a < 3; a++) { double c = (tb[a] + tb[3 + a]) * 0.5; if (c != c) c = 0.0; X.cen_a[(size_t)a * X.n_in + i] = c; }
}

// One level of the median splits (clusters.cpp: split).  The recursion's SHAPE depends on the list length only: a range of `len` items with leaf
// size L splits into (ceil(len / L) / 2) * L items and the rest, so an entry finds the range it is in at split number `depth` of its own path by
// walking down from the whole list (first with L = 64 to the super-clusters, then with L = 8 inside each).  Which entries go left is decided by
// rank: the left part is the `left` smallest by (centroid on the widest axis, list position) -- a strict total order, so the parts are sets that do
// not depend on how they are found (the host uses nth_element).  Here every entry of a range is written to the position of its rank in the range.
__device__ __forceinline__ bool walk_to_range(uint32_t n, uint32_t p, uint32_t depth, uint32_t& b, uint32_t& e) {
    b = 0; e = n;
    uint32_t L = kSuperTris, steps = depth;
    for (;;) {
        const uint32_t len = e - b;
        if (len <= L) { if (L == kSuperTris) { L = kClusterTris; continue; } return false; }
        if (steps == 0) return true;
        const uint32_t parts = (len + L - 1) / L;
        uint32_t left = (parts / 2) * L;
        if (left == 0 || left >= len) left = len / 2;
        if (p < b + left) e = b + left; else b = b + left;
        --steps;
    }
}
// A wave's 64 consecutive entries belong to one range (long lists) or a few (short ones); the wave handles its distinct ranges one after the other,
// all 64 lanes working on the range in hand.  Bounds: the lanes stride over the range's entries and reduce (min and max are exact: any order gives the
// same bits).  Ranks: the range is taken in tiles of 64 entries, one per lane (key on the chosen axis, list position; the next tile's loads are issued
// before the current tile is compared), and every lane counts the tile's entries that sort before its own, lane q's entry being broadcast through
// scalar registers (v_readlane): no memory access inside the comparison loop (through LDS the loop waited ~100 cycles per entry: 1.4 ms for the root).
// The first form of this kernel let every lane walk its own range through the vector memory path: 3.4 ms for the 1 M soup's 10 961-triangle root
// list, 7.3 ms for all splits; a wave-uniform loop over scalar loads is latency-bound the same way (one s_load per entry, waited for at once).
// Centroids are stored axis by axis (cen[a * n_in + entry]): a tile's keys are one coalesced load.
__device__ __forceinline__ double wave_min_f64(double v) { for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = w < v ? w : v; } return v; }
__device__ __forceinline__ double wave_max_f64(double v) { for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = v < w ? w : v; } return v; }
__global__ void __launch_bounds__(kBlock) k_idx_split(Idx X, const uint32_t* __restrict__ ps, const double* __restrict__ cs, uint32_t* __restrict__ pd, double* __restrict__ cd, uint32_t depth) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x, lane = threadIdx.x & 63u;
    const size_t N = X.n_in;
    const bool valid = i < X.n_in;
    uint32_t b = 0, e = 0, gb = 0, ge = 0, dst = i, pi = 0;
    double ci[3] = {0.0, 0.0, 0.0};
    bool pending = false;
    if (valid) {
        const uint32_t F = X.skey[i], b0 = X.own_off[F], n = X.own_off[F + 1] - b0;
        pi = ps[i];
        for (int a = 0; a < 3; a++) ci[a] = cs[a * N + i];
        pending = n > kClusterTris && walk_to_range(n, i - b0, depth, b, e);
        gb = b0 + b; ge = b0 + e;
    }
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (!pm) break;
        const int leader = __ffsll((long long)pm) - 1;
        const uint32_t ugb = (uint32_t)__builtin_amdgcn_readlane((int)gb, leader), uge = (uint32_t)__builtin_amdgcn_readlane((int)ge, leader);
        const bool mine = pending && gb == ugb;
        // ---- centroid bounds of the range -> the widest axis (clusters.cpp: split)
        double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
#pragma unroll 8
        for (uint32_t j = ugb + lane; j < uge; j += 64u)
            for (int a = 0; a < 3; a++) { const double c = cs[a * N + j]; lo[a] = c < lo[a] ? c : lo[a]; hi[a] = hi[a] < c ? c : hi[a]; }
        int axis = 0;
        double best = 0.0;
        for (int a = 0; a < 3; a++) { const double l = wave_min_f64(lo[a]), h = wave_max_f64(hi[a]), ext = h - l; if (a == 0 || ext > best) { best = ext; axis = a; } }
        const double c0 = axis == 0 ? ci[0] : axis == 1 ? ci[1] : ci[2];
        // ---- rank of every entry of the range by (centroid on that axis, list position)
        const double* keys = cs + (size_t)axis * N;
        uint32_t r = 0;
        uint32_t jn = ugb + lane;
        const double kInfKey = __builtin_huge_val();                     // a tile's unused lanes: (+inf, 0xFFFFFFFF) sorts before nothing
        double nk = jn < uge ? keys[jn] : kInfKey; uint32_t np = jn < uge ? ps[jn] : kNone;
        for (uint32_t tile = ugb; tile < uge; tile += 64u) {
            const double k = nk; const uint32_t p = np;
            jn = tile + 64u + lane;
            nk = jn < uge ? keys[jn] : kInfKey; np = jn < uge ? ps[jn] : kNone;   // (the next tile's loads are in flight while this one is compared)
            const int klo = __double2loint(k), khi = __double2hiint(k);
#pragma unroll
            for (int q = 0; q < 64; q++) {                                // lane q's entry, broadcast through scalar registers: no memory in the loop
                const double cj = __hiloint2double(__builtin_amdgcn_readlane(khi, q), __builtin_amdgcn_readlane(klo, q));
                const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)p, q);
                r += (cj < c0 || (cj == c0 && pj < pi)) ? 1u : 0u;
            }
        }
        if (mine) { dst = ugb + r; pending = false; }
    }
    if (valid) { pd[dst] = pi; cd[dst] = ci[0]; cd[N + dst] = ci[1]; cd[2 * N + dst] = ci[2]; }
}

// After the splits: clusters are the runs of 8 list positions; inside a cluster the triangles keep list order (clusters.cpp: std::sort of the cluster).
__global__ void __launch_bounds__(kBlock) k_idx_scatter(Idx X, int flip) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= X.n_in) return;
    const uint32_t* ps = flip ? X.perm_b : X.perm_a;
    const uint32_t F = X.skey[i], b0 = X.own_off[F], n = X.own_off[F + 1] - b0, p = i - b0, cb = p & ~7u, ce = min(cb + 8u, n);
    const uint32_t pi = ps[i];
    uint32_t r = 0;
    for (uint32_t j = cb; j < ce; j++) r += ps[b0 + j] < pi ? 1u : 0u;
    const uint32_t slot = X.slot_base[F] + cb + r;
    X.slot_tri[slot] = X.own_idx[b0 + pi]; X.slot_pos[slot] = pi;
    X.cluster_node[slot >> 3] = F;
    if (p == n - 1u) for (uint32_t s = X.slot_base[F] + n; s < X.slot_base[F] + ((n + 7u) & ~7u); s++) { X.slot_tri[s] = kNone; X.slot_pos[s] = 0u; }   // padding to the cluster boundary
}
__global__ void __launch_bounds__(kBlock) k_idx_leaf_slots(Idx X) {   // the second, dense slot of every single-triangle leaf below the root (clusters.cpp)
    const uint32_t F = blockIdx.x * kBlock + threadIdx.x;
    if (F >= X.n_nodes || !X.a_leaf[F]) return;
    const uint32_t s = X.n_list_slots + X.leaf_rank[F];
    X.slot_tri[s] = X.own_idx[X.own_off[F]]; X.slot_pos[s] = 0u;
}

__global__ void __launch_bounds__(kBlock) k_idx_slots(Idx X) {   // per slot: geometry, attributes, padded per-triangle box (api.cpp's fill loops, clusters.cpp)
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= X.n_slots_total + 8u) return;
    if (s >= X.n_slots_total) {                                 // the 8 spare box records after the list slots (all-zero boxes in centre/half form)
        DevClusterBox B{}; to_centre_half(B.lo, B.hi); X.tboxes[X.n_list_slots + (s - X.n_slots_total)] = B;
        return;
    }
    const uint32_t tri = X.slot_tri[s];
    DevTriGeom g; DevTriAttr a;
    if (tri == kNone) { memset(&g, 0, sizeof g); memset(&a, 0, sizeof a); a.orig = kNone; }
    else {
        const Triangle& t = X.tris[tri];
        g.v1[0] = t.v1.x; g.v1[1] = t.v1.y; g.v1[2] = t.v1.z;
        g.e1[0] = t.v2.x - t.v1.x; g.e1[1] = t.v2.y - t.v1.y; g.e1[2] = t.v2.z - t.v1.z;   // ray.rs:60
        g.e2[0] = t.v3.x - t.v1.x; g.e2[1] = t.v3.y - t.v1.y; g.e2[2] = t.v3.z - t.v1.z;   // ray.rs:61
        g.pos = X.slot_pos[s]; g._pad = 0;
        a.uv[0] = t.t1.x; a.uv[1] = t.t1.y; a.uv[2] = t.t2.x; a.uv[3] = t.t2.y; a.uv[4] = t.t3.x; a.uv[5] = t.t3.y;
        a.nrm[0] = t.n1.x; a.nrm[1] = t.n1.y; a.nrm[2] = t.n1.z; a.nrm[3] = t.n2.x; a.nrm[4] = t.n2.y; a.nrm[5] = t.n2.z;
        a.nrm[6] = t.n3.x; a.nrm[7] = t.n3.y; a.nrm[8] = t.n3.z;
        a.mat = t.mat; a.orig = tri;
    }
    X.geom[s] = g; X.attr[s] = a;
    if (s < X.n_list_slots) {
        DevClusterBox B{};
        if (tri == kNone) { for (int k = 0; k < 3; k++) { B.lo[k] = FLT_MAX; B.hi[k] = -FLT_MAX; } }
        else {
            const double* tb = &X.tbox[6 * (size_t)tri];
            for (int k = 0; k < 3; k++) { B.lo[k] = X.enable_cull ? round_down_f(tb[k] - X.pad) : -FLT_MAX; B.hi[k] = X.enable_cull ? round_up_f(tb[3 + k] + X.pad) : FLT_MAX; }
        }
        to_centre_half(B.lo, B.hi);
        X.tboxes[s] = B;
    }
}

__global__ void __launch_bounds__(kBlock) k_idx_clusters(Idx X) {   // per cluster of 8 slots: its padded box; the node's own bounds
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x, n_cl = X.n_list_slots >> 3;
    if (c >= n_cl + 8u) return;
    if (c >= n_cl) { DevClusterBox B{}; to_centre_half(B.lo, B.hi); X.cboxes[c] = B; return; }
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (uint32_t s = 8u * c; s < 8u * c + 8u; s++) {
        const uint32_t tri = X.slot_tri[s];
        if (tri == kNone) continue;
        const double* tb = &X.tbox[6 * (size_t)tri];
        for (int k = 0; k < 3; k++) { lo[k] = tb[k] < lo[k] ? tb[k] : lo[k]; hi[k] = hi[k] < tb[3 + k] ? tb[3 + k] : hi[k]; }
    }
    DevClusterBox B{};
    for (int k = 0; k < 3; k++) { B.lo[k] = X.enable_cull ? round_down_f(lo[k] - X.pad) : -FLT_MAX; B.hi[k] = X.enable_cull ? round_up_f(hi[k] + X.pad) : FLT_MAX; }
    for (int k = 0; k < 3; k++) { X.cl_lohi[6 * (size_t)c + k] = B.lo[k]; X.cl_lohi[6 * (size_t)c + 3 + k] = B.hi[k]; }
    to_centre_half(B.lo, B.hi);
    X.cboxes[c] = B;
    const uint32_t F = X.cluster_node[c];
    for (int k = 0; k < 3; k++) { atomicMin(&X.nb[6 * (size_t)F + k], ord_f64(lo[k])); atomicMax(&X.nb[6 * (size_t)F + 3 + k], ord_f64(hi[k])); }
}

__global__ void __launch_bounds__(kBlock) k_idx_supers(Idx X) {   // per super-cluster (and group of 8 of them): the union of its clusters' boxes
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x, n_cl = X.n_list_slots >> 3;
    if (c >= n_cl) return;
    const uint32_t F = X.cluster_node[c], lc = c - (X.slot_base[F] >> 3);
    if (lc & 7u) return;
    const uint32_t n = X.own_off[F + 1] - X.own_off[F], node_cl = (n + 7u) >> 3, n_sup = (n + kSuperTris - 1) / kSuperTris;
    const bool grouped = X.enable_cull && n_sup > kGroupThreshold;
    const uint32_t j = lc >> 3;
    auto unite = [&](uint32_t c0, uint32_t c1, float* lo, float* hi) {
        for (int k = 0; k < 3; k++) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; }
        for (uint32_t q = c0; q < c1; q++) for (int k = 0; k < 3; k++) { const float l = X.cl_lohi[6 * (size_t)q + k], h = X.cl_lohi[6 * (size_t)q + 3 + k]; lo[k] = l < lo[k] ? l : lo[k]; hi[k] = hi[k] < h ? h : hi[k]; }
    };
    const uint32_t cbase = X.slot_base[F] >> 3;
    {
        DevSuper S{};
        unite(cbase + lc, cbase + min(lc + 8u, node_cl), S.lo, S.hi);
        S.tri_begin = X.slot_base[F] + kSuperTris * j; S.tri_count = min(kSuperTris, n - kSuperTris * j);
        to_centre_half(S.lo, S.hi);
        X.supers[X.sup_base[F] + j + (grouped ? j / kGroupSupers + 1u : 0u)] = S;
    }
    if (grouped && (j % kGroupSupers) == 0u) {
        const uint32_t g = j / kGroupSupers;
        DevSuper G{};
        unite(cbase + lc, cbase + min(lc + 8u * kGroupSupers, node_cl), G.lo, G.hi);
        G.tri_begin = min(kGroupSupers, n_sup - kGroupSupers * g); G.tri_count = 0u;
        to_centre_half(G.lo, G.hi);
        X.supers[X.sup_base[F] + g * (kGroupSupers + 1u)] = G;
    }
}

// subtree bounds, bottom-up: one launch per level (temporary ids of a level are contiguous)
__global__ void __launch_bounds__(kBlock) k_idx_sweep(Idx X, const uint32_t* tmp2final, uint32_t lb, uint32_t le) {
    const uint32_t m = lb + blockIdx.x * kBlock + threadIdx.x;
    if (m >= le) return;
    const uint32_t F = tmp2final[m], fc = X.ffc[F];
    if (!fc) return;
    unsigned long long* d = &X.nb[6 * (size_t)F];
    for (uint32_t c = fc; c < fc + 8u; c++) {
        const unsigned long long* s = &X.nb[6 * (size_t)c];
        for (int k = 0; k < 3; k++) { d[k] = s[k] < d[k] ? s[k] : d[k]; d[3 + k] = d[3 + k] < s[3 + k] ? s[3 + k] : d[3 + k]; }
    }
}
__global__ void __launch_bounds__(kBlock) k_idx_child_boxes(Idx X) {
    const uint32_t c = 1u + blockIdx.x * kBlock + threadIdx.x;   // node id c >= 1 -> child_boxes[c - 1]
    if (c >= X.n_nodes + 8u) return;
    DevClusterBox B{};
    if (c < X.n_nodes) {
        double lo[3], hi[3];
        for (int k = 0; k < 3; k++) { lo[k] = unord_f64(X.nb[6 * (size_t)c + k]); hi[k] = unord_f64(X.nb[6 * (size_t)c + 3 + k]); }
        const bool empty = X.ftc[c] == 0u || lo[0] > hi[0];
        for (int k = 0; k < 3; k++) {
            B.lo[k] = !X.enable_cull ? -FLT_MAX : empty ? FLT_MAX : round_down_f(lo[k] - X.pad);
            B.hi[k] = !X.enable_cull ? FLT_MAX : empty ? -FLT_MAX : round_up_f(hi[k] + X.pad);
        }
    }
    to_centre_half(B.lo, B.hi);
    X.child_boxes[c - 1u] = B;
}

__global__ void __launch_bounds__(kBlock) k_idx_nodes(Idx X, const double* fbox) {   // DevNode records (api.cpp's fill loop)
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= X.n_nodes) return;
    DevNode d;
    bool plain = true;
    for (int k = 0; k < 3; k++) {
        d.lo[k] = fbox[6 * (size_t)i + k]; d.hi[k] = fbox[6 * (size_t)i + 3 + k];
        d.mid[k] = d.lo[k] + (d.hi[k] - d.lo[k]) / 2.0;          // octree.rs:136-146 (== the lower corner of child TFR)
        const double v[3] = {d.lo[k], d.mid[k], d.hi[k]};
        for (int q = 0; q < 3; q++) if (!(v[q] == 0.0 || (fabs(v[q]) > 0x1p-200 && fabs(v[q]) < 0x1p200))) plain = false;
    }
    if (!plain) X.flags[2] = 1u;
    d.first_child = X.ffc[i]; d.sup_begin = X.sup_base[i]; d.sup_count = X.a_sup[i];
    d.s0_begin = d.sup_count ? X.supers[d.sup_begin].tri_begin : 0u;
    d.flags = (X.ftc[i] ? 0x100u : 0u) | ((d.sup_count ? X.supers[d.sup_begin].tri_count : 0u) << 24);
    d.leaf_base = X.a_leaf[i] ? X.n_list_slots + X.leaf_rank[i] : 0u;
    if (d.first_child) for (uint32_t k = 8; k-- > 0;) {
        const uint32_t c = d.first_child + k;
        if (X.ftc[c]) d.flags |= 1u << k;
        if (X.a_leaf[c]) { d.flags |= 1u << (9 + k); d.leaf_base = X.n_list_slots + X.leaf_rank[c]; }
    }
    X.nodes[i] = d;
}

// exactness guard of the index: clusters.cpp find_origin_suspects, per triangle of the tree
struct SuspectOut { uint32_t tri; uint32_t _pad; DevSuspect s; };
__global__ void __launch_bounds__(kBlock) k_suspects(const Triangle* tris, const uint32_t* own, uint32_t n, double ox, double oy, double oz, double pad, uint32_t* count, SuspectOut* out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n || own[i] == kNone) return;
    const Triangle& t = tris[i];
    const double eps = 0x1p-53;
    const double e1[3] = {t.v2.x - t.v1.x, t.v2.y - t.v1.y, t.v2.z - t.v1.z}, e2[3] = {t.v3.x - t.v1.x, t.v3.y - t.v1.y, t.v3.z - t.v1.z};
    const double s[3] = {ox - t.v1.x, oy - t.v1.y, oz - t.v1.z};
    const double nn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double l1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
    const double ln = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]), ls = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    if (!(l1 > 0) || !(l2 > 0)) return;
    if (!isfinite(l1 + l2 + ls)) return;
    const double R = ls + fmax(l1, l2);
    const double sinphi = ln / (l1 * l2);
    double alpha, delta;
    if (!(sinphi > 1e-300)) { alpha = 2.0; delta = INFINITY; }
    else { alpha = 8.0 * 64.0 * eps * R / (pad * sinphi); delta = 2.0 * (alpha * R + 64.0 * eps * R) / sinphi; }
    const double rho = ln > 0 ? fabs(s[0] * nn[0] + s[1] * nn[1] + s[2] * nn[2]) / ln : 0.0;
    if (!(rho <= delta)) return;
    const uint32_t k = atomicAdd(count, 1u);
    if (k > RRT_MAX_SUSPECTS) return;
    SuspectOut q{};
    q.tri = i;
    for (int a = 0; a < 3; a++) q.s.n[a] = ln > 0 ? nn[a] / ln : 0.0;
    q.s.alpha2 = alpha >= 1.0 ? 4.0 : alpha * alpha;
    out[k] = q;
}

// the caller's arrays -> Triangle records (rrt_raytracer_create_from_arrays): the layout rrt_model_from_arrays produces on the host
__global__ void __launch_bounds__(kBlock) k_pack_triangles(const double* __restrict__ pos, const double* __restrict__ uv, const double* __restrict__ nrm, const uint32_t* __restrict__ mat, uint32_t n, Triangle* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    auto rd = [](const double* p) { Vec3 v; v.x = p[0]; v.y = p[1]; v.z = p[2]; return v; };
    Triangle t;
    t.v1 = rd(pos + 9 * (size_t)i); t.v2 = rd(pos + 9 * (size_t)i + 3); t.v3 = rd(pos + 9 * (size_t)i + 6);
    t.t1 = rd(uv + 9 * (size_t)i);  t.t2 = rd(uv + 9 * (size_t)i + 3);  t.t3 = rd(uv + 9 * (size_t)i + 6);
    t.n1 = rd(nrm + 9 * (size_t)i); t.n2 = rd(nrm + 9 * (size_t)i + 3); t.n3 = rd(nrm + 9 * (size_t)i + 6);
    t.mat = mat[i]; t._pad = 0;
    out[i] = t;
}
__global__ void k_set_root(Oct S) {
    for (int a = 0; a < 3; a++) { S.nbox[a] = S.rlo[a]; S.nbox[3 + a] = S.rhi[a]; }
    S.first[0] = kNone; S.second[0] = kNone; S.cnt[0] = 0u; S.child_base[0] = 0u; S.ctr[0] = 0u; S.ctr[1] = 0u; S.ctr[2] = 0u;
}

// ------------------------------------------------------------------------------------------------ host side
struct DevArena {
    char* base = nullptr; size_t cap = 0, used = 0;
    template <class T> T* take(size_t count) {
        const size_t bytes = (sizeof(T) * (count ? count : 1) + 255) & ~(size_t)255;
        if (used + bytes > cap) throw Error{RRT_ERR_OOM, "internal: set-up arena too small"};
        T* p = reinterpret_cast<T*>(base + used); used += bytes; return p;
    }
};
struct DevFree { void* p = nullptr; ~DevFree() { if (p) (void)hipFree(p); } };

// ---- pinned staging: one ring of page-locked chunks per device, shared by every upload and every pageable-framebuffer download of the process.
// A slot's event says when the DMA that last used it has finished; a slot is waited for right before it is reused, never at the end of a call.
#ifndef RRT_RING_SLOTS
#define RRT_RING_SLOTS 8
#endif
#ifndef RRT_RING_SLOT_MB
#define RRT_RING_SLOT_MB 4
#endif
struct StagingRing {
    static constexpr int kSlots = RRT_RING_SLOTS; static constexpr size_t kSlotBytes = (size_t)RRT_RING_SLOT_MB << 20;
    char* mem = nullptr; hipEvent_t ev[kSlots] = {}; bool busy[kSlots] = {}; size_t next = 0; hipStream_t stream = nullptr, stream2 = nullptr;
    void ensure() {
        if (mem) return;
        HB_TRY(hipHostMalloc((void**)&mem, kSlots * kSlotBytes, hipHostMallocDefault));
        for (auto& e : ev) HB_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HB_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));   // the device's set-up stream (creating one costs ~3 ms: done once, by the warm-up thread when it runs)
        HB_TRY(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));  // uploads beside the build (upload_stream)
    }
    int acquire() {                                                     // next slot, free to be written
        const int s = (int)(next++ % kSlots);
        if (busy[s]) { HB_TRY(hipEventSynchronize(ev[s])); busy[s] = false; }
        return s;
    }
    void release(int s, hipStream_t st) { HB_TRY(hipEventRecord(ev[s], st)); busy[s] = true; }
};
constexpr int kMaxDevices = 64;
std::mutex g_ring_mu;
StagingRing g_rings[kMaxDevices];
StagingRing& ring_of_current_device() {                                 // (caller holds g_ring_mu)
    int dev = 0;
    HB_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) throw Error{RRT_ERR_INVALID_ARG, "device index beyond the staging table"};
    g_rings[dev].ensure();
    return g_rings[dev];
}
// parallel memcpy on the host pool (copies out of the ring on the frame path: one core moves ~10 GB/s)
void copy_bytes(char* dst, const char* src, size_t len) {
    if (len < ((size_t)2 << 20)) { std::memcpy(dst, src, len); return; }
    parallel_ranges(len, (len + 3) / 4, [&](size_t b, size_t e, size_t) { std::memcpy(dst + b, src + b, e - b); });
}

}  // namespace

void staged_upload_warm() { try { std::lock_guard<std::mutex> lk(g_ring_mu); (void)ring_of_current_device(); } catch (...) { (void)hipGetLastError(); } }

void* setup_stream() { std::lock_guard<std::mutex> lk(g_ring_mu); return ring_of_current_device().stream; }
void* upload_stream() { std::lock_guard<std::mutex> lk(g_ring_mu); return ring_of_current_device().stream2; }

void staged_upload(void* dst, const void* src, size_t bytes, void* stream_) {
    if (!bytes) return;
    hipStream_t stream = (hipStream_t)stream_;
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost) {   // already page-locked: one DMA
        HB_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        return;
    }
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_ring_mu);
    StagingRing& R = ring_of_current_device();
    const size_t S = StagingRing::kSlotBytes, n_chunks = (bytes + S - 1) / S;
    // One task per ring slot (host pool), each an independent pipeline: wait for the slot's last DMA, fill the slot from `src`, enqueue its DMA, take the next
    // chunk -- so the copies into page-locked memory (the slow part: one core moves ~10 GB/s) run on several cores while the DMA engine drains
    // the finished slots.  Chunks land at disjoint destinations: their order on the stream does not matter.
    const unsigned workers = (unsigned)std::min<size_t>(std::min<size_t>(StagingRing::kSlots, n_chunks), host_threads());
    int dev = 0;
    HB_TRY(hipGetDevice(&dev));
    std::atomic<size_t> next_chunk{0};
    std::vector<int> err(workers, 0);
    auto run = [&](unsigned w) {
        if (hipSetDevice(dev) != hipSuccess) { err[w] = (int)hipGetLastError(); return; }        // (HIP's current device is per thread, and a pool worker keeps its last one)
        char* stage = R.mem + (size_t)w * S;
        for (size_t c; (c = next_chunk.fetch_add(1)) < n_chunks;) {
            const size_t off = c * S, len = std::min(S, bytes - off);
            hipError_t e = hipSuccess;
            if (R.busy[w]) { e = hipEventSynchronize(R.ev[w]); R.busy[w] = false; }
            if (e == hipSuccess) { std::memcpy(stage, static_cast<const char*>(src) + off, len); e = hipMemcpyAsync(static_cast<char*>(dst) + off, stage, len, hipMemcpyHostToDevice, stream); }
            if (e == hipSuccess) { e = hipEventRecord(R.ev[w], stream); R.busy[w] = true; }
            if (e != hipSuccess) { err[w] = (int)e; return; }
        }
    };
    parallel_ranges(workers, 1, [&](size_t b, size_t e, size_t) { for (size_t w = b; w < e; w++) run((unsigned)w); });
    for (int e : err) if (e) throw HipBuildFail{e, "staged_upload (pinned-staging host-to-device copy)"};
    // src has been read completely: it may be freed.  dst is complete once `stream` has drained; the slots guard themselves (busy + event).
}

// Device -> pageable host memory through the ring: chunk DMAs run ahead while the finished chunks are copied out (a pageable hipMemcpy stages through
// the runtime's own bounce buffers serially; a frame-sized pinned buffer of the caller's own costs milliseconds to allocate -- more than the
// reference's one frame takes to trace).  Blocking: dst is complete on return.  Everything enqueued on `stream` before the call is waited for.
void staged_download(void* dst, const void* src_dev, size_t bytes, void* stream_) {
    if (!bytes) return;
    hipStream_t stream = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lk(g_ring_mu);
    StagingRing& R = ring_of_current_device();
    size_t chunk = (bytes / 8 + 0xFFFFF) & ~(size_t)0xFFFFF;              // about 8 chunks per frame, whole MiB, at most a slot
    chunk = std::min(std::max(chunk, (size_t)1 << 20), StagingRing::kSlotBytes);
    const size_t n_chunks = (bytes + chunk - 1) / chunk;
    int slot_of[StagingRing::kSlots];
    size_t issued = 0;
    auto issue = [&](size_t c) {
        const int slot = R.acquire();
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        HB_TRY(hipMemcpyAsync(R.mem + (size_t)slot * StagingRing::kSlotBytes, static_cast<const char*>(src_dev) + off, len, hipMemcpyDeviceToHost, stream));
        R.release(slot, stream);
        slot_of[c % StagingRing::kSlots] = slot;
    };
    for (; issued < n_chunks && issued < (size_t)StagingRing::kSlots; issued++) issue(issued);
    for (size_t c = 0; c < n_chunks; c++) {
        const int slot = slot_of[c % StagingRing::kSlots];
        HB_TRY(hipEventSynchronize(R.ev[slot])); R.busy[slot] = false;
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        copy_bytes(static_cast<char*>(dst) + off, R.mem + (size_t)slot * StagingRing::kSlotBytes, len);
        if (issued < n_chunks) issue(issued++);
    }
}

void gpu_build_scene(const TriSource& src, uint32_t n, const Box& root, bool enable_cull, const double origin[3], void* stream_, GpuScene& out, const std::function<void()>& after_upload) {
    hipStream_t st = (hipStream_t)stream_;
    const bool trace = std::getenv("RRT_SETUP_TRACE") != nullptr;          // developer: host wall time of every stage (synchronising: not the production timing)
    auto lap = [&, last = std::chrono::steady_clock::now()](const char* what) mutable {
        if (!trace) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[gpu set-up] %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count()); last = now;
    };
    hipEvent_t evs[4] = {};
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int i = 0; i < 4; i++) if (e[i]) (void)hipEventDestroy(e[i]); } } evg{evs};
    for (auto& e : evs) HB_TRY(hipEventCreate(&e));

    // ---- temporaries, first part: everything whose size follows from the triangle count.  A subdivision has a distinct trigger triangle and
    // triangle 0 triggers none, so there are at most n - 1 of them: 1 + 8 (n - 1) nodes bound every scene.
    const size_t cap = 1 + 8 * (size_t)(n > 1 ? n - 1 : 0) + 8;
    if (cap > 0xFFFFFF00ull) throw Error{RRT_ERR_UNSUPPORTED, "more than 2^29 triangles: node ids are 32 bits wide"};
    size_t scan_bytes = 0, sort_bytes = 0;
    {
        size_t b = 0;
        HB_TRY(rocprim::exclusive_scan(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, std::max<size_t>(cap + 1, n + 1), rocprim::plus<uint32_t>(), st)); scan_bytes = b;
        HB_TRY(rocprim::radix_sort_pairs(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, n ? n : 1, 0, 32, st)); sort_bytes = b;
    }
    const size_t prim_bytes = std::max(scan_bytes, sort_bytes) + 256;
    DevFree t1;
    DevArena A1;
    A1.cap = (size_t)n * (sizeof(Triangle) + (src.tris ? 0 : 27 * 8 + 4) + 48 + 3 * 4 + 4 /*rank*/ + 4 * 4 /*key,val in/out*/) + cap * (48 + 4 * 4) + (cap / 8 + 1) * 8 + prim_bytes + (64 << 10);
    lap("events, rocPRIM size queries");
    HB_TRY(hipMalloc(&t1.p, A1.cap)); A1.base = static_cast<char*>(t1.p);
    lap("hipMalloc temporaries 1");

    HB_TRY(hipEventRecord(evs[0], st));
    Triangle* d_tris = A1.take<Triangle>(n);
    if (src.tris) staged_upload(d_tris, src.tris, sizeof(Triangle) * (size_t)n, st);
    // The caller's own arrays: the octree needs the positions only, so they go up first and the tree is built while the other two thirds (texture
    // coordinates, normals, materials) follow on the upload stream, on a pool thread; the triangle records are packed when the index needs them.
    double* d_pos = nullptr, *d_uv = nullptr, *d_nrm = nullptr; uint32_t* d_mat = nullptr;
    hipEvent_t ev_attr = nullptr;
    struct EvOne { hipEvent_t& e; ~EvOne() { if (e) (void)hipEventDestroy(e); } } ev_attr_guard{ev_attr};
    std::unique_ptr<AsyncTask> attr_task;
    if (!src.tris && n) {
        d_pos = A1.take<double>(9 * (size_t)n); d_uv = A1.take<double>(9 * (size_t)n); d_nrm = A1.take<double>(9 * (size_t)n); d_mat = A1.take<uint32_t>(n);
        staged_upload(d_pos, src.pos, 72 * (size_t)n, st);
        HB_TRY(hipEventCreateWithFlags(&ev_attr, hipEventDisableTiming));
        hipStream_t st2 = (hipStream_t)upload_stream();
        int dev = 0;
        HB_TRY(hipGetDevice(&dev));
        attr_task.reset(new AsyncTask([=, &src] {
            HB_TRY(hipSetDevice(dev));                                  // (HIP's current device is per thread, and a pool worker keeps its last one)
            staged_upload(d_uv, src.uv, 72 * (size_t)n, st2); staged_upload(d_nrm, src.nrm, 72 * (size_t)n, st2); staged_upload(d_mat, src.mat, 4 * (size_t)n, st2);
            HB_TRY(hipEventRecord(ev_attr, st2));
        }));
    }
    HB_TRY(hipEventRecord(evs[1], st));
    if (after_upload) after_upload();
    lap("triangle upload (pinned staging)");

    Oct S{};
    S.tris = d_tris; S.n = n; S.pos = d_pos;
    S.tbox = A1.take<double>(6 * (size_t)n); S.cur = A1.take<uint32_t>(n); S.own = A1.take<uint32_t>(n); S.trig = A1.take<uint32_t>(n + 1);
    S.nbox = A1.take<double>(6 * cap); S.first = A1.take<uint32_t>(cap); S.second = A1.take<uint32_t>(cap); S.cnt = A1.take<uint32_t>(cap); S.child_base = A1.take<uint32_t>(cap);
    S.block_parent = A1.take<uint32_t>(cap / 8 + 1);
    S.ctr = A1.take<uint32_t>(8);
    uint32_t* rank = A1.take<uint32_t>(n + 1);
    uint32_t* key_in = A1.take<uint32_t>(n), *val_in = A1.take<uint32_t>(n), *key_out = A1.take<uint32_t>(n), *val_out = A1.take<uint32_t>(n);
    void* prim_tmp = A1.take<char>(prim_bytes);
    for (int a = 0; a < 3; a++) { S.rlo[a] = root.lo[a]; S.rhi[a] = root.hi[a]; }

    uint32_t* h_ctr = nullptr;                                         // pinned read-back words
    HB_TRY(hipHostMalloc((void**)&h_ctr, 64 * sizeof(uint32_t), hipHostMallocDefault));
    struct HostFree { void* p; ~HostFree() { if (p) (void)hipHostFree(p); } } hf{h_ctr};
    lap("hipHostMalloc read-back words");
    auto read_words = [&](const uint32_t* dev, int count) { HB_TRY(hipMemcpyAsync(h_ctr, dev, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, st)); HB_TRY(hipStreamSynchronize(st)); };

    // ---- level loop (octree.rs:54-108 for every triangle at once)
    hipLaunchKernelGGL(k_set_root, dim3(1), dim3(1), 0, st, S);
    const unsigned oct_grid = std::max(1u, std::min(kOctMaxGrid, (unsigned)((n + kOctBlock - 1) / kOctBlock)));
    if (n) hipLaunchKernelGGL(k_oct_init, dim3(oct_grid), dim3(kOctBlock), 0, st, S);
    std::vector<uint32_t> level_begin{0u};
    uint32_t n_nodes = 1, lb = 0, le = 1, n_blocks = 0;
    read_words(S.ctr, 3);
    uint32_t active = h_ctr[1];
    out.all_inside_root = h_ctr[2] ? 0u : 1u;
    while (n && active) {
        hipLaunchKernelGGL(k_oct_second, dim3(oct_grid), dim3(kOctBlock), 0, st, S, lb, le);
        hipLaunchKernelGGL(k_oct_split, dim3(grid_for(le - lb)), dim3(kBlock), 0, st, S, lb, le);
        hipLaunchKernelGGL(k_oct_descend, dim3(oct_grid), dim3(kOctBlock), 0, st, S, le, n_blocks);
        read_words(S.ctr, 2);
        const uint32_t blocks_now = h_ctr[0]; active = h_ctr[1];
        if (blocks_now == n_blocks) break;                             // no node of this level subdivided: it was the last one
        n_blocks = blocks_now;
        lb = le; le = 1u + 8u * n_blocks; n_nodes = le;
        level_begin.push_back(lb);
        if (level_begin.size() > RRT_MAX_OCTREE_DEPTH)
            throw Error{RRT_ERR_DEPTH, "octree depth exceeds RRT_MAX_OCTREE_DEPTH (" + std::to_string(RRT_MAX_OCTREE_DEPTH) + "): coincident triangles? (octree.rs:79-92)"};
    }
    HB_TRY(hipGetLastError());
    lap("octree level loop");
    level_begin.push_back(n_nodes);                                     // level L = temporary ids [level_begin[L], level_begin[L + 1])
    out.n_nodes = n_nodes; out.max_depth = (uint32_t)level_begin.size() - 1;

    // ---- second allocation: the octree in the reference's numbering (kept), and the node-sized temporaries
    const size_t oct_bytes = (size_t)n_nodes * (48 + 4 + 4) + ((size_t)n_nodes + 1) * 4 + (size_t)n * 4 + 4096;
    DevFree t2;
    DevArena A2;
    A2.cap = oct_bytes + (size_t)n_nodes * (4 /*tmp2final*/ + 4 /*newblock*/ + 6 * 4 /*a_*, bases*/ + 48 /*nb*/) + (size_t)(n_nodes + 1) * 4 * 4 + (size_t)n * (2 * 4 + 2 * 24) + (64 << 10);
    HB_TRY(hipMalloc(&t2.p, A2.cap)); A2.base = static_cast<char*>(t2.p);
    lap("hipMalloc temporaries 2");
    Remap R{};
    R.n_nodes = n_nodes; R.n_blocks = n_blocks; R.n = n;
    R.second = S.second; R.cnt = S.cnt; R.child_base = S.child_base; R.block_parent = S.block_parent; R.rank = rank; R.own = S.own; R.nbox = S.nbox;
    R.newblock = A2.take<uint32_t>(n_blocks + 1); R.tmp2final = A2.take<uint32_t>(n_nodes);
    R.fbox = A2.take<double>(6 * (size_t)n_nodes); R.ffc = A2.take<uint32_t>(n_nodes); R.ftc = A2.take<uint32_t>(n_nodes);
    uint32_t* own_count = A2.take<uint32_t>(n_nodes + 2);
    uint32_t* own_off = A2.take<uint32_t>(n_nodes + 2);
    R.key = key_in; R.val = val_in; R.own_count = own_count;
    if (n) {
        size_t b = prim_bytes;
        HB_TRY(rocprim::exclusive_scan(prim_tmp, b, S.trig, rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
    }
    if (n_blocks) hipLaunchKernelGGL(k_newblock, dim3(grid_for(n_blocks)), dim3(kBlock), 0, st, R);
    hipLaunchKernelGGL(k_remap_nodes, dim3(grid_for(n_nodes)), dim3(kBlock), 0, st, R);
    HB_TRY(hipMemsetAsync(own_count, 0, sizeof(uint32_t) * ((size_t)n_nodes + 2), st));
    unsigned key_bits = 1; while ((1ull << key_bits) <= (unsigned long long)n_nodes) key_bits++;
    DevFree sort_tmp;                                                   // only when this bit range needs more than the up-front query said
    if (n) {
        hipLaunchKernelGGL(k_tri_keys, dim3(grid_for(n)), dim3(kBlock), 0, st, R);
        // rocPRIM picks the sort's algorithm (and with it the size of its temporary storage) from the element count AND the bit range, so the need is
        // asked again for the range actually sorted: the 32-bit query above is a lower bound only (4 M triangles, 23 bits: more).
        size_t b = 0;
        HB_TRY(rocprim::radix_sort_pairs(nullptr, b, key_in, key_out, val_in, val_out, (size_t)n, 0u, key_bits, st));
        void* tmp = prim_tmp;
        if (b > prim_bytes) { HB_TRY(hipMalloc(&sort_tmp.p, b)); tmp = sort_tmp.p; } else b = prim_bytes;
        HB_TRY(rocprim::radix_sort_pairs(tmp, b, key_in, key_out, val_in, val_out, (size_t)n, 0u, key_bits, st));
    }
    {
        size_t b = prim_bytes;
        HB_TRY(rocprim::exclusive_scan(prim_tmp, b, own_count, own_off, 0u, (size_t)n_nodes + 1, rocprim::plus<uint32_t>(), st));
    }
    read_words(own_off + n_nodes, 1);
    const uint32_t n_in = h_ctr[0];
    out.n_in_tree = n_in;
    HB_TRY(hipEventRecord(evs[2], st));
    lap("remap, own-list sort");

    // ---- index sizes
    double mag = 0;
    for (int k = 0; k < 3; k++) mag = std::max(mag, std::max(std::fabs(root.lo[k]), std::fabs(root.hi[k])));
    out.scene_magnitude = mag; out.pad = mag * kPadFraction;
    Idx X{};
    X.n_nodes = n_nodes; X.n_in = n_in; X.enable_cull = enable_cull ? 1u : 0u; X.inline_leaves = n_nodes < (1u << 24) ? 1u : 0u; X.pad = out.pad;
    X.ffc = R.ffc; X.ftc = R.ftc; X.own_off = own_off; X.own_idx = val_out; X.skey = key_out; X.tbox = S.tbox; X.tris = d_tris;
    X.a_slots = A2.take<uint32_t>(n_nodes + 1); X.a_sup = A2.take<uint32_t>(n_nodes + 1); X.a_leaf = A2.take<uint32_t>(n_nodes + 1);
    X.slot_base = A2.take<uint32_t>(n_nodes + 1); X.sup_base = A2.take<uint32_t>(n_nodes + 1); X.leaf_rank = A2.take<uint32_t>(n_nodes + 1);
    X.flags = A2.take<uint32_t>(8);
    X.nb = A2.take<unsigned long long>(6 * (size_t)n_nodes);
    X.perm_a = A2.take<uint32_t>(n_in); X.perm_b = A2.take<uint32_t>(n_in); X.cen_a = A2.take<double>(3 * (size_t)n_in); X.cen_b = A2.take<double>(3 * (size_t)n_in);
    HB_TRY(hipMemsetAsync(X.flags, 0, 8 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_idx_sizes, dim3(grid_for((size_t)n_nodes + 1)), dim3(kBlock), 0, st, X);
    for (auto pr : {std::pair<uint32_t*, uint32_t*>{X.a_slots, X.slot_base}, {X.a_sup, X.sup_base}, {X.a_leaf, X.leaf_rank}}) {
        size_t b = prim_bytes;
        HB_TRY(rocprim::exclusive_scan(prim_tmp, b, pr.first, pr.second, 0u, (size_t)n_nodes + 1, rocprim::plus<uint32_t>(), st));
    }
    HB_TRY(hipMemcpyAsync(h_ctr, X.slot_base + n_nodes, 4, hipMemcpyDeviceToHost, st));
    HB_TRY(hipMemcpyAsync(h_ctr + 1, X.sup_base + n_nodes, 4, hipMemcpyDeviceToHost, st));
    HB_TRY(hipMemcpyAsync(h_ctr + 2, X.leaf_rank + n_nodes, 4, hipMemcpyDeviceToHost, st));
    HB_TRY(hipMemcpyAsync(h_ctr + 3, X.flags, 8, hipMemcpyDeviceToHost, st));
    HB_TRY(hipStreamSynchronize(st));
    const uint32_t n_list_slots = h_ctr[0], n_sup = h_ctr[1], n_leaves = h_ctr[2], has_groups = h_ctr[3], max_own = h_ctr[4];
    const uint32_t n_slots_total = n_list_slots + n_leaves, n_cl = n_list_slots / 8;
    out.n_list_slots = n_list_slots; out.n_slots_total = n_slots_total; out.n_sup_records = n_sup; out.n_clusters = n_cl; out.has_groups = has_groups; out.inline_leaves = X.inline_leaves; out.max_own = max_own;
    X.n_list_slots = n_list_slots; X.n_slots_total = n_slots_total;
    lap("index sizes + scans");

    // ---- third allocation: what the trace kernels read (kept) + the octree (kept) in one piece; slot-sized temporaries in a piece of their own
    {
        size_t need = (size_t)1 << 16;
        need += (size_t)n_nodes * sizeof(DevNode) + (size_t)(n_slots_total + 1) * (sizeof(DevTriGeom) + sizeof(DevTriAttr));
        need += ((size_t)n_sup + 1 + n_cl + 8 + n_nodes + 8 + n_list_slots + 8) * 32 + (RRT_MAX_SUSPECTS + 2) * sizeof(DevSuspect);
        need += oct_bytes + (size_t)(n_slots_total + 8) * 8 + 16 * 256;
        HB_TRY(hipMalloc(&out.scene_alloc, need)); out.scene_alloc_bytes = need;
    }
    DevArena A3; A3.base = static_cast<char*>(out.scene_alloc); A3.cap = out.scene_alloc_bytes;
    out.nodes = A3.take<DevNode>(n_nodes); out.geom = A3.take<DevTriGeom>(n_slots_total); out.attr = A3.take<DevTriAttr>(n_slots_total);
    out.supers = A3.take<DevSuper>(n_sup); out.cboxes = A3.take<DevClusterBox>(n_cl + 8); out.child_boxes = A3.take<DevClusterBox>((n_nodes > 1 ? n_nodes - 1 : 0) + 8);
    out.tboxes = A3.take<DevClusterBox>(n_list_slots + 8); out.suspects = A3.take<DevSuspect>(RRT_MAX_SUSPECTS + 1);
    out.oct_box = A3.take<double>(6 * (size_t)n_nodes); out.oct_first_child = A3.take<uint32_t>(n_nodes); out.oct_tri_count = A3.take<uint32_t>(n_nodes);
    out.oct_own_off = A3.take<uint32_t>(n_nodes + 1); out.oct_own_idx = A3.take<uint32_t>(n_in);
    out.slot_tri = A3.take<uint32_t>(n_slots_total + 8); out.slot_pos = A3.take<uint32_t>(n_slots_total + 8);
    HB_TRY(hipMemcpyAsync(out.oct_box, R.fbox, sizeof(double) * 6 * n_nodes, hipMemcpyDeviceToDevice, st));
    HB_TRY(hipMemcpyAsync(out.oct_first_child, R.ffc, 4 * (size_t)n_nodes, hipMemcpyDeviceToDevice, st));
    HB_TRY(hipMemcpyAsync(out.oct_tri_count, R.ftc, 4 * (size_t)n_nodes, hipMemcpyDeviceToDevice, st));
    HB_TRY(hipMemcpyAsync(out.oct_own_off, own_off, 4 * ((size_t)n_nodes + 1), hipMemcpyDeviceToDevice, st));
    if (n_in) HB_TRY(hipMemcpyAsync(out.oct_own_idx, val_out, 4 * (size_t)n_in, hipMemcpyDeviceToDevice, st));
    DevFree t3; DevArena A4;
    A4.cap = (size_t)(n_cl + 8) * (4 + 24) + (RRT_MAX_SUSPECTS + 2) * sizeof(SuspectOut) + 4096;
    HB_TRY(hipMalloc(&t3.p, A4.cap)); A4.base = static_cast<char*>(t3.p);
    X.slot_tri = out.slot_tri; X.slot_pos = out.slot_pos; X.cluster_node = A4.take<uint32_t>(n_cl + 8); X.cl_lohi = A4.take<float>(6 * (size_t)(n_cl + 8));
    X.supers = out.supers; X.cboxes = out.cboxes; X.tboxes = out.tboxes; X.child_boxes = out.child_boxes; X.nodes = out.nodes; X.geom = out.geom; X.attr = out.attr;
    lap("hipMalloc scene + temporaries 3, octree copies");

    // ---- median splits, level by level: ceil(log2(super-clusters of the longest list)) splits down to super-clusters, 3 more down to clusters
    int flip = 0;
    if (n_in) {
        hipLaunchKernelGGL(k_idx_items, dim3(grid_for(n_in)), dim3(kBlock), 0, st, X);
        if (enable_cull && max_own > kClusterTris) {
            uint32_t parts = (max_own + kSuperTris - 1) / kSuperTris, depth64 = 0;
            while ((1u << depth64) < parts) depth64++;
            const uint32_t passes = depth64 + 3;
            for (uint32_t d = 0; d < passes; d++) {
                hipLaunchKernelGGL(k_idx_split, dim3(grid_for(n_in)), dim3(kBlock), 0, st, X, flip ? X.perm_b : X.perm_a, flip ? X.cen_b : X.cen_a, flip ? X.perm_a : X.perm_b, flip ? X.cen_a : X.cen_b, d);
                flip ^= 1;
            }
        }
        lap("median splits");
        hipLaunchKernelGGL(k_idx_scatter, dim3(grid_for(n_in)), dim3(kBlock), 0, st, X, flip);
    }
    hipLaunchKernelGGL(k_idx_leaf_slots, dim3(grid_for(n_nodes)), dim3(kBlock), 0, st, X);
    if (attr_task) {                                                   // the records are needed from here on (k_idx_slots, k_suspects)
        attr_task->wait();
        HB_TRY(hipStreamWaitEvent(st, ev_attr, 0));
        hipLaunchKernelGGL(k_pack_triangles, dim3(grid_for(n)), dim3(kBlock), 0, st, d_pos, d_uv, d_nrm, d_mat, n, d_tris);
    }
    hipLaunchKernelGGL(k_idx_slots, dim3(grid_for((size_t)n_slots_total + 8)), dim3(kBlock), 0, st, X);
    hipLaunchKernelGGL(k_idx_clusters, dim3(grid_for((size_t)n_cl + 8)), dim3(kBlock), 0, st, X);
    if (n_cl) hipLaunchKernelGGL(k_idx_supers, dim3(grid_for(n_cl)), dim3(kBlock), 0, st, X);
    for (size_t L = level_begin.size() - 1; L-- > 0;)
        hipLaunchKernelGGL(k_idx_sweep, dim3(grid_for(level_begin[L + 1] - level_begin[L])), dim3(kBlock), 0, st, X, R.tmp2final, level_begin[L], level_begin[L + 1]);
    hipLaunchKernelGGL(k_idx_child_boxes, dim3(grid_for((size_t)n_nodes + 8)), dim3(kBlock), 0, st, X);
    hipLaunchKernelGGL(k_idx_nodes, dim3(grid_for(n_nodes)), dim3(kBlock), 0, st, X, R.fbox);

    // ---- exactness guard (clusters.cpp: find_origin_suspects)
    SuspectOut* d_sus = A4.take<SuspectOut>(RRT_MAX_SUSPECTS + 2);
    uint32_t* d_sus_count = X.flags + 3;
    if (enable_cull && n && out.pad > 0) hipLaunchKernelGGL(k_suspects, dim3(grid_for(n)), dim3(kBlock), 0, st, d_tris, S.own, n, origin[0], origin[1], origin[2], out.pad, d_sus_count, d_sus);
    HB_TRY(hipGetLastError());
    HB_TRY(hipEventRecord(evs[3], st));
    lap("scatter, records, boxes, sweep, suspects");
    read_words(X.flags, 4);
    out.bounds_plain = h_ctr[2] ? 0u : 1u;
    out.n_suspects = h_ctr[3];
    if (out.n_suspects && out.n_suspects <= RRT_MAX_SUSPECTS) {          // deterministic order (by triangle index): the appends above land in any order
        std::vector<SuspectOut> hs(out.n_suspects);
        HB_TRY(hipMemcpy(hs.data(), d_sus, sizeof(SuspectOut) * hs.size(), hipMemcpyDeviceToHost));
        std::sort(hs.begin(), hs.end(), [](const SuspectOut& a, const SuspectOut& b) { return a.tri < b.tri; });
        std::vector<DevSuspect> ds(hs.size());
        for (size_t i = 0; i < hs.size(); i++) ds[i] = hs[i].s;
        HB_TRY(hipMemcpy(out.suspects, ds.data(), sizeof(DevSuspect) * ds.size(), hipMemcpyHostToDevice));
    }
    HB_TRY(hipStreamSynchronize(st));
    float ms = 0;
    HB_TRY(hipEventElapsedTime(&ms, evs[0], evs[1])); out.ms_upload = ms;
    HB_TRY(hipEventElapsedTime(&ms, evs[1], evs[2])); out.ms_octree = ms;
    HB_TRY(hipEventElapsedTime(&ms, evs[2], evs[3])); out.ms_index = ms;
    lap("read-backs, suspects");
    for (DevFree* f : {&t3, &t2, &t1}) { if (f->p) (void)hipFree(f->p); f->p = nullptr; }
    lap("hipFree temporaries");
}

}  // namespace rrt
