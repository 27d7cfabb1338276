// octree.cpp -- host-side octree construction, reference-identical in structure and ordering.
//
// What it must reproduce (src/collision/octree.rs:41-241, src/collision/aabb.rs:25-60): triangles are inserted
// one at a time in push order; a triangle whose box misses the root is dropped (octree.rs:71-73); every node
// passed on the way down counts it (octree.rs:75); an empty leaf keeps it (octree.rs:77-78); a leaf that already
// holds one splits into 8 children appended contiguously in the order BBL,BFL,BFR,BBR,TBL,TFL,TFR,TBR
// (octree.rs:216-238) and the resident triangle is NOT pushed down; the newcomer descends iff its box touches
// exactly one child (inclusive test, aabb.rs:49-60), else it stays (octree.rs:82-104).  The per-pixel result
// depends on this structure (ray.rs:116-167 is not an exact nearest-hit query), so nothing here is "improved".
//
// The descent is written as a loop over a growing node table; the GPU consumes the flattened form directly.
#include <chrono>
#include <cmath>
#include <functional>

#include "model.hpp"
#include "parallel.hpp"

namespace rrt {
namespace {

inline Box triangle_box(const Triangle& t) {   // Aabb::from_triangle, aabb.rs:25-47 (f64::min/max == fmin/fmax)
    Box b;
    b.lo[0] = std::fmin(t.v1.x, std::fmin(t.v2.x, t.v3.x)); b.hi[0] = std::fmax(t.v1.x, std::fmax(t.v2.x, t.v3.x));
    b.lo[1] = std::fmin(t.v1.y, std::fmin(t.v2.y, t.v3.y)); b.hi[1] = std::fmax(t.v1.y, std::fmax(t.v2.y, t.v3.y));
    b.lo[2] = std::fmin(t.v1.z, std::fmin(t.v2.z, t.v3.z)); b.hi[2] = std::fmax(t.v1.z, std::fmax(t.v2.z, t.v3.z));
    return b;
}

inline bool touches(const Box& a, const Box& b) {   // Aabb::intersects, aabb.rs:49-60
    for (int k = 0; k < 3; k++)
        if (a.hi[k] < b.lo[k] || a.lo[k] > b.hi[k]) return false;
    return true;
}

// The reference tests the newcomer's box against each of the 8 child boxes (octree.rs:82-104, aabb.rs:49-60).  A triangle that reaches a node
// touches that node's box (checked at the root, true by induction below it), and the children are the node's box cut at its three mid planes, so
// per axis "touches the lower half" is exactly !(tb.lo > mid) and "touches the upper half" is exactly !(tb.hi < mid) -- the other two comparisons of
// the inclusive box test are the ones the parent already passed (NaN coordinates included: every comparison is written as the reference's negation).
// 6 comparisons per level instead of 48; the number of touched children is the product of the per-axis counts.
struct Builder {
    std::vector<Box> box;
    std::vector<uint32_t> first_child, tri_count, own_count;
    std::vector<double> mid;                      // 3 per node, valid for internal nodes
    std::vector<uint32_t> tri_node;               // node whose `triangles` Vec holds triangle i (kNowhere: dropped outside the root)
    static constexpr uint32_t kNowhere = 0xFFFFFFFFu;

    uint32_t add_node(const Box& b) {
        box.push_back(b); first_child.push_back(0); tri_count.push_back(0); own_count.push_back(0); mid.insert(mid.end(), 3, 0.0);
        return (uint32_t)box.size() - 1;
    }

    void split(uint32_t node) {   // Octree::subdivide, octree.rs:121-241
        const Box p = box[node];
        double m[3];
        for (int k = 0; k < 3; k++) m[k] = p.lo[k] + (p.hi[k] - p.lo[k]) / 2.0;   // octree.rs:136-138, 142-146
        // (x-half, y-half, z-half) per child in the reference's order; 0 = [lo,mid], 1 = [mid,hi]
        static const int half[8][3] = {{0, 0, 0}, {0, 0, 1}, {1, 0, 1}, {1, 0, 0}, {0, 1, 0}, {0, 1, 1}, {1, 1, 1}, {1, 1, 0}};
        const uint32_t base = (uint32_t)box.size();
        for (int c = 0; c < 8; c++) {
            Box b;
            for (int k = 0; k < 3; k++) {
                b.lo[k] = half[c][k] ? m[k] : p.lo[k];
                b.hi[k] = half[c][k] ? p.hi[k] : m[k];
            }
            add_node(b);
        }
        first_child[node] = base;
        for (int k = 0; k < 3; k++) mid[3 * (size_t)node + k] = m[k];
    }

    void insert(uint32_t tri, const Box& tb) {   // push_triangle + push_at_octant, octree.rs:41-108
        if (!touches(tb, box[0])) return;         // octree.rs:71-73: silently dropped
        static const uint32_t child_of[8] = {0, 1, 4, 5, 3, 2, 7, 6};   // index (x-half << 2 | y-half << 1 | z-half) -> child number of `half` above
        uint32_t node = 0;
        for (uint32_t level = 1;; level++) {
            // (every level below RRT_MAX_OCTREE_DEPTH - 1 that a triangle reaches exists or is about to: a chain of coincident triangles would otherwise
            //  be followed to its end -- quadratic in their number -- before the depth check could refuse it)
            if (level > RRT_MAX_OCTREE_DEPTH) throw Error{RRT_ERR_DEPTH, "octree depth exceeds RRT_MAX_OCTREE_DEPTH (" + std::to_string(RRT_MAX_OCTREE_DEPTH) + "): coincident triangles? (octree.rs:79-92)"};
            tri_count[node] += 1;
            const bool leaf = first_child[node] == 0;
            if (leaf && own_count[node] == 0) { own_count[node] = 1; tri_node[tri] = node; return; }
            if (leaf) split(node);
            const double* m = &mid[3 * (size_t)node];
            const bool lx = !(tb.lo[0] > m[0]), hx = !(tb.hi[0] < m[0]), ly = !(tb.lo[1] > m[1]), hy = !(tb.hi[1] < m[1]), lz = !(tb.lo[2] > m[2]), hz = !(tb.hi[2] < m[2]);
            const int n_touch = ((int)lx + (int)hx) * ((int)ly + (int)hy) * ((int)lz + (int)hz);
            if (n_touch != 1) { own_count[node] += 1; tri_node[tri] = node; return; }
            node = first_child[node] + child_of[((uint32_t)hx << 2) | ((uint32_t)hy << 1) | (uint32_t)hz];
        }
    }
};

}  // namespace

void build_octree(const Triangle* tris, size_t n_tris, const Box& root, FlatOctree& out) {
    Builder b;
    b.tri_node.assign(n_tris, Builder::kNowhere);
    b.add_node(root);                                 // Octree::new, octree.rs:23-39
    std::vector<Box> tb(n_tris);
    parallel_ranges(n_tris, 1 << 15, [&](size_t lo, size_t hi, size_t) { for (size_t i = lo; i < hi; i++) tb[i] = triangle_box(tris[i]); });
    // One triangle after the other, as the reference does.  (The insertion is not serial by nature -- scene_build.hip builds the same tree level by
    // level on the GPU, which is what rrt_raytracer_create uses; this loop serves the host-side getters and RRT_FLAG_HOST_SETUP.)
    for (uint32_t i = 0; i < n_tris; i++) b.insert(i, tb[i]);

    const uint32_t n = (uint32_t)b.box.size();
    out.box = std::move(b.box);
    out.first_child = std::move(b.first_child);
    out.tri_count = std::move(b.tri_count);
    // every node's `triangles` Vec in insertion order == increasing triangle index: a counting sort by node
    out.own_off.assign(n + 1, 0);
    for (uint32_t i = 0; i < n; i++) out.own_off[i + 1] = out.own_off[i] + b.own_count[i];
    out.own_idx.assign(out.own_off[n], 0);
    {
        std::vector<uint32_t> fill(out.own_off.begin(), out.own_off.end() - 1);
        for (uint32_t i = 0; i < n_tris; i++) if (b.tri_node[i] != Builder::kNowhere) out.own_idx[fill[b.tri_node[i]]++] = i;
    }

    // depth (root = 1): children always have larger ids than their parent, so one forward sweep suffices
    std::vector<uint32_t> depth(n, 0);
    depth[0] = 1; out.max_depth = 1;
    for (uint32_t i = 0; i < n; i++) {
        if (!out.first_child[i]) continue;
        for (uint32_t c = 0; c < 8; c++) depth[out.first_child[i] + c] = depth[i] + 1;
        if (depth[i] + 1 > out.max_depth) out.max_depth = depth[i] + 1;
    }
}

const FlatOctree& host_tree(const Model& m) {
    std::lock_guard<std::mutex> lk(m.tree_mu);
    if (!m.tree_ready) {
        const auto t0 = std::chrono::steady_clock::now();
        FlatOctree t;
        build_octree(m.triangles.data(), m.triangles.size(), m.root, t);
        if (t.max_depth > RRT_MAX_OCTREE_DEPTH) throw Error{RRT_ERR_DEPTH, "octree depth " + std::to_string(t.max_depth) + " exceeds RRT_MAX_OCTREE_DEPTH"};
        m.tree = std::move(t);
        m.octree_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        m.tree_ready = true;
    }
    return m.tree;
}

}  // namespace rrt
