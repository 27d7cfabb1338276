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
#include <cmath>
#include <functional>

#include "model.hpp"

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

struct Builder {
    std::vector<Box> box;
    std::vector<uint32_t> first_child, tri_count;
    std::vector<std::vector<uint32_t>> own;

    uint32_t add_node(const Box& b) {
        box.push_back(b); first_child.push_back(0); tri_count.push_back(0); own.emplace_back();
        return (uint32_t)box.size() - 1;
    }

    void split(uint32_t node) {   // Octree::subdivide, octree.rs:121-241
        const Box p = box[node];
        double mid[3];
        for (int k = 0; k < 3; k++) mid[k] = p.lo[k] + (p.hi[k] - p.lo[k]) / 2.0;   // octree.rs:136-138, 142-146
        // (x-half, y-half, z-half) per child in the reference's order; 0 = [lo,mid], 1 = [mid,hi]
        static const int half[8][3] = {{0, 0, 0}, {0, 0, 1}, {1, 0, 1}, {1, 0, 0}, {0, 1, 0}, {0, 1, 1}, {1, 1, 1}, {1, 1, 0}};
        const uint32_t base = (uint32_t)box.size();
        for (int c = 0; c < 8; c++) {
            Box b;
            for (int k = 0; k < 3; k++) {
                b.lo[k] = half[c][k] ? mid[k] : p.lo[k];
                b.hi[k] = half[c][k] ? p.hi[k] : mid[k];
            }
            add_node(b);
        }
        first_child[node] = base;
    }

    void insert(uint32_t tri, const Box& tb) {   // push_triangle + push_at_octant, octree.rs:41-108
        uint32_t node = 0;
        for (;;) {
            if (!touches(tb, box[node])) return;
            tri_count[node] += 1;
            const bool leaf = first_child[node] == 0;
            if (leaf && own[node].empty()) { own[node].push_back(tri); return; }
            if (leaf) split(node);
            const uint32_t base = first_child[node];
            int n_touch = 0; uint32_t only = 0;
            for (uint32_t c = 0; c < 8; c++)
                if (touches(box[base + c], tb)) { n_touch++; only = base + c; }
            if (n_touch != 1) { own[node].push_back(tri); return; }
            node = only;
        }
    }
};

}  // namespace

void build_octree(const std::vector<Triangle>& tris, const Box& root, FlatOctree& out) {
    Builder b;
    b.add_node(root);                                 // Octree::new, octree.rs:23-39
    for (uint32_t i = 0; i < tris.size(); i++) b.insert(i, triangle_box(tris[i]));

    const uint32_t n = (uint32_t)b.box.size();
    out.box = std::move(b.box);
    out.first_child = std::move(b.first_child);
    out.tri_count = std::move(b.tri_count);
    out.own_off.assign(n + 1, 0);
    out.own_idx.clear();
    for (uint32_t i = 0; i < n; i++) {
        out.own_off[i] = (uint32_t)out.own_idx.size();
        out.own_idx.insert(out.own_idx.end(), b.own[i].begin(), b.own[i].end());
    }
    out.own_off[n] = (uint32_t)out.own_idx.size();

    // depth (root = 1): children always have larger ids than their parent, so one forward sweep suffices
    std::vector<uint32_t> depth(n, 0);
    depth[0] = 1; out.max_depth = 1;
    for (uint32_t i = 0; i < n; i++) {
        if (!out.first_child[i]) continue;
        for (uint32_t c = 0; c < 8; c++) depth[out.first_child[i] + c] = depth[i] + 1;
        if (depth[i] + 1 > out.max_depth) out.max_depth = depth[i] + 1;
    }
}

}  // namespace rrt
