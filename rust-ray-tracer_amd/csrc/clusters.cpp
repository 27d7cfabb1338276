// clusters.cpp -- result-preserving index over each node's OWN triangle list.
//
// The reference tests every triangle of a node's list and keeps the arg-min of t with the first list position winning ties
// (src/collision/ray.rs:116-129).  That is an exact arg-min over a SET, so the triangles may be visited in any order and any
// triangle the ray cannot hit may be skipped, provided ties are broken by the original list position.  Here each list is
// cut into spatially compact clusters of <= kClusterTris triangles (median splits of the centroid bounds), clusters are grouped
// into super-clusters of <= kSuperClusters clusters, and both carry a padded f32 bounding box that the kernel tests with a cheap
// conservative fp32 slab test before touching the triangles.  Padding (kPadFraction of the scene magnitude, boxes rounded
// outward) is >100x the fp32 error of that test; DESIGN.md section 4 states the exactness argument and its one caveat.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <numeric>

#include "device_scene.hpp"
#include "model.hpp"
#include "parallel.hpp"

namespace rrt {
namespace {

constexpr uint32_t kClusterTris = 8;
constexpr uint32_t kSuperClusters = 8;
constexpr size_t kGroupSupers = 8;          // super-clusters per group record
constexpr size_t kGroupThreshold = 24;      // lists with more super-clusters than this get group records
constexpr double kPadFraction = 1.0 / 32768.0;   // 2^-15 of the scene magnitude

struct TriBox { double lo[3], hi[3], c[3]; };

TriBox tri_box(const Triangle& t) {
    TriBox b;
    const double x[3][3] = {{t.v1.x, t.v2.x, t.v3.x}, {t.v1.y, t.v2.y, t.v3.y}, {t.v1.z, t.v2.z, t.v3.z}};
    for (int k = 0; k < 3; k++) {
        b.lo[k] = std::min(x[k][0], std::min(x[k][1], x[k][2]));
        b.hi[k] = std::max(x[k][0], std::max(x[k][1], x[k][2]));
        b.c[k] = (b.lo[k] + b.hi[k]) * 0.5;
        if (b.c[k] != b.c[k]) b.c[k] = 0.0;   // NaN coordinates: keep (centroid, position) a strict total order (the splits below rank by it)
    }
    return b;
}

float round_down(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -FLT_MAX); return std::isfinite(f) ? f : -FLT_MAX; }
float round_up(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, FLT_MAX); return std::isfinite(f) ? f : FLT_MAX; }

// (lo, hi) -> (centre, half-extent), in place, for one record's six floats (device_scene.hpp).  lo/hi are already padded and rounded outward;
// the half-extent is rounded up once more, so the new box contains the old one whatever the rounding of the centre.
void to_centre_half(float* lo, float* hi) {
    for (int k = 0; k < 3; k++) {
        const double L = lo[k], H = hi[k];
        float c, h;
        if (L > H) { c = 0.0f; h = -FLT_MAX; }                                   // empty: never hit
        else if (L <= -FLT_MAX || H >= FLT_MAX) { c = 0.0f; h = FLT_MAX; }     // unbounded (no-cull index): always hit
        else {
            c = (float)((L + H) * 0.5);
            h = round_up(std::max((double)c - L, H - (double)c));
            h = std::nextafterf(h, FLT_MAX);                                   // (the differences above are rounded doubles: one more ulp covers that)
            if (!std::isfinite(c) || !std::isfinite(h)) { c = 0.0f; h = FLT_MAX; }
        }
        lo[k] = c; hi[k] = h;
    }
}
template <class Rec> void boxes_to_centre_half(std::vector<Rec>& v) {
    parallel_ranges(v.size(), 1 << 16, [&](size_t b, size_t e, size_t) { for (size_t i = b; i < e; i++) to_centre_half(v[i].lo, v[i].hi); });
}

// recursively split items[begin,end) (indices into `boxes`) until every part holds <= leaf items; emits [begin,end) ranges in order
void split(std::vector<uint32_t>& items, size_t begin, size_t end, size_t leaf, const std::vector<TriBox>& boxes, std::vector<std::pair<size_t, size_t>>& out) {
    if (end - begin <= leaf) { out.emplace_back(begin, end); return; }
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (size_t i = begin; i < end; i++) for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], boxes[items[i]].c[k]); hi[k] = std::max(hi[k], boxes[items[i]].c[k]); }
    int axis = 0;
    for (int k = 1; k < 3; k++) if (hi[k] - lo[k] > hi[axis] - lo[axis]) axis = k;
    // split at a multiple of `leaf` nearest the middle so that parts fill up
    size_t n = end - begin, parts = (n + leaf - 1) / leaf, left = (parts / 2) * leaf;
    if (left == 0 || left >= n) left = n / 2;
    std::nth_element(items.begin() + begin, items.begin() + begin + left, items.begin() + end, [&](uint32_t a, uint32_t b) {
        const double ca = boxes[a].c[axis], cb = boxes[b].c[axis];
        return ca < cb || (ca == cb && a < b);
    });
    split(items, begin, begin + left, leaf, boxes, out);
    split(items, begin + left, end, leaf, boxes, out);
}

// One node's own list -> super-clusters / clusters / per-triangle boxes, appended to R with slot numbers local to R (a multiple of 8 per node, so
// a range of nodes can be indexed on its own worker and spliced in afterwards).  Also the bounds of the node's own triangles (for the subtree sweep).
struct RangeOut {
    std::vector<DevSuper> supers; std::vector<DevClusterBox> tboxes, cboxes; std::vector<uint32_t> slot_tri, slot_pos;
    bool has_groups = false;
};

void index_node(const Model& m, size_t node, bool enable_cull, double pad, RangeOut& R, uint32_t& sup_begin, uint32_t& sup_count, double own_lo[3], double own_hi[3],
                std::vector<TriBox>& boxes, std::vector<uint32_t>& items) {
    const FlatOctree& T = host_tree(m);
    const uint32_t b = T.own_off[node], e = T.own_off[node + 1];
    sup_begin = (uint32_t)R.supers.size(); sup_count = 0;
    for (int k = 0; k < 3; k++) { own_lo[k] = DBL_MAX; own_hi[k] = -DBL_MAX; }
    if (b == e) return;
    const uint32_t n = e - b;
    boxes.resize(n); items.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        boxes[i] = tri_box(m.triangles[T.own_idx[b + i]]);
        for (int k = 0; k < 3; k++) { own_lo[k] = std::min(own_lo[k], boxes[i].lo[k]); own_hi[k] = std::max(own_hi[k], boxes[i].hi[k]); }
    }
    std::iota(items.begin(), items.end(), 0u);

    std::vector<std::pair<size_t, size_t>> sup_ranges;
    if (enable_cull) split(items, 0, n, (size_t)kClusterTris * kSuperClusters, boxes, sup_ranges);
    else for (size_t s0 = 0; s0 < n; s0 += (size_t)kClusterTris * kSuperClusters) sup_ranges.emplace_back(s0, std::min<size_t>(n, s0 + kClusterTris * kSuperClusters));   // list order
    const size_t node_first_super = R.supers.size();
    std::vector<std::pair<size_t, size_t>> cl_ranges;
    for (auto [sb, se] : sup_ranges) {
        cl_ranges.clear();
        if (enable_cull) split(items, sb, se, kClusterTris, boxes, cl_ranges);
        else for (size_t c0 = sb; c0 < se; c0 += kClusterTris) cl_ranges.emplace_back(c0, std::min<size_t>(se, c0 + kClusterTris));
        DevSuper S{};
        S.tri_begin = (uint32_t)R.slot_tri.size();                       // a multiple of 8 by construction
        S.tri_count = (uint32_t)(se - sb);
        for (int k = 0; k < 3; k++) { S.lo[k] = FLT_MAX; S.hi[k] = -FLT_MAX; }
        for (size_t ci = 0; ci < cl_ranges.size(); ci++) {
            auto [cb, ce] = cl_ranges[ci];
            // every cluster but the last of a super-cluster must be full, so that cluster c starts at slot tri_begin + 8c
            if (ci + 1 < cl_ranges.size() && ce - cb != kClusterTris) throw Error{RRT_ERR_INVALID_ARG, "internal: cluster split is not 8-aligned"};
            std::sort(items.begin() + cb, items.begin() + ce);          // list order inside a cluster
            DevClusterBox C{};
            double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
            for (size_t i = cb; i < ce; i++) {
                const uint32_t it = items[i];
                for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], boxes[it].lo[k]); hi[k] = std::max(hi[k], boxes[it].hi[k]); }
                R.slot_tri.push_back(T.own_idx[b + it]);
                R.slot_pos.push_back(it);                                // position in the node's `triangles` Vec (ray.rs:119)
                DevClusterBox TB{};
                for (int k = 0; k < 3; k++) {
                    TB.lo[k] = enable_cull ? round_down(boxes[it].lo[k] - pad) : -FLT_MAX;
                    TB.hi[k] = enable_cull ? round_up(boxes[it].hi[k] + pad) : FLT_MAX;
                }
                R.tboxes.push_back(TB);
            }
            for (int k = 0; k < 3; k++) {
                C.lo[k] = enable_cull ? round_down(lo[k] - pad) : -FLT_MAX;
                C.hi[k] = enable_cull ? round_up(hi[k] + pad) : FLT_MAX;
                S.lo[k] = std::min(S.lo[k], C.lo[k]); S.hi[k] = std::max(S.hi[k], C.hi[k]);
            }
            R.cboxes.push_back(C);
        }
        while (R.slot_tri.size() % kClusterTris) {                       // pad to the next cluster boundary; a padding slot's box is empty (never hit)
            R.slot_tri.push_back(kPadSlot); R.slot_pos.push_back(0);
            DevClusterBox TB{};
            for (int k = 0; k < 3; k++) { TB.lo[k] = FLT_MAX; TB.hi[k] = -FLT_MAX; }
            R.tboxes.push_back(TB);
        }
        R.supers.push_back(S);
    }
    // A long list (the root of a large soup holds thousands of straddlers) gets one more level: every run of kGroupSupers consecutive
    // super-clusters -- spatially compact, they come out of the median splits in order -- is preceded by a GROUP record (tri_count = 0,
    // tri_begin = number of super-clusters it covers) carrying the union of their boxes.  The lane-filter kernel skips a group no lane can
    // reach; the boxes-in-lanes paths ignore group records.
    const size_t n_sup = R.supers.size() - node_first_super;
    if (enable_cull && n_sup > kGroupThreshold) {
        R.has_groups = true;
        std::vector<DevSuper> plain(R.supers.begin() + node_first_super, R.supers.end());
        R.supers.resize(node_first_super);
        for (size_t g0 = 0; g0 < plain.size(); g0 += kGroupSupers) {
            const size_t g1 = std::min(plain.size(), g0 + kGroupSupers);
            DevSuper G{};
            for (int k = 0; k < 3; k++) { G.lo[k] = FLT_MAX; G.hi[k] = -FLT_MAX; }
            for (size_t i = g0; i < g1; i++) for (int k = 0; k < 3; k++) { G.lo[k] = std::min(G.lo[k], plain[i].lo[k]); G.hi[k] = std::max(G.hi[k], plain[i].hi[k]); }
            G.tri_begin = (uint32_t)(g1 - g0); G.tri_count = 0;
            R.supers.push_back(G);
            R.supers.insert(R.supers.end(), plain.begin() + g0, plain.begin() + g1);
        }
    }
    sup_count = (uint32_t)R.supers.size() - sup_begin;
}

}  // namespace

void build_clusters(const Model& m, bool enable_cull, ClusterSet& out) {
    const FlatOctree& T = host_tree(m);
    const size_t n_nodes = T.box.size();
    out = ClusterSet{};
    out.node_sup_begin.assign(n_nodes, 0); out.node_sup_count.assign(n_nodes, 0);
    // Single-triangle leaves below the root (a non-empty leaf never holds more than one triangle, octree.rs:77-92): the lane-filter kernel
    // tests such a child's triangle while it is at the PARENT (render.hip, "leaf children") and never enters the leaf.  For that the triangle
    // gets a SECOND slot in a dense run after all list slots, in node-id order (siblings' triangles are adjacent in memory); the leaf keeps its
    // ordinary own list as well, which the bundle-filter kernel (that enters leaves like any node) uses.  Both slots carry the same geometry and
    // attributes.  The frame word that carries the leaf-hit mask keeps 24 bits for first_child, hence the node-count bound.
    out.node_leaf_slot.assign(n_nodes, kPadSlot);
    const bool inline_leaves = n_nodes < (1u << 24);
    out.inline_leaves = inline_leaves;
    auto is_inline_leaf = [&](size_t node) { return inline_leaves && node >= 1 && T.first_child[node] == 0 && T.own_off[node + 1] - T.own_off[node] == 1; };

    double mag = 0;
    for (int k = 0; k < 3; k++) mag = std::max(mag, std::max(std::fabs(m.root.lo[k]), std::fabs(m.root.hi[k])));
    out.scene_magnitude = mag;
    const double pad = mag * kPadFraction;
    out.pad = pad;

    // ---- own lists: every node is independent, so contiguous node ranges are indexed on the host's cores and spliced together in node order
    // (the result does not depend on the number of workers: every node's slots are a multiple of 8 and are numbered within its range first)
    std::vector<double> slo(3 * n_nodes, DBL_MAX), shi(3 * n_nodes, -DBL_MAX);
    const size_t max_parts = host_threads();
    std::vector<RangeOut> parts(max_parts);
    std::vector<std::pair<size_t, size_t>> part_nodes(max_parts, {0, 0});
    parallel_ranges(n_nodes, 4096, [&](size_t nb, size_t ne, size_t p) {
        RangeOut& R = parts[p];
        part_nodes[p] = {nb, ne};
        R.slot_tri.reserve(T.own_off[ne] - T.own_off[nb] + 8 * (ne - nb) / 4);
        std::vector<TriBox> boxes; std::vector<uint32_t> items;
        for (size_t node = nb; node < ne; node++)
            index_node(m, node, enable_cull, pad, R, out.node_sup_begin[node], out.node_sup_count[node], &slo[3 * node], &shi[3 * node], boxes, items);
    });
    // splice: sizes first, then every range copies itself into place
    std::vector<size_t> slot_base(max_parts + 1, 0), sup_base(max_parts + 1, 0);
    for (size_t p = 0; p < max_parts; p++) { slot_base[p + 1] = slot_base[p] + parts[p].slot_tri.size(); sup_base[p + 1] = sup_base[p] + parts[p].supers.size(); out.has_groups = out.has_groups || parts[p].has_groups; }
    const size_t n_slots = slot_base[max_parts];
    out.supers.resize(sup_base[max_parts]); out.slot_tri.resize(n_slots); out.slot_pos.resize(n_slots); out.tboxes.resize(n_slots); out.cboxes.resize(n_slots / kClusterTris);
    parallel_ranges(max_parts, 1, [&](size_t pb, size_t pe, size_t) {
        for (size_t p = pb; p < pe; p++) {
            RangeOut& R = parts[p];
            for (DevSuper& S : R.supers) if (S.tri_count != 0) S.tri_begin += (uint32_t)slot_base[p];      // (a group record's tri_begin counts super-clusters)
            for (size_t node = part_nodes[p].first; node < part_nodes[p].second; node++) out.node_sup_begin[node] += (uint32_t)sup_base[p];
            std::copy(R.supers.begin(), R.supers.end(), out.supers.begin() + sup_base[p]);
            std::copy(R.slot_tri.begin(), R.slot_tri.end(), out.slot_tri.begin() + slot_base[p]);
            std::copy(R.slot_pos.begin(), R.slot_pos.end(), out.slot_pos.begin() + slot_base[p]);
            std::copy(R.tboxes.begin(), R.tboxes.end(), out.tboxes.begin() + slot_base[p]);
            std::copy(R.cboxes.begin(), R.cboxes.end(), out.cboxes.begin() + slot_base[p] / kClusterTris);
            R = RangeOut{};
        }
    });
    for (int i = 0; i < 8; i++) { out.cboxes.push_back(DevClusterBox{}); out.tboxes.push_back(DevClusterBox{}); }   // spare records: box bursts never leave the buffers
    out.n_list_slots = (uint32_t)out.slot_tri.size();
    for (size_t node = 1; node < n_nodes; node++) {
        if (!is_inline_leaf(node)) continue;
        out.node_leaf_slot[node] = (uint32_t)out.slot_tri.size();
        out.slot_tri.push_back(T.own_idx[T.own_off[node]]); out.slot_pos.push_back(0);
    }

    // ---- tight bounds of every subtree (all triangles counted by triangle_count, octree.rs:75), as padded f32 boxes grouped by sibling set.
    // A child whose subtree the ray cannot reach returns None (ray.rs:112-167 finds no triangle), exactly like an empty child, so the walk may
    // drop it from the candidate list before the exact slab test; children always have larger ids than their parent, so one reverse sweep
    // accumulates the bounds bottom-up (slo/shi hold each node's own triangles' bounds from the pass above).
    for (size_t node = n_nodes; node-- > 0;) {
        if (T.first_child[node]) for (uint32_t c = T.first_child[node]; c < T.first_child[node] + 8; c++)
            for (int k = 0; k < 3; k++) { slo[3 * node + k] = std::min(slo[3 * node + k], slo[3 * c + k]); shi[3 * node + k] = std::max(shi[3 * node + k], shi[3 * c + k]); }
    }
    out.child_boxes.assign(n_nodes > 1 ? n_nodes - 1 : 0, DevClusterBox{});           // node id c >= 1 -> child_boxes[c - 1]; siblings are 8 consecutive records
    parallel_ranges(n_nodes > 1 ? n_nodes - 1 : 0, 1 << 16, [&](size_t b0, size_t e0, size_t) {
        for (size_t c = b0 + 1; c < e0 + 1; c++) {
            DevClusterBox& B = out.child_boxes[c - 1];
            const bool empty = T.tri_count[c] == 0 || slo[3 * c] > shi[3 * c];
            for (int k = 0; k < 3; k++) {
                B.lo[k] = !enable_cull ? -FLT_MAX : empty ? FLT_MAX : round_down(slo[3 * c + k] - pad);
                B.hi[k] = !enable_cull ? FLT_MAX : empty ? -FLT_MAX : round_up(shi[3 * c + k] + pad);
            }
        }
    });
    for (int i = 0; i < 8; i++) out.child_boxes.push_back(DevClusterBox{});
    // ---- device form of every box record (the spare all-zero records become the point box at the origin: they are only ever read past a count)
    boxes_to_centre_half(out.supers); boxes_to_centre_half(out.cboxes); boxes_to_centre_half(out.tboxes); boxes_to_centre_half(out.child_boxes);
}

namespace {
}  // namespace

// ---- exactness guard ------------------------------------------------------------------------------------------------------------------
// The box filters drop a (ray, triangle) pair when the ray misses the triangle's padded box.  That is exact as long as a pair the reference's
// Moller-Trumbore (ray.rs:56-94) ACCEPTS has its computed hit point within the padding of the true line -- which fails only when a = e1.(d x e2)
// is rounding noise, i.e. the ray lies in the triangle's plane.  With eps = 2^-53, R = |o - v1| + max|e|, sin(phi) = |e1 x e2| / (|e1||e2|):
//   * the computed u, v, t carry absolute errors <= ~64 eps R |d| |e1||e2| / |a| relative to the triangle's extent, so the computed hit point
//     is within pad/2 of the line whenever |sin(angle(d, plane))| > alpha = 8 * 64 eps R / (pad sin(phi))           (not near-parallel: safe);
//   * for a near-parallel ray (below alpha) the reference can only accept if also |s.(d x e2)| <= |a| and |d.(s x e1)| <= |a| up to rounding,
//     which bounds the distance of the ray's ORIGIN from the plane by delta = 2 (alpha R + 64 eps R) / sin(phi).
// So a ray whose origin is farther than delta from a triangle's plane can never be dropped wrongly.  For the raytracer's origin (every primary
// ray) the triangles within delta are found here, once; a ray from the origin that is within alpha of parallel to one of their planes runs with
// the filters off (render.hip, origin_ray_in_suspect_plane).  Degenerate and sliver triangles (sin(phi) -> 0) get alpha >= 1: every direction.
void find_origin_suspects(const Model& m, const double origin[3], double pad, std::vector<DevSuspect>& out) {
    out.clear();
    if (!(pad > 0)) return;
    const double eps = 0x1p-53;
    const std::vector<uint32_t>& idx = host_tree(m).own_idx;
    std::vector<std::vector<DevSuspect>> found(host_threads());
    parallel_ranges(idx.size(), 1 << 15, [&](size_t ib, size_t ie, size_t part) {
    std::vector<DevSuspect>& out = found[part];
    for (size_t ii = ib; ii < ie; ii++) {
        const Triangle& t = m.triangles[idx[ii]];
        const double e1[3] = {t.v2.x - t.v1.x, t.v2.y - t.v1.y, t.v2.z - t.v1.z}, e2[3] = {t.v3.x - t.v1.x, t.v3.y - t.v1.y, t.v3.z - t.v1.z};
        const double s[3] = {origin[0] - t.v1.x, origin[1] - t.v1.y, origin[2] - t.v1.z};
        const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
        const double ln = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), ls = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
        if (!(l1 > 0) || !(l2 > 0)) continue;             // a zero edge makes a == 0 exactly for every ray: rejected as parallel (ray.rs:66)
        if (!std::isfinite(l1 + l2 + ls)) continue;        // non-finite geometry: the filter is off for such magnitudes anyway (cull_limit)
        const double R = ls + std::max(l1, l2);
        const double sinphi = ln / (l1 * l2);
        double alpha, delta;
        if (!(sinphi > 1e-300)) { alpha = 2.0; delta = INFINITY; }
        else { alpha = 8.0 * 64.0 * eps * R / (pad * sinphi); delta = 2.0 * (alpha * R + 64.0 * eps * R) / sinphi; }
        const double rho = ln > 0 ? std::fabs(s[0] * n[0] + s[1] * n[1] + s[2] * n[2]) / ln : 0.0;
        if (!(rho <= delta)) continue;
        DevSuspect q{};
        for (int k = 0; k < 3; k++) q.n[k] = ln > 0 ? n[k] / ln : 0.0;
        q.alpha2 = alpha >= 1.0 ? 4.0 : alpha * alpha;
        out.push_back(q);
    }
    });
    for (auto& f : found) out.insert(out.end(), f.begin(), f.end());
}

}  // namespace rrt
