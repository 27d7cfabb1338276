// render.hip -- the per-pixel hot path of conor722/rust-ray-tracer as hand-written HIP for gfx950 (MI355X).
//
// What runs here (reference file:line):
//   Ray::intersect_aabb                       src/collision/ray.rs:21-54
//   Ray::intersect_with_triangle              src/collision/ray.rs:56-94     (Moller-Trumbore, f64, no culling)
//   Ray::intersect_with_octant_with_max_t     src/collision/ray.rs:104-168   (the "own list, then first child that hits" walk)
//   RayTracer::get_ray_colour(_recursive)     src/scene/raytracer.rs:29-112
//   get_normal_at_intersection                src/scene/raytracer.rs:114-162
//   triangle_exists_between_points            src/scene/raytracer.rs:164-188
//   compute_lighting_intensity/diffuse/spec   src/scene/raytracer.rs:192-304
//   Scene::draw_scene pixel grid + put_pixel  src/scene/engine.rs:186-255, 146-158; Color::mix entities.rs:49-69
//
// Arithmetic is IEEE f64, op for op in the reference's order, compiled with -ffp-contract=off (Rust never fuses);
// f64 divide and sqrt are correctly rounded on both sides; only pow() (raytracer.rs:295) may differ by an ulp.
//
// Execution model (MI355X-first, not a translation of the recursive CPU code):
//   * one lane = one ray; one 64-lane wave = the 8x8 sub-sample rays of a 4x4-pixel tile; one wave per workgroup.
//   * the recursion of ray.rs:104-168 is an explicit per-lane stack in LDS (own_t, own_slot, sorted-children word,
//     first_child per level), laid out [level][field][lane] so every access is bank-conflict free.
//   * the wave walks the octree NODE-COHERENTLY: each step picks one pending node (the first pending lane's), and
//     every lane parked at that node processes it together.  Node record, the node's triangles and its children's
//     boxes are then wave-uniform, so they are fetched with SCALAR loads (s_load via the constant address space)
//     into SGPRs and cost no vector-memory traffic; VALU does only the f64 math.
//   * primary, reflection and shadow rays share ONE traversal call site: each lane runs a small state machine
//     (segment ray -> per-light shadow rays -> reflection) so lanes in different phases still traverse together.
#include <hip/hip_runtime.h>

#include <cmath>

#include "device_scene.hpp"

// The library compiles this file THREE times (csrc/Makefile), same device functions, different code-generation switches per group of kernels:
//   -DRRT_TU=1  the bundle-filter frame kernel, detile_kernel and the launchers (issue-bound: max-ILP scheduler);
//   -DRRT_TU=3  the lane-filter and ray-walk frame kernels (as 1, plus the structurizer / load-store-vectorizer switches that gain the 1 M soup 5 %
//               and cost the bundle-filter kernel 2.6 % on the teapot);
//   -DRRT_TU=2  the per-ray kernels ray_colour_kernel, intersect_kernel (default scheduler: max-ILP costs scattered rays 17 %).
// Without RRT_TU: everything in one unit (developer builds, tools).
#ifndef RRT_TU
#define RRT_TU 0
#endif
#define RRT_TU_FRAME (RRT_TU == 0 || RRT_TU == 1)      /* bundle-filter render kernel, detile, launch_render, preload */
#define RRT_TU_LANE (RRT_TU == 0 || RRT_TU == 3)       /* lane-filter and ray-walk render kernels */
#define RRT_TU_RAYS (RRT_TU == 0 || RRT_TU == 2)

namespace rrt {
namespace {

#define RRT_CONSTANT __attribute__((address_space(4)))
constexpr double kEps = 2.220446049250313e-16;   // f64::EPSILON, ray.rs:66,89
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr double kInf = __builtin_huge_val();

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }   // engine.rs:16-26
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }   // engine.rs:36-46
__device__ __forceinline__ V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }     // engine.rs:48-58
__device__ __forceinline__ V3 operator/(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }     // engine.rs:72-82
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }                              // engine.rs:60-70
__device__ __forceinline__ double dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); } // engine.rs:85-87
__device__ __forceinline__ double length(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }    // engine.rs:89-91
__device__ __forceinline__ V3 cross(V3 a, V3 b) {                                                     // engine.rs:93-99
    return mk(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
// (a.x/b, a.y/b, a.z/b), each correctly rounded exactly as the `/` operator rounds it.  hipcc lowers one f64 divide to
//   s0 = div_scale(b,b,a); r = rcp(s0); two Newton steps on r; s1 = div_scale(a,b,a); q = s1*r; q = div_fmas(fma(-s0,q,s1), r, q); div_fixup(q,b,a)
// and the first half depends on the denominator only whenever v_div_scale leaves its operands alone (every exponent within +-500 here).
// Sharing it between the three numerators is therefore the SAME instruction sequence per quotient, 18 instructions instead of 39.
// (v_div_scale_f64 leaves both operands alone and clears VCC when the denominator is normal with 1/den normal, the numerator is not tiny
// (exponent > -970), num/den is normal and exponent(num) - exponent(den) < 768; a zero numerator makes it return NaN, but v_div_fixup_f64 then
// returns the signed zero whatever the quotient register holds.  |den| in [2^-500, 2^500] and |num| in [2^-252, 2^250] or 0 satisfy all of it.)
__device__ __forceinline__ bool div_plain(double x) { const double a = fabs(x); return x == 0.0 || (a > 0x1p-250 && a < 0x1p250); }
__device__ __forceinline__ bool den_plain(double x) { const double a = fabs(x); return a > 0x1p-500 && a < 0x1p500; }
__device__ __forceinline__ V3 div3(V3 a, double b) {
    if (!(den_plain(b) && div_plain(a.x) && div_plain(a.y) && div_plain(a.z))) return mk(a.x / b, a.y / b, a.z / b);
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
    double qx = a.x * r, qy = a.y * r, qz = a.z * r;
    qx = __builtin_fma(__builtin_fma(-b, qx, a.x), r, qx);
    qy = __builtin_fma(__builtin_fma(-b, qy, a.y), r, qy);
    qz = __builtin_fma(__builtin_fma(-b, qz, a.z), r, qz);
    return mk(__builtin_amdgcn_div_fixup(qx, b, a.x), __builtin_amdgcn_div_fixup(qy, b, a.y), __builtin_amdgcn_div_fixup(qz, b, a.z));
}
__device__ __forceinline__ V3 normalised(V3 a) { return div3(a, length(a)); }                          // engine.rs:101-103

// The same sharing across a whole walk: every slab quotient (bound - o)/d of ray.rs:22-27 divides by one of the ray's three direction
// components, so the denominator half of the divide (rcp + two Newton steps) is done once per ray and axis, and each quotient is the
// numerator half only -- mul, fma, fma, div_fixup: the instructions hipcc emits for `/`, hence the same bits.  `plain` says that no
// v_div_scale in those divides would rescale: d plain as a denominator, every origin component 0 or in [2^-200, 2^200], and (host-checked,
// DevScene::bounds_plain) every node plane 0 or in [2^-200, 2^200] -- a non-zero difference of two such numbers is at least 2^-252 in
// magnitude.  Lanes that are not plain use the ordinary divide.
struct RayRcp { double rx, ry, rz; bool plain; };
__device__ __forceinline__ double rcp_refined(double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
    return r;
}
__device__ __forceinline__ bool org_plain(double x) { const double a = fabs(x); return x == 0.0 || (a > 0x1p-200 && a < 0x1p200); }
__device__ __forceinline__ RayRcp make_ray_rcp(V3 o, V3 d, bool bounds_plain) {
    RayRcp R;
    R.plain = bounds_plain && den_plain(d.x) && den_plain(d.y) && den_plain(d.z) && org_plain(o.x) && org_plain(o.y) && org_plain(o.z);
    R.rx = rcp_refined(d.x); R.ry = rcp_refined(d.y); R.rz = rcp_refined(d.z);
    return R;
}
// num / den for a plain pair, given r = rcp_refined(den)
__device__ __forceinline__ double quot(double num, double den, double r) {
    double q = num * r;
    q = __builtin_fma(__builtin_fma(-den, q, num), r, q);
    return __builtin_amdgcn_div_fixup(q, den, num);
}

// The scene descriptor arrives in the kernarg segment and hipcc fetches its pointers with one s_load_dwordx16; a value that is a slice of such a
// 16-SGPR tuple is spilled and reloaded WITH the tuple (16 v_writelane / v_readlane per use of one pointer once SGPRs run short, which they do in
// the own-list loops: box bursts take 32-64 of the 102).  An empty asm makes each pointer a value of its own, 2 SGPRs, spilled alone.
template <class T> __device__ __forceinline__ T* own_sgprs(T* p) {
#ifdef RRT_NO_OWN_SGPRS
    return p;
#else
    unsigned long long r = (unsigned long long)p;
    asm volatile("" : "+s"(r));
    return (T*)r;
#endif
}

// ------------------------------------------------------------------------------------------------ LDS stack
// per wave: levels x 768 B, level record = own_slot[64] u32 | meta[64] u32 | fc[64] u32.  A frame does not keep the t of its own hit: the few
// times an older frame's own hit is compared or returned, t is recomputed from the slot (t_of_slot: same arithmetic, same bits), which
// keeps the stack at 12 bytes per lane and level so that LDS does not cap the number of resident waves.
constexpr uint32_t kLevelBytes = 3 * 64 * 4;
constexpr uint32_t kParkBytes = 64;       // in front of the stack: the wave's ray bundle of the current walk, parked by the lane-filter kernel (see traverse)
struct Stack {
    char* base; uint32_t lane;
    __device__ __forceinline__ float* park() const { return reinterpret_cast<float*>(base - kParkBytes); }
    __device__ __forceinline__ uint32_t& own_slot(uint32_t l) const { return *reinterpret_cast<uint32_t*>(base + l * kLevelBytes + lane * 4); }
    __device__ __forceinline__ uint32_t& meta(uint32_t l) const { return *reinterpret_cast<uint32_t*>(base + l * kLevelBytes + 256 + lane * 4); }
    __device__ __forceinline__ uint32_t& fc(uint32_t l) const { return *reinterpret_cast<uint32_t*>(base + l * kLevelBytes + 512 + lane * 4); }
};

// ------------------------------------------------------------------------------------------------ wave-uniform record loads
// Every record below is fetched with ONE scalar load burst of its full size (s_load_dwordx8/x16) so that a record costs one memory
// round trip; letting the compiler load fields one by one split a 32-byte record into three dependent round trips.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ double mkd(uint32_t lo, uint32_t hi) { return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); }
__device__ __forceinline__ float mkf(uint32_t x) { return __builtin_bit_cast(float, x); }

// The node record in two scalar loads: the 32-byte tail (third split plane, child/list descriptors, flags) that every visit needs at once, and the
// 64-byte head (box and split planes as 8 doubles) that only the exact child slab tests read.  Loading the head after the fp32 reach filter keeps
// its 16 SGPRs out of the way of the child-box bursts (they were spilled to VGPR lanes and reloaded on every visit: 32 VALU instructions) and
// skips the load when no child is reachable.
struct UHead { uint32_t first_child, sup_begin, sup_count, flags, s0_begin, s0_count, leaf_base; double mid2; };
__device__ __forceinline__ UHead load_uhead(const RRT_CONSTANT DevNode* p) {
    const u32x8 b = *(const RRT_CONSTANT u32x8*)((const RRT_CONSTANT char*)p + 64);
    UHead n;
    n.mid2 = mkd(b[0], b[1]); n.first_child = b[2]; n.sup_begin = b[3]; n.sup_count = b[4]; n.flags = b[5]; n.s0_begin = b[6]; n.s0_count = b[5] >> 24; n.leaf_base = b[7];
    return n;
}
struct UPlanes { double lo[3], hi[3], mid[3]; };
// (four independent 16-byte loads, issued together: one round trip like a single 64-byte load, but four register tuples of 4 instead of one of 16 --
// the allocator evicts a tuple WHOLE when it wants two of its registers, which with the 16-tuple cost 16 v_writelane + 16 v_readlane on every visit)
__device__ __forceinline__ UPlanes load_uplanes(const RRT_CONSTANT DevNode* p, double mid2) {
    // (each address passes through an empty asm: otherwise the load/store optimizer fuses the four loads back into one s_load_dwordx16)
    const RRT_CONSTANT u32x4* q = (const RRT_CONSTANT u32x4*)p;
    const u32x4 a = *own_sgprs(q), b = *own_sgprs(q + 1), c = *own_sgprs(q + 2), d = *own_sgprs(q + 3);
    UPlanes n;
    n.lo[0] = mkd(a[0], a[1]); n.lo[1] = mkd(a[2], a[3]); n.lo[2] = mkd(b[0], b[1]);
    n.hi[0] = mkd(b[2], b[3]); n.hi[1] = mkd(c[0], c[1]); n.hi[2] = mkd(c[2], c[3]);
    n.mid[0] = mkd(d[0], d[1]); n.mid[1] = mkd(d[2], d[3]); n.mid[2] = mid2;
    return n;
}

struct UBox { float cx, cy, cz, hx, hy, hz; uint32_t a, b; };   // centre and half-extent (device_scene.hpp); DevSuper (a = tri_begin, b = tri_count) or DevClusterBox
__device__ __forceinline__ UBox mk_ubox(u32x8 r) { UBox x; x.cx = mkf(r[0]); x.cy = mkf(r[1]); x.cz = mkf(r[2]); x.hx = mkf(r[3]); x.hy = mkf(r[4]); x.hz = mkf(r[5]); x.a = r[6]; x.b = r[7]; return x; }
__device__ __forceinline__ UBox load_ubox(const RRT_CONSTANT void* p) { return mk_ubox(*(const RRT_CONSTANT u32x8*)p); }

// One triangle held in SGPRs (wave-uniform): v1 and the two precomputed edges, plus its position in the node's own list.
struct UTri { double v1x, v1y, v1z, e1x, e1y, e1z, e2x, e2y, e2z; uint32_t pos; };
__device__ __forceinline__ UTri load_utri(const RRT_CONSTANT DevTriGeom* g) {
    const u32x16 a = *(const RRT_CONSTANT u32x16*)g;
    const u32x4 b = *(const RRT_CONSTANT u32x4*)((const RRT_CONSTANT char*)g + 64);
    UTri t;
    t.v1x = mkd(a[0], a[1]); t.v1y = mkd(a[2], a[3]); t.v1z = mkd(a[4], a[5]);
    t.e1x = mkd(a[6], a[7]); t.e1y = mkd(a[8], a[9]); t.e1z = mkd(a[10], a[11]);
    t.e2x = mkd(a[12], a[13]); t.e2y = mkd(a[14], a[15]); t.e2z = mkd(b[0], b[1]);
    t.pos = b[2];
    return t;
}

// ------------------------------------------------------------------------------------------------ primitives
// Ray::intersect_with_triangle, ray.rs:56-94, against a wave-uniform triangle (SGPR operands).  Returns t only; the
// winning triangle's (u,v) are recomputed once per hit by mt_full (same arithmetic => same bits).
//
// The reference divides (f = 1.0/a, ray.rs:71) before its first rejection test, and an IEEE f64 divide costs as much as
// everything before it.  Nearly every (ray, triangle) pair is rejected, so the rejections are decided here from the
// NUMERATORS su = s.h, dq = d.q, eq = e2.q -- computed exactly as the reference computes them -- by conservative filters
// that never reject a pair the reference accepts.  With f = fl(1/a) (sign of a; normal, relative error <= 2^-53, whenever
// eps <= |a| < 1e100) and u = fl(f*su), v = fl(f*dq), t = fl(f*eq):
//   (F1) sign(x) != sign(a), |x| > 1e-200, |a| < 1e100        =>  fl(f*x) is a non-zero negative number: u < 0 resp. v < 0.
//   (F2) |x| > |a|*(1+2^-40)                                  =>  |fl(f*x)| > 1: u > 1 or u < 0; v > 1 (=> u+v > 1 as u >= 0) or v < 0.
//   (F3) su, dq of a's sign and |su|+|dq| > |a|*(1+2^-30)     =>  fl(u+v) > 1.
//   (F4) sign(eq) != sign(a)                                  =>  t <= 0, so `t > eps` fails.
// Whatever the filters let through takes the reference's exact path (divide included), so borderline cases are decided by
// the reference's own arithmetic.  NaNs fail every filter comparison and fall through to the exact path.
// A padding slot (all-zero record, clusters.cpp) has a = 0 and is rejected as parallel (ray.rs:66).
constexpr double kC40 = 1.0 + 0x1p-40, kC30 = 1.0 + 0x1p-30;
__device__ __forceinline__ bool mt_uniform(const UTri& g, V3 o, V3 d, double& t_out) {
    const double hx = d.y * g.e2z - d.z * g.e2y;
    const double hy = -(d.x * g.e2z - d.z * g.e2x);
    const double hz = d.x * g.e2y - d.y * g.e2x;
    const double a = (g.e1x * hx + g.e1y * hy) + g.e1z * hz;
    if (a > -kEps && a < kEps) return false;                   // ray.rs:66-69
    const double sx = o.x - g.v1x, sy = o.y - g.v1y, sz = o.z - g.v1z;
    const double su = (sx * hx + sy * hy) + sz * hz;
    const double abs_a = fabs(a), abs_su = fabs(su);
    const bool a_neg = a < 0.0, a_ok = abs_a < 1e100;
    const double lim = abs_a * kC40;
    if ((((su < 0.0) != a_neg) && abs_su > 1e-200 && a_ok) || abs_su > lim) return false;        // F1, F2 on u (ray.rs:75)
    const double qx = sy * g.e1z - sz * g.e1y;
    const double qy = -(sx * g.e1z - sz * g.e1x);
    const double qz = sx * g.e1y - sy * g.e1x;
    const double dq = (d.x * qx + d.y * qy) + d.z * qz;
    const double abs_dq = fabs(dq);
    if ((((dq < 0.0) != a_neg) && abs_dq > 1e-200 && a_ok) || abs_dq > lim) return false;        // F1, F2 on v (ray.rs:82)
    if (((su < 0.0) == a_neg) && ((dq < 0.0) == a_neg) && (abs_su + abs_dq) > abs_a * kC30) return false;   // F3 on u+v (ray.rs:82)
    const double eq = (g.e2x * qx + g.e2y * qy) + g.e2z * qz;
    if ((eq < 0.0) != a_neg) return false;                     // F4 (ray.rs:89)
    // exact path, ray.rs:71-93
    const double f = 1.0 / a;
    const double u = f * su;
    if (u < 0.0 || u > 1.0) return false;                      // ray.rs:75-77
    const double v = f * dq;
    if (v < 0.0 || u + v > 1.0) return false;                  // ray.rs:82-84
    const double t = f * eq;
    t_out = t;
    return t > kEps;                                           // ray.rs:89-93
}

// same test on a per-lane triangle, returning u and v as well
__device__ __forceinline__ bool mt_full(const DevTriGeom* g, V3 o, V3 d, double& t_out, double& u_out, double& v_out) {
    const V3 v1 = ld3(g->v1), e1 = ld3(g->e1), e2 = ld3(g->e2);
    const V3 h = cross(d, e2);
    const double a = dot(e1, h);
    if (a > -kEps && a < kEps) return false;
    const double f = 1.0 / a;
    const V3 s = o - v1;
    const double u = f * dot(s, h);
    if (u < 0.0 || u > 1.0) return false;
    const V3 q = cross(s, e1);
    const double v = f * dot(d, q);
    if (v < 0.0 || u + v > 1.0) return false;
    const double t = f * dot(e2, q);
    t_out = t; u_out = u; v_out = v;
    return t > kEps;
}

// t of a triangle already known to be hit by this ray (the own hit of an older stack frame): ray.rs:60-87 again, same bits as the first time
__device__ __forceinline__ double t_of_slot(const DevTriGeom* g, V3 o, V3 d) {
    const V3 v1 = ld3(g->v1), e1 = ld3(g->e1), e2 = ld3(g->e2);
    const V3 h = cross(d, e2);
    const double f = 1.0 / dot(e1, h);
    const V3 q = cross(o - v1, e1);
    return f * dot(e2, q);
}

// Ray::intersect_aabb, ray.rs:21-54, on one child box given the six quotients t1..t6 = (bound - o)/d of its planes (ray.rs:22-27).
// fmin/fmax == Rust f64::min/max (NaN-ignoring).
__device__ __forceinline__ bool slab_from_quotients(double t1, double t2, double t3, double t4, double t5, double t6, double& t_out) {
    const double tmin = fmax(fmax(fmin(t1, t2), fmin(t3, t4)), fmin(t5, t6));
    const double tmax = fmin(fmin(fmax(t1, t2), fmax(t3, t4)), fmax(t5, t6));
    if (tmax < 0.0) return false;        // ray.rs:39-41
    if (tmin > tmax) return false;       // ray.rs:44-46
    t_out = (tmin < 0.0) ? tmax : tmin;  // ray.rs:49-53
    return true;
}

// ------------------------------------------------------------------------------------------------ own-list index: fp32 box filter
// The ray in fp32 for the conservative box filter over the cluster boxes of clusters.cpp: t = lo*inv - o*inv per slab as one FMA.
// A direction component smaller than 1e-20 is treated as parallel (inv = 1e30): over any t that matters the ray does not move along
// that axis by more than the box padding.  Rays with non-finite or out-of-scale components (and every ray in RRT_FLAG_NO_CULL mode) get
// inv = n = 0: all six slab values are then 0 and every box tests as hit, i.e. the filter is off for that lane.
// The ray as the packed FMAs below want it: register PAIRS {ix, iy}, {nx, ny}, {|ix|, |iy|}, {iz, |iz|}, {nz, 0}.  The parameter is scaled by
// sigma = max|d| / (2 limit): every box of the scene is entered before t' = sigma t reaches 1 (|o| < limit = 4 x the scene magnitude, boxes within
// the scene: at most 1.25 limit along the dominant axis), which lets the near-plane maximum be clamped to [0, 1] by the instruction that forms it.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Ray32 {
    f32x2 i01, n01, a01, izaz, nz0; float sigma;
    __device__ __forceinline__ float ix() const { return i01.x; }
    __device__ __forceinline__ float iy() const { return i01.y; }
    __device__ __forceinline__ float iz() const { return izaz.x; }
};
__device__ __forceinline__ Ray32 make_ray32(V3 o, V3 d, float limit, float half_over_limit, bool enabled) {
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z, dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    const float dmax = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
    // (dmax > 1e-10: a component below 1e-20 is treated as parallel, which is only right while the ray cannot cross the scene -- t <= 2 limit / dmax -- by
    // moving along that axis; dmax < 1e8: the parallel axes' slab values are bounded by 1.25e30 dmax and must stay finite)
    const bool cull = enabled && fabsf(ox) < limit && fabsf(oy) < limit && fabsf(oz) < limit && dmax < limit && dmax > 1e-10f && dmax < 1e8f;
    const float sigma = cull ? dmax * half_over_limit : 0.0f;
    // v_rcp_f32 (1 ulp) is plenty for a filter whose boxes are padded by ~1e-5 of the scene: an IEEE divide would cost ten instructions each
    const float ix = !cull ? 0.0f : (fabsf(dx) < 1e-20f ? 1e30f : __builtin_amdgcn_rcpf(dx)) * sigma;
    const float iy = !cull ? 0.0f : (fabsf(dy) < 1e-20f ? 1e30f : __builtin_amdgcn_rcpf(dy)) * sigma;
    const float iz = !cull ? 0.0f : (fabsf(dz) < 1e-20f ? 1e30f : __builtin_amdgcn_rcpf(dz)) * sigma;
    Ray32 r;
    r.i01.x = ix; r.i01.y = iy; r.izaz.x = iz; r.izaz.y = fabsf(iz);
    r.n01.x = -ox * ix; r.n01.y = -oy * iy; r.nz0.x = -oz * iz; r.nz0.y = 0.0f;
    r.a01.x = fabsf(ix); r.a01.y = fabsf(iy);
    r.sigma = sigma;
    return r;
}
// Near-plane maximum (clamped to [0, 1]) and far-plane minimum of one padded box (centre c, half-extent h) along the ray: per axis the slab values of
// the planes c -/+ h are (c*inv + n) -/+ h*|inv|, the smaller being the near plane whatever the sign of the direction.  The ray cannot reach the box
// at t >= 0 unless tmin <= tmax.  (tmin is only ever clamped DOWN from above 1, which can only let a box through.)
// Per-lane box (ray walk):
__device__ __forceinline__ bool slab32(const UBox& b, const Ray32& r) {
    const float tx = __builtin_fmaf(b.cx, r.i01.x, r.n01.x), ty = __builtin_fmaf(b.cy, r.i01.y, r.n01.y), tz = __builtin_fmaf(b.cz, r.izaz.x, r.nz0.x);
    const float nx = __builtin_fmaf(-b.hx, r.a01.x, tx), fx = __builtin_fmaf(b.hx, r.a01.x, tx);
    const float ny = __builtin_fmaf(-b.hy, r.a01.y, ty), fy = __builtin_fmaf(b.hy, r.a01.y, ty);
    const float nz = __builtin_fmaf(-b.hz, r.izaz.y, tz), fz = __builtin_fmaf(b.hz, r.izaz.y, tz);
    const float tmin = fmaxf(fmaxf(fmaxf(nx, ny), nz), 0.0f);
    const float tmax = fminf(fminf(fx, fy), fz);
    return tmin <= tmax;
}
// Wave-uniform box, its record's first six dwords as three SGPR pairs p0 = {cx, cy}, p1 = {cz, hx}, p2 = {hy, hz}: five packed FMAs give all nine
// values (v_pk_fma_f32 with an SGPR-pair operand issues in the time of ONE v_fma_f32 with an SGPR operand -- 4.5 cycles per SIMD, measured,
// tools/probes/issue_probe.hip; min/max cost as much, hence the clamp in place of a separate max with 0): 7 VALU instructions where the plain form
// above takes 13.
__device__ __forceinline__ void slab32_pk(unsigned long long p0, unsigned long long p1, unsigned long long p2, const Ray32& r, float& tmin, float& tmax) {
    f32x2 T, TZ, X, Y, Z;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(T) : "s"(p0), "v"(r.i01), "v"(r.n01));                                        // {cx ix + nx, cy iy + ny}
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(TZ) : "s"(p1), "v"(r.izaz), "v"(r.nz0));                                     // {cz iz + nz, (unused)}
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_lo:[1,0,0]" : "=v"(X) : "s"(p1), "v"(r.a01), "v"(T));    // tcx -/+ hx |ix|
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(Y) : "s"(p2), "v"(r.a01), "v"(T));    // tcy -/+ hy |iy|
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0] neg_lo:[1,0,0]" : "=v"(Z) : "s"(p2), "v"(r.izaz), "v"(TZ));  // tcz -/+ hz |iz|
    asm("v_max3_f32 %0, %1, %2, %3 clamp" : "=v"(tmin) : "v"(X.x), "v"(Y.x), "v"(Z.x));
    tmax = fminf(fminf(X.y, Y.y), Z.y);
}
__device__ __forceinline__ unsigned long long sgpr_pair(uint32_t lo, uint32_t hi) { return ((unsigned long long)hi << 32) | lo; }
// ---- the box test as ONE asm block (round 3).  Written as one asm statement per instruction (round 2), every dependent pair picked up a compiler
// s_nop: the hazard recognizer cannot see into inline asm, counts an asm statement as zero wait states and assumes the worst (a partial-register
// write) of every asm that produces a VGPR -- 3-4 s_nop per box, 0.36 G per 100 k-soup frame, a quarter of the test's instructions, none of them needed
// by the hardware (full-width VALU results forward with the ordinary interlock).  In one block nothing is inserted.  The temporaries are PINNED
// (v118..v125): sub-registers of a 64-bit operand cannot be named in an asm template, v_max3/v_min3 need the halves of the packed results.
#ifndef RRT_NO_FUSED_BOX
#define RRT_FUSED_BOX 1
#endif
#define RRT_BOX_TMP "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125"
#define RRT_BOX_SLAB                                                                                                                     \
    "v_pk_fma_f32 v[118:119], %[p0], %[i01], %[n01]\n\t"                                          /* T  = {cx ix + nx, cy iy + ny} */   \
    "v_pk_fma_f32 v[120:121], %[p1], %[izaz], %[nz0]\n\t"                                         /* TZ = {cz iz + nz, -}        */   \
    "v_pk_fma_f32 v[122:123], %[p2], %[a01], v[118:119] op_sel:[0,1,1] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"   /* Y = tcy -/+ hy |iy| */  \
    "v_pk_fma_f32 v[118:119], %[p1], %[a01], v[118:119] op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_lo:[1,0,0]\n\t"   /* X = tcx -/+ hx |ix| */  \
    "v_pk_fma_f32 v[120:121], %[p2], %[izaz], v[120:121] op_sel:[1,1,0] op_sel_hi:[1,1,0] neg_lo:[1,0,0]\n\t"  /* Z = tcz -/+ hz |iz| */  \
    "v_max3_f32 v124, v118, v122, v120 clamp\n\t"                                                 /* near-plane maximum in [0, 1] */   \
    "v_min3_f32 v125, v119, v123, v121\n\t"                                                       /* far-plane minimum */
#define RRT_BOX_IN(p0, p1, p2, r) [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2), [i01] "v"(r.i01), [n01] "v"(r.n01), [a01] "v"(r.a01), [izaz] "v"(r.izaz), [nz0] "v"(r.nz0)
// hit mask of one wave-uniform box over the active lanes (== ballot of the per-lane test)
__device__ __forceinline__ unsigned long long box_mask(unsigned long long p0, unsigned long long p1, unsigned long long p2, const Ray32& r) {
    unsigned long long m;
    asm(RRT_BOX_SLAB "v_cmp_le_f32_e64 %[m], v124, v125" : [m] "=s"(m) : RRT_BOX_IN(p0, p1, p2, r) : RRT_BOX_TMP);
    return m;
}
// A burst of boxes: bit i of the lane's accumulator and of the wave's says whether this lane's ray / some lane's ray may hit box i.  The boxes of a
// burst are tested from the LAST to the first and each result is shifted in from below (acc = 2 acc + hit: one v_addc_co_u32 with the compare's mask
// as carry-in; the wave's word likewise with s_addc_u32), so box i ends up at bit i whatever the count.
__device__ __forceinline__ void box_test_shift(unsigned long long p0, unsigned long long p1, unsigned long long p2, const Ray32& r, uint32_t& lane_acc, uint32_t& wave_acc) {
#ifdef RRT_FUSED_BOX
    unsigned long long m;
    asm(RRT_BOX_SLAB
        "v_cmp_le_f32_e64 %[m], v124, v125\n\t"
        "v_addc_co_u32_e64 %[la], vcc, %[la], %[la], %[m]"
        : [la] "+v"(lane_acc), [m] "=&s"(m) : RRT_BOX_IN(p0, p1, p2, r) : "vcc", RRT_BOX_TMP);
    // (the wave's word in a statement of its own: with the scalar accumulator as an in/out operand of the block above the backend fails with
    //  "illegal VGPR to SGPR copy"; an SGPR dependency between two asm statements costs no s_nop)
    asm("s_cmp_lg_u64 %1, 0\n\ts_addc_u32 %0, %0, %0" : "+s"(wave_acc) : "s"(m) : "scc");
#else
    float tmin, tmax;
    slab32_pk(p0, p1, p2, r, tmin, tmax);
    unsigned long long m;
    asm("v_cmp_le_f32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(lane_acc), "=&s"(m) : "v"(tmin), "v"(tmax) : "vcc");
    asm("s_cmp_lg_u64 %1, 0\n\ts_addc_u32 %0, %0, %0" : "+s"(wave_acc) : "s"(m) : "scc");
#endif
}
// the reach filter's form: the result lands at bit k (children are often absent, so nothing is shifted)
template <int k> __device__ __forceinline__ void box_test_bit(unsigned long long p0, unsigned long long p1, unsigned long long p2, const Ray32& r, uint32_t& lane_bits, uint32_t& wave_bits) {
#ifdef RRT_FUSED_BOX
    unsigned long long m; uint32_t t;
    asm(RRT_BOX_SLAB
        "v_cmp_le_f32_e64 %[m], v124, v125\n\t"
        "v_cndmask_b32_e64 v124, 0, 1, %[m]\n\t"
        "v_lshl_or_b32 %[lb], v124, %[k], %[lb]\n\t"
        "s_cmp_lg_u64 %[m], 0\n\t"
        "s_cselect_b32 %[t], %[bit], 0\n\t"
        "s_or_b32 %[wb], %[wb], %[t]"
        : [lb] "+v"(lane_bits), [wb] "+s"(wave_bits), [m] "=&s"(m), [t] "=&s"(t) : RRT_BOX_IN(p0, p1, p2, r), [k] "n"(k), [bit] "n"(1 << k) : "scc", RRT_BOX_TMP);
#else
    float tmin, tmax;
    slab32_pk(p0, p1, p2, r, tmin, tmax);
    const bool h = tmin <= tmax;
    lane_bits |= h ? (1u << k) : 0u;
    wave_bits |= (__builtin_amdgcn_ballot_w64(h) != 0ull) ? (1u << k) : 0u;
#endif
}
__device__ __forceinline__ unsigned long long slab32_u_mask(const UBox& b, const Ray32& r) {      // a wave-uniform box held as a UBox (super-cluster records): hit mask
    const unsigned long long p0 = sgpr_pair(__builtin_bit_cast(uint32_t, b.cx), __builtin_bit_cast(uint32_t, b.cy)), p1 = sgpr_pair(__builtin_bit_cast(uint32_t, b.cz), __builtin_bit_cast(uint32_t, b.hx)),
                             p2 = sgpr_pair(__builtin_bit_cast(uint32_t, b.hy), __builtin_bit_cast(uint32_t, b.hz));
#ifdef RRT_FUSED_BOX
    return box_mask(p0, p1, p2, r);
#else
    float tmin, tmax;
    slab32_pk(p0, p1, p2, r, tmin, tmax);
    return __builtin_amdgcn_ballot_w64(tmin <= tmax);
#endif
}

// Exactness guard: must the index filter stay off for this ray?  Only rays that start at the raytracer's origin can be
// flagged (every primary ray does); the list of suspect planes is empty for all but constructed scenes, so this is one scalar branch.
__device__ __forceinline__ bool origin_ray_in_suspect_plane(const DevScene& S, V3 o, V3 d) {
    if (S.n_suspects == 0u) return false;
    const bool from_origin = o.x == S.origin[0] && o.y == S.origin[1] && o.z == S.origin[2];
    if (S.n_suspects > RRT_MAX_SUSPECTS) return from_origin;
    const double dd = dot(d, d);
    bool in_plane = false;
    for (uint32_t i = 0; i < S.n_suspects; ++i) {
        const DevSuspect q = S.suspects[i];
        const double c = (d.x * q.n[0] + d.y * q.n[1]) + d.z * q.n[2];
        in_plane = in_plane || (c * c <= q.alpha2 * dd);
    }
    return from_origin && in_plane;
}

// ------------------------------------------------------------------------------------------------ developer counters
#ifdef RRT_PROFILE
struct Prof { unsigned long long c[16]; unsigned long long t[8]; unsigned long long last; unsigned long long b[4]; double pad; bool secondary; };
#define PROF_DECL Prof& prof,
#define PROF_ARG prof,
// Counters are per WAVE: whichever lane is the first active one at the increment adds to its own copy, and every lane's copies are summed at the end
// (an increment under divergent control flow therefore counts once per wave that reaches it, whatever lanes are active).
__device__ __forceinline__ bool prof_leader() {
    const unsigned long long e = __builtin_amdgcn_ballot_w64(true);
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(e >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)e, 0u)) == 0u;
}
#ifdef RRT_PROF_HIST   /* histogram build: only the histogram counters (written as 100 + index) count */
#define PROF_ADD(i, x) do { const unsigned long long _v = (unsigned long long)(x); if ((i) >= 100) prof.c[(i) >= 100 ? (i) - 100 : 0] += prof_leader() ? _v : 0ull; } while (0)
#else
#define PROF_ADD(i, x) do { const unsigned long long _v = (unsigned long long)(x); prof.c[i] += prof_leader() ? _v : 0ull; } while (0)   /* x (a ballot, usually) is evaluated by every active lane */
#endif
#define PROF_T(i) do { const unsigned long long _n = __builtin_amdgcn_s_memtime(); prof.t[i] += _n - prof.last; prof.last = _n; } while (0)
#else
#define PROF_DECL
#define PROF_ARG
#define PROF_ADD(i, x) ((void)0)
#define PROF_T(i) ((void)0)
#endif
// Developer build `make band` (-DRRT_PROFILE -DRRT_BAND_COUNT, run with RRT_FLAG_NO_CULL so that every listed pair is tested): counts the (ray,
// triangle) pairs that the reference's Moller-Trumbore ACCEPTS and, of those, the pairs inside the (alpha, delta) band of DESIGN.md section 4 --
// direction within alpha of the triangle's plane AND origin within delta of it -- the only pairs the index's box filters could drop wrongly.
// b[0]/b[1]: accepted / in-band pairs of SECONDARY rays (origin != the raytracer's origin: shadow and reflection rays, unguarded);
// b[2]/b[3]: the same for rays from the origin (guarded: clusters.cpp find_origin_suspects).  tools/band_count.py reports them per BASELINE frame.
#if defined(RRT_PROFILE) && defined(RRT_BAND_COUNT)
__device__ __noinline__ void band_count(Prof& prof, double v1x, double v1y, double v1z, V3 e1, V3 e2, V3 o, V3 d) {
    const double eps = 0x1p-53;
    const V3 s = mk(o.x - v1x, o.y - v1y, o.z - v1z);
    const V3 n = cross(e1, e2);
    const double l1 = length(e1), l2 = length(e2), ln = length(n), ls = length(s), ld = length(d);
    const double R = ls + fmax(l1, l2), sinphi = ln / (l1 * l2);
    double alpha = 2.0, delta = kInf;
    if (sinphi > 1e-300) { alpha = 8.0 * 64.0 * eps * R / (prof.pad * sinphi); delta = 2.0 * (alpha * R + 64.0 * eps * R) / sinphi; }
    const double sin_theta = fabs(dot(d, n)) / (ld * ln), rho = fabs(dot(s, n)) / ln;
    const bool in_band = (alpha >= 1.0 || sin_theta <= alpha) && rho <= delta;
    prof.b[prof.secondary ? 0 : 2] += 1ull;
    if (in_band) prof.b[prof.secondary ? 1 : 3] += 1ull;
}
#define MT_UNIFORM(tri, o, d, t) (mt_uniform(tri, o, d, t) && (band_count(prof, tri.v1x, tri.v1y, tri.v1z, mk(tri.e1x, tri.e1y, tri.e1z), mk(tri.e2x, tri.e2y, tri.e2z), o, d), true))
#else
#define MT_UNIFORM(tri, o, d, t) mt_uniform(tri, o, d, t)
#endif

// ------------------------------------------------------------------------------------------------ traversal
// Wave reductions with in-place DPP: `v_min_* v, v, v row_shr:k` -- a lane whose DPP source is out of range keeps its own value, so six steps
// (row_shr 1,2,4,8, row_bcast 15,31) leave the reduction of all 64 lanes in lane 63.  hipcc's own lowering of the same pattern spends four
// instructions per step (identity move, DPP move, NaN-canonicalise, min).  Inline asm gets no automatic wait states (cdna guide 5.7): a VALU
// write followed by a DPP read of the same VGPR needs 2, hence the s_nop 1 between dependent steps.
#define RRT_DPP_STEP_U32(v, ctrl) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 " ctrl : "+v"(v))
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    RRT_DPP_STEP_U32(v, "row_shr:1 row_mask:0xf bank_mask:0xf");
    RRT_DPP_STEP_U32(v, "row_shr:2 row_mask:0xf bank_mask:0xf");
    RRT_DPP_STEP_U32(v, "row_shr:4 row_mask:0xf bank_mask:0xf");
    RRT_DPP_STEP_U32(v, "row_shr:8 row_mask:0xf bank_mask:0xf");
    RRT_DPP_STEP_U32(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    RRT_DPP_STEP_U32(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Twelve f32 min-reductions at once, step by step across all twelve registers: dependent DPP operations are then 12 instructions apart and need
// no wait states (only the first step follows ordinary VALU writes).  All inputs are finite or +-inf, never NaN.
#define RRT_DPP12(pre, ctrl)                                                                                         \
    asm volatile(pre "v_min_f32_dpp %0, %0, %0 " ctrl "\n\tv_min_f32_dpp %1, %1, %1 " ctrl "\n\tv_min_f32_dpp %2, %2, %2 " ctrl "\n\t"    \
                 "v_min_f32_dpp %3, %3, %3 " ctrl "\n\tv_min_f32_dpp %4, %4, %4 " ctrl "\n\tv_min_f32_dpp %5, %5, %5 " ctrl "\n\t"    \
                 "v_min_f32_dpp %6, %6, %6 " ctrl "\n\tv_min_f32_dpp %7, %7, %7 " ctrl "\n\tv_min_f32_dpp %8, %8, %8 " ctrl "\n\t"    \
                 "v_min_f32_dpp %9, %9, %9 " ctrl "\n\tv_min_f32_dpp %10, %10, %10 " ctrl "\n\tv_min_f32_dpp %11, %11, %11 " ctrl     \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]))
__device__ __forceinline__ void wave_min12_f32(float (&x)[12]) {
    // (the wait states sit INSIDE the block that holds the first DPP read: an s_nop in a statement of its own can be scheduled away from it, and the
    //  compiler's hazard recognizer does not look into inline asm -- the x[] producers may be the instructions directly before this block)
    RRT_DPP12("s_nop 1\n\t", "row_shr:1 row_mask:0xf bank_mask:0xf");
    RRT_DPP12("", "row_shr:2 row_mask:0xf bank_mask:0xf");
    RRT_DPP12("", "row_shr:4 row_mask:0xf bank_mask:0xf");
    RRT_DPP12("", "row_shr:8 row_mask:0xf bank_mask:0xf");
    RRT_DPP12("", "row_bcast:15 row_mask:0xa bank_mask:0xf");
    RRT_DPP12("", "row_bcast:31 row_mask:0xc bank_mask:0xf");
#pragma unroll
    for (int i = 0; i < 12; i++) x[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x[i]), 63));
}

// ---- the wave's ray bundle, for the boxes-in-lanes filter (RRT_BUNDLE).  A slab value of lane l is a_l(b) = b*inv_l - o_l*inv_l
// = (b - c)*inv_l + m_l with m_l = (c - o_l)*inv_l and c a common reference point (the first active lane's origin: m_l = 0 for a bundle
// with one origin).  With inv_l in [imin, imax] (same sign) and m_l in [mmin, mmax], a_l(b) lies in (b-c)*[imin,imax] + [mmin,mmax]: an
// interval that bounds every lane's near/far slab values, so a box whose interval test fails is missed by every lane's own test.
struct Bundle { float cx, cy, cz, ilx, ihx, ily, ihy, ilz, ihz, mlx, mhx, mly, mhy, mlz, mhz, amx, amy, amz; };   // am = max |inv| over the lanes
// The rays are anchored at their point of parameter tau: q_l = o_l + tau d_l, and (b - o_l) inv_l = (b - q_l) inv_l + tau.  tau = 0 (the origin) is
// tight for rays that start together (a primary tile), tau = 1 for shadow rays, which end together at the light (raytracer.rs:170-174: origin + dir
// = light + 1e-4 n): their origins are spread along the tile's view rays, their far ends are not.
__device__ __forceinline__ Bundle make_bundle(bool active, V3 o, V3 d, const Ray32& r, float tau) {
    Bundle B;
    const unsigned long long act = __builtin_amdgcn_ballot_w64(active);
    const int leader = act ? __builtin_ctzll(act) : 0;
    const float ox = (float)o.x + tau * (float)d.x, oy = (float)o.y + tau * (float)d.y, oz = (float)o.z + tau * (float)d.z;
    B.cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ox), leader));
    B.cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, oy), leader));
    B.cz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, oz), leader));
    const float ts = tau * r.sigma;                                        // the anchor's parameter in the ray's scaled units (make_ray32)
    const float mx = (B.cx - ox) * r.ix() + ts, my = (B.cy - oy) * r.iy() + ts, mz = (B.cz - oz) * r.iz() + ts;
    const float pinf = __builtin_huge_valf();
    // min over the active lanes of x and of -x (max = -min(-x)); inactive lanes hold +inf
    float x[12] = {active ? r.ix() : pinf, active ? -r.ix() : pinf, active ? r.iy() : pinf, active ? -r.iy() : pinf, active ? r.iz() : pinf, active ? -r.iz() : pinf,
                   active ? mx : pinf,   active ? -mx : pinf,   active ? my : pinf,   active ? -my : pinf,   active ? mz : pinf,   active ? -mz : pinf};
    wave_min12_f32(x);
    B.ilx = x[0]; B.ihx = -x[1]; B.ily = x[2]; B.ihy = -x[3]; B.ilz = x[4]; B.ihz = -x[5];
    B.mlx = x[6]; B.mhx = -x[7]; B.mly = x[8]; B.mhy = -x[9]; B.mlz = x[10]; B.mhz = -x[11];
    // An axis whose directions have mixed signs in the wave gives no bound: its inv interval becomes [-FLT_MAX, FLT_MAX], whose products are
    // -inf/+inf (or 0 for a plane through c) -- no constraint, no NaN.  A lane with the filter off (inv = 0) switches the whole bundle test
    // off: all-zero intervals make every slab value 0 and every box a hit.
    constexpr float kBig = 3.0e38f;
    if (B.ilx < 0.0f && B.ihx > 0.0f) { B.ilx = -kBig; B.ihx = kBig; }
    if (B.ily < 0.0f && B.ihy > 0.0f) { B.ily = -kBig; B.ihy = kBig; }
    if (B.ilz < 0.0f && B.ihz > 0.0f) { B.ilz = -kBig; B.ihz = kBig; }
    if (act == 0ull || __builtin_amdgcn_ballot_w64(active && r.ix() == 0.0f && r.iy() == 0.0f && r.iz() == 0.0f) != 0ull) {
        B.ilx = B.ihx = B.ily = B.ihy = B.ilz = B.ihz = 0.0f; B.mlx = B.mhx = B.mly = B.mhy = B.mlz = B.mhz = 0.0f;
    }
    B.amx = fmaxf(fabsf(B.ilx), fabsf(B.ihx)); B.amy = fmaxf(fabsf(B.ily), fabsf(B.ihy)); B.amz = fmaxf(fabsf(B.ilz), fabsf(B.ihz));
    return B;
}
// Is the bundle worth testing boxes against (lane-filter kernel: long own lists switch to boxes in lanes when it is)?  Every axis: directions of
// one sign, reciprocal directions within RRT_TIGHT_INV of the smallest, anchor spread below `slack` (RRT_TIGHT_SLACK of the filter's coordinate limit)
// in space.  A speed heuristic only: both filters are exact.  Tuned on the soups with the final kernels (1 M soup @4K, slack as a fraction of the
// limit: 1/16384 27.5 ms, 1/1024 26.9, 1/384 26.5, 1/256 26.3, 1/192 26.7, 1/128 27.6, 1/64 33.8 -- a cliff, hence the margin; reciprocal spread
// 0.1 -> 1.6: -0.3 ms; the 100 k soup moves by 1 %, the teapot's kernel does not use it).
#ifndef RRT_TIGHT_INV
#define RRT_TIGHT_INV 1.6f
#endif
#ifndef RRT_TIGHT_SLACK
#define RRT_TIGHT_SLACK (1.0f / 384.0f)
#endif
__device__ __forceinline__ bool bundle_is_tight(const Bundle& B, float slack) {
    bool ok = true;
#define RRT_TIGHT(il, ih, ml, mh)                                                                               \
    {                                                                                                           \
        const float lo = fminf(fabsf(il), fabsf(ih));                                                           \
        ok = ok && (il > 0.0f) == (ih > 0.0f) && lo > 0.0f && (ih - il) <= RRT_TIGHT_INV * lo && (mh - ml) <= slack * lo;  \
    }
    RRT_TIGHT(B.ilx, B.ihx, B.mlx, B.mhx) RRT_TIGHT(B.ily, B.ihy, B.mly, B.mhy) RRT_TIGHT(B.ilz, B.ihz, B.mlz, B.mhz)
#undef RRT_TIGHT
    return ok;
}
// per-LANE box (centre c, half-extent h in VGPRs) against the bundle: false only if no active lane's own slab test could pass.  For a plane
// coordinate b = cb + s, |s| <= h, lane l's slab value is (b - c) inv_l + m_l = p inv_l + s inv_l + m_l with p = cb - c: the first term lies
// between p*il and p*ih (linear in inv_l), the second within h*am of 0, the third in [ml, mh].
__device__ __forceinline__ bool bundle_hit(const Bundle& B, float cx, float cy, float cz, float hx, float hy, float hz) {
#define RRT_AX(cb, h, c, il, ih, ml, mh, am, tn, tf)                                                           \
    float tn, tf;                                                                                               \
    {                                                                                                           \
        const float p = cb - c;                                                                                 \
        const float a1 = p * il, a2 = p * ih;                                                                   \
        tn = __builtin_fmaf(-h, am, fminf(a1, a2) + ml);                                                        \
        tf = __builtin_fmaf(h, am, fmaxf(a1, a2) + mh);                                                         \
    }
    RRT_AX(cx, hx, B.cx, B.ilx, B.ihx, B.mlx, B.mhx, B.amx, tnx, tfx)
    RRT_AX(cy, hy, B.cy, B.ily, B.ihy, B.mly, B.mhy, B.amy, tny, tfy)
    RRT_AX(cz, hz, B.cz, B.ilz, B.ihz, B.mlz, B.mhz, B.amz, tnz, tfz)
#undef RRT_AX
    return fmaxf(fmaxf(fmaxf(tnx, tny), tnz), 0.0f) <= fminf(fminf(tfx, tfy), tfz);
}
// stream compaction across the wave (all 64 lanes must execute it): entry k of the survivors ends up in lane k; returns their number
__device__ __forceinline__ uint32_t wave_compact2(bool keep, uint32_t a, uint32_t b, uint32_t lane, uint32_t& out_a, uint32_t& out_b) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    const uint32_t n = (uint32_t)__popcll(m);
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));   // kept lanes below this one
    const uint32_t dest = keep ? below : n + (lane - below);                                                            // a permutation of 0..63
    out_a = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)a);
    out_b = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)b);
    return n;
}
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }

// One cluster of an own list (the <= 8 slots from cb0) for the lane-filter kernel: per-triangle boxes (same conservative fp32 filter, one level
// down), then only the triangles whose box some lane may hit are fetched (80-byte f64 records) and tested.
__device__ __forceinline__ void own_cluster_lane(PROF_DECL const RRT_CONSTANT DevClusterBox* tboxes, const RRT_CONSTANT DevTriGeom* geom, uint32_t cb0, uint32_t cn,
                                                 const Ray32& r32, V3 o, V3 d, double& own_t, uint32_t& own_slot, uint32_t& own_pos) {
    const RRT_CONSTANT u32x16* tbx = (const RRT_CONSTANT u32x16*)(tboxes + cb0);
    uint32_t lane_tri = 0, wave_tri = 0;
#define RRT_TB(v, off)                                                                                                             \
    {                                                                                                                              \
        PROF_ADD(4, 1);                                                                                                            \
        box_test_shift(sgpr_pair(v[off], v[off + 1]), sgpr_pair(v[off + 2], v[off + 3]), sgpr_pair(v[off + 4], v[off + 5]), r32, lane_tri, wave_tri);  \
    }
    {
        // The last box first, results shifted in from below.
        // (Lanes that missed the cluster's box are not masked out: a triangle's box lies inside its cluster's, so they miss here too bar rounding,
        // and a lane that does slip through merely tests a triangle it cannot hit.)
        const u32x16 t01 = tbx[0], t23 = tbx[1];
        if (cn > 4u) {
            const u32x16 t45 = tbx[2], t67 = tbx[3];
            if (cn > 7u) RRT_TB(t67, 8)
            if (cn > 6u) RRT_TB(t67, 0)
            if (cn > 5u) RRT_TB(t45, 8)
            RRT_TB(t45, 0) RRT_TB(t23, 8) RRT_TB(t23, 0) RRT_TB(t01, 8) RRT_TB(t01, 0)
        } else {
            if (cn > 3u) RRT_TB(t23, 8)
            if (cn > 2u) RRT_TB(t23, 0)
            if (cn > 1u) RRT_TB(t01, 8)
            RRT_TB(t01, 0)
        }
    }
#undef RRT_TB
    if (wave_tri) {
        uint32_t s = __builtin_ctz(wave_tri);
        wave_tri &= wave_tri - 1u;
        UTri tri = load_utri(geom + cb0 + s);
        for (;;) {
            const uint32_t s_next = wave_tri ? (uint32_t)__builtin_ctz(wave_tri) : s;
            UTri nxt = tri;
            if (wave_tri) nxt = load_utri(geom + cb0 + s_next);                 // scalar prefetch of the next candidate triangle
            double t;
            const bool ht = (lane_tri >> s) & 1u;
            PROF_ADD(2, 1); PROF_ADD(3, __popcll(__ballot(ht)));
            if (ht && MT_UNIFORM(tri, o, d, t) && (t < own_t || (t == own_t && tri.pos < own_pos))) { own_t = t; own_slot = cb0 + s; own_pos = tri.pos; }
            if (!wave_tri) break;
            wave_tri &= wave_tri - 1u;
            tri = nxt; s = s_next;
        }
    }
}

// Ray::intersect_with_octant_with_max_t(octree, 0, max_t), ray.rs:104-168, for all 64 lanes at once.
// Must be called from wave-uniform control flow; lanes with active == false take no part.
// Result: slot == kNone <=> None; otherwise (t, slot) of the returned triangle.
// any_ok: the caller only uses Some/None of the result (shadow query, raytracer.rs:181-187).
// kBundle selects the own-list filter: false = every lane tests each box against its own ray (64 rays x 1 box per instruction);
// true = boxes in lanes against the wave's ray bundle (64 boxes x 1 bundle per instruction).  Same results either way.
template <bool kBundle, bool kGroups>
__device__ __forceinline__ void traverse(PROF_DECL const DevScene& S, const Stack& stk, bool active, bool any_ok, bool filter_ok, V3 o, V3 d, double max_t,
                                         double& out_t, uint32_t& out_slot) {
    constexpr bool kLeaf = !kBundle;   // leaf children are tested at their parent by the lane-filter kernel only (measured: the extra code costs the bundle kernel 12 % on the teapot)
    bool done = !active;
    uint32_t cur = 0;        // node this lane has to enter next
    uint32_t crank = 0;      // position of `cur` among its parent's sorted children (0 = the nearest): tie-break of the pick below
    uint32_t sp = 0;         // number of frames on this lane's stack == depth of `cur`
#ifdef RRT_PREFETCH_NODES
    // Node records come through dependent SCALAR loads at the start of every visit, and on the soups most of them miss the L2 (140 k nodes x 96 B):
    // the s_memtime stamps put 42 % of a wave's time between "pick the node" and "its record is here".  A lane knows the node it will enter next as
    // soon as it has pushed / unwound -- usually many visits before the wave gets to it -- so it touches that record with a VECTOR load then (the
    // texture-address path idles in this kernel: TA busy 8 %, profiles/r03_mem_lane100k.json); the scalar load later finds the line in the L2.  The
    // loaded words are kept (fake use at the next prefetch) so that the register is not recycled while the load is in flight.
    uint32_t pf_tail = 0;
#if RRT_PREFETCH_NODES > 1
    uint32_t pf_head = 0;
#endif
#endif
    double ret_t = kInf; uint32_t ret_slot = kNone;
    const RRT_CONSTANT DevNode* nodes = (const RRT_CONSTANT DevNode*)own_sgprs(S.nodes);
    const RRT_CONSTANT DevTriGeom* geom = (const RRT_CONSTANT DevTriGeom*)own_sgprs(S.geom);
    const RRT_CONSTANT DevSuper* supers = (const RRT_CONSTANT DevSuper*)own_sgprs(S.supers);
    const RRT_CONSTANT DevClusterBox* cboxes = (const RRT_CONSTANT DevClusterBox*)own_sgprs(S.cboxes);
    const RRT_CONSTANT DevClusterBox* child_boxes = (const RRT_CONSTANT DevClusterBox*)own_sgprs(S.child_boxes);
    const RRT_CONSTANT DevClusterBox* tboxes = (const RRT_CONSTANT DevClusterBox*)own_sgprs(S.tboxes);
    const Ray32 r32 = make_ray32(o, d, S.cull_limit, S.cull_half_over_limit, S.cull_enabled != 0 && filter_ok);   // filter_ok == false: this lane walks every list in full
    // (lane-filter kernel only: with ~2 slab tests per walk on coherent frames the per-walk set-up costs the bundle-filter kernel what the
    // cheaper quotients save -- measured, rocprofv3 SQ_INSTS_VALU 502.8 M -> 495.6 M per teapot frame but 2 % slower; the soups gain 2 %)
    RayRcp RR; RR.rx = RR.ry = RR.rz = 0.0; RR.plain = false;
    if constexpr (!kBundle) RR = make_ray_rcp(o, d, S.bounds_plain != 0);
    // The lane-filter kernel builds the bundle too (about 150 instructions per walk): where the wave's rays form a tight bundle -- a primary tile,
    // the shadow rays of a tile towards one light -- its LONG own lists (the straddler lists of the upper nodes: thousands of triangles at the root
    // of a large soup) are searched with boxes in lanes, 64 boxes per instruction, instead of one wave-uniform box at a time.
#ifndef RRT_BUNDLE_TAU
#define RRT_BUNDLE_TAU (any_ok ? 1.0f : 0.0f)
#endif
#ifndef RRT_HYBRID_MIN_SUPERS
#define RRT_HYBRID_MIN_SUPERS 2u
#endif
#ifndef RRT_HYBRID_MIN_TRIS
#define RRT_HYBRID_MIN_TRIS 8u
#endif
    Bundle BU{};
    bool long_lists_in_lanes = false;
    if constexpr (kBundle) BU = make_bundle(active, o, d, r32, RRT_BUNDLE_TAU);
#ifndef RRT_NO_HYBRID
    if constexpr (!kBundle) {
        // (parked in LDS rather than held in 15 SGPRs for the whole walk: scalar registers are what the lane-filter kernel is shortest of)
        const Bundle B0 = make_bundle(active, o, d, r32, any_ok ? 1.0f : 0.0f);
        long_lists_in_lanes = bundle_is_tight(B0, S.cull_limit * RRT_TIGHT_SLACK);
        if (stk.lane == 0u) {
            float* q = stk.park();
            q[0] = B0.cx; q[1] = B0.cy; q[2] = B0.cz; q[3] = B0.ilx; q[4] = B0.ihx; q[5] = B0.ily; q[6] = B0.ihy; q[7] = B0.ilz; q[8] = B0.ihz;
            q[9] = B0.mlx; q[10] = B0.mhx; q[11] = B0.mly; q[12] = B0.mhy; q[13] = B0.mlz; q[14] = B0.mhz;
        }
    }
#endif

#if defined(RRT_PROFILE) && defined(RRT_BAND_COUNT)
    prof.secondary = !(o.x == S.origin[0] && o.y == S.origin[1] && o.z == S.origin[2]); prof.pad = (double)S.cull_limit / 131072.0;   // pad = magnitude / 2^15, cull_limit = 4 magnitude
#endif
    PROF_ADD(6, 1); PROF_ADD(7, __popcll(__ballot(active)));
    PROF_T(5);                                                           // [5] traverse set-up (ray32) + whatever ran since the last stamp outside
    for (;;) {
        const unsigned long long pending = __builtin_amdgcn_ballot_w64(!done);
        if (pending == 0) break;
#ifndef RRT_LEADER_FIRST   // default: a wave-wide reduction picks the node (below); -DRRT_LEADER_FIRST = the first pending lane's node (round 1: 6 % slower than smallest-id on the 100k soup)
        // WHICH pending node the wave visits next decides how many lanes share a visit, and a visit costs the same ~800 instructions whoever takes part.
        // Round 3: the DEEPEST pending lane's node (ties: smallest id) -- depth-first for the wave as a whole.  Lanes deep in the tree finish their
        // subtrees and come back up to where the others are parked before those nodes are processed, so a node is visited once with everybody who will
        // ever need it, instead of early with few lanes and again for the late-comers (rounds 1-2 took the smallest id, a breadth-first-like order:
        // 17.8 of 64 lanes per visit on the 100 k soup).  Any order gives the same per-lane results.  100 k soup 11.8 -> 8.9 ms, 1 M soup 26.0 -> 22.9 ms,
        // teapot 0.894 -> 0.882 ms (profiles/r03_ab_pick_policy.txt; largest id first: 8.95 / 23.5; shallowest first: 11.87 / 26.06).
#ifndef RRT_WAVE_FOOTPRINT
#define RRT_WAVE_FOOTPRINT 0
#endif
#ifndef RRT_PICK_POLICY
#define RRT_PICK_POLICY 5
#endif
#ifndef RRT_PICK_POLICY_BUNDLE
#define RRT_PICK_POLICY_BUNDLE 5   /* the bundle-filter kernel (coherent frames) hardly cares: teapot 0.872 -> 0.865 ms with the lane-filter kernel's order, 4K unchanged */
#endif
        constexpr int kPick = kBundle ? RRT_PICK_POLICY_BUNDLE : RRT_PICK_POLICY;
        uint32_t unode;
        if constexpr (kPick == 5) {
            // deepest pending lane first; among those the lane that has come least far through its parent's sorted children (its node is the nearer one: lanes that
            // finish it may still move on to the farther ones, never back), then the smaller id.  23 bits of the id in the key: larger scenes only lose the last tie-break.
            const uint32_t key = done ? 0xFFFFFFFFu : (((63u - sp) << 26) | (crank << 23) | (cur & 0x007FFFFFu));
            const uint32_t kmin = wave_min_u32(key);
            unode = (uint32_t)__builtin_amdgcn_readlane((int)cur, __builtin_ctzll(__builtin_amdgcn_ballot_w64(key == kmin)));
        } else if constexpr (kPick == 3) {
            // deepest pending lane first, ties to the smaller id.  The key keeps 26 bits of the id; the node itself is read from a lane that holds the
            // winning key, so larger scenes only lose the tie-break (any pending lane's node is a valid pick).
            const uint32_t key = done ? 0xFFFFFFFFu : (((63u - sp) << 26) | (cur & 0x03FFFFFFu));
            const uint32_t kmin = wave_min_u32(key);
            unode = (uint32_t)__builtin_amdgcn_readlane((int)cur, __builtin_ctzll(__builtin_amdgcn_ballot_w64(key == kmin)));
        } else if constexpr (kPick == 0) unode = wave_min_u32(done ? 0xFFFFFFFFu : cur);                          // rounds 1-2: smallest id
        else if constexpr (kPick == 1) unode = ~wave_min_u32(done ? 0xFFFFFFFFu : ~cur);                          // largest id (newest nodes first)
        else if constexpr (kPick == 2) unode = wave_min_u32(done ? 0xFFFFFFFFu : ((sp << 26) | cur)) & 0x03FFFFFFu;   // shallowest first (ids < 2^26)
        else unode = 0x03FFFFFFu - (wave_min_u32(done ? 0xFFFFFFFFu : (((63u - sp) << 26) | (0x03FFFFFFu - cur))) & 0x03FFFFFFu);   // deepest first, ties to the larger id
#else
        const int leader = __builtin_ctzll(pending);
        const uint32_t unode = __builtin_amdgcn_readlane(cur, leader);   // wave-uniform node id
#endif
        const UHead N = load_uhead(nodes + unode);
        const uint32_t fc = N.first_child, sb = N.sup_begin, sc = N.sup_count, fl = N.flags;
        PROF_ADD(0, 1); PROF_ADD(1, __popcll(__ballot(!done && cur == unode)));
        PROF_T(0);                                                       // [0] pick node + node record load
        const bool mine = !done && cur == unode;
#ifdef RRT_PROF_HIST   /* developer histogram of lanes parked at the visited node (tools/visit_hist.py) */
        { const int pc = __popcll(__ballot(mine)); PROF_ADD(100 + (pc <= 1 ? 0 : pc <= 3 ? 1 : pc <= 7 ? 2 : pc <= 15 ? 3 : pc <= 31 ? 4 : 5), 1); PROF_ADD(100 + (fc != 0 ? 6 : 7), 1); PROF_ADD(108, pc); }
#endif
        uint32_t order = 0, nchild = 0;
        uint32_t leaf_hit = 0;                                           // bit k: child k is a leaf whose triangle this lane's ray hits
        if (mine && (fl & 0x100u)) {
            {
                // ---- children first (their boxes are this node's lo/mid/hi, which can then leave the SGPRs): slab test in child order
                // (ray.rs:135-144).  Children whose triangle_count is 0 return None at once (ray.rs:112) and dropping entries does not disturb a
                // stable sort, so they are skipped untested.
                if (fc != 0) {
                    PROF_ADD(12, 1);
                    // (1) conservative fp32 filter against the TIGHT bounds of each non-empty child's subtree (clusters.cpp): a child the ray cannot
                    // reach returns None like an empty one, so it is dropped here; `reach` = children some lane may still hit (wave-uniform).
                    uint32_t lane_reach = 0, reach = 0;
                    {
                        const RRT_CONSTANT u32x16* cbx = (const RRT_CONSTANT u32x16*)(child_boxes + (fc - 1u));
#define RRT_CB(k, v, off)                                                                                                          \
                        if (fl & (1u << k)) {                                                                                      \
                            if constexpr (kBundle) {                                                                               \
                                UBox B; B.cx = mkf(v[off]); B.cy = mkf(v[off + 1]); B.cz = mkf(v[off + 2]);                     \
                                B.hx = mkf(v[off + 3]); B.hy = mkf(v[off + 4]); B.hz = mkf(v[off + 5]); B.a = 0; B.b = 0;       \
                                const bool h = slab32(B, r32);                                                                    \
                                lane_reach |= h ? (1u << k) : 0u;                                                                  \
                                reach |= (__builtin_amdgcn_ballot_w64(h) != 0ull) ? (1u << k) : 0u;                                \
                            } else {                                                                                               \
                                box_test_bit<k>(sgpr_pair(v[off], v[off + 1]), sgpr_pair(v[off + 2], v[off + 3]), sgpr_pair(v[off + 4], v[off + 5]), r32, lane_reach, reach);  \
                            }                                                                                                      \
                        }
                        // (Children are often absent, so the results are OR-ed in at their own bit rather than shifted in as box_hit_shift does: the
                        // zeros to shift in for absent children cost more than the shift saves.  Measured, teapot / 100 k soup / 1 M soup: the packed
                        // arithmetic gains the lane-filter kernel 2 % on the soups and costs the bundle-filter kernel 4 % on the teapot -- register
                        // pairs in a kernel that is short of VGPRs -- so each kernel gets the form that suits it.)
                        if (fl & 0x0Fu) { const u32x16 b01 = cbx[0], b23 = cbx[1]; RRT_CB(0, b01, 0) RRT_CB(1, b01, 8) RRT_CB(2, b23, 0) RRT_CB(3, b23, 8) }
                        if (fl & 0xF0u) { const u32x16 b45 = cbx[2], b67 = cbx[3]; RRT_CB(4, b45, 0) RRT_CB(5, b45, 8) RRT_CB(6, b67, 0) RRT_CB(7, b67, 8) }
#undef RRT_CB
                    }
                    // (2) The eight children are cut from this node's box by its three mid planes (octree.rs:136-225), so their 48 slab bounds are
                    // only nine distinct planes {lo, mid, hi} x {x, y, z}, all in the node record: the reference's quotient (bound - o)/d
                    // (ray.rs:22-27) is formed ONCE per plane that some reachable child uses (6..9 IEEE divides per node instead of 6 per
                    // child), with the reference's operands, and each child's test is then the reference's min/max on those quotients.
                    // child k = BBL,BFL,BFR,BBR,TBL,TFL,TFR,TBR (octree.rs:216-225): upper x half for k in {2,3,6,7}, y {4..7}, z {1,2,5,6}
                    if (reach) {
                    uint32_t after = reach;                                                                // (an address that depends on the filter's result: the load is issued after it)
                    asm volatile("s_and_b32 %0, %0, 0" : "+s"(after));
                    const RRT_CONSTANT char* rec = (const RRT_CONSTANT char*)(nodes + unode + after);
                    // The planes are subtracted from the origin IN THE BLOCK THAT LOADS THEM (the empty asm pins the differences there): left to
                    // itself the compiler keeps the planes in SGPRs across the branches below, runs out of SGPRs and moves all sixteen through VGPR lanes
                    // (16 v_writelane + 16..18 v_readlane per visit, a third of the kernel's spill traffic).
                    if ((reach & (reach - 1u)) == 0u) {
                        // a single candidate child (the usual case after the reach filter): its six planes only, fetched by offset
                        // (DevNode: lo[a] at 8a, hi[a] at 24 + 8a, mid[a] at 48 + 8a), its six quotients, nothing to sort
                        {
                            const uint32_t k = (uint32_t)__builtin_ctz(reach);                           // wave-uniform
                            const bool ux = (0xCCu >> k) & 1u, uy = (0xF0u >> k) & 1u, uz = (0x66u >> k) & 1u;   // upper half per axis
                            const double clx = *(const RRT_CONSTANT double*)(rec + (ux ? 48u : 0u)), chx = *(const RRT_CONSTANT double*)(rec + (ux ? 24u : 48u));
                            const double cly = *(const RRT_CONSTANT double*)(rec + (uy ? 56u : 8u)), chy = *(const RRT_CONSTANT double*)(rec + (uy ? 32u : 56u));
                            const double clz = *(const RRT_CONSTANT double*)(rec + (uz ? 64u : 16u)), chz = *(const RRT_CONSTANT double*)(rec + (uz ? 40u : 64u));
                            double n1 = clx - o.x, n2 = chx - o.x, n3 = cly - o.y, n4 = chy - o.y, n5 = clz - o.z, n6 = chz - o.z;
                            asm volatile("" : "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6));
                            double t = kInf;
                            PROF_ADD(8, 1); PROF_ADD(9, __popcll(__ballot(1)));
                            // The child's subtree box shrunk by twice the filter's pad (stored: tight + pad; tested here: tight - pad).  A ray whose fp32 slab test
                            // passes THAT box passes the reference's f64 test of the octant box around it: the subtree box lies inside the octant box when no triangle
                            // pokes out of the root (S.inner_shrink is 0 otherwise), and the margin -- pad = scene / 2^15 -- is 128 x the fp32 evaluation error
                            // (~ scene / 2^22; the same bound the filter's outward pad relies on) and 10^10 x the f64 rounding of the reference's quotients; parallel
                            // and non-finite cases fail the comparison and take the exact path.  If it holds for every lane that reaches the child, nobody needs the
                            // six quotients: with one candidate there is nothing to sort.  Bundle-filter kernel only: teapot 0.860 -> 0.840 ms; the lane-filter kernel
                            // loses 0.7 % on the 100 k soup to the extra box load (profiles/r03_ab_certain_hit.txt).
                            bool need_exact = true;
                            if constexpr (kBundle) {
                                if (S.inner_shrink > 0.0f) {
                                    UBox CB = load_ubox(child_boxes + (fc - 1u) + k);
                                    CB.hx -= S.inner_shrink; CB.hy -= S.inner_shrink; CB.hz -= S.inner_shrink;
                                    const bool certain = lane_reach && r32.sigma > 0.0f && CB.hx >= 0.0f && CB.hy >= 0.0f && CB.hz >= 0.0f && slab32(CB, r32);
                                    need_exact = __builtin_amdgcn_ballot_w64(lane_reach && !certain) != 0ull;
                                    PROF_ADD(5, need_exact ? 0 : 1);              // [5] single-candidate visits decided without the exact test
                                    if (!need_exact && lane_reach) { order = k; nchild = 1u; }
                                }
                            }
                            if (need_exact)
                            if (lane_reach) {
                                double t1, t2, t3, t4, t5, t6;
                                if (RR.plain) {
                                    t1 = quot(n1, d.x, RR.rx); t2 = quot(n2, d.x, RR.rx); t3 = quot(n3, d.y, RR.ry);
                                    t4 = quot(n4, d.y, RR.ry); t5 = quot(n5, d.z, RR.rz); t6 = quot(n6, d.z, RR.rz);
                                } else {
                                    t1 = n1 / d.x; t2 = n2 / d.x; t3 = n3 / d.y; t4 = n4 / d.y; t5 = n5 / d.z; t6 = n6 / d.z;
                                }
                                if (slab_from_quotients(t1, t2, t3, t4, t5, t6, t)) { order = k; nchild = 1u; }
                            }
                            if (kLeaf && ((fl >> (9u + k)) & 1u)) {
                                // leaf child (see below): its one triangle is tested here; a miss means the child returns None
                                const UTri tri = load_utri(geom + N.leaf_base + (uint32_t)__builtin_popcount((fl >> 9) & ((1u << k) - 1u)));
                                double tl;
                                PROF_ADD(2, 1); PROF_ADD(3, __popcll(__ballot(nchild != 0u)));
                                if (nchild) { if (MT_UNIFORM(tri, o, d, tl)) leaf_hit = 1u << k; else nchild = 0u; }
                            }
                        }
                    } else {
                    const UPlanes NP = load_uplanes((const RRT_CONSTANT DevNode*)rec, N.mid2);
                    double dlx = NP.lo[0] - o.x, dmx = NP.mid[0] - o.x, dhx = NP.hi[0] - o.x, dly = NP.lo[1] - o.y, dmy = NP.mid[1] - o.y, dhy = NP.hi[1] - o.y,
                           dlz = NP.lo[2] - o.z, dmz = NP.mid[2] - o.z, dhz = NP.hi[2] - o.z;
                    asm volatile("" : "+v"(dlx), "+v"(dmx), "+v"(dhx), "+v"(dly), "+v"(dmy), "+v"(dhy), "+v"(dlz), "+v"(dmz), "+v"(dhz));
                    double qlx = 0, qmx = 0, qhx = 0, qly = 0, qmy = 0, qhy = 0, qlz = 0, qmz = 0, qhz = 0;
                        if (RR.plain) {
                            qmx = quot(dmx, d.x, RR.rx); qmy = quot(dmy, d.y, RR.ry); qmz = quot(dmz, d.z, RR.rz);
                            if (reach & 0x33u) qlx = quot(dlx, d.x, RR.rx);
                            if (reach & 0xCCu) qhx = quot(dhx, d.x, RR.rx);
                            if (reach & 0x0Fu) qly = quot(dly, d.y, RR.ry);
                            if (reach & 0xF0u) qhy = quot(dhy, d.y, RR.ry);
                            if (reach & 0x99u) qlz = quot(dlz, d.z, RR.rz);
                            if (reach & 0x66u) qhz = quot(dhz, d.z, RR.rz);
                        } else {
                            qmx = dmx / d.x; qmy = dmy / d.y; qmz = dmz / d.z;
                            if (reach & 0x33u) qlx = dlx / d.x;
                            if (reach & 0xCCu) qhx = dhx / d.x;
                            if (reach & 0x0Fu) qly = dly / d.y;
                            if (reach & 0xF0u) qhy = dhy / d.y;
                            if (reach & 0x99u) qlz = dlz / d.z;
                            if (reach & 0x66u) qhz = dhz / d.z;
                        }
                        double tk[8]; bool vk[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            vk[k] = false; tk[k] = kInf;
                            if (reach & (1u << k)) {
                                constexpr int hx[8] = {0, 0, 1, 1, 0, 0, 1, 1}, hy[8] = {0, 0, 0, 0, 1, 1, 1, 1}, hz[8] = {0, 1, 1, 0, 0, 1, 1, 0};
                                double t = kInf;
                                PROF_ADD(8, 1); PROF_ADD(9, __popcll(__ballot(1)));
                                if (((lane_reach >> k) & 1u) &&
                                    slab_from_quotients(hx[k] ? qmx : qlx, hx[k] ? qhx : qmx, hy[k] ? qmy : qly, hy[k] ? qhy : qmy, hz[k] ? qmz : qlz, hz[k] ? qhz : qmz, t)) {
                                    vk[k] = true; tk[k] = (t != t) ? kInf : t;                                              // NaN sorts last (reference panics, ray.rs:147)
                                }
                            }
                        }
                        // LEAF CHILDREN.  A non-empty leaf holds exactly one triangle (octree.rs:77-92), so entering it (ray.rs:112-129 with
                        // max_t = +inf, no children) returns Some(that triangle) iff the ray hits it.  Such children are not visited as nodes: their
                        // triangle (slot leaf_base + rank among the leaf children, geometry adjacent in memory) is tested here by the lanes whose exact
                        // box test entered the child; a miss drops the child from the sorted list (it would have returned None), a hit keeps it and
                        // marks it in the frame, and the unwind takes (t, slot) from there when the walk reaches it in sorted order.
                        uint32_t lm = kLeaf ? ((fl >> 9) & reach) : 0u;                                     // wave-uniform
                        if (lm) {
                            uint32_t vmask = 0;
#pragma unroll
                            for (int k = 0; k < 8; ++k) vmask |= vk[k] ? (1u << k) : 0u;
                            uint32_t k = (uint32_t)__builtin_ctz(lm);
                            lm &= lm - 1u;
                            UTri tri = load_utri(geom + N.leaf_base + (uint32_t)__builtin_popcount((fl >> 9) & ((1u << k) - 1u)));
                            for (;;) {
                                const uint32_t k_next = lm ? (uint32_t)__builtin_ctz(lm) : k;
                                UTri nxt = tri;
                                if (lm) nxt = load_utri(geom + N.leaf_base + (uint32_t)__builtin_popcount((fl >> 9) & ((1u << k_next) - 1u)));   // scalar prefetch
                                double tl;
                                const bool cand = (vmask >> k) & 1u;
                                PROF_ADD(2, 1); PROF_ADD(3, __popcll(__ballot(cand)));
                                if (cand) { if (MT_UNIFORM(tri, o, d, tl)) leaf_hit |= 1u << k; else vmask &= ~(1u << k); }
                                if (!lm) break;
                                lm &= lm - 1u;
                                tri = nxt; k = k_next;
                            }
#pragma unroll
                            for (int k2 = 0; k2 < 8; ++k2) vk[k2] = (vmask >> k2) & 1u;
                        }
                        // stable ascending sort by t (ray.rs:146-147) as a rank computation over the candidates only
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            if (!(reach & (1u << k))) continue;
                            uint32_t rank = 0;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                if (j == k || !(reach & (1u << j))) continue;
                                const bool before = (j < k) ? (tk[j] <= tk[k]) : (tk[j] < tk[k]);
                                rank += (vk[j] && before) ? 1u : 0u;
                            }
                            if (vk[k]) { order |= (uint32_t)k << (3u * rank); nchild++; }
                        }
                    }
                    }

                }
            }
        }
        PROF_T(1);                                                       // [1] children: reach filter, plane quotients, slab tests, rank
        // ---- own list: ray.rs:119-129 as an arg-min over the list: super-cluster box -> its <= 8 cluster boxes -> triangle boxes -> triangles.
        // A later list position never replaces an equal t (strict < in the reference keeps the first), so ties go to the smaller `pos`.
        double own_t = (sp == 0) ? max_t : kInf;                         // ray.rs:117 (children are entered with +inf, ray.rs:96-102,153)
        uint32_t own_slot = kNone;
        uint32_t own_pos = 0;
        // Boxes in lanes: all 64 lanes work here (wave-uniform control flow), each lane testing ONE box of the current level against the
        // wave's ray bundle; survivors are compacted and expanded to the next level; only the triangles that survive are tested, by the
        // lanes parked at this node.
        bool in_lanes = kBundle;
#ifndef RRT_NO_HYBRID
        if constexpr (!kBundle) in_lanes = long_lists_in_lanes && (sc >= RRT_HYBRID_MIN_SUPERS || N.s0_count > RRT_HYBRID_MIN_TRIS);
#endif
        if (in_lanes) {
        if ((fl & 0x100u) && sc) {
            const uint32_t lane = stk.lane, sub = lane & 7u, grp = lane >> 3;
            Bundle BL = BU;
            if constexpr (!kBundle) {
                const float* q = stk.park();
                BL.cx = q[0]; BL.cy = q[1]; BL.cz = q[2]; BL.ilx = q[3]; BL.ihx = q[4]; BL.ily = q[5]; BL.ihy = q[6]; BL.ilz = q[7]; BL.ihz = q[8];
                BL.mlx = q[9]; BL.mhx = q[10]; BL.mly = q[11]; BL.mhy = q[12]; BL.mlz = q[13]; BL.mhz = q[14];
                BL.amx = fmaxf(fabsf(BL.ilx), fabsf(BL.ihx)); BL.amy = fmaxf(fabsf(BL.ily), fabsf(BL.ihy)); BL.amz = fmaxf(fabsf(BL.ilz), fabsf(BL.ihz));
            }
            for (uint32_t s0 = 0; s0 < sc; s0 += 64u) {
                uint32_t l1a = N.s0_begin, l1b = N.s0_count, n1 = 1;             // a single super-cluster: its slot range is in the node record
                if (sc > 1u) {
                    const uint32_t si = s0 + lane;
                    bool h = false; uint32_t tb = 0, tn = 0;
                    if (si < sc) {
                        const DevSuper* P = (const DevSuper*)supers + sb + si;
                        tb = P->tri_begin; tn = P->tri_count;
                        h = (!kGroups || tn != 0u) && bundle_hit(BL, P->c[0], P->c[1], P->c[2], P->h[0], P->h[1], P->h[2]);   // (tn == 0: a group record, clusters.cpp -- the lane-filter kernel's business)
                    }
                    PROF_ADD(10, 1);
                    n1 = wave_compact2(h, tb, tn, lane, l1a, l1b);
                }
                for (uint32_t e0 = 0; e0 < n1; e0 += 8u) {                      // 8 super-clusters x 8 clusters per batch
                    const uint32_t e = e0 + grp;
                    const uint32_t etb = lane_read(l1a, e < n1 ? e : 0u), etn = lane_read(l1b, e < n1 ? e : 0u);
                    bool h2 = false;
                    if (e < n1 && sub * 8u < etn) {
                        const DevClusterBox* P = (const DevClusterBox*)cboxes + (etb >> 3) + sub;
                        h2 = bundle_hit(BL, P->c[0], P->c[1], P->c[2], P->h[0], P->h[1], P->h[2]);
                    }
                    PROF_ADD(11, 1);
                    uint32_t l2a, l2b;
                    const uint32_t n2 = wave_compact2(h2, etb + 8u * sub, (etn - 8u * sub < 8u) ? etn - 8u * sub : 8u, lane, l2a, l2b);
                    for (uint32_t f0 = 0; f0 < n2; f0 += 8u) {                  // 8 clusters x 8 triangles per batch
                        const uint32_t f = f0 + grp;
                        const uint32_t fs = lane_read(l2a, f < n2 ? f : 0u), fn = lane_read(l2b, f < n2 ? f : 0u);
                        bool h3 = false;
                        if (f < n2 && sub < fn) {
                            const DevClusterBox* P = (const DevClusterBox*)tboxes + fs + sub;
                            h3 = bundle_hit(BL, P->c[0], P->c[1], P->c[2], P->h[0], P->h[1], P->h[2]);
                        }
                        PROF_ADD(4, 1);
                        uint32_t l3a, l3b;
                        const uint32_t n3 = wave_compact2(h3, fs + sub, 0u, lane, l3a, l3b);
                        if (n3) {
                            uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)l3a, 0);
                            UTri tri = load_utri(geom + slot);
                            for (uint32_t i = 0; i < n3; ++i) {
                                UTri nxt = tri; uint32_t slot_n = slot;
                                if (i + 1 < n3) { slot_n = (uint32_t)__builtin_amdgcn_readlane((int)l3a, (int)(i + 1)); nxt = load_utri(geom + slot_n); }   // scalar prefetch
                                double t;
                                PROF_ADD(2, 1); PROF_ADD(3, __popcll(__ballot(mine)));
                                if (mine && MT_UNIFORM(tri, o, d, t) && (t < own_t || (t == own_t && tri.pos < own_pos))) { own_t = t; own_slot = slot; own_pos = tri.pos; }
                                tri = nxt; slot = slot_n;
                            }
                        }
                    }
                }
            }
        }
        } else {
        if (mine && (fl & 0x100u)) {
                if (sc) {
                // a node with a single super-cluster carries its slot range in the node record and skips the super-cluster box
                UBox SP; SP.a = N.s0_begin; SP.b = N.s0_count; SP.cx = SP.cy = SP.cz = SP.hx = SP.hy = SP.hz = 0.0f;
                if (sc > 1) SP = load_ubox(supers + sb);
                for (uint32_t si = 0; si < sc; ++si) {
                    UBox SN = SP;
                    if (si + 1 < sc) SN = load_ubox(supers + sb + si + 1);                            // scalar prefetch of the next super-cluster
                    const unsigned long long hs_mask = (sc == 1) ? __builtin_amdgcn_ballot_w64(true) : slab32_u_mask(SP, r32);
                    const uint32_t tb = SP.a, tn = SP.b;
                    PROF_ADD(10, 1);
                    if constexpr (kGroups) {
                        if (sc > 1 && tn == 0u) {
                            // a group record (clusters.cpp): the box of the next `tb` super-clusters; if no lane can reach it they are all skipped
                            if (hs_mask == 0ull) {
                                si += tb;
                                if (si + 1 < sc) SN = load_ubox(supers + sb + si + 1);
                            }
                            SP = SN;
                            continue;
                        }
                    }
#ifdef RRT_SKIP_CBOX_MAX
                    // A short list in one cluster (90 % of a soup's internal nodes hold <= 8 triangles of their own): its cluster box is one more dependent
                    // scalar load and test in front of the triangle boxes, which are few -- go to them directly.
                    if (sc == 1 && tn <= RRT_SKIP_CBOX_MAX) { own_cluster_lane(PROF_ARG tboxes, geom, tb, tn, r32, o, d, own_t, own_slot, own_pos); SP = SN; continue; }
#endif
                    if (hs_mask != 0ull) {
                        // the (up to) 8 cluster boxes of this super-cluster in bursts of 4; cluster c covers slots tb+8c .. tb+8c+7
                        const RRT_CONSTANT u32x16* cb = (const RRT_CONSTANT u32x16*)(cboxes + (tb >> 3));
                        const uint32_t nc = (tn + 7u) >> 3;
                        uint32_t lane_hits = 0, wave_hits = 0;
#define RRT_CL(v, off)                                                                                                             \
                        {                                                                                                      \
                            PROF_ADD(11, 1);                                                                                   \
                            box_test_shift(sgpr_pair(v[off], v[off + 1]), sgpr_pair(v[off + 2], v[off + 3]), sgpr_pair(v[off + 4], v[off + 5]), r32, lane_hits, wave_hits);  \
                        }
                        {
                            // (as for the triangle boxes: no mask for the lanes that missed the super-cluster's box)
                            const u32x16 c01 = cb[0], c23 = cb[1];
                            if (nc > 4u) {
                                const u32x16 c45 = cb[2], c67 = cb[3];
                                if (nc > 7u) RRT_CL(c67, 8)
                                if (nc > 6u) RRT_CL(c67, 0)
                                if (nc > 5u) RRT_CL(c45, 8)
                                RRT_CL(c45, 0) RRT_CL(c23, 8) RRT_CL(c23, 0) RRT_CL(c01, 8) RRT_CL(c01, 0)
                            } else {
                                if (nc > 3u) RRT_CL(c23, 8)
                                if (nc > 2u) RRT_CL(c23, 0)
                                if (nc > 1u) RRT_CL(c01, 8)
                                RRT_CL(c01, 0)
                            }
                        }
#undef RRT_CL
                        while (wave_hits) {
                            const uint32_t c = __builtin_ctz(wave_hits);
                            wave_hits &= wave_hits - 1u;
                            const uint32_t cb0 = tb + 8u * c, cn = (tn - 8u * c < 8u) ? tn - 8u * c : 8u;
                            own_cluster_lane(PROF_ARG tboxes, geom, cb0, cn, r32, o, d, own_t, own_slot, own_pos);
                        }
                    }
                    SP = SN;
                }
            }
        }
        }
        PROF_T(2);                                                       // [2] own list: boxes + Moller-Trumbore
        if (mine) {
            bool returning;
            if (!(fl & 0x100u)) {                                        // triangle_count == 0 -> None, ray.rs:112-114
                returning = true; ret_slot = kNone; ret_t = kInf;
            }
            // A shadow query at the root that already holds an own hit (t < max_t) returns Some whatever the children do
            // (ray.rs:163-167 picks child or own, both Some), and only Some/None is used (raytracer.rs:183-187): stop here.
            // A node none of whose children is entered (leaf, or no child box hit) returns its own result: ray.rs:163-167 with child_dist = inf.
            else if (nchild == 0 || (any_ok && sp == 0 && own_slot != kNone)) {
                returning = true; ret_slot = own_slot; ret_t = own_t;
            } else {
                stk.own_slot(sp) = own_slot; stk.meta(sp) = order | (nchild << 24); stk.fc(sp) = fc | (leaf_hit << 24);   // first_child < 2^24 whenever leaf bits exist (clusters.cpp)
                sp++;
                returning = false;
            }
            // unwind until this lane has a next node to enter or the root has returned (ray.rs:152-167)
            for (;;) {
                if (!returning) {
                    const uint32_t m = stk.meta(sp - 1);
                    const uint32_t cursor = m >> 28, n = (m >> 24) & 15u;
                    if (cursor < n) {                                    // next sorted child, entered with max_t = +inf (ray.rs:153)
                        stk.meta(sp - 1) = m + (1u << 28);
                        const uint32_t fcw = stk.fc(sp - 1), k = (m >> (3u * cursor)) & 7u;
                        const uint32_t fcm = kLeaf ? S.fc_mask : 0xFFFFFFFFu;
                        if (!(((fcw & ~fcm) >> (24u + k)) & 1u)) { cur = (fcw & fcm) + k; crank = cursor; break; }
                        // a leaf child whose triangle this ray hits (tested at the parent, above): it returns Some(t, triangle) without a visit;
                        // a leaf's own record holds its dense slot in leaf_base (it has no children to describe)
                        ret_slot = ((const DevNode*)nodes)[(fcw & fcm) + k].leaf_base;
                        ret_t = t_of_slot((const DevTriGeom*)geom + ret_slot, o, d);
                    } else {
                        ret_slot = stk.own_slot(sp - 1);                     // no child hit: child_dist = inf -> own (ray.rs:163-167)
                        ret_t = (ret_slot != kNone) ? t_of_slot((const DevTriGeom*)geom + ret_slot, o, d) : kInf;
                        sp--;
                    }
                    returning = true;
                }
                if (sp == 0) { done = true; break; }
                if (ret_slot != kNone) {                                 // a child returned Some -> `break` (ray.rs:155-160), then ray.rs:163-167
                    const uint32_t ps = stk.own_slot(sp - 1);               // the parent's `closest`: its own hit's t, else its threshold (ray.rs:117)
                    const double pt = (ps != kNone) ? t_of_slot((const DevTriGeom*)geom + ps, o, d) : ((sp == 1) ? max_t : kInf);
                    if (!(ret_t < pt)) { ret_t = pt; ret_slot = ps; }
                    sp--;
                } else {
                    returning = false;                                   // child returned None -> try the next child
                }
            }
#ifdef RRT_PREFETCH_NODES
            asm volatile("" :: "v"(pf_tail));
            if (!done) pf_tail = *reinterpret_cast<const volatile uint32_t*>(reinterpret_cast<const char*>((const DevNode*)nodes + cur) + 64);
#if RRT_PREFETCH_NODES > 1
            asm volatile("" :: "v"(pf_head));
            if (!done) pf_head = *reinterpret_cast<const volatile uint32_t*>((const DevNode*)nodes + cur);
#endif
#endif
            PROF_T(3);
        }
    }
    PROF_T(3);                                                           // [3] stack push / unwind (attributed at traverse exit: includes the last unwind only)
    out_t = ret_t; out_slot = ret_slot;
}

// ------------------------------------------------------------------------------------------------ traversal, one node per LANE (ray walk)
// The walk above keeps the wave together: one node per step, processed by the lanes parked at it, its records in SGPRs.  That is the right shape
// while the rays of a wave share nodes (a teapot frame: 93 % of the lanes take part in a visit) and the wrong one once they scatter -- on the
// triangle soups half the visits serve fewer than 8 lanes and every visit costs about a thousand wave instructions whoever takes part.  Here every
// pending lane visits ITS OWN node in every step: records come through vector memory (per-lane addresses), every test is per lane, the control
// flow diverges only in loop trip counts.  Same index, same arithmetic, same results (the per-lane state -- cur, sp, the LDS stack frames -- is
// the same as above; only `which lanes process which node when` differs, and no lane's result depends on another lane).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ UHead load_lhead(const DevNode* nodes, uint32_t id) {
    const u32x4* p = (const u32x4*)((const char*)nodes + (size_t)id * sizeof(DevNode) + 64);
    const u32x4 a = p[0], b = p[1];
    UHead n;
    n.mid2 = mkd(a[0], a[1]); n.first_child = a[2]; n.sup_begin = a[3]; n.sup_count = b[0]; n.flags = b[1]; n.s0_begin = b[2]; n.s0_count = b[1] >> 24; n.leaf_base = b[3];
    return n;
}
__device__ __forceinline__ UPlanes load_lplanes(const DevNode* nodes, uint32_t id, double mid2) {
    const u32x4* q = (const u32x4*)((const char*)nodes + (size_t)id * sizeof(DevNode));
    const u32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    UPlanes n;
    n.lo[0] = mkd(a[0], a[1]); n.lo[1] = mkd(a[2], a[3]); n.lo[2] = mkd(b[0], b[1]);
    n.hi[0] = mkd(b[2], b[3]); n.hi[1] = mkd(c[0], c[1]); n.hi[2] = mkd(c[2], c[3]);
    n.mid[0] = mkd(d[0], d[1]); n.mid[1] = mkd(d[2], d[3]); n.mid[2] = mid2;
    return n;
}
__device__ __forceinline__ UBox load_lbox6(const void* rec) {      // centre and half-extent only (24 of the record's 32 bytes)
    const u32x4 a = *(const u32x4*)rec; const u32x2 b = *(const u32x2*)((const char*)rec + 16);
    UBox x; x.cx = mkf(a[0]); x.cy = mkf(a[1]); x.cz = mkf(a[2]); x.hx = mkf(a[3]); x.hy = mkf(b[0]); x.hz = mkf(b[1]); x.a = 0; x.b = 0;
    return x;
}
__device__ __forceinline__ UBox load_lbox8(const void* rec) {
    const u32x4 a = *(const u32x4*)rec, b = *(const u32x4*)((const char*)rec + 16);
    UBox x; x.cx = mkf(a[0]); x.cy = mkf(a[1]); x.cz = mkf(a[2]); x.hx = mkf(a[3]); x.hy = mkf(b[0]); x.hz = mkf(b[1]); x.a = b[2]; x.b = b[3];
    return x;
}
__device__ __forceinline__ UTri load_ltri(const DevTriGeom* geom, uint32_t slot) {
    const u32x4* q = (const u32x4*)((const char*)geom + (size_t)slot * sizeof(DevTriGeom));
    const u32x4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    UTri t;
    t.v1x = mkd(a[0], a[1]); t.v1y = mkd(a[2], a[3]); t.v1z = mkd(b[0], b[1]);
    t.e1x = mkd(b[2], b[3]); t.e1y = mkd(c[0], c[1]); t.e1z = mkd(c[2], c[3]);
    t.e2x = mkd(d[0], d[1]); t.e2y = mkd(d[2], d[3]); t.e2z = mkd(e[0], e[1]);
    t.pos = e[2];
    return t;
}

// Ray::intersect_with_octant_with_max_t(octree, 0, max_t), ray.rs:104-168: same contract as traverse() above.
template <bool kGroups>
__device__ __forceinline__ void traverse_ray(PROF_DECL const DevScene& S, const Stack& stk, bool active, bool any_ok, bool filter_ok, V3 o, V3 d, double max_t,
                                             double& out_t, uint32_t& out_slot) {
    bool done = !active;
    uint32_t cur = 0, sp = 0;
    double ret_t = kInf; uint32_t ret_slot = kNone;
    const DevNode* nodes = S.nodes; const DevTriGeom* geom = S.geom; const DevSuper* supers = S.supers;
    const DevClusterBox* cboxes = S.cboxes; const DevClusterBox* child_boxes = S.child_boxes; const DevClusterBox* tboxes = S.tboxes;
    const Ray32 r32 = make_ray32(o, d, S.cull_limit, S.cull_half_over_limit, S.cull_enabled != 0 && filter_ok);
    const RayRcp RR = make_ray_rcp(o, d, S.bounds_plain != 0);
    PROF_ADD(6, 1); PROF_ADD(7, __popcll(__ballot(active)));
    PROF_T(5);
    while (__builtin_amdgcn_ballot_w64(!done) != 0ull) {
        if (done) continue;
        const UHead N = load_lhead(nodes, cur);
        const uint32_t fc = N.first_child, sb = N.sup_begin, sc = N.sup_count, fl = N.flags;
        PROF_ADD(0, 1); PROF_ADD(1, __popcll(__ballot(1)));
        PROF_T(0);
        uint32_t order = 0, nchild = 0, leaf_hit = 0;
        double own_t = (sp == 0) ? max_t : kInf;                         // ray.rs:117 (children are entered with +inf, ray.rs:96-102,153)
        uint32_t own_slot = kNone, own_pos = 0;
        if (fl & 0x100u) {
            if (fc != 0) {
                // (1) conservative fp32 filter against the tight bounds of each non-empty child's subtree, as in traverse()
                uint32_t reach = 0;
                const DevClusterBox* cb = child_boxes + (fc - 1u);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const UBox B = load_lbox6(cb + k); reach |= slab32(B, r32) ? (1u << k) : 0u; }
                reach &= fl & 0xFFu;                                     // children with triangle_count 0 return None at once (ray.rs:112)
                if (reach) {
                    // (2) exact slab tests from the nine plane quotients (see traverse())
                    const UPlanes NP = load_lplanes(nodes, cur, N.mid2);
                    double qlx, qmx, qhx, qly, qmy, qhy, qlz, qmz, qhz;
                    if (RR.plain) {
                        qlx = quot(NP.lo[0] - o.x, d.x, RR.rx); qmx = quot(NP.mid[0] - o.x, d.x, RR.rx); qhx = quot(NP.hi[0] - o.x, d.x, RR.rx);
                        qly = quot(NP.lo[1] - o.y, d.y, RR.ry); qmy = quot(NP.mid[1] - o.y, d.y, RR.ry); qhy = quot(NP.hi[1] - o.y, d.y, RR.ry);
                        qlz = quot(NP.lo[2] - o.z, d.z, RR.rz); qmz = quot(NP.mid[2] - o.z, d.z, RR.rz); qhz = quot(NP.hi[2] - o.z, d.z, RR.rz);
                    } else {
                        qlx = (NP.lo[0] - o.x) / d.x; qmx = (NP.mid[0] - o.x) / d.x; qhx = (NP.hi[0] - o.x) / d.x;
                        qly = (NP.lo[1] - o.y) / d.y; qmy = (NP.mid[1] - o.y) / d.y; qhy = (NP.hi[1] - o.y) / d.y;
                        qlz = (NP.lo[2] - o.z) / d.z; qmz = (NP.mid[2] - o.z) / d.z; qhz = (NP.hi[2] - o.z) / d.z;
                    }
                    double tk[8]; uint32_t vmask = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        constexpr int hx[8] = {0, 0, 1, 1, 0, 0, 1, 1}, hy[8] = {0, 0, 0, 0, 1, 1, 1, 1}, hz[8] = {0, 1, 1, 0, 0, 1, 1, 0};
                        double t = kInf;
                        const bool v = slab_from_quotients(hx[k] ? qmx : qlx, hx[k] ? qhx : qmx, hy[k] ? qmy : qly, hy[k] ? qhy : qmy, hz[k] ? qmz : qlz, hz[k] ? qhz : qmz, t) && ((reach >> k) & 1u);
                        tk[k] = (!v || t != t) ? kInf : t;                                               // NaN sorts last (reference panics, ray.rs:147)
                        vmask |= v ? (1u << k) : 0u;
                    }
                    // leaf children: their one triangle is tested here, at the parent (see traverse())
                    uint32_t lm = (fl >> 9) & vmask;
                    while (lm) {
                        const uint32_t k = (uint32_t)__builtin_ctz(lm);
                        lm &= lm - 1u;
                        const UTri tri = load_ltri(geom, N.leaf_base + (uint32_t)__builtin_popcount((fl >> 9) & ((1u << k) - 1u)));
                        double tl;
                        PROF_ADD(2, 1); PROF_ADD(3, __popcll(__ballot(1)));
                        if (mt_uniform(tri, o, d, tl)) leaf_hit |= 1u << k; else vmask &= ~(1u << k);
                    }
                    // stable ascending sort by t (ray.rs:146-147) as a rank computation: every pair (j < k) once -- j goes first iff t_j <= t_k
                    // (no NaN left in tk) -- and nothing to do for the lanes of a wave none of which has two candidates
                    if (__builtin_amdgcn_ballot_w64((vmask & (vmask - 1u)) != 0u) == 0ull) {
                        if (vmask) { order = (uint32_t)__builtin_ctz(vmask); nchild = 1u; }
                    } else {
                        uint32_t rank[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int k = 1; k < 8; ++k) {
#pragma unroll
                            for (int j = 0; j < k; ++j) {
                                const bool both = ((vmask >> j) & (vmask >> k) & 1u) != 0u;
                                const bool j_first = tk[j] <= tk[k];
                                rank[k] += (both && j_first) ? 1u : 0u;
                                rank[j] += (both && !j_first) ? 1u : 0u;
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k) if ((vmask >> k) & 1u) { order |= (uint32_t)k << (3u * rank[k]); nchild++; }
                    }
                }
            }
            PROF_T(1);
            // ---- own list (ray.rs:119-129): super-cluster box -> cluster boxes -> triangle boxes -> triangles, all per lane
            // (the record of the next super-cluster is requested before this one is tested: one memory round trip per iteration instead of two in a row)
            UBox SP; SP.a = N.s0_begin; SP.b = N.s0_count; SP.cx = SP.cy = SP.cz = SP.hx = SP.hy = SP.hz = 0.0f;
            if (sc > 1u) SP = load_lbox8(supers + sb);
            for (uint32_t si = 0; si < sc; ++si) {
                UBox SN = SP;
                if (si + 1u < sc) SN = load_lbox8(supers + sb + si + 1u);
                const uint32_t tb = SP.a, tn = SP.b;
                const bool hs = (sc == 1u) || slab32(SP, r32);
                PROF_ADD(10, 1);
                if constexpr (kGroups) {
                    if (sc > 1u && tn == 0u) {                                                           // group record (clusters.cpp): skip its super-clusters if out of reach
                        if (!hs) { si += tb; if (si + 1u < sc) SN = load_lbox8(supers + sb + si + 1u); }
                        SP = SN;
                        continue;
                    }
                }
                SP = SN;
                if (!hs) continue;
                const uint32_t nc = (tn + 7u) >> 3;
                UBox CB = load_lbox6(cboxes + (tb >> 3));
                for (uint32_t c = 0; c < nc; ++c) {
                    UBox CN = CB;
                    if (c + 1u < nc) CN = load_lbox6(cboxes + (tb >> 3) + c + 1u);
                    PROF_ADD(11, 1);
                    const bool hc = slab32(CB, r32);
                    CB = CN;
                    if (!hc) continue;
                    const uint32_t cb0 = tb + 8u * c, cn = (tn - 8u * c < 8u) ? tn - 8u * c : 8u;
                    UBox TB = load_lbox6(tboxes + cb0);
                    for (uint32_t i = 0; i < cn; ++i) {
                        UBox TN = TB;
                        if (i + 1u < cn) TN = load_lbox6(tboxes + cb0 + i + 1u);
                        PROF_ADD(4, 1);
                        const bool ht = slab32(TB, r32);
                        TB = TN;
                        if (!ht) continue;
                        const UTri tri = load_ltri(geom, cb0 + i);
                        double t;
                        PROF_ADD(2, 1); PROF_ADD(3, __popcll(__ballot(1)));
                        if (mt_uniform(tri, o, d, t) && (t < own_t || (t == own_t && tri.pos < own_pos))) { own_t = t; own_slot = cb0 + i; own_pos = tri.pos; }
                    }
                }
            }
        }
        PROF_T(2);
        // ---- return / push / unwind: ray.rs:152-167, as in traverse()
        bool returning;
        if (!(fl & 0x100u)) { returning = true; ret_slot = kNone; ret_t = kInf; }                    // triangle_count == 0 -> None, ray.rs:112-114
        else if (nchild == 0 || (any_ok && sp == 0 && own_slot != kNone)) { returning = true; ret_slot = own_slot; ret_t = own_t; }
        else {
            stk.own_slot(sp) = own_slot; stk.meta(sp) = order | (nchild << 24); stk.fc(sp) = fc | (leaf_hit << 24);
            sp++;
            returning = false;
        }
        for (;;) {
            if (!returning) {
                const uint32_t m = stk.meta(sp - 1);
                const uint32_t cursor = m >> 28, n = (m >> 24) & 15u;
                if (cursor < n) {
                    stk.meta(sp - 1) = m + (1u << 28);
                    const uint32_t fcw = stk.fc(sp - 1), k = (m >> (3u * cursor)) & 7u;
                    const uint32_t fcm = S.fc_mask;
                    if (!(((fcw & ~fcm) >> (24u + k)) & 1u)) { cur = (fcw & fcm) + k; break; }
                    ret_slot = nodes[(fcw & fcm) + k].leaf_base;
                    ret_t = t_of_slot(geom + ret_slot, o, d);
                } else {
                    ret_slot = stk.own_slot(sp - 1);
                    ret_t = (ret_slot != kNone) ? t_of_slot(geom + ret_slot, o, d) : kInf;
                    sp--;
                }
                returning = true;
            }
            if (sp == 0) { done = true; break; }
            if (ret_slot != kNone) {
                const uint32_t ps = stk.own_slot(sp - 1);
                const double pt = (ps != kNone) ? t_of_slot(geom + ps, o, d) : ((sp == 1) ? max_t : kInf);
                if (!(ret_t < pt)) { ret_t = pt; ret_slot = ps; }
                sp--;
            } else {
                returning = false;
            }
        }
        PROF_T(3);
    }
    out_t = ret_t; out_slot = ret_slot;
}

// ------------------------------------------------------------------------------------------------ shading helpers
__device__ __forceinline__ uint64_t f64_as_usize(double x) {   // Rust `as usize`: saturating, NaN -> 0 (raytracer.rs:52-53)
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551616.0) return ~0ull;
    return (uint64_t)x;
}
__device__ __forceinline__ uint64_t umod(uint64_t x, uint32_t m) {   // x % m without the 64-bit software divide when it can be avoided
    if ((m & (m - 1u)) == 0u) return x & (uint64_t)(m - 1u);        // power-of-two texture sizes
    if (x <= 0xFFFFFFFFull) return (uint32_t)x % m;
    return x % m;
}
__device__ __forceinline__ uint32_t clamp_u8(double x) {       // clamp(0.0, 255.0) as u8 (raytracer.rs:97-108)
    if (x < 0.0) x = 0.0;
    if (x > 255.0) x = 255.0;
    if (!(x > 0.0)) return 0;
    return (uint32_t)x;
}
// len_n = normal.length(), len_l = l.length(), len_v = v.length(): the reference recomputes them in every call (raytracer.rs:276,295); the
// value is the same each time, so they are computed once per hit (len_n, len_v) / once per light (len_l) and passed in.
__device__ __forceinline__ V3 diffuse_term(double intensity, double n_dot_l, double len_n, double len_l, V3 kd) {   // raytracer.rs:260-277
    if (n_dot_l <= 0.0) return mk(0.0, 0.0, 0.0);
    return div3((kd * intensity) * n_dot_l, len_n * len_l);
}
__device__ __forceinline__ V3 specular_term(double sw, double intensity, V3 normal, V3 v, double len_v, V3 l, V3 ks) { // raytracer.rs:279-304
    if (sw != -1.0) {
        const V3 r = ((normal * 2.0) * dot(normal, l)) - l;
        const double r_dot_v = dot(r, v);
#ifdef RRT_ABL_NOPOW     /* ablation build only: timing experiment, wrong pixels */
        if (r_dot_v > 0.0) return (ks * intensity) * (r_dot_v / (length(r) * len_v));
#else
        if (r_dot_v > 0.0) return (ks * intensity) * pow(r_dot_v / (length(r) * len_v), sw);
#endif
    }
    return mk(0.0, 0.0, 0.0);
}

// RayTracer::get_ray_colour (raytracer.rs:29-112) for 64 lanes; wave-uniform call.  Returns 0x00RRGGBB.
// kWalk: 0 = node-coherent walk, lane filter; 1 = node-coherent walk, bundle filter; 2 = ray walk (one node per lane)
constexpr int kWalkLane = 0, kWalkBundle = 1, kWalkRay = 2;
template <int kWalk, bool kGroups>
__device__ __forceinline__ uint32_t trace_colour(PROF_DECL const DevScene& S, const Stack& stk, bool active, V3 origin, V3 direction) {
    bool live = active;
    bool in_shadow = false;                 // false: the ray in flight is a segment (primary/reflection) ray; true: a shadow ray
    V3 ro = origin, rd = direction; double rmax = kInf;
    V3 seg_d = direction;
    V3 p = mk(0, 0, 0), n = mk(0, 0, 0);
    uint32_t col = 0, mat = 0, li = 0, depth = 0;
    uint32_t n_eval = 0;                    // lights [0, n_eval) contribute (an occluded point light ends the loop, raytracer.rs:235-237)
    uint32_t term = 0x00FFFFFFu;            // colour of the last segment
    double st_local[RRT_MAX_REFLECT][3]; double st_kr[RRT_MAX_REFLECT];
    const bool first_unfiltered = origin_ray_in_suspect_plane(S, origin, direction);   // exactness guard for the primary segment (false for all but constructed scenes)

    while (__any(live)) {
        double t; uint32_t slot;
        PROF_T(4);                                                       // [4] shading / state machine between traversals
        if constexpr (kWalk == kWalkRay) traverse_ray<kGroups>(PROF_ARG S, stk, live, in_shadow, !(first_unfiltered && depth == 0u && !in_shadow), ro, rd, rmax, t, slot);
        else
        traverse<kWalk == kWalkBundle, kGroups>(PROF_ARG S, stk, live, in_shadow, !(first_unfiltered && depth == 0u && !in_shadow), ro, rd, rmax, t, slot);
        if (live) {
            const bool found = slot != kNone;
            if (!in_shadow) {
                if (!found) {
                    term = 0x00FFFFFFu; live = false;                                    // WHITE, raytracer.rs:109-111
                } else {
                    PROF_ADD(13, 1);
                    // --- hit: raytracer.rs:39-57
                    double u = 0, v = 0, t2;
                    mt_full(S.geom + slot, ro, rd, t2, u, v);
                    const DevTriAttr& A = S.attr[slot];
                    seg_d = rd;
                    p = ro + rd * t;                                                     // raytracer.rs:39
                    mat = A.mat;
                    const DevMaterial& M = S.mats[mat];
                    const DevTexture T = M.tex_desc;
                    const double w = 1.0 - u - v;                                        // raytracer.rs:43
                    const double tex_x = A.uv[2] * u + A.uv[4] * v + A.uv[0] * w;        // raytracer.rs:45-47
                    const double tex_y = A.uv[3] * u + A.uv[5] * v + A.uv[1] * w;        // raytracer.rs:48-50
                    const uint64_t txi = umod(f64_as_usize(tex_x * (double)T.width), T.width);    // raytracer.rs:52
                    const uint64_t tyi = umod(f64_as_usize(tex_y * (double)T.height), T.height);  // raytracer.rs:53
                    const uint8_t* tp = T.rgb + 3ull * ((uint64_t)T.width * tyi + txi);  // raytracer.rs:55
                    col = ((uint32_t)tp[0] << 16) | ((uint32_t)tp[1] << 8) | (uint32_t)tp[2];
                    // get_normal_at_intersection, raytracer.rs:114-162
                    V3 nn = (ld3(A.nrm + 3) * u + ld3(A.nrm + 6) * v) + ld3(A.nrm) * w;  // raytracer.rs:122-124
                    if (M.bump >= 0) {
                        const DevTexture B = M.bump_desc;
                        const uint8_t* bp = B.rgb + 3ull * ((uint64_t)B.width * tyi + txi);  // raytracer.rs:127-128 (colour-texture indices, bump width)
                        V3 bv = mk((double)bp[0], (double)bp[1], (double)bp[2]);
                        bv = normalised(bv);
                        bv = (bv * 2.0) - mk(1.0, 1.0, 1.0);                             // raytracer.rs:130-135
                        V3 tg = cross(nn, mk(0.0, 1.0, 0.0));                            // raytracer.rs:137-141
                        double len_tg = length(tg);
                        if (len_tg == 0.0) { tg = cross(nn, mk(0.0, 0.0, 1.0)); len_tg = length(tg); }   // raytracer.rs:143-149
                        tg = div3(tg, len_tg);                                           // raytracer.rs:151 (t.length() again: same value)
                        const V3 bt = normalised(cross(nn, tg));                         // raytracer.rs:152
                        nn = mk(dot(bv, tg), dot(bv, bt), dot(bv, nn));                  // raytracer.rs:154-158
                    }
                    n = normalised(nn);                                                  // raytracer.rs:161
                    li = 0;                                                              // compute_lighting_intensity, raytracer.rs:199-203
#ifdef RRT_ABL_NOLIGHTS  /* ablation build only: primary hit set-up, then stop */
                    li = S.n_lights;
#endif
                }
            } else {
                // --- result of the shadow ray for point light li (raytracer.rs:232-237): occluded -> `break` out of the whole light loop
                if (found) { n_eval = li; li = 0xFFFFu; }                                // every point light below n_eval was tested and is lit
                else li++;
                in_shadow = false;
            }
            if (live) {
                // --- walk the light list (raytracer.rs:205-255) up to the next point light, whose shadow ray (raytracer.rs:164-188) is traced next.
                // The lighting sum itself is formed once all shadow rays of this hit are known (same lights, same order, same additions): the
                // walks then carry only p, n and the segment direction, not the running intensity.
                while (li < S.n_lights && !in_shadow) {
                    const DevLight& L = S.lights[li];
                    if (L.kind == 1u) {
                        const V3 dir = ld3(L.v) - p;
                        ro = p + n * S.surface_offset;
                        rd = dir;
                        rmax = length(dir);
                        in_shadow = true;
                    } else {
                        li++;
                    }
                }
                if (!in_shadow) {
                    if (li != 0xFFFFu) n_eval = S.n_lights;
                    // --- compute_lighting_intensity, raytracer.rs:192-258
                    const DevMaterial& M = S.mats[mat];
                    const V3 vdir = neg(seg_d);
                    const double len_n = length(n), len_v = length(vdir);                // |normal|, |v|: the reference recomputes them per light, same value
                    V3 I = mk(0.0, 0.0, 0.0);
#ifdef RRT_ABL_NOLIGHTS
                    I = mk(1.0, 1.0, 1.0); n_eval = 0;
#endif
#ifdef RRT_ABL_NOLIGHTMATH   /* ablation build only: shadow rays traced, light arithmetic skipped */
                    I = mk(0.5, 0.5, 0.5); n_eval = 0;
#endif
                    for (uint32_t k = 0; k < n_eval; ++k) {
                        const DevLight& L = S.lights[k];
                        if (L.kind == 0u) {                                              // Ambient, raytracer.rs:207-209
                            I = I + ld3(M.ka) * L.intensity;
                        } else {                                                         // Directional (raytracer.rs:210-227) / unoccluded Point (raytracer.rs:239-252)
                            const V3 l = (L.kind == 2u) ? ld3(L.v) : ld3(L.v) - p;
                            const double n_dot_l = dot(n, l);
                            I = I + diffuse_term(L.intensity, n_dot_l, len_n, length(l), ld3(M.kd));
                            I = I + specular_term(M.ns, L.intensity, n, vdir, len_v, l, ld3(M.ks));
                        }
                    }
                    // --- raytracer.rs:67-108
                    const V3 local = mk((double)((col >> 16) & 255u) * I.x, (double)((col >> 8) & 255u) * I.y, (double)(col & 255u) * I.z);
                    const double kr = M.kr;
                    if (kr > 0.0 && depth < S.max_reflection_depth) {                    // raytracer.rs:76
                        st_local[depth][0] = local.x; st_local[depth][1] = local.y; st_local[depth][2] = local.z; st_kr[depth] = kr;
                        const double d_dot_n = dot(seg_d, n);
                        rd = normalised(seg_d - (n * 2.0) * d_dot_n);                    // raytracer.rs:79
                        ro = p + n * S.surface_offset;                                   // raytracer.rs:82
                        rmax = kInf;
                        depth++;
                    } else {
                        term = (clamp_u8(local.x) << 16) | (clamp_u8(local.y) << 8) | clamp_u8(local.z);   // raytracer.rs:104-108
                        live = false;
                    }
                }
            }
        }
    }
    // unwind the reflection chain, innermost first (raytracer.rs:85-101): every level quantises to u8 before blending
    uint32_t c = term;
    for (uint32_t k = depth; k-- > 0;) {
        const double kr = st_kr[k];
        const double fx = st_local[k][0] * (1.0 - kr) + (double)((c >> 16) & 255u) * kr;
        const double fy = st_local[k][1] * (1.0 - kr) + (double)((c >> 8) & 255u) * kr;
        const double fz = st_local[k][2] * (1.0 - kr) + (double)(c & 255u) * kr;
        c = (clamp_u8(fx) << 16) | (clamp_u8(fy) << 8) | clamp_u8(fz);
    }
    return c;
}

// ------------------------------------------------------------------------------------------------ kernels
// One wave per workgroup; workgroup b renders quadrant (b & 3) of this rank's local tile (b >> 2).
// Waves per SIMD (register budget) per traversal variant, measured on MI355X with the code-generation switches of the Makefile: 4 everywhere
// (128 VGPRs).  The lane-filter kernel ran best at 5 (96 VGPRs) until its spills were cut down; now 4 is 2 % faster on both soups (12.3 -> 12.1 ms,
// 28.2 -> 27.7 ms) and 6 is 20 % slower.  The bundle-filter kernel: 3 -> +20 %, 5 -> +11 %.  The ray walk: 3 -> +11 %, 5 -> +21 %.
#ifndef RRT_WAVES_LANE
#define RRT_WAVES_LANE 4
#endif
#ifndef RRT_WAVES_BUNDLE
#define RRT_WAVES_BUNDLE 4
#endif
#ifndef RRT_WAVES_RAY
#define RRT_WAVES_RAY 4
#endif
#if RRT_TU_FRAME || RRT_TU_LANE
template <int kWalk, bool kGroups>
__global__ __launch_bounds__(64, kWalk == kWalkBundle ? RRT_WAVES_BUNDLE : kWalk == kWalkLane ? RRT_WAVES_LANE : RRT_WAVES_RAY) void render_kernel(const DevScene S, const FrameParams F, uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x;
    const Stack stk{lds + kParkBytes, lane};
    // Blocks that share an XCD (equal blockIdx % 8 under the observed round-robin placement: a speed matter only) take whole chunks of consecutive tiles, so that
    // an XCD's 4 MB L2 serves one screen region's part of a large scene at a time instead of all of it (F.xcd_chunk blocks per chunk, 0 = off; api.cpp decides).
    uint32_t blk = blockIdx.x;
    if (F.xcd_chunk) {
        const uint32_t C = F.xcd_chunk, full = (gridDim.x / (8u * C)) * (8u * C);
        if (blk < full) { const uint32_t xcd = blk & 7u, i = blk >> 3; blk = ((i / C) * 8u + xcd) * C + (i % C); }
    }
    const uint32_t local_tile = blk >> 2, quad = blk & 3u;
    const uint32_t pix = lane >> 2, sub = lane & 3u;
    const uint32_t tile = F.tile_begin + local_tile * F.world + F.rank;
    const bool tile_ok = tile < F.tile_end;
    const uint32_t tx = tile_ok ? tile % F.tiles_x : 0, ty = tile_ok ? tile / F.tiles_x : 0;
#if RRT_WAVE_FOOTPRINT == 1        // 8 x 2 pixels per wave (experiment: 32-byte row segments instead of 16)
    const uint32_t px = tx * 8 + (pix & 7u);
    const uint32_t py = ty * 8 + quad * 2 + (pix >> 3);
#else                               // 4 x 4 pixels per wave
    const uint32_t px = tx * 8 + (quad & 1u) * 4 + (pix & 3u);
    const uint32_t py = ty * 8 + (quad >> 1) * 4 + (pix >> 2);
#endif
    const int32_t W = (int32_t)F.width, H = (int32_t)F.height;
    // put_pixel (engine.rs:146-158): new_x = x + W/2, new_y = H - (y + H/2); draw_scene loops x in [-W/2, W/2), y in [-H/2, H/2)
    // (engine.rs:198,205).  Pixels with no (x,y) in range stay 0 (Canvas::new, engine.rs:135): row 0 (rows 0,1 for odd H) and,
    // for odd W, the last column.  The scene row y = -H/2 maps to new_y = H and is rejected (engine.rs:152-155), so it is not traced.
    const bool in_fb = tile_ok && (int32_t)px < W && py >= F.row_begin && py < F.row_end;    // row_end <= height
    const bool traced = in_fb && (int32_t)px < 2 * (W / 2) && (int32_t)py >= H - 2 * (H / 2) + 1;
    const int32_t x = (int32_t)px - W / 2;
    const int32_t y = (H - H / 2) - (int32_t)py;
    const double xd = (sub & 1u) ? ((double)x + 0.5) : (double)x;                      // engine.rs:207-236: sub-samples (x,y),(x+.5,y),(x,y+.5),(x+.5,y+.5)
    const double yd = (sub & 2u) ? ((double)y + 0.5) : (double)y;
    const V3 dir = mk(xd * F.x_scale, yd * F.y_scale, F.z_value);
#ifdef RRT_PROFILE
    Prof prof{}; prof.last = __builtin_amdgcn_s_memtime();
    const unsigned long long wave_rt0 = __builtin_amdgcn_s_memrealtime(); (void)wave_rt0;
#endif
    const uint32_t c = trace_colour<kWalk, kGroups>(PROF_ARG S, stk, traced, ld3(S.origin), dir);
#if defined(RRT_PROFILE) && defined(RRT_PROF_WAVETIME)
#ifndef RRT_MEMTIME_TICKS_PER_US
#define RRT_MEMTIME_TICKS_PER_US 100ull
#endif
    {   // developer build: how long each wave lived (s_memrealtime: the constant 100 MHz clock; s_memtime counts shader cycles), one count per wave into
        // power-of-two buckets of microseconds, the longest in slot 16
        const unsigned long long ticks = __builtin_amdgcn_s_memrealtime() - wave_rt0;
        const uint32_t us = (uint32_t)(ticks / RRT_MEMTIME_TICKS_PER_US);
        const uint32_t bucket = us < 2u ? 0u : (uint32_t)(31 - __builtin_clz(us));                 // 0: < 2 us, b: [2^b, 2^(b+1)) us
        if (lane == 0) { atomicAdd(S.prof + (bucket < 15u ? bucket : 15u), 1ull); atomicMax(S.prof + 16, ticks); }
    }
#elif defined(RRT_PROFILE)
    PROF_T(4);
    for (int i = 0; i < 16; i++) if (prof.c[i]) atomicAdd(S.prof + i, prof.c[i]);
    if (lane == 0) { for (int i = 0; i < 8; i++) if (prof.t[i]) atomicAdd(S.prof + 16 + i, prof.t[i]); }
    for (int i = 0; i < 4; i++) if (prof.b[i]) atomicAdd(S.prof + 24 + i, prof.b[i]);        // (per lane: band_count events are per ray)
#endif
    // Color::mix over the 4 sub-samples of the pixel = 4 consecutive lanes (entities.rs:49-69): u64 sums, truncating /4
    uint32_t r = (c >> 16) & 255u, g = (c >> 8) & 255u, b = c & 255u;
    r += __shfl_xor(r, 1); g += __shfl_xor(g, 1); b += __shfl_xor(b, 1);
    r += __shfl_xor(r, 2); g += __shfl_xor(g, 2); b += __shfl_xor(b, 2);
    const uint32_t mixed = traced ? (((r >> 2) << 16) | ((g >> 2) << 8) | (b >> 2)) : 0u;   // Into<u32>, entities.rs:32-36
    if (sub == 0) {
        if (F.tiled_output) {
            out[(size_t)local_tile * 64 + ((py & 7u) * 8 + (px & 7u))] = in_fb ? mixed : 0u;
        } else if (in_fb) {
            out[(size_t)py * F.width + px] = mixed;
        }
    }
}

#endif   // RRT_TU_FRAME || RRT_TU_LANE
#if RRT_TU_FRAME
__global__ __launch_bounds__(256) void detile_kernel(uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t world, uint32_t tiles_per_rank,
                                                     const uint32_t* __restrict__ gathered, uint32_t* __restrict__ fb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t px = i % width, py = i / width;
    const uint32_t tile = (py >> 3) * tiles_x + (px >> 3);
    const uint32_t r = tile % world, lt = tile / world;
    fb[i] = gathered[((size_t)r * tiles_per_rank + lt) * 64 + ((py & 7u) * 8 + (px & 7u))];
}
#endif   // RRT_TU_FRAME

#if RRT_TU_RAYS
template <int kWalk>
__global__ __launch_bounds__(64) void ray_colour_kernel(const DevScene S, uint32_t n, const double* __restrict__ origins, const double* __restrict__ dirs,
                                                        uint32_t* __restrict__ colours) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const Stack stk{lds + kParkBytes, threadIdx.x};
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    const bool ok = i < n;
    const V3 o = ok ? ld3(origins + 3 * (size_t)i) : mk(0, 0, 0), d = ok ? ld3(dirs + 3 * (size_t)i) : mk(0, 0, 1);
#ifdef RRT_PROFILE
    Prof prof{}; prof.last = 0;
#endif
    const uint32_t c = trace_colour<kWalk, true>(PROF_ARG S, stk, ok, o, d);
#if defined(RRT_PROFILE) && defined(RRT_BAND_COUNT)
    for (int k = 0; k < 4; k++) if (prof.b[k]) atomicAdd(S.prof + 24 + k, prof.b[k]);
#endif
    if (ok) colours[i] = c;
}

template <int kWalk>
__global__ __launch_bounds__(64) void intersect_kernel(const DevScene S, uint32_t n, const double* __restrict__ origins, const double* __restrict__ dirs,
                                                       const double* __restrict__ max_t, uint8_t* __restrict__ hit, double* __restrict__ t_out,
                                                       double* __restrict__ u_out, double* __restrict__ v_out, uint32_t* __restrict__ tri_out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const Stack stk{lds + kParkBytes, threadIdx.x};
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    const bool ok = i < n;
    const V3 o = ok ? ld3(origins + 3 * (size_t)i) : mk(0, 0, 0), d = ok ? ld3(dirs + 3 * (size_t)i) : mk(0, 0, 1);
    const double mt = (ok && max_t) ? max_t[i] : kInf;
    double t; uint32_t slot;
#ifdef RRT_PROFILE
    Prof prof{}; prof.last = 0;
#endif
    if constexpr (kWalk == kWalkRay) traverse_ray<true>(PROF_ARG S, stk, ok, false, !origin_ray_in_suspect_plane(S, o, d), o, d, mt, t, slot);
    else traverse<kWalk == kWalkBundle, true>(PROF_ARG S, stk, ok, false, !origin_ray_in_suspect_plane(S, o, d), o, d, mt, t, slot);
#if defined(RRT_PROFILE) && defined(RRT_BAND_COUNT)
    for (int k = 0; k < 4; k++) if (prof.b[k]) atomicAdd(S.prof + 24 + k, prof.b[k]);
#endif
    if (!ok) return;
    if (slot == kNone) { hit[i] = 0; t_out[i] = 0; u_out[i] = 0; v_out[i] = 0; tri_out[i] = kNone; return; }
    double t2, u = 0, v = 0;
    mt_full(S.geom + slot, o, d, t2, u, v);
    hit[i] = 1; t_out[i] = t; u_out[i] = u; v_out[i] = v; tri_out[i] = S.attr[slot].orig;
}
#endif   // RRT_TU_RAYS

}  // namespace

#if RRT_TU_FRAME
// Forces the code object of this library onto the current device (HIP loads it lazily, ~80 ms for these kernels): called from a helper thread
// while the host is still parsing the scene (api.cpp, warm_device_async).
void preload_kernels_lane_ray();
int launch_render_lane_ray(const DevScene& s, const FrameParams& f, uint32_t* d_out, void* stream, int walk, uint32_t n_blocks, uint32_t lds);
void preload_kernels() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&render_kernel<kWalkBundle, false>));
    (void)hipGetLastError();
    preload_kernels_lane_ray();
}

uint32_t stack_bytes_per_wave(uint32_t levels) { return kParkBytes + levels * kLevelBytes; }

#ifdef RRT_DEV_HSACO
// Developer build only (tools/bbprof.py): launch the render kernels from an external code object (the product's own assembly with a counter per
// basic block) and append the block counts of every launch to a text file.  Never compiled into librrt_hip.so.
}  // namespace rrt
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace rrt {
static int dev_hsaco_launch(const DevScene& s, const FrameParams& f, uint32_t* d_out, hipStream_t stream, int walk, dim3 grid, uint32_t lds) {
    static hipModule_t mod = nullptr; static uint32_t* d_cnt = nullptr;
    constexpr size_t kCnt = 16 * 64;
    if (!mod) {
        if (hipModuleLoad(&mod, getenv("RRT_DEV_HSACO")) != hipSuccess) { fprintf(stderr, "RRT_DEV_HSACO: cannot load %s\n", getenv("RRT_DEV_HSACO")); abort(); }
        if (hipMalloc((void**)&d_cnt, kCnt * 4) != hipSuccess) abort();
    }
    char name[160];
    snprintf(name, sizeof name, "_ZN3rrt12_GLOBAL__N_113render_kernelILi%dELb%dEEEvNS_8DevSceneENS_11FrameParamsEPj", walk, s.has_groups ? 1 : 0);
    hipFunction_t fn;
    if (hipModuleGetFunction(&fn, mod, name) != hipSuccess) { fprintf(stderr, "RRT_DEV_HSACO: no kernel %s\n", name); abort(); }
    struct Args { DevScene s; FrameParams f; uint32_t* out; } a{s, f, d_out};
    a.s.tex = reinterpret_cast<const DevTexture*>(d_cnt);            // the kernels never read DevScene::tex (texture descriptors are inline in the materials)
    (void)hipMemsetAsync(d_cnt, 0, kCnt * 4, stream);
    size_t size = sizeof a;
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    hipError_t e = hipModuleLaunchKernel(fn, grid.x, 1, 1, 64, 1, 1, lds, stream, nullptr, cfg);
    if (e != hipSuccess) return (int)e;
    std::vector<uint32_t> h(kCnt);
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(h.data(), d_cnt, kCnt * 4, hipMemcpyDeviceToHost) != hipSuccess) return (int)hipGetLastError();
    if (const char* path = getenv("RRT_DEV_BBPROF_OUT")) {
        FILE* fp = fopen(path, "a");
        if (fp) { fprintf(fp, "%s", name); for (uint32_t v : h) fprintf(fp, " %u", v); fprintf(fp, "\n"); fclose(fp); }
    }
    return 0;
}
#endif

int launch_render(const DevScene& s, const FrameParams& f, uint32_t* d_out, void* stream, int walk) {
    const uint32_t n_tiles = f.tile_end > f.tile_begin ? f.tile_end - f.tile_begin : 0u;
    const uint32_t local_tiles = (n_tiles + f.world - 1) / f.world;
    if (local_tiles == 0) return 0;
#ifdef RRT_DEV_HSACO
    if (getenv("RRT_DEV_HSACO")) return dev_hsaco_launch(s, f, d_out, (hipStream_t)stream, walk, dim3(local_tiles * 4), stack_bytes_per_wave(s.stack_levels));
#endif
    // (two instantiations per walk: the group-record handling of long own lists, clusters.cpp, is compiled in only for scenes that have such lists --
    // its few instructions in the super-cluster loop cost the other scenes 5 % through register allocation alone, measured)
    const dim3 grid(local_tiles * 4), block(64);
    const uint32_t lds = stack_bytes_per_wave(s.stack_levels);
    const hipStream_t q = (hipStream_t)stream;
    if (walk != kWalkBundle) return launch_render_lane_ray(s, f, d_out, stream, walk, local_tiles * 4, lds);
    if (s.has_groups) hipLaunchKernelGGL((render_kernel<kWalkBundle, true>), grid, block, lds, q, s, f, d_out);
    else hipLaunchKernelGGL((render_kernel<kWalkBundle, false>), grid, block, lds, q, s, f, d_out);
    return (int)hipGetLastError();
}

int launch_detile(uint32_t width, uint32_t height, uint32_t world, const uint32_t* d_gathered, uint32_t* d_fb, void* stream) {
    const uint32_t tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    const uint32_t tpr = (tiles_x * tiles_y + world - 1) / world;
    const uint32_t n = width * height;
    if (n == 0) return 0;
    hipLaunchKernelGGL(detile_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, width, height, tiles_x, world, tpr, d_gathered, d_fb);
    return (int)hipGetLastError();
}

#endif   // RRT_TU_FRAME

#if RRT_TU_LANE
void preload_kernels_lane_ray() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&render_kernel<kWalkLane, false>));
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&render_kernel<kWalkRay, false>));
    (void)hipGetLastError();
}
int launch_render_lane_ray(const DevScene& s, const FrameParams& f, uint32_t* d_out, void* stream, int walk, uint32_t n_blocks, uint32_t lds) {
    const dim3 grid(n_blocks), block(64);
    const hipStream_t q = (hipStream_t)stream;
    if (walk == kWalkRay) {
        if (s.has_groups) hipLaunchKernelGGL((render_kernel<kWalkRay, true>), grid, block, lds, q, s, f, d_out);
        else hipLaunchKernelGGL((render_kernel<kWalkRay, false>), grid, block, lds, q, s, f, d_out);
    } else {
        if (s.has_groups) hipLaunchKernelGGL((render_kernel<kWalkLane, true>), grid, block, lds, q, s, f, d_out);
        else hipLaunchKernelGGL((render_kernel<kWalkLane, false>), grid, block, lds, q, s, f, d_out);
    }
    return (int)hipGetLastError();
}
#endif   // RRT_TU_LANE

#if RRT_TU_RAYS
int launch_ray_colours(const DevScene& s, uint32_t n, const double* d_origins, const double* d_dirs, uint32_t* d_colours, void* stream, int walk) {
    if (n == 0) return 0;
    const dim3 grid((n + 63) / 64), block(64);
    const uint32_t lds = stack_bytes_per_wave(s.stack_levels);
    if (walk == kWalkBundle) hipLaunchKernelGGL(ray_colour_kernel<kWalkBundle>, grid, block, lds, (hipStream_t)stream, s, n, d_origins, d_dirs, d_colours);
    else if (walk == kWalkRay) hipLaunchKernelGGL(ray_colour_kernel<kWalkRay>, grid, block, lds, (hipStream_t)stream, s, n, d_origins, d_dirs, d_colours);
    else hipLaunchKernelGGL(ray_colour_kernel<kWalkLane>, grid, block, lds, (hipStream_t)stream, s, n, d_origins, d_dirs, d_colours);
    return (int)hipGetLastError();
}

int launch_intersect(const DevScene& s, uint32_t n, const double* d_origins, const double* d_dirs, const double* d_max_t,
                     uint8_t* d_hit, double* d_t, double* d_u, double* d_v, uint32_t* d_tri, void* stream, int walk) {
    if (n == 0) return 0;
    const dim3 grid((n + 63) / 64), block(64);
    const uint32_t lds = stack_bytes_per_wave(s.stack_levels);
    if (walk == kWalkBundle) hipLaunchKernelGGL(intersect_kernel<kWalkBundle>, grid, block, lds, (hipStream_t)stream, s, n, d_origins, d_dirs, d_max_t, d_hit, d_t, d_u, d_v, d_tri);
    else if (walk == kWalkRay) hipLaunchKernelGGL(intersect_kernel<kWalkRay>, grid, block, lds, (hipStream_t)stream, s, n, d_origins, d_dirs, d_max_t, d_hit, d_t, d_u, d_v, d_tri);
    else hipLaunchKernelGGL(intersect_kernel<kWalkLane>, grid, block, lds, (hipStream_t)stream, s, n, d_origins, d_dirs, d_max_t, d_hit, d_t, d_u, d_v, d_tri);
    return (int)hipGetLastError();
}
#endif   // RRT_TU_RAYS

}  // namespace rrt
