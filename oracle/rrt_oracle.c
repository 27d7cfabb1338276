/*
 * rrt_oracle.c -- CPU restatement (plain C, IEEE f64, no FMA contraction, no fast-math) of the reference
 * renderer's per-pixel hot path.  TEST INFRASTRUCTURE ONLY -- see rrt_oracle.h for the rules and for what
 * pins it (no runnable reference: "parity unpinned"; octree vectors and the reference's one image do pin it).  Build: see oracle/Makefile (gcc -O2 -ffp-contract=off).
 *
 * Structure deliberately follows the reference (recursive octree walk on an arena of nodes with
 * per-node Vec-like lists), NOT the product's flattened/iterative GPU form, so that the two are
 * independent statements of the same algorithm.
 */
#include "rrt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Vector3d, engine.rs:16-104 */
static inline ovec3 v3(double x, double y, double z) { ovec3 r = {x, y, z}; return r; }
static inline ovec3 vadd(ovec3 a, ovec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }      /* engine.rs:16-26 */
static inline ovec3 vsub(ovec3 a, ovec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }      /* engine.rs:36-46 */
static inline ovec3 vmul(ovec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }            /* engine.rs:48-58 */
static inline ovec3 vneg(ovec3 a) { return v3(-a.x, -a.y, -a.z); }                               /* engine.rs:60-70 */
static inline ovec3 vdiv(ovec3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }            /* engine.rs:72-82 */
static inline double vdot(ovec3 a, ovec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }  /* engine.rs:85-87 */
static inline double vlength(ovec3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }        /* engine.rs:89-91, powi(2) == x*x */
static inline ovec3 vcross(ovec3 a, ovec3 b) {                                                   /* engine.rs:93-99 */
    return v3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
static inline ovec3 vnormalised(ovec3 a) { return vdiv(a, vlength(a)); }                         /* engine.rs:101-103 */

/* Rust f64::min / f64::max: if one operand is NaN the other is returned == C fmin/fmax (IEEE minNum/maxNum). */
#define RMIN(a, b) fmin((a), (b))
#define RMAX(a, b) fmax((a), (b))

#define F64_EPSILON 2.220446049250313e-16 /* f64::EPSILON, ray.rs:66,89 */
static const double SURFACE_OFFSET = 0.0001;       /* raytracer.rs:17 */
#define MAX_REFLECTION_DEPTH 5u                    /* raytracer.rs:20 */

/* ------------------------------------------------------------------ data model */
typedef struct { double min[3], max[3]; } oaabb;   /* aabb.rs:4-8 */

typedef struct {            /* entities.rs:72-84 (material as an index instead of Arc<Material>) */
    ovec3 v1, v2, v3;
    ovec3 t1, t2, t3;
    ovec3 n1, n2, n3;
    uint32_t mat;
} otriangle;

typedef struct {            /* octree.rs:5-12 */
    uint32_t aabb_index;
    uint32_t *triangles; uint32_t n_triangles, cap_triangles;
    uint32_t children[8]; uint32_t n_children;
    uint32_t triangle_count;
} onode;

struct oracle_scene {
    /* Octree, octree.rs:14-20 */
    onode *nodes; uint32_t n_nodes, cap_nodes;
    oaabb *aabbs; uint32_t n_aabbs, cap_aabbs;
    otriangle *triangles; uint32_t n_triangles;
    /* RayTracer, raytracer.rs:22-26 */
    omaterial *mats; uint32_t n_mats;
    otexture *tex; uint32_t n_tex;
    olight *lights; uint32_t n_lights;
    ovec3 origin;
};

/* ------------------------------------------------------------------ Aabb, aabb.rs:25-60 */
static oaabb aabb_from_triangle(const otriangle *t) {   /* aabb.rs:25-47 */
    oaabb b;
    b.min[0] = RMIN(t->v1.x, RMIN(t->v2.x, t->v3.x)); b.max[0] = RMAX(t->v1.x, RMAX(t->v2.x, t->v3.x));
    b.min[1] = RMIN(t->v1.y, RMIN(t->v2.y, t->v3.y)); b.max[1] = RMAX(t->v1.y, RMAX(t->v2.y, t->v3.y));
    b.min[2] = RMIN(t->v1.z, RMIN(t->v2.z, t->v3.z)); b.max[2] = RMAX(t->v1.z, RMAX(t->v2.z, t->v3.z));
    return b;
}
static int aabb_intersects(const oaabb *a, const oaabb *o) {   /* aabb.rs:49-60, inclusive */
    if (a->max[0] < o->min[0] || a->min[0] > o->max[0]) return 0;
    if (a->max[1] < o->min[1] || a->min[1] > o->max[1]) return 0;
    if (a->max[2] < o->min[2] || a->min[2] > o->max[2]) return 0;
    return 1;
}

/* ------------------------------------------------------------------ Octree build, octree.rs:23-241 */
static uint32_t push_aabb(oracle_scene *s, oaabb b) {
    if (s->n_aabbs == s->cap_aabbs) { s->cap_aabbs = s->cap_aabbs ? s->cap_aabbs * 2 : 64; s->aabbs = realloc(s->aabbs, sizeof(oaabb) * s->cap_aabbs); }
    s->aabbs[s->n_aabbs] = b; return s->n_aabbs++;
}
static uint32_t push_node(oracle_scene *s, uint32_t aabb_index) {
    if (s->n_nodes == s->cap_nodes) { s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 64; s->nodes = realloc(s->nodes, sizeof(onode) * s->cap_nodes); }
    onode *n = &s->nodes[s->n_nodes]; memset(n, 0, sizeof *n); n->aabb_index = aabb_index; return s->n_nodes++;
}
static void node_push_triangle(onode *n, uint32_t ti) {
    if (n->n_triangles == n->cap_triangles) { n->cap_triangles = n->cap_triangles ? n->cap_triangles * 2 : 4; n->triangles = realloc(n->triangles, sizeof(uint32_t) * n->cap_triangles); }
    n->triangles[n->n_triangles++] = ti;
}
static oaabb aabb_new(double min_x, double max_x, double min_y, double max_y, double min_z, double max_z) { /* aabb.rs:10-23 */
    oaabb b; b.min[0] = min_x; b.min[1] = min_y; b.min[2] = min_z; b.max[0] = max_x; b.max[1] = max_y; b.max[2] = max_z; return b;
}

static void subdivide(oracle_scene *s, uint32_t octant_index, uint32_t child_indices[8]) {   /* octree.rs:121-241 */
    oaabb a = s->aabbs[s->nodes[octant_index].aabb_index];
    double x_min = a.min[0], y_min = a.min[1], z_min = a.min[2];
    double x_max = a.max[0], y_max = a.max[1], z_max = a.max[2];
    double hx = (x_max - x_min) / 2.0, hy = (y_max - y_min) / 2.0, hz = (z_max - z_min) / 2.0;   /* octree.rs:136-138 */
    oaabb kids[8] = {                                                                            /* order: octree.rs:216-225 */
        aabb_new(x_min, x_min + hx, y_min, y_min + hy, z_min, z_min + hz),   /* bottom_back_left   */
        aabb_new(x_min, x_min + hx, y_min, y_min + hy, z_min + hz, z_max),   /* bottom_front_left  */
        aabb_new(x_min + hx, x_max, y_min, y_min + hy, z_min + hz, z_max),   /* bottom_front_right */
        aabb_new(x_min + hx, x_max, y_min, y_min + hy, z_min, z_min + hz),   /* bottom_back_right  */
        aabb_new(x_min, x_min + hx, y_min + hy, y_max, z_min, z_min + hz),   /* top_back_left      */
        aabb_new(x_min, x_min + hx, y_min + hy, y_max, z_min + hz, z_max),   /* top_front_left     */
        aabb_new(x_min + hx, x_max, y_min + hy, y_max, z_min + hz, z_max),   /* top_front_right    */
        aabb_new(x_min + hx, x_max, y_min + hy, y_max, z_min, z_min + hz),   /* top_back_right     */
    };
    s->nodes[octant_index].n_children = 0;                                   /* octree.rs:214 */
    for (int k = 0; k < 8; k++) {                                            /* octree.rs:216-238 */
        uint32_t new_aabb_index = push_aabb(s, kids[k]);
        uint32_t new_node_index = push_node(s, new_aabb_index);
        onode *n = &s->nodes[octant_index];                                  /* re-take: push_node may realloc */
        n->children[n->n_children++] = new_node_index;
        child_indices[k] = new_node_index;
    }
}

static void push_at_octant(oracle_scene *s, uint32_t triangle_index, uint32_t aabb_index, uint32_t octant_index) { /* octree.rs:54-108 */
    int intersects, current_octant_has_triangle, is_leaf_octant;
    uint32_t children[8]; uint32_t n_children;
    {
        const oaabb *aabb = &s->aabbs[aabb_index];
        const onode *node = &s->nodes[octant_index];
        const oaabb *octant_aabb = &s->aabbs[node->aabb_index];
        intersects = aabb_intersects(aabb, octant_aabb);                     /* octree.rs:65 */
        current_octant_has_triangle = node->n_triangles != 0;                /* octree.rs:66 */
        n_children = node->n_children; memcpy(children, node->children, sizeof children);
        is_leaf_octant = n_children == 0;                                    /* octree.rs:68 */
    }
    if (!intersects) return;                                                 /* octree.rs:71-73 */
    s->nodes[octant_index].triangle_count += 1;                              /* octree.rs:75 */

    if (is_leaf_octant && !current_octant_has_triangle) {                    /* octree.rs:77-78 */
        node_push_triangle(&s->nodes[octant_index], triangle_index);
    } else {
        if (is_leaf_octant) { subdivide(s, octant_index, children); n_children = 8; }  /* octree.rs:79-80 */
        uint32_t hit[8]; uint32_t n_hit = 0;                                 /* octree.rs:82-86 / 94-98 */
        for (uint32_t k = 0; k < n_children; k++) {
            const oaabb *octant_aabb = &s->aabbs[s->nodes[children[k]].aabb_index];  /* octree.rs:110-119 */
            if (aabb_intersects(octant_aabb, &s->aabbs[aabb_index])) hit[n_hit++] = children[k];
        }
        if (n_hit == 1) push_at_octant(s, triangle_index, aabb_index, hit[0]);        /* octree.rs:88-89 / 100-101 */
        else node_push_triangle(&s->nodes[octant_index], triangle_index);             /* octree.rs:90-92 / 102-104 */
    }
}

static void push_triangle(oracle_scene *s, uint32_t triangle_index) {        /* octree.rs:41-52; the triangle is already stored at triangles[triangle_index] */
    oaabb b = aabb_from_triangle(&s->triangles[triangle_index]);
    uint32_t aabb_index = push_aabb(s, b);
    push_at_octant(s, triangle_index, aabb_index, 0);
}

/* ------------------------------------------------------------------ scene lifetime */
static ovec3 rd3(const double *p) { return v3(p[0], p[1], p[2]); }

oracle_scene *oracle_scene_create(uint32_t n_tris, const double *pos, const double *uv, const double *nrm,
                                  const uint32_t *mat, uint32_t n_mats, const omaterial *mats,
                                  uint32_t n_tex, const otexture *tex, uint32_t n_lights, const olight *lights,
                                  ovec3 origin, const double root[6]) {
    oracle_scene *s = calloc(1, sizeof *s);
    s->triangles = malloc(sizeof(otriangle) * (n_tris ? n_tris : 1));
    s->mats = malloc(sizeof(omaterial) * (n_mats ? n_mats : 1)); memcpy(s->mats, mats, sizeof(omaterial) * n_mats); s->n_mats = n_mats;
    s->tex = malloc(sizeof(otexture) * (n_tex ? n_tex : 1)); memcpy(s->tex, tex, sizeof(otexture) * n_tex); s->n_tex = n_tex;
    s->lights = malloc(sizeof(olight) * (n_lights ? n_lights : 1)); memcpy(s->lights, lights, sizeof(olight) * n_lights); s->n_lights = n_lights;
    s->origin = origin;
    /* Octree::new, octree.rs:23-39 */
    push_aabb(s, aabb_new(root[0], root[1], root[2], root[3], root[4], root[5]));
    push_node(s, 0);
    for (uint32_t i = 0; i < n_tris; i++) {
        otriangle *t = &s->triangles[i];
        t->v1 = rd3(pos + 9 * i); t->v2 = rd3(pos + 9 * i + 3); t->v3 = rd3(pos + 9 * i + 6);
        t->t1 = rd3(uv + 9 * i);  t->t2 = rd3(uv + 9 * i + 3);  t->t3 = rd3(uv + 9 * i + 6);
        t->n1 = rd3(nrm + 9 * i); t->n2 = rd3(nrm + 9 * i + 3); t->n3 = rd3(nrm + 9 * i + 6);
        t->mat = mat[i];
        s->n_triangles = i + 1;
        push_triangle(s, i);
    }
    return s;
}

void oracle_scene_destroy(oracle_scene *s) {
    if (!s) return;
    for (uint32_t i = 0; i < s->n_nodes; i++) free(s->nodes[i].triangles);
    free(s->nodes); free(s->aabbs); free(s->triangles); free(s->mats); free(s->tex); free(s->lights); free(s);
}

uint32_t oracle_octree_num_nodes(const oracle_scene *s) { return s->n_nodes; }
static uint32_t depth_of(const oracle_scene *s, uint32_t n) {
    uint32_t d = 0;
    for (uint32_t k = 0; k < s->nodes[n].n_children; k++) { uint32_t c = depth_of(s, s->nodes[n].children[k]); if (c > d) d = c; }
    return d + 1;
}
uint32_t oracle_octree_max_depth(const oracle_scene *s) { return depth_of(s, 0); }
uint32_t oracle_octree_own_total(const oracle_scene *s) {
    uint32_t t = 0; for (uint32_t i = 0; i < s->n_nodes; i++) t += s->nodes[i].n_triangles; return t;
}
void oracle_octree_export(const oracle_scene *s, double *aabb, uint32_t *first_child, uint32_t *tri_count,
                          uint32_t *own_off, uint32_t *own_idx) {
    uint32_t off = 0;
    for (uint32_t i = 0; i < s->n_nodes; i++) {
        const onode *n = &s->nodes[i]; const oaabb *b = &s->aabbs[n->aabb_index];
        for (int k = 0; k < 3; k++) { aabb[6 * i + k] = b->min[k]; aabb[6 * i + 3 + k] = b->max[k]; }
        first_child[i] = n->n_children ? n->children[0] : 0;
        tri_count[i] = n->triangle_count;
        own_off[i] = off;
        for (uint32_t k = 0; k < n->n_triangles; k++) own_idx[off++] = n->triangles[k];
    }
    own_off[s->n_nodes] = off;
}

/* ------------------------------------------------------------------ Ray queries, ray.rs:20-169 */
typedef struct { ovec3 origin, direction; } oray;
typedef struct { double t, u, v; const otriangle *triangle; } otrihit;   /* ray.rs:5-10 */

static int intersect_aabb(const oray *r, const oaabb *b, double *t_out) {   /* ray.rs:21-54 */
    double t1 = (b->min[0] - r->origin.x) / r->direction.x;
    double t2 = (b->max[0] - r->origin.x) / r->direction.x;
    double t3 = (b->min[1] - r->origin.y) / r->direction.y;
    double t4 = (b->max[1] - r->origin.y) / r->direction.y;
    double t5 = (b->min[2] - r->origin.z) / r->direction.z;
    double t6 = (b->max[2] - r->origin.z) / r->direction.z;
    double tmin = RMAX(RMAX(RMIN(t1, t2), RMIN(t3, t4)), RMIN(t5, t6));
    double tmax = RMIN(RMIN(RMAX(t1, t2), RMAX(t3, t4)), RMAX(t5, t6));
    if (tmax < 0.0) return 0;                       /* ray.rs:39-41 */
    if (tmin > tmax) return 0;                      /* ray.rs:44-46 */
    if (tmin < 0.0) { *t_out = tmax; return 1; }    /* ray.rs:49-51 */
    *t_out = tmin; return 1;                        /* ray.rs:53 */
}

static int intersect_with_triangle(const oray *r, const otriangle *tri, otrihit *out) {   /* ray.rs:56-94 */
    ovec3 edge1 = vsub(tri->v2, tri->v1);
    ovec3 edge2 = vsub(tri->v3, tri->v1);
    ovec3 h = vcross(r->direction, edge2);
    double a = vdot(edge1, h);
    if (a > -F64_EPSILON && a < F64_EPSILON) return 0;   /* ray.rs:66-69 */
    double f = 1.0 / a;
    ovec3 s = vsub(r->origin, tri->v1);
    double u = f * vdot(s, h);
    if (u < 0.0 || u > 1.0) return 0;                    /* ray.rs:75-77 */
    ovec3 q = vcross(s, edge1);
    double v = f * vdot(r->direction, q);
    if (v < 0.0 || u + v > 1.0) return 0;                /* ray.rs:82-84 */
    double t = f * vdot(edge2, q);
    if (t > F64_EPSILON) { out->t = t; out->u = u; out->v = v; out->triangle = tri; return 1; }  /* ray.rs:89-91 */
    return 0;
}

/* ray.rs:104-168.  Deviation, documented: the reference sorts with partial_cmp().unwrap() (ray.rs:147), which
 * PANICS if an AABB distance is NaN; here a NaN distance is ordered after every number (children keep index
 * order among themselves), so the oracle and the GPU agree on inputs the reference aborts on. */
static int intersect_with_octant_with_max_t(const oracle_scene *s, const oray *r, uint32_t octant_index,
                                            double max_t, otrihit *out, ocounters *c) {
    const onode *node = &s->nodes[octant_index];
    c->nodes_entered++;
    if (node->triangle_count == 0) return 0;                               /* ray.rs:112-114 */

    int have_own = 0; otrihit own;
    double closest = max_t;                                                /* ray.rs:117 */
    for (uint32_t k = 0; k < node->n_triangles; k++) {                     /* ray.rs:119-129 */
        otrihit h;
        c->tri_tests++;
        if (intersect_with_triangle(r, &s->triangles[node->triangles[k]], &h)) {
            if (h.t < closest) { closest = h.t; own = h; have_own = 1; }
        }
    }

    double dist[8]; uint32_t idx[8]; uint32_t num_children = 0;            /* ray.rs:132-144 */
    for (uint32_t k = 0; k < node->n_children; k++) {
        uint32_t coi = node->children[k];
        double t;
        c->aabb_tests++;
        if (intersect_aabb(r, &s->aabbs[s->nodes[coi].aabb_index], &t)) { dist[num_children] = t; idx[num_children] = coi; num_children++; }
    }
    /* stable ascending sort (insertion sort is stable), ray.rs:146-147 */
    for (uint32_t i = 1; i < num_children; i++) {
        double kd = dist[i]; uint32_t ki = idx[i]; uint32_t j = i;
        while (j > 0 && (dist[j - 1] > kd || (isnan(dist[j - 1]) && !isnan(kd)))) { dist[j] = dist[j - 1]; idx[j] = idx[j - 1]; j--; }
        dist[j] = kd; idx[j] = ki;
    }

    int have_child = 0; otrihit child; double child_dist = INFINITY;       /* ray.rs:149-161 */
    for (uint32_t i = 0; i < num_children; i++) {
        otrihit h;
        if (intersect_with_octant_with_max_t(s, r, idx[i], INFINITY, &h, c)) {   /* ray.rs:153 -> 96-102: max_t reset to +inf */
            child_dist = h.t; child = h; have_child = 1;
            break;
        }
    }
    if (child_dist < closest) { if (have_child) *out = child; return have_child; }   /* ray.rs:163-164 */
    if (have_own) *out = own;                                                        /* ray.rs:165-167 */
    return have_own;
}

/* ------------------------------------------------------------------ casts */
uint64_t oracle_f64_as_usize(double x) {          /* Rust `f64 as usize`: saturating, NaN -> 0 */
    if (!(x > 0.0)) return 0;                     /* NaN, negatives, -0, 0 */
    if (x >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)x;
}
uint8_t oracle_clamp_u8(double x) {               /* f64::clamp(0.0,255.0) as u8, raytracer.rs:97-108 */
    if (x < 0.0) x = 0.0;                         /* f64::clamp keeps NaN */
    if (x > 255.0) x = 255.0;
    if (!(x > 0.0)) return 0;                     /* NaN as u8 == 0 */
    return (uint8_t)x;                            /* truncation */
}

/* ------------------------------------------------------------------ RayTracer, raytracer.rs:28-305 */
typedef struct { uint8_t r, g, b; } ocolor;
static inline uint32_t color_to_u32(ocolor c) { return (uint32_t)c.b + ((uint32_t)c.g << 8) + ((uint32_t)c.r << 16); } /* entities.rs:32-36 */

static ocolor texel(const otexture *t, uint64_t index) { ocolor c = { t->rgb[3 * index], t->rgb[3 * index + 1], t->rgb[3 * index + 2] }; return c; } /* utils.rs:353-361 */

static ovec3 get_normal_at_intersection(const oracle_scene *s, const otrihit *hit, uint64_t tex_x_index, uint64_t tex_y_index) { /* raytracer.rs:114-162 */
    const otriangle *tr = hit->triangle;
    double w = 1.0 - hit->u - hit->v;
    ovec3 n = vadd(vadd(vmul(tr->n2, hit->u), vmul(tr->n3, hit->v)), vmul(tr->n1, w));   /* raytracer.rs:122-124 */
    const omaterial *m = &s->mats[tr->mat];
    if (m->bump >= 0) {                                                                   /* raytracer.rs:126 */
        const otexture *bm = &s->tex[m->bump];
        ocolor bc = texel(bm, (uint64_t)bm->width * tex_y_index + tex_x_index);           /* raytracer.rs:127-128 */
        ovec3 bump_vector = v3((double)bc.r, (double)bc.g, (double)bc.b);                 /* entities.rs:38-46 */
        bump_vector = vnormalised(bump_vector);
        bump_vector = vsub(vmul(bump_vector, 2.0), v3(1.0, 1.0, 1.0));                    /* raytracer.rs:130-135 */
        ovec3 t = vcross(n, v3(0.0, 1.0, 0.0));                                           /* raytracer.rs:137-141 */
        if (vlength(t) == 0.0) t = vcross(n, v3(0.0, 0.0, 1.0));                          /* raytracer.rs:143-149 */
        t = vnormalised(t);
        ovec3 b = vnormalised(vcross(n, t));                                              /* raytracer.rs:152 */
        n = v3(vdot(bump_vector, t), vdot(bump_vector, b), vdot(bump_vector, n));         /* raytracer.rs:154-158 */
    }
    return vnormalised(n);                                                                /* raytracer.rs:161 */
}

static int light_reaches_point(const oracle_scene *s, ovec3 origin, ovec3 normal, ovec3 target, ocounters *c) { /* raytracer.rs:164-188: true when NOT occluded */
    ovec3 direction = vsub(target, origin);
    ovec3 new_origin = vadd(origin, vmul(normal, SURFACE_OFFSET));
    oray ray = { new_origin, direction };
    double max_t = vlength(direction);
    otrihit h;
    c->rays_shadow++;
    return !intersect_with_octant_with_max_t(s, &ray, 0, max_t, &h, c);
}

static ovec3 diffuse(double intensity, double n_dot_l, ovec3 normal, ovec3 l, const omaterial *m) {   /* raytracer.rs:260-277 */
    if (n_dot_l <= 0.0) return v3(0.0, 0.0, 0.0);
    return vdiv(vmul(vmul(m->kd, intensity), n_dot_l), vlength(normal) * vlength(l));
}
static ovec3 specular(double sw, double intensity, ovec3 normal, ovec3 v, ovec3 l, const omaterial *m) { /* raytracer.rs:279-304 */
    if (sw != -1.0) {
        ovec3 r = vsub(vmul(vmul(normal, 2.0), vdot(normal, l)), l);
        double r_dot_v = vdot(r, v);
        if (r_dot_v > 0.0) return vmul(vmul(m->ks, intensity), pow(r_dot_v / (vlength(r) * vlength(v)), sw));
    }
    return v3(0.0, 0.0, 0.0);
}

static ovec3 compute_lighting_intensity(const oracle_scene *s, ovec3 point, ovec3 normal, ovec3 v, const omaterial *m, ocounters *c) { /* raytracer.rs:192-258 */
    ovec3 i = v3(0.0, 0.0, 0.0);
    for (uint32_t k = 0; k < s->n_lights; k++) {
        const olight *L = &s->lights[k];
        if (L->kind == 0) {                                       /* Ambient, raytracer.rs:207-209 */
            i = vadd(i, vmul(m->ka, L->intensity));
        } else if (L->kind == 2) {                                /* Directional, raytracer.rs:210-227 */
            double n_dot_l = vdot(normal, L->v);
            i = vadd(i, diffuse(L->intensity, n_dot_l, normal, L->v, m));
            i = vadd(i, specular(m->ns, L->intensity, normal, v, L->v, m));
        } else {                                                  /* Point, raytracer.rs:228-253 */
            if (!light_reaches_point(s, point, normal, L->v, c)) break;   /* raytracer.rs:235-237: leaves the WHOLE loop */
            ovec3 l = vsub(L->v, point);
            double n_dot_l = vdot(normal, l);
            i = vadd(i, diffuse(L->intensity, n_dot_l, normal, l, m));
            i = vadd(i, specular(m->ns, L->intensity, normal, v, l, m));
        }
    }
    return i;
}

static ocolor get_ray_colour_recursive(const oracle_scene *s, ovec3 origin, ovec3 direction, uint32_t depth, ocounters *c) { /* raytracer.rs:33-112 */
    oray ray = { origin, direction };
    otrihit hit;
    ocolor white = { 255, 255, 255 };                                          /* raytracer.rs:10-14 */
    if (!intersect_with_octant_with_max_t(s, &ray, 0, INFINITY, &hit, c)) return white;   /* raytracer.rs:36, 109-111 */
    c->hits_shaded++;

    ovec3 p = vadd(origin, vmul(direction, hit.t));                            /* raytracer.rs:39 */
    const otriangle *tr = hit.triangle;
    const omaterial *m = &s->mats[tr->mat];
    const otexture *tex = &s->tex[m->tex];
    double w = 1.0 - hit.u - hit.v;
    double tex_x = tr->t2.x * hit.u + tr->t3.x * hit.v + tr->t1.x * w;         /* raytracer.rs:45-47 */
    double tex_y = tr->t2.y * hit.u + tr->t3.y * hit.v + tr->t1.y * w;         /* raytracer.rs:48-50 */
    uint64_t tex_x_index = oracle_f64_as_usize(tex_x * (double)tex->width) % tex->width;    /* raytracer.rs:52 */
    uint64_t tex_y_index = oracle_f64_as_usize(tex_y * (double)tex->height) % tex->height;  /* raytracer.rs:53 */
    ocolor col = texel(tex, (uint64_t)tex->width * tex_y_index + tex_x_index); /* raytracer.rs:55 */

    ovec3 n = get_normal_at_intersection(s, &hit, tex_x_index, tex_y_index);   /* raytracer.rs:57 */
    ovec3 li = compute_lighting_intensity(s, p, n, vneg(direction), m, c);     /* raytracer.rs:59-64 */
    ovec3 local = v3((double)col.r * li.x, (double)col.g * li.y, (double)col.b * li.z);   /* raytracer.rs:67-71 */

    double reflectivity = m->kr;
    if (reflectivity > 0.0 && depth < MAX_REFLECTION_DEPTH) {                  /* raytracer.rs:76 */
        double d_dot_n = vdot(direction, n);
        ovec3 reflect_dir = vnormalised(vsub(direction, vmul(vmul(n, 2.0), d_dot_n)));    /* raytracer.rs:79 */
        ovec3 reflect_origin = vadd(p, vmul(n, SURFACE_OFFSET));               /* raytracer.rs:82 */
        c->rays_reflect++;
        ocolor rc = get_ray_colour_recursive(s, reflect_origin, reflect_dir, depth + 1, c);
        ovec3 reflected = v3((double)rc.r, (double)rc.g, (double)rc.b);
        ovec3 fin = vadd(vmul(local, 1.0 - reflectivity), vmul(reflected, reflectivity)); /* raytracer.rs:95 */
        ocolor out = { oracle_clamp_u8(fin.x), oracle_clamp_u8(fin.y), oracle_clamp_u8(fin.z) };
        return out;
    }
    ocolor out = { oracle_clamp_u8(local.x), oracle_clamp_u8(local.y), oracle_clamp_u8(local.z) };  /* raytracer.rs:104-108 */
    return out;
}

static ocolor color_mix4(ocolor a, ocolor b, ocolor c, ocolor d) {   /* entities.rs:49-69 */
    uint64_t r = (uint64_t)a.r + b.r + c.r + d.r, g = (uint64_t)a.g + b.g + c.g + d.g, bl = (uint64_t)a.b + b.b + c.b + d.b;
    ocolor o = { (uint8_t)(r / 4), (uint8_t)(g / 4), (uint8_t)(bl / 4) }; return o;
}

/* ------------------------------------------------------------------ exported primitives */
static ocolor unpack(uint32_t c) { ocolor o = { (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c }; return o; }
uint32_t oracle_color_mix4(uint32_t c1, uint32_t c2, uint32_t c3, uint32_t c4) { return color_to_u32(color_mix4(unpack(c1), unpack(c2), unpack(c3), unpack(c4))); }
int oracle_intersect_aabb(ovec3 o, ovec3 d, const double box[6], double *t) {
    oray r = { o, d }; oaabb b; for (int k = 0; k < 3; k++) { b.min[k] = box[k]; b.max[k] = box[3 + k]; } return intersect_aabb(&r, &b, t);
}
int oracle_intersect_triangle(ovec3 o, ovec3 d, const double v[9], double *t, double *u, double *vv) {
    oray r = { o, d }; otriangle tr; memset(&tr, 0, sizeof tr); tr.v1 = rd3(v); tr.v2 = rd3(v + 3); tr.v3 = rd3(v + 6);
    otrihit h; if (!intersect_with_triangle(&r, &tr, &h)) return 0; *t = h.t; *u = h.u; *vv = h.v; return 1;
}
int oracle_intersect_scene(const oracle_scene *s, ovec3 o, ovec3 d, double max_t, double *t, double *u, double *v, uint32_t *tri) {
    oray r = { o, d }; otrihit h; ocounters c; memset(&c, 0, sizeof c);
    if (!intersect_with_octant_with_max_t(s, &r, 0, max_t, &h, &c)) return 0;
    *t = h.t; *u = h.u; *v = h.v; *tri = (uint32_t)(h.triangle - s->triangles); return 1;
}
uint32_t oracle_get_ray_colour(const oracle_scene *s, ovec3 o, ovec3 d) {
    ocounters c; memset(&c, 0, sizeof c); return color_to_u32(get_ray_colour_recursive(s, o, d, 0, &c));
}

/* ------------------------------------------------------------------ Scene::draw_scene, engine.rs:186-255 */
typedef struct {
    const oracle_scene *s; int32_t width, height; double x_scale, y_scale, z_value;
    uint32_t *fb; int32_t chunk_start, chunk_end; atomic_int next_row; pthread_barrier_t *bar; int n_chunks; int32_t first;
    ocounters *per_thread;
} render_job;
typedef struct { render_job *job; int tid; } worker_arg;

static void put_pixel(uint32_t *fb, int32_t width, int32_t height, int32_t x, int32_t y, uint32_t color) {  /* engine.rs:146-158 */
    int32_t new_x = x + width / 2;
    int32_t new_y = height - (y + height / 2);
    if (new_x < 0 || new_x >= width || new_y < 0 || new_y >= height) return;
    fb[(size_t)new_y * (size_t)width + (size_t)new_x] = color;
}

static void render_row(const render_job *j, int32_t y, ocounters *c) {   /* engine.rs:203-243 */
    const oracle_scene *s = j->s;
    for (int32_t x = -(j->width / 2); x < j->width / 2; x++) {
        ovec3 d1 = v3((double)x * j->x_scale, (double)y * j->y_scale, j->z_value);
        ovec3 d2 = v3(((double)x + 0.5) * j->x_scale, (double)y * j->y_scale, j->z_value);
        ovec3 d3 = v3((double)x * j->x_scale, ((double)y + 0.5) * j->y_scale, j->z_value);
        ovec3 d4 = v3(((double)x + 0.5) * j->x_scale, ((double)y + 0.5) * j->y_scale, j->z_value);
        c->rays_primary += 4;
        ocolor c1 = get_ray_colour_recursive(s, s->origin, d1, 0, c);
        ocolor c2 = get_ray_colour_recursive(s, s->origin, d2, 0, c);
        ocolor c3 = get_ray_colour_recursive(s, s->origin, d3, 0, c);
        ocolor c4 = get_ray_colour_recursive(s, s->origin, d4, 0, c);
        /* rows write disjoint pixels, so applying put_pixel here instead of after the chunk (engine.rs:246-250) is equivalent */
        put_pixel(j->fb, j->width, j->height, x, y, color_to_u32(color_mix4(c1, c2, c3, c4)));
    }
}

static void *worker(void *argp) {
    worker_arg *a = argp; render_job *j = a->job; ocounters *c = &j->per_thread[a->tid];
    int32_t half = j->height / 2;
    for (int32_t chunk_start = -half; chunk_start < half; chunk_start += 50) {         /* engine.rs:196-199, sequential chunks */
        int32_t chunk_end = chunk_start + 50 < half ? chunk_start + 50 : half;
        for (;;) {                                                                     /* engine.rs:201-203, rows handed out dynamically */
            int32_t y = atomic_fetch_add(&j->next_row, 1);
            if (y >= chunk_end) break;
            render_row(j, y, c);
        }
        int r = pthread_barrier_wait(j->bar);                                          /* chunk barrier (engine.rs:246-253 run between chunks) */
        if (r == PTHREAD_BARRIER_SERIAL_THREAD) atomic_store(&j->next_row, chunk_end);
        pthread_barrier_wait(j->bar);
    }
    return NULL;
}

int oracle_render(const oracle_scene *s, uint32_t width, uint32_t height, const double vp[3],
                  uint32_t n_threads, uint32_t *fb, ocounters *counters_out) {
    if (n_threads == 0) n_threads = 1;
    memset(fb, 0, sizeof(uint32_t) * (size_t)width * height);                          /* engine.rs:135 */
    render_job j; memset(&j, 0, sizeof j);
    j.s = s; j.width = (int32_t)width; j.height = (int32_t)height; j.fb = fb;
    j.x_scale = vp[0] / (double)width;                                                 /* engine.rs:189 */
    j.y_scale = vp[1] / (double)height;                                                /* engine.rs:190 */
    j.z_value = vp[2];                                                                 /* engine.rs:191 */
    atomic_init(&j.next_row, -(j.height / 2));
    pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, n_threads); j.bar = &bar;
    j.per_thread = calloc(n_threads, sizeof(ocounters));
    pthread_t *th = malloc(sizeof(pthread_t) * n_threads); worker_arg *args = malloc(sizeof(worker_arg) * n_threads);
    for (uint32_t t = 0; t < n_threads; t++) { args[t].job = &j; args[t].tid = (int)t; pthread_create(&th[t], NULL, worker, &args[t]); }
    for (uint32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    if (counters_out) {
        memset(counters_out, 0, sizeof *counters_out);
        for (uint32_t t = 0; t < n_threads; t++) {
            const ocounters *c = &j.per_thread[t];
            counters_out->rays_primary += c->rays_primary; counters_out->rays_shadow += c->rays_shadow; counters_out->rays_reflect += c->rays_reflect;
            counters_out->tri_tests += c->tri_tests; counters_out->aabb_tests += c->aabb_tests; counters_out->nodes_entered += c->nodes_entered;
            counters_out->hits_shaded += c->hits_shaded;
        }
    }
    pthread_barrier_destroy(&bar); free(j.per_thread); free(th); free(args);
    return 0;
}
