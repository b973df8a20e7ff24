/*
 * yahr_oracle.c -- CPU restatement of the yahr render hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Literal restatement in plain C of the Haskell reference pstiasny/yahr (no GHC exists in this
 * environment, so the reference itself cannot be built or run).  Every function cites the reference
 * file:line it follows; float operation ORDER is kept exactly (Haskell fixities noted inline).
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 *
 * Pinned by the reference's own known-answer tests (test/Spec.hs) -- see oracle/kat.c and
 * tests/golden/spec_kat.json.  Third-party arithmetic that is NOT in /root/reference:
 *   - `linear` (unpinned in yahr.cabal:22): M44 !*! and !* -- restated as left folds from zero
 *     (Linear.Matrix: foldl' (^+^) zero; Foldable.sum as foldl).  Pinned only to 1e-4 by the camera
 *     KATs (Spec.hs:62-108); ulp-level association order is "parity unpinned".
 *   - libm powf/sqrtf via GHC primops (BSDF.hs:25, Vectors.hs:40).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include "yahr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Vectors.hs */

static inline ovec3 v3(float x, float y, float z) { ovec3 r = {x, y, z}; return r; }
static inline ovec3 vof(float l) { return v3(l, l, l); }                                   /* :20-21 */
static inline ovec3 vadd(ovec3 a, ovec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); } /* :24 */
static inline ovec3 vsub(ovec3 a, ovec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); } /* :25 */
static inline ovec3 vmul(ovec3 a, ovec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); } /* :26 */
static inline ovec3 vneg(ovec3 a) { return v3(-a.x, -a.y, -a.z); }                         /* :27 */
/* (.*) :33  x1*y1 + x2*y2 + x3*y3, + is infixl 6 */
static inline float dot(ovec3 a, ovec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float lensq(ovec3 v) { return dot(v, v); }                                    /* :36-37 */
static inline float len(ovec3 v) { return sqrtf(lensq(v)); }                                /* :39-40 */
static inline ovec3 scale(float f, ovec3 v) { return vmul(vof(f), v); }                     /* (@*) :42-43 */
static inline ovec3 norm(ovec3 v) { return scale(1.0f / len(v), v); }                       /* :46-47 */
static inline ovec3 cross(ovec3 a, ovec3 b) {                                               /* :49-53 */
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float getdim(int d, ovec3 v) { return d == 0 ? v.x : (d == 1 ? v.y : v.z); }   /* :57-60 */
static inline int maxdim(ovec3 v) {                                                          /* :62-66 */
  if (v.x > v.y && v.x > v.z) return 0;
  if (v.y > v.z) return 1;
  return 2;
}
/* Haskell's class-default Ord methods (GHC.Classes): NaN-asymmetric, do not replace by fminf/fmaxf */
static inline float hmax(float x, float y) { return (x <= y) ? y : x; }
static inline float hmin(float x, float y) { return (x <= y) ? x : y; }
static inline ovec3 vzip_min(ovec3 a, ovec3 b) { return v3(hmin(a.x, b.x), hmin(a.y, b.y), hmin(a.z, b.z)); }
static inline ovec3 vzip_max(ovec3 a, ovec3 b) { return v3(hmax(a.x, b.x), hmax(a.y, b.y), hmax(a.z, b.z)); }

static inline ovec3 ld3(const float p[3]) { return v3(p[0], p[1], p[2]); }
static inline void st3(float p[3], ovec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

/* ------------------------------------------------------------------ Rays.hs */

static inline ovec3 ray_at(oray r, float t) { return vadd(r.x0, scale(t, r.u)); }           /* :35 */

/* ------------------------------------------------------------------ AABBs.hs */

static obox box_empty(void) {                                                                /* :10-11 */
  obox b; b.bmin = vof(INFINITY); b.bmax = vof(-INFINITY); return b;
}
static obox box_from_points(ovec3 x, ovec3 y) {                                              /* :13-14 */
  obox b; b.bmin = vzip_min(x, y); b.bmax = vzip_max(x, y); return b;
}
static obox box_include(obox b, ovec3 x) {                                                   /* :16-18 */
  obox r; r.bmin = vzip_min(b.bmin, x); r.bmax = vzip_max(b.bmax, x); return r;
}
static obox box_join(obox a, obox b) {                                                       /* :25-27 */
  obox r; r.bmin = vzip_min(a.bmin, b.bmin); r.bmax = vzip_max(a.bmax, b.bmax); return r;
}
static ovec3 box_centroid(obox b) { return vadd(scale(0.5f, b.bmin), scale(0.5f, b.bmax)); } /* :48-49 */
static float box_surf(obox b) {                                                              /* :51-53 */
  ovec3 d = vsub(b.bmax, b.bmin);
  return 2.0f * ((d.x * d.y + d.x * d.z) + d.y * d.z);
}
static int box_max_extent(obox b) { return maxdim(vsub(b.bmax, b.bmin)); }                   /* :45-46 */

/* bbRayIntersection :29-40.  foldl over [X,Y,Z] starting at (0, tMax). */
static int bb_ray(obox b, oray r, float *tnear_out) {
  float tnear = 0.0f, tfar = r.tmax;
  for (int d = 0; d < 3; ++d) {
    float inv = 1.0f / getdim(d, r.u);
    float t0 = (getdim(d, b.bmin) - getdim(d, r.x0)) * inv;
    float t1 = (getdim(d, b.bmax) - getdim(d, r.x0)) * inv;
    float dnear = hmin(t0, t1);
    float dfar = hmax(t0, t1);
    tnear = hmax(tnear, dnear);
    tfar = hmin(tfar, dfar);
  }
  if (tnear > tfar) return 0;
  if (tnear_out) *tnear_out = tnear;
  return 1;
}

/* ------------------------------------------------------------------ Shapes.hs */

typedef struct {
  int kind;               /* 0 triangle, 1 sphere */
  ovec3 p0, p1, p2, n0, n1, n2; /* triangle; sphere: p0 = centre */
  float r;
  int mat;
} oprim;

/* collideSphere :13-29 */
static int collide_sphere(int what, float r, ovec3 s, oray ray, ohit *out) {
  ovec3 d = vsub(ray.x0, s);
  float a = dot(ray.u, ray.u);
  float b = dot(vmul(vof(2.0f), d), ray.u); /* 2 * d .* u : both infixl 7 -> (2*d) .* u */
  float c = dot(d, d) - r * r;
  float delta = b * b - (4.0f * a) * c;      /* 4 * a * c = (4*a)*c */
  if (delta < 0.0f) return 0;
  float ts[2];
  ts[0] = ((-b) - sqrtf(delta)) / (2.0f * a);
  ts[1] = ((-b) + sqrtf(delta)) / (2.0f * a);
  for (int i = 0; i < 2; ++i) {
    float t = ts[i];
    if (t > 0.0f && t <= ray.tmax) {
      ovec3 x = ray_at(ray, t);
      ovec3 n = norm(vsub(x, s));
      out->t = t;
      out->dg.point = x;
      out->dg.normal = n;
      out->dg.dpdu = cross(n, v3(0, 0, 1));
      out->dg.dpdv = cross(n, v3(0, 1, 0));
      out->what = what;
      return 1;
    }
  }
  return 0;
}
static obox bound_sphere(float r, ovec3 s) {                                                 /* :32-33 */
  return box_from_points(vadd(s, vof(r)), vsub(s, vof(r)));
}

/* collideTriangle :36-59 */
static int collide_triangle(int what, ovec3 p0, ovec3 p1, ovec3 p2, ovec3 n0, ovec3 n1, ovec3 n2,
                            oray ray, ohit *out) {
  ovec3 e1 = vsub(p1, p0);
  ovec3 e2 = vsub(p2, p0);
  ovec3 s = vsub(ray.x0, p0);
  ovec3 s1 = cross(ray.u, e2);
  ovec3 s2 = cross(s, e1);
  float inv = 1.0f / dot(s1, e1);
  float t = dot(s2, e2) * inv;
  float b1 = dot(s1, s) * inv;
  float b2 = dot(s2, ray.u) * inv;
  float b0 = (1.0f - b1) - b2;
  ovec3 ns = vadd(vadd(scale(b0, n0), scale(b1, n1)), scale(b2, n2));
  if (b0 >= 0.0f && b0 <= 1.0f && b1 >= 0.0f && b1 <= 1.0f && b2 >= 0.0f && b2 <= 1.0f &&
      t > 0.0f && t <= ray.tmax && dot(ray.u, ns) < 0.0f) {
    ovec3 ss = norm(e2);
    ovec3 ts = norm(cross(ss, ns));
    out->t = t;
    out->dg.point = vadd(ray.x0, scale(t, ray.u));
    out->dg.normal = ns;
    out->dg.dpdu = cross(ts, ns);
    out->dg.dpdv = ts;
    out->what = what;
    return 1;
  }
  return 0;
}
static obox bound_triangle(ovec3 p0, ovec3 p1, ovec3 p2) {                                   /* :62-63 */
  return box_include(box_from_points(p0, p1), p2);
}

/* ------------------------------------------------------------------ BSDF.hs / Shaders.hs */

typedef struct {
  int raw_blinn;          /* test-only: BSDF = Blinn shininess (Spec.hs:279) */
  ovec3 diffuse, specular;
  float shininess;
} omat;

static const float O_PI = 3.14159274101257324f; /* pi :: Float */

typedef struct { ovec3 nn, sn, tn; } oframe;
static oframe bsdf_frame(const odg *dg) {                                                    /* BSDF.hs:36-38 */
  oframe f;
  f.nn = dg->normal;
  f.sn = norm(dg->dpdu);
  f.tn = cross(f.nn, f.sn);
  return f;
}
static ovec3 world_to_loc(const oframe *f, ovec3 v) {                                        /* :40 */
  return v3(dot(v, f->sn), dot(v, f->tn), dot(v, f->nn));
}
static ovec3 loc_lambertian(ovec3 i, ovec3 o) {                                              /* :15-17 */
  if (i.z > 0.0f && o.z > 0.0f) return vof(1.0f / O_PI);
  return vof(0.0f);
}
static ovec3 loc_blinn(float e, ovec3 i, ovec3 o) {                                          /* :18-30 */
  if (i.z > 0.0f && o.z > 0.0f) {
    ovec3 h = norm(vadd(i, o));
    float cos_o = fabsf(o.z);
    float cos_i = fabsf(i.z);
    float cos_h = dot(i, h);
    float o_dot_h = dot(o, h);
    float d = ((e + 2.0f) * powf(fabsf(h.z), e)) / (2.0f * O_PI);
    float g = hmin(1.0f, hmin(((2.0f * cos_h) * cos_o) / o_dot_h, ((2.0f * cos_h) * cos_i) / o_dot_h));
    float f = 1.0f;                                                                          /* fresnel :34 */
    return vof(((d * g) * f) / ((4.0f * cos_i) * cos_o));
  }
  return vof(0.0f);
}
/* blinnPhong (Shaders.hs:12-14) = Composite [Scaled diffuse Lambertian, Scaled specular (Blinn s)];
 * Composite = sum from 0 in list order (BSDF.hs:32) */
static ovec3 mat_bsdf_at(const omat *m, const odg *dg, ovec3 wi, ovec3 wo) {                  /* BSDF.hs:12-13 */
  oframe f = bsdf_frame(dg);
  ovec3 i = world_to_loc(&f, wi), o = world_to_loc(&f, wo);
  if (m->raw_blinn) return loc_blinn(m->shininess, i, o);
  ovec3 acc = vof(0.0f);
  acc = vadd(acc, vmul(m->diffuse, loc_lambertian(i, o)));
  acc = vadd(acc, vmul(m->specular, loc_blinn(m->shininess, i, o)));
  return acc;
}

/* ------------------------------------------------------------------ Culling.hs */

enum { N_INTERIOR = 0, N_SINGLE = 1, N_MULTI = 2, N_NULL = 3 };
typedef struct onode {
  int kind;
  obox bb;
  struct onode *l, *r;
  int32_t *prims;
  int nprims;
} onode;

typedef struct { obox bb; int32_t prim; } oitem;

typedef struct { ovec3 pos, spectrum; } olight;

struct oscene {
  float imW, imH, focal; ovec3 look, up, pos;
  float tf[16], vtf[16]; int cam_ready;
  int depth;
  int bvh_depth, split;
  omat *mats; int n_mats, cap_mats;
  olight *lights; int n_lights, cap_lights;
  oprim *prims; int n_prims, cap_prims;
  onode *root; int built;
};

static onode *node_new(int kind) {
  onode *n = (onode *)calloc(1, sizeof(onode));
  n->kind = kind;
  n->bb = box_empty();
  return n;
}
static void node_free(onode *n) {
  if (!n) return;
  node_free(n->l); node_free(n->r);
  free(n->prims); free(n);
}

/* multiLeaf :52-54 */
static onode *multi_leaf(const oitem *xs, int n) {
  onode *nd = node_new(N_MULTI);
  obox bb = box_empty();
  for (int i = 0; i < n; ++i) bb = box_join(bb, xs[i].bb);   /* foldl join empty bbs */
  nd->bb = bb;
  nd->nprims = n;
  nd->prims = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) nd->prims[i] = xs[i].prim;
  return nd;
}

/* boundAllPoints (map centroid bbs)  AABBs.hs:20-23 */
static obox bound_centroids(const oitem *xs, int n) {
  ovec3 c0 = box_centroid(xs[0].bb);
  if (n == 1) return box_from_points(c0, c0);
  obox b = box_from_points(c0, box_centroid(xs[1].bb));
  for (int i = 2; i < n; ++i) b = box_include(b, box_centroid(xs[i].bb));
  return b;
}

/* midpointSplit :56-60 -- stable partition, (<= midpoint) goes left */
static int midpoint_split(const oitem *xs, int n, int dim, obox cb, oitem *left, oitem *right, int *nr) {
  float mid = getdim(dim, box_centroid(cb));
  int nl = 0; *nr = 0;
  for (int i = 0; i < n; ++i) {
    if (getdim(dim, box_centroid(xs[i].bb)) <= mid) left[nl++] = xs[i]; else right[(*nr)++] = xs[i];
  }
  return nl;
}

/* bucketId :69-79.  GHC's `floor :: Float -> Int` lowers to a truncating float2Int (cvttss2si) plus a
 * sign fix-up; NaN and +-inf come out as minBound, which the following `max 0` turns into 0. */
static int sah_bucket_id(obox bb, int dim, float cmin, float cmax) {
  float length = cmax - cmin;
  float d = getdim(dim, box_centroid(bb)) - cmin;
  float frac = d / length;
  float v = 16.0f * frac;
  long fl = (isnan(v) || isinf(v)) ? -1L : (long)floorf(v);
  if (fl < 0) fl = 0;             /* max 0 */
  if (fl > 15) fl = 15;           /* min (nBuckets - 1) */
  return (int)fl;
}

/* Ord Float `compare` (GHC.Classes): x<y -> LT; x==y -> EQ; otherwise GT (so any NaN gives GT) */
static int hcompare(float x, float y) { return x < y ? -1 : (x == y ? 0 : 1); }

/* sahSplit :62-112 */
static int sah_split(const oitem *xs, int n, int dim, obox cb, oitem *left, oitem *right, int *nr) {
  if (!(n > 4)) return midpoint_split(xs, n, dim, cb, left, right, nr);     /* :110-112 */
  enum { NB = 16 };
  float cmin = getdim(dim, cb.bmin), cmax = getdim(dim, cb.bmax);
  int cnt[NB]; obox bbs[NB];
  for (int i = 0; i < NB; ++i) { cnt[i] = 0; bbs[i] = box_empty(); }
  for (int i = 0; i < n; ++i) {                                              /* :82-90 */
    int b = sah_bucket_id(xs[i].bb, dim, cmin, cmax);
    cnt[b] += 1; bbs[b] = box_join(bbs[b], xs[i].bb);
  }
  float wrapped = 0.0f;                                                      /* :92-93 */
  for (int i = 0; i < NB; ++i) wrapped = wrapped + box_surf(bbs[i]);
  int best_i = 0; float best_c = 0.0f;
  for (int i = 0; i <= NB - 2; ++i) {                                        /* :95-108 */
    obox b0 = box_empty(), b1 = box_empty();
    int c0 = 0, c1 = 0;
    for (int k = 0; k <= i; ++k) { b0 = box_join(b0, bbs[k]); c0 += cnt[k]; }
    for (int k = i + 1; k < NB; ++k) { b1 = box_join(b1, bbs[k]); c1 += cnt[k]; }
    float cost = 0.125f + ((float)c0 * box_surf(b0) + (float)c1 * box_surf(b1)) / wrapped;
    /* minimumBy = foldl1 (\x y -> case cmp x y of GT -> y; _ -> x) */
    if (i == 0) { best_i = 0; best_c = cost; }
    else if (hcompare(best_c, cost) > 0) { best_i = i; best_c = cost; }
  }
  int nl = 0; *nr = 0;
  for (int i = 0; i < n; ++i) {
    if (sah_bucket_id(xs[i].bb, dim, cmin, cmax) <= best_i) left[nl++] = xs[i]; else right[(*nr)++] = xs[i];
  }
  return nl;
}

/* buildTree :31-50 */
static onode *build_tree(int max_depth, int split, const oitem *xs, int n) {
  if (n == 0) return node_new(N_NULL);                                       /* :32 */
  if (n == 1) {                                                              /* :33 */
    onode *nd = node_new(N_SINGLE);
    nd->bb = xs[0].bb; nd->nprims = 1;
    nd->prims = (int32_t *)malloc(sizeof(int32_t)); nd->prims[0] = xs[0].prim;
    return nd;
  }
  if (max_depth == 0) return multi_leaf(xs, n);                              /* :34 */
  obox cb = bound_centroids(xs, n);                                          /* :49 */
  int dim = box_max_extent(cb);                                              /* :48 */
  oitem *left = (oitem *)malloc(sizeof(oitem) * n), *right = (oitem *)malloc(sizeof(oitem) * n);
  int nr = 0, nl;
  if (split == OSPLIT_SAH) nl = sah_split(xs, n, dim, cb, left, right, &nr);
  else nl = midpoint_split(xs, n, dim, cb, left, right, &nr);
  onode *nd;
  if (nr == 0) {                                                             /* :36-37 */
    nd = multi_leaf(left, nl);
  } else {                                                                   /* :38-41 */
    nd = node_new(N_INTERIOR);
    nd->l = build_tree(max_depth - 1, split, left, nl);
    nd->r = build_tree(max_depth - 1, split, right, nr);
    nd->bb = box_join(nd->l->bb, nd->r->bb);
  }
  free(left); free(right);
  return nd;
}

typedef struct { ostats st; } octx;

static int collide_prim(const oscene *s, int pi, oray ray, ohit *out, octx *cx) {
  const oprim *p = &s->prims[pi];
  if (p->kind == 1) { cx->st.sphere_tests++; return collide_sphere(pi, p->r, p->p0, ray, out); }
  cx->st.tri_tests++;
  return collide_triangle(pi, p->p0, p->p1, p->p2, p->n0, p->n1, p->n2, ray, out);
}

/* the closure tree of Culling.hs:24-25,33,38,52 + AABBs.hs:42-43 + Rays.hs:39-46 */
static int collide_node(const oscene *s, const onode *nd, oray ray, ohit *out, octx *cx) {
  if (nd->kind == N_NULL) return 0;                                          /* nullCollider */
  cx->st.box_tests++;
  if (!bb_ray(nd->bb, ray, NULL)) return 0;                                  /* wrapCollider */
  if (nd->kind == N_SINGLE) return collide_prim(s, nd->prims[0], ray, out, cx);
  int have = 0; ohit h;
  if (nd->kind == N_MULTI) {                                                 /* collideAll colliders */
    for (int i = 0; i < nd->nprims; ++i)
      if (collide_prim(s, nd->prims[i], ray, &h, cx)) { have = 1; *out = h; ray.tmax = h.t; }
    return have;
  }
  /* collideBVHNodes: collideAll [col1, col2] */
  if (collide_node(s, nd->l, ray, &h, cx)) { have = 1; *out = h; ray.tmax = h.t; }
  if (collide_node(s, nd->r, ray, &h, cx)) { have = 1; *out = h; ray.tmax = h.t; }
  return have;
}

/* Optional first-hit variant of the same traversal, used ONLY to validate the HIP kernel's box/primitive
 * counters (its shadow rays stop at the first accepted primitive).  Same Bool as the full traversal:
 * once collideAll holds a Just it never returns to Nothing (Rays.hs:42-46). */
static int g_shadow_any_hit = 0;
void oracle_set_shadow_any_hit(int on) { g_shadow_any_hit = on; }
static int any_node(const oscene *s, const onode *nd, oray ray, octx *cx) {
  ohit h;
  if (nd->kind == N_NULL) return 0;
  cx->st.box_tests++;
  if (!bb_ray(nd->bb, ray, NULL)) return 0;
  if (nd->kind == N_SINGLE) return collide_prim(s, nd->prims[0], ray, &h, cx);
  if (nd->kind == N_MULTI) {
    for (int i = 0; i < nd->nprims; ++i) if (collide_prim(s, nd->prims[i], ray, &h, cx)) return 1;
    return 0;
  }
  if (any_node(s, nd->l, ray, cx)) return 1;
  return any_node(s, nd->r, ray, cx);
}

static int root_collide(const oscene *s, oray ray, ohit *out, octx *cx) {
  if (!s->root) return 0;                                                    /* bvh _ _ [] = const Nothing :28 */
  return collide_node(s, s->root, ray, out, cx);
}

/* reachable Rays.hs:49-54 */
static int reachable(const oscene *s, ovec3 p0, ovec3 p1, octx *cx) {
  oray probe; ohit h;
  probe.x0 = p0;
  probe.u = norm(vsub(p1, p0));
  probe.tmax = len(vsub(p1, p0));
  if (g_shadow_any_hit) return !(s->root && any_node(s, s->root, probe, cx));
  return !root_collide(s, probe, &h, cx);
}

/* ------------------------------------------------------------------ Integrators.hs / Lights.hs */

static ovec3 reflection_dir(ovec3 u, ovec3 n) {                                               /* :46-47 */
  return vsub(u, scale(2.0f * dot(u, n), n));
}

static ovec3 direct_illumination(const oscene *s, const odg *dg, oray ray, const omat *m, octx *cx) { /* :50-61 */
  ovec3 x = dg->point, n = dg->normal;
  ovec3 acc = vof(0.0f);                                      /* sum = foldl (+) 0 */
  for (int li = 0; li < s->n_lights; ++li) {
    const olight *L = &s->lights[li];
    /* illuminationAtPoint Lights.hs:15-24 */
    ovec3 p2l = vsub(L->pos, x);
    ovec3 ldir = norm(p2l);
    ovec3 intensity = scale(1.0f / lensq(p2l), L->spectrum);
    ovec3 k = mat_bsdf_at(m, dg, ldir, vneg(ray.u));
    ovec3 contrib = vof(0.0f);
    if (lensq(k) > 0.0f) {                                    /* && is lazy: probe only if k != 0 */
      cx->st.shadow_rays++;
      int unoccluded = reachable(s, vadd(x, scale(0.001f, ldir)), L->pos, cx);
      if (unoccluded) contrib = vmul(scale(fabsf(dot(ldir, n)), k), intensity);
    }
    acc = vadd(acc, contrib);
  }
  return acc;
}

static ovec3 vcast(const oscene *s, int max_depth, oray ray, octx *cx, int is_primary) {        /* :25-30 */
  if (max_depth == 0) return v3(0, 0, 0);
  if (is_primary) cx->st.primary_rays++; else cx->st.reflection_rays++;
  ohit hit;
  if (!root_collide(s, ray, &hit, cx)) return v3(0, 0, 0);
  /* vhit :32-43 */
  cx->st.shaded_hits++;
  const omat *m = &s->mats[s->prims[hit.what].mat];
  ovec3 x = hit.dg.point, n = hit.dg.normal;
  ovec3 r = reflection_dir(ray.u, n);
  oray rr; rr.x0 = vadd(x, scale(0.001f, r)); rr.u = r; rr.tmax = 1e6f;
  ovec3 rs = vcast(s, max_depth - 1, rr, cx, 0);
  ovec3 fr = mat_bsdf_at(m, &hit.dg, r, vneg(ray.u));
  ovec3 direct = direct_illumination(s, &hit.dg, ray, m, cx);
  /* (n .* r) @* f r * rs + direct   -- .*, @*, * all infixl 7 */
  return vadd(vmul(scale(dot(n, r), fr), rs), direct);
}

/* ------------------------------------------------------------------ Cameras.hs */

/* Linear.Matrix (!*!): row_i = foldl' (^+^) zero (liftI2 (*^) f_i g) */
static void m44_mul(const float f[16], const float g[16], float out[16]) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < 4; ++k) acc = acc + f[i * 4 + k] * g[k * 4 + j];
      out[i * 4 + j] = acc;
    }
}
/* transformPoint :40-44 : (tf !* V4 x y z 1) then normalizePoint = (1/w) *^ V3 */
static ovec3 transform_point(const float m[16], ovec3 p) {
  float v[4] = {p.x, p.y, p.z, 1.0f}, tv[4];
  for (int i = 0; i < 4; ++i) {
    float acc = 0.0f;
    for (int k = 0; k < 4; ++k) acc = acc + m[i * 4 + k] * v[k];
    tv[i] = acc;
  }
  float iw = 1.0f / tv[3];
  return v3(iw * tv[0], iw * tv[1], iw * tv[2]);
}
static void camera_prepare(oscene *s) {
  /* cameraToWorld :58-65 */
  ovec3 forward = norm(s->look);
  ovec3 left = norm(cross(forward, s->up));
  ovec3 up = cross(left, forward);
  ovec3 a = vneg(left), b = up, c = forward;
  float dir_tf[16] = {a.x, b.x, c.x, 0, a.y, b.y, c.y, 0, a.z, b.z, c.z, 0, 0, 0, 0, 1};  /* fromBasis :24-29 */
  float tr[16] = {1, 0, 0, s->pos.x, 0, 1, 0, s->pos.y, 0, 0, 1, s->pos.z, 0, 0, 0, 1};  /* translate :32-37 */
  m44_mul(tr, dir_tf, s->tf);
  /* rasterToCamera :68-75 */
  float frame_w = s->imW / s->imH, frame_h = 1.0f;
  float r2c[16] = {frame_w / s->imW, 0, 0, -(frame_w / 2.0f),
                   0, -(frame_h / s->imH), 0, frame_h / 2.0f,
                   0, 0, 1, 0,
                   0, 0, 0, 1};
  m44_mul(s->tf, r2c, s->vtf);                                                /* :81 */
  s->cam_ready = 1;
}
static oray camera_ray(const oscene *s, float u, float v) {                                  /* :83-86 */
  ovec3 origin = transform_point(s->tf, v3(0, 0, 0));
  ovec3 direction = vsub(transform_point(s->vtf, v3(u, v, s->focal)), origin);
  oray r; r.x0 = origin; r.u = norm(direction); r.tmax = 1e6f;
  return r;
}

/* ------------------------------------------------------------------ Sampling.hs / main.hs */

void oracle_batch_window(int w, int h, int num, int count, int32_t out[4]) {                 /* Sampling.hs:9-21 */
  long nx = count, ny = 1;
  while (nx % 2 == 0 && 2L * w * ny < (long)h * nx) { nx = nx / 2; ny = ny * 2; }
  long j = num / nx, i = num % nx;  /* quotRem */
  out[0] = (int32_t)floor((double)i / (double)nx * (double)w);
  out[1] = (int32_t)floor((double)j / (double)ny * (double)h);
  out[2] = (int32_t)floor(((double)i + 1.0) / (double)nx * (double)w);
  out[3] = (int32_t)floor(((double)j + 1.0) / (double)ny * (double)h);
}
int oracle_n_batches(int num_threads, int width, int height) {                               /* main.hs:109-110,128-130 */
  /* roundUpPow2 $ max (32*numThreads) width * height `div` (16*16)
   * parses as roundUpPow2 (((max (32*numThreads) width) * height) `div` 256) */
  long m = 32L * num_threads > width ? 32L * num_threads : width;
  long x = (m * height) / 256;
  double e = ceil(log((double)x) / log(2.0));                                 /* logBase 2 */
  return (int)llround(pow(2.0, e));
}

/* Sampler.  spp == 1 is the reference: the integer pixel corner (main.hs:73), no jitter.
 * spp > 1 is a BUILD EXTENSION: n x n strata when spp = n*n, else one stratum; jitter from
 * a counter-based integer hash.  Shared definition with the HIP kernel (DESIGN.md "Sampler"). */
static uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
void oracle_sample_offset(int u, int v, int k, int spp, uint32_t seed, float *du, float *dv) {
  if (spp <= 1) { *du = 0.0f; *dv = 0.0f; return; }
  int n = (int)floor(sqrt((double)spp) + 0.5);
  if (n * n != spp) n = 1;
  int i = (n > 1) ? (k % n) : 0, j = (n > 1) ? ((k / n) % n) : 0;
  uint32_t h = lowbias32((uint32_t)u + lowbias32((uint32_t)v + lowbias32((uint32_t)k + lowbias32(seed))));
  uint32_t h2 = lowbias32(h ^ 0x68bc21ebU);
  float xi1 = (float)(h >> 8) * (1.0f / 16777216.0f);
  float xi2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
  *du = ((float)i + xi1) / (float)n;
  *dv = ((float)j + xi2) / (float)n;
}

static ovec3 render_pixel(const oscene *s, int u, int v, int spp, uint32_t seed, octx *cx) {
  if (spp <= 1) {                                                             /* main.hs:73 */
    return vcast(s, s->depth, camera_ray(s, (float)u, (float)v), cx, 1);
  }
  ovec3 acc = vof(0.0f);
  for (int k = 0; k < spp; ++k) {
    float du, dv;
    oracle_sample_offset(u, v, k, spp, seed, &du, &dv);
    acc = vadd(acc, vcast(s, s->depth, camera_ray(s, (float)u + du, (float)v + dv), cx, 1));
  }
  return scale(1.0f / (float)spp, acc);
}

typedef struct {
  oscene *s; int w, h, row0, row1, spp; uint32_t seed; float *rgb;
  int n_batches; int use_rows; atomic_int next; ostats st; pthread_mutex_t mu;
} ojob;

static void stats_add(ostats *a, const ostats *b) {
  a->primary_rays += b->primary_rays; a->reflection_rays += b->reflection_rays;
  a->shadow_rays += b->shadow_rays; a->box_tests += b->box_tests; a->tri_tests += b->tri_tests;
  a->sphere_tests += b->sphere_tests; a->shaded_hits += b->shaded_hits;
}

static void *worker(void *arg) {
  ojob *job = (ojob *)arg;
  octx cx; memset(&cx, 0, sizeof cx);
  for (;;) {
    int b = atomic_fetch_add(&job->next, 1);
    if (job->use_rows) {                         /* lineBatches (Sampling.hs:3): one batch per row */
      int v = job->row0 + b;
      if (v >= job->row1) break;
      for (int u = 0; u < job->w; ++u) {
        ovec3 c = render_pixel(job->s, u, v, job->spp, job->seed, &cx);
        float *px = job->rgb + ((size_t)(v - job->row0) * job->w + u) * 3;
        px[0] = c.x; px[1] = c.y; px[2] = c.z;
      }
    } else {                                     /* squareBatches (Sampling.hs:5-7), u-major inside a tile */
      if (b >= job->n_batches) break;
      int32_t win[4];
      oracle_batch_window(job->w, job->h, b, job->n_batches, win);
      for (int u = win[0]; u < win[2]; ++u)
        for (int v = win[1]; v < win[3]; ++v) {
          ovec3 c = render_pixel(job->s, u, v, job->spp, job->seed, &cx);
          float *px = job->rgb + ((size_t)job->w * v + u) * 3;                /* main.hs:100 */
          px[0] = c.x; px[1] = c.y; px[2] = c.z;
        }
    }
  }
  pthread_mutex_lock(&job->mu);
  stats_add(&job->st, &cx.st);
  pthread_mutex_unlock(&job->mu);
  return NULL;
}

static void run_job(ojob *job, int n_threads, ostats *stats) {
  if (n_threads < 1) n_threads = 1;
  atomic_init(&job->next, 0);
  memset(&job->st, 0, sizeof job->st);
  pthread_mutex_init(&job->mu, NULL);
  if (n_threads == 1) {
    worker(job);
  } else {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, worker, job);
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    free(th);
  }
  pthread_mutex_destroy(&job->mu);
  if (stats) *stats = job->st;
}

/* ------------------------------------------------------------------ public API */

oscene *oracle_scene_new(void) {
  oscene *s = (oscene *)calloc(1, sizeof(oscene));
  s->depth = 1; s->bvh_depth = 16; s->split = OSPLIT_MIDPOINT;
  return s;
}
void oracle_scene_free(oscene *s) {
  if (!s) return;
  node_free(s->root);
  free(s->mats); free(s->lights); free(s->prims); free(s);
}
void oracle_scene_set_camera(oscene *s, float imW, float imH, float focal,
                             const float look[3], const float up[3], const float pos[3]) {
  s->imW = imW; s->imH = imH; s->focal = focal;
  s->look = ld3(look); s->up = ld3(up); s->pos = ld3(pos);
  camera_prepare(s);
}
void oracle_scene_set_integrator(oscene *s, int d) { s->depth = d; }
void oracle_scene_set_culling(oscene *s, int max_depth, int split) { s->bvh_depth = max_depth; s->split = split; }

#define GROW(arr, n, cap, T) do { if ((n) == (cap)) { (cap) = (cap) ? (cap) * 2 : 16; \
  (arr) = (T *)realloc((arr), sizeof(T) * (cap)); } } while (0)

int oracle_scene_add_material(oscene *s, const float diffuse[3], const float specular[3], float shininess) {
  GROW(s->mats, s->n_mats, s->cap_mats, omat);
  omat *m = &s->mats[s->n_mats]; m->raw_blinn = 0;
  m->diffuse = ld3(diffuse); m->specular = ld3(specular); m->shininess = shininess;
  return s->n_mats++;
}
int oracle_scene_add_material_raw_blinn(oscene *s, float e) {
  GROW(s->mats, s->n_mats, s->cap_mats, omat);
  omat *m = &s->mats[s->n_mats]; memset(m, 0, sizeof *m); m->raw_blinn = 1; m->shininess = e;
  return s->n_mats++;
}
void oracle_scene_add_light(oscene *s, const float pos[3], const float spectrum[3]) {
  GROW(s->lights, s->n_lights, s->cap_lights, olight);
  s->lights[s->n_lights].pos = ld3(pos); s->lights[s->n_lights].spectrum = ld3(spectrum);
  s->n_lights++;
}
int oracle_scene_add_sphere(oscene *s, const float c[3], float r, int mat) {
  GROW(s->prims, s->n_prims, s->cap_prims, oprim);
  oprim *p = &s->prims[s->n_prims]; memset(p, 0, sizeof *p);
  p->kind = 1; p->p0 = ld3(c); p->r = r; p->mat = mat;
  return s->n_prims++;
}
static int add_tri(oscene *s, ovec3 p0, ovec3 p1, ovec3 p2, ovec3 n0, ovec3 n1, ovec3 n2, int mat) {
  GROW(s->prims, s->n_prims, s->cap_prims, oprim);
  oprim *p = &s->prims[s->n_prims]; memset(p, 0, sizeof *p);
  p->kind = 0; p->p0 = p0; p->p1 = p1; p->p2 = p2; p->n0 = n0; p->n1 = n1; p->n2 = n2; p->mat = mat;
  return s->n_prims++;
}
int oracle_scene_add_triangle(oscene *s, const float p0[3], const float p1[3], const float p2[3],
                              const float n0[3], const float n1[3], const float n2[3], int mat) {
  return add_tri(s, ld3(p0), ld3(p1), ld3(p2), ld3(n0), ld3(n1), ld3(n2), mat);
}
/* expand (TriangleMesh ...)  Scene.hs:62-79 */
int oracle_scene_add_mesh(oscene *s, int n_pts, const float *pts, const float *normals,
                          int n_tris, const int32_t *tris, const uint8_t *smooth, int n_smooth, int mat) {
  (void)n_pts;
  int n = n_tris;
  if (smooth && n_smooth < n) n = n_smooth;                 /* zipWith truncates; Nothing = repeat False */
  for (int k = 0; k < n; ++k) {
    int i0 = tris[k * 3], i1 = tris[k * 3 + 1], i2 = tris[k * 3 + 2];
    ovec3 p0 = ld3(pts + 3 * i0), p1 = ld3(pts + 3 * i1), p2 = ld3(pts + 3 * i2);
    ovec3 nf = norm(cross(vsub(p2, p0), vsub(p1, p0)));     /* :78 */
    int sm = smooth ? smooth[k] : 0;
    if (normals && sm) add_tri(s, p0, p1, p2, ld3(normals + 3 * i0), ld3(normals + 3 * i1), ld3(normals + 3 * i2), mat);
    else add_tri(s, p0, p1, p2, nf, nf, nf, mat);
  }
  return n;
}
/* expand (Subsampled ...)  Scene.hs:80-85 */
int oracle_subsample_pick(int n, float size, int32_t *out_idx) {
  long of_hundred = (long)ceilf(size * 100.0f);
  int cnt = 0; long pos = 0;
  while (pos < n) {                                         /* pick xs */
    long take = of_hundred < 0 ? 0 : of_hundred;            /* splitAt n<=0 = ([], xs) */
    for (long i = 0; i < take && pos + i < n; ++i) out_idx[cnt++] = (int32_t)(pos + i);
    pos += take; if (pos > n) pos = n;
    long drop = 100 - of_hundred; if (drop < 0) drop = 0;   /* drop n<=0 = id */
    if (take == 0 && drop == 0) break;                      /* unreachable: take==0 => drop>=100 */
    pos += drop;
  }
  return cnt;
}
int oracle_scene_n_prims(const oscene *s) { return s->n_prims; }
void oracle_scene_get_prim(const oscene *s, int i, float out[19], int32_t *mat) {
  const oprim *p = &s->prims[i];
  memset(out, 0, sizeof(float) * 19);
  out[0] = (float)p->kind;
  if (p->kind == 1) { st3(out + 1, p->p0); out[4] = p->r; }
  else { st3(out + 1, p->p0); st3(out + 4, p->p1); st3(out + 7, p->p2); st3(out + 10, p->n0); st3(out + 13, p->n1); st3(out + 16, p->n2); }
  if (mat) *mat = p->mat;
}

/* cull (BVH d mode) (buildCollisionModel scene)   main.hs:41-60,118; Culling.hs:21-29 */
void oracle_scene_build(oscene *s) {
  node_free(s->root); s->root = NULL;
  if (s->n_prims > 0) {
    oitem *xs = (oitem *)malloc(sizeof(oitem) * s->n_prims);
    for (int i = 0; i < s->n_prims; ++i) {
      const oprim *p = &s->prims[i];
      xs[i].prim = i;
      xs[i].bb = (p->kind == 1) ? bound_sphere(p->r, p->p0) : bound_triangle(p->p0, p->p1, p->p2);
    }
    s->root = build_tree(s->bvh_depth, s->split, xs, s->n_prims);
    free(xs);
  }
  s->built = 1;
}

void oracle_scene_size(const oscene *s, int32_t *w, int32_t *h) {
  *w = (int32_t)floorf(s->imW); *h = (int32_t)floorf(s->imH);                /* main.hs:122-123 */
}

void oracle_render(oscene *s, int spp, uint32_t seed, int n_threads, float *rgb, ostats *stats) {
  if (!s->built) oracle_scene_build(s);
  ojob job; memset(&job, 0, sizeof job);
  oracle_scene_size(s, &job.w, &job.h);
  job.s = s; job.spp = spp; job.seed = seed; job.rgb = rgb; job.use_rows = 0;
  job.n_batches = oracle_n_batches(n_threads, job.w, job.h);
  run_job(&job, n_threads, stats);
}
void oracle_render_rows(oscene *s, int row0, int row1, int spp, uint32_t seed, int n_threads,
                        float *rgb_rows, ostats *stats) {
  if (!s->built) oracle_scene_build(s);
  ojob job; memset(&job, 0, sizeof job);
  oracle_scene_size(s, &job.w, &job.h);
  job.s = s; job.spp = spp; job.seed = seed; job.rgb = rgb_rows; job.use_rows = 1;
  job.row0 = row0; job.row1 = row1;
  run_job(&job, n_threads, stats);
}

void oracle_cross(const float a[3], const float b[3], float out[3]) { st3(out, cross(ld3(a), ld3(b))); }
void oracle_norm(const float a[3], float out[3]) { st3(out, norm(ld3(a))); }
void oracle_camera_ray(const oscene *s, float u, float v, float x0[3], float dir[3], float *tmax) {
  oray r = camera_ray(s, u, v); st3(x0, r.x0); st3(dir, r.u); *tmax = r.tmax;
}
void oracle_camera_matrices(const oscene *s, float tf[16], float vtf[16]) {
  memcpy(tf, s->tf, sizeof s->tf); memcpy(vtf, s->vtf, sizeof s->vtf);
}
void oracle_bound_sphere(float r, const float c[3], float bmin[3], float bmax[3]) {
  obox b = bound_sphere(r, ld3(c)); st3(bmin, b.bmin); st3(bmax, b.bmax);
}
void oracle_bound_triangle(const float p0[3], const float p1[3], const float p2[3], float bmin[3], float bmax[3]) {
  obox b = bound_triangle(ld3(p0), ld3(p1), ld3(p2)); st3(bmin, b.bmin); st3(bmax, b.bmax);
}
static void put_hit(const ohit *h, float *t, float point[3], float normal[3], float dpdu[3], float dpdv[3]) {
  if (t) *t = h->t;
  if (point) st3(point, h->dg.point);
  if (normal) st3(normal, h->dg.normal);
  if (dpdu) st3(dpdu, h->dg.dpdu);
  if (dpdv) st3(dpdv, h->dg.dpdv);
}
int oracle_collide_triangle(const float p0[3], const float p1[3], const float p2[3],
                            const float n0[3], const float n1[3], const float n2[3],
                            const float x0[3], const float u[3], float tmax,
                            float *t, float point[3], float normal[3], float dpdu[3], float dpdv[3]) {
  oray r; r.x0 = ld3(x0); r.u = ld3(u); r.tmax = tmax; ohit h;
  if (!collide_triangle(0, ld3(p0), ld3(p1), ld3(p2), ld3(n0), ld3(n1), ld3(n2), r, &h)) return 0;
  put_hit(&h, t, point, normal, dpdu, dpdv); return 1;
}
int oracle_collide_sphere(float rad, const float c[3], const float x0[3], const float u[3], float tmax,
                          float *t, float point[3], float normal[3], float dpdu[3], float dpdv[3]) {
  oray r; r.x0 = ld3(x0); r.u = ld3(u); r.tmax = tmax; ohit h;
  if (!collide_sphere(0, rad, ld3(c), r, &h)) return 0;
  put_hit(&h, t, point, normal, dpdu, dpdv); return 1;
}
void oracle_box_from_points(const float a[3], const float b[3], float bmin[3], float bmax[3]) {
  obox r = box_from_points(ld3(a), ld3(b)); st3(bmin, r.bmin); st3(bmax, r.bmax);
}
void oracle_box_include(float bmin[3], float bmax[3], const float p[3]) {
  obox b; b.bmin = ld3(bmin); b.bmax = ld3(bmax); b = box_include(b, ld3(p)); st3(bmin, b.bmin); st3(bmax, b.bmax);
}
void oracle_box_join(const float amin[3], const float amax[3], const float bmin[3], const float bmax[3],
                     float omin[3], float omax[3]) {
  obox a, b; a.bmin = ld3(amin); a.bmax = ld3(amax); b.bmin = ld3(bmin); b.bmax = ld3(bmax);
  obox r = box_join(a, b); st3(omin, r.bmin); st3(omax, r.bmax);
}
void oracle_box_centroid(const float bmin[3], const float bmax[3], float out[3]) {
  obox b; b.bmin = ld3(bmin); b.bmax = ld3(bmax); st3(out, box_centroid(b));
}
float oracle_box_surf(const float bmin[3], const float bmax[3]) {
  obox b; b.bmin = ld3(bmin); b.bmax = ld3(bmax); return box_surf(b);
}
int oracle_bb_ray(const float bmin[3], const float bmax[3], const float x0[3], const float u[3],
                  float tmax, float *tnear) {
  obox b; b.bmin = ld3(bmin); b.bmax = ld3(bmax);
  oray r; r.x0 = ld3(x0); r.u = ld3(u); r.tmax = tmax;
  return bb_ray(b, r, tnear);
}
/* collideAll (Rays.hs:39-46) over constant colliders, as Spec.hs:49-60 builds them */
int oracle_collide_all_const(int n, const uint8_t *is_hit, const float *ts, float tmax) {
  int what = -1; float cur = tmax; (void)cur;
  for (int i = 0; i < n; ++i) {
    if (is_hit[i]) { what = i; cur = ts[i]; }   /* Just h -> (Just h, cutRay (tHit h) ray); const colliders ignore the ray */
  }
  return what;
}
int oracle_scene_intersect(oscene *s, const float x0[3], const float u[3], float tmax, float *t) {
  if (!s->built) oracle_scene_build(s);
  oray r; r.x0 = ld3(x0); r.u = ld3(u); r.tmax = tmax; ohit h; octx cx; memset(&cx, 0, sizeof cx);
  if (!root_collide(s, r, &h, &cx)) return -1;
  if (t) *t = h.t;
  return h.what;
}
int oracle_scene_reachable(oscene *s, const float p0[3], const float p1[3]) {
  if (!s->built) oracle_scene_build(s);
  octx cx; memset(&cx, 0, sizeof cx);
  return reachable(s, ld3(p0), ld3(p1), &cx);
}
/* `**` on Float (BSDF.hs:25) is the C library's powf: this is the very call the restatement above makes, on arrays -- the
 * tests hold the device's restatement of glibc's routine (yahr_amd/csrc/device_math.hpp: glibc_powf) against it. */
void oracle_powf_array(int n, const float *x, const float *y, float *out) {
  for (int i = 0; i < n; ++i) out[i] = powf(x[i], y[i]);
}

void oracle_radiance(oscene *s, const float x0[3], const float u[3], float tmax, float out[3]) {
  if (!s->built) oracle_scene_build(s);
  oray r; r.x0 = ld3(x0); r.u = ld3(u); r.tmax = tmax; octx cx; memset(&cx, 0, sizeof cx);
  st3(out, vcast(s, s->depth, r, &cx, 1));
}
void oracle_bsdf_at(int kind, float exponent, const float normal[3], const float dpdu[3],
                    const float wi[3], const float wo[3], float out[3]) {
  odg dg; memset(&dg, 0, sizeof dg); dg.normal = ld3(normal); dg.dpdu = ld3(dpdu);
  oframe f = bsdf_frame(&dg);
  ovec3 i = world_to_loc(&f, ld3(wi)), o = world_to_loc(&f, ld3(wo));
  st3(out, kind == OBSDF_LAMBERTIAN ? loc_lambertian(i, o) : loc_blinn(exponent, i, o));
}
void oracle_blinn_phong_at(const float diffuse[3], const float specular[3], float shininess,
                           const float normal[3], const float dpdu[3],
                           const float wi[3], const float wo[3], float out[3]) {
  omat m; m.raw_blinn = 0; m.diffuse = ld3(diffuse); m.specular = ld3(specular); m.shininess = shininess;
  odg dg; memset(&dg, 0, sizeof dg); dg.normal = ld3(normal); dg.dpdu = ld3(dpdu);
  st3(out, mat_bsdf_at(&m, &dg, ld3(wi), ld3(wo)));
}

typedef struct { int32_t *kind; float *boxes; int32_t *first, *count, *prim_out; int n, np; } odump;
static void dump_rec(const onode *nd, odump *d) {
  int i = d->n++;
  if (d->kind) {
    d->kind[i] = nd->kind;
    st3(d->boxes + 6 * i, nd->bb.bmin); st3(d->boxes + 6 * i + 3, nd->bb.bmax);
    d->first[i] = (nd->kind == N_SINGLE || nd->kind == N_MULTI) ? d->np : -1;
    d->count[i] = (nd->kind == N_SINGLE || nd->kind == N_MULTI) ? nd->nprims : 0;
  }
  if (nd->kind == N_SINGLE || nd->kind == N_MULTI) {
    for (int k = 0; k < nd->nprims; ++k) { if (d->prim_out) d->prim_out[d->np] = nd->prims[k]; d->np++; }
  } else if (nd->kind == N_INTERIOR) {
    dump_rec(nd->l, d); dump_rec(nd->r, d);
  }
}
int oracle_tree_dump(const oscene *s, int32_t *kind, float *boxes, int32_t *first, int32_t *count,
                     int32_t *prim_out, int32_t *n_prim_out) {
  odump d; d.kind = kind; d.boxes = boxes; d.first = first; d.count = count; d.prim_out = prim_out; d.n = 0; d.np = 0;
  if (s->root) dump_rec(s->root, &d);
  if (n_prim_out) *n_prim_out = d.np;
  return d.n;
}
