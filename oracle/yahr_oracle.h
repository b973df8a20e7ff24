/*
 * yahr_oracle.h -- CPU restatement of the yahr render hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle: a literal, operation-order-preserving restatement in plain C of the
 * Haskell reference (pstiasny/yahr).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (libyahr_hip.so) never links or calls it.
 *
 * Pinning: checked against every known-answer test of the reference's own test/Spec.hs
 * (oracle/kat.c + tests/test_oracle_kat.py + tests/golden/spec_kat.json).  The Haskell reference
 * cannot be executed in this environment (no GHC), so there are no image-level goldens.
 *
 * All arithmetic is IEEE binary32, compiled -O2 -ffp-contract=off -fno-fast-math.
 */
#ifndef YAHR_ORACLE_H
#define YAHR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } ovec3;                 /* Vectors.hs:5 */
typedef struct { ovec3 x0, u; float tmax; } oray;        /* Rays.hs:11 */
typedef struct { ovec3 bmin, bmax; } obox;               /* AABBs.hs:8 */
typedef struct { ovec3 point, normal, dpdu, dpdv; } odg; /* DifferentialGeometry.hs:5-10 */
typedef struct { float t; odg dg; int32_t what; } ohit;  /* Rays.hs:16; what = primitive index */

enum { OSPLIT_MIDPOINT = 0, OSPLIT_SAH = 1 };            /* Culling.hs:18 */
enum { OBSDF_LAMBERTIAN = 0, OBSDF_BLINN = 1 };          /* BSDF.hs:6-8 */

typedef struct {
  uint64_t primary_rays, reflection_rays, shadow_rays;
  uint64_t box_tests, tri_tests, sphere_tests, shaded_hits;
} ostats;

typedef struct oscene oscene;

/* ---- scene assembly (what main.hs:41-60 buildCollisionModel + Scene.hs:61-86 expand produce) */
oscene *oracle_scene_new(void);
void oracle_scene_free(oscene *s);
void oracle_scene_set_camera(oscene *s, float imW, float imH, float focal,
                             const float look[3], const float up[3], const float pos[3]);
void oracle_scene_set_integrator(oscene *s, int recursion_depth);
void oracle_scene_set_culling(oscene *s, int max_depth, int split_mode);
int oracle_scene_add_material(oscene *s, const float diffuse[3], const float specular[3], float shininess);
/* test-only material: the BSDF is a bare Blinn e (Spec.hs:265-297 uses `Blinn 10` directly) */
int oracle_scene_add_material_raw_blinn(oscene *s, float exponent);
void oracle_scene_add_light(oscene *s, const float pos[3], const float spectrum[3]);
int oracle_scene_add_sphere(oscene *s, const float c[3], float r, int mat);
int oracle_scene_add_triangle(oscene *s, const float p0[3], const float p1[3], const float p2[3],
                              const float n0[3], const float n1[3], const float n2[3], int mat);
/* Scene.hs:62-79: expand a TriangleMesh; normals may be NULL (Nothing); smooth may be NULL (Nothing);
 * n_smooth shorter than n_tris truncates the mesh (zipWith).  Returns number of triangles added. */
int oracle_scene_add_mesh(oscene *s, int n_pts, const float *pts, const float *normals,
                          int n_tris, const int32_t *tris, const uint8_t *smooth, int n_smooth, int mat);
/* Scene.hs:80-85: which indices of an n-element expanded list `Subsampled size` keeps. Returns count. */
int oracle_subsample_pick(int n, float size, int32_t *out_idx);
int oracle_scene_n_prims(const oscene *s);
/* read back primitive i as 19 floats: kind(0 tri,1 sphere), p0 p1 p2 n0 n1 n2 (or c, r) */
void oracle_scene_get_prim(const oscene *s, int i, float out[19], int32_t *mat);
void oracle_scene_build(oscene *s);                      /* Culling.hs:21-54, forced eagerly */
/* counters only: let shadow probes stop at the first accepted primitive (same Bool; see .c) */
void oracle_set_shadow_any_hit(int on);

/* ---- rendering (main.hs:65-107) */
void oracle_scene_size(const oscene *s, int32_t *w, int32_t *h);
/* rgb: w*h*3 floats, index (w*v+u)*3 (main.hs:100).  spp=1 is the reference's sampling (main.hs:73). */
void oracle_render(oscene *s, int spp, uint32_t seed, int n_threads, float *rgb, ostats *stats);
void oracle_render_rows(oscene *s, int row0, int row1, int spp, uint32_t seed, int n_threads,
                        float *rgb_rows, ostats *stats);
int oracle_n_batches(int num_threads, int width, int height);            /* main.hs:109-110,128-130 */
void oracle_batch_window(int w, int h, int num, int count, int32_t out[4]); /* Sampling.hs:9-21 */

/* ---- unit-level entry points for the known-answer tests */
void oracle_cross(const float a[3], const float b[3], float out[3]);      /* Vectors.hs:49-53 */
void oracle_norm(const float a[3], float out[3]);                         /* Vectors.hs:46-47 */
void oracle_camera_ray(const oscene *s, float u, float v, float x0[3], float dir[3], float *tmax);
void oracle_camera_matrices(const oscene *s, float tf[16], float vtf[16]);
void oracle_bound_sphere(float r, const float c[3], float bmin[3], float bmax[3]);
void oracle_bound_triangle(const float p0[3], const float p1[3], const float p2[3], float bmin[3], float bmax[3]);
int oracle_collide_triangle(const float p0[3], const float p1[3], const float p2[3],
                            const float n0[3], const float n1[3], const float n2[3],
                            const float x0[3], const float u[3], float tmax,
                            float *t, float point[3], float normal[3], float dpdu[3], float dpdv[3]);
int oracle_collide_sphere(float r, const float c[3], const float x0[3], const float u[3], float tmax,
                          float *t, float point[3], float normal[3], float dpdu[3], float dpdv[3]);
void oracle_box_from_points(const float a[3], const float b[3], float bmin[3], float bmax[3]);
void oracle_box_include(float bmin[3], float bmax[3], const float p[3]);
void oracle_box_join(const float amin[3], const float amax[3], const float bmin[3], const float bmax[3],
                     float omin[3], float omax[3]);
void oracle_box_centroid(const float bmin[3], const float bmax[3], float out[3]);
float oracle_box_surf(const float bmin[3], const float bmax[3]);
void oracle_powf_array(int n, const float *x, const float *y, float *out);   /* libm powf, elementwise */
int oracle_bb_ray(const float bmin[3], const float bmax[3], const float x0[3], const float u[3],
                  float tmax, float *tnear);                                /* AABBs.hs:29-40 */
/* Rays.hs:39-46 with constant colliders: is_hit[i] ? Just (Hit ts[i] _ i) : Nothing. Returns `what` or -1 */
int oracle_collide_all_const(int n, const uint8_t *is_hit, const float *ts, float tmax);
/* closest hit through the built BVH; returns primitive index or -1 */
int oracle_scene_intersect(oscene *s, const float x0[3], const float u[3], float tmax, float *t);
int oracle_scene_reachable(oscene *s, const float p0[3], const float p1[3]); /* Rays.hs:49-54 */
void oracle_radiance(oscene *s, const float x0[3], const float u[3], float tmax, float out[3]);
void oracle_bsdf_at(int kind, float exponent, const float normal[3], const float dpdu[3],
                    const float wi[3], const float wo[3], float out[3]);     /* BSDF.hs:12-46 */
void oracle_blinn_phong_at(const float diffuse[3], const float specular[3], float shininess,
                           const float normal[3], const float dpdu[3],
                           const float wi[3], const float wo[3], float out[3]); /* Shaders.hs:12-14 */
void oracle_sample_offset(int u, int v, int k, int spp, uint32_t seed, float *du, float *dv);

/* ---- tree dump in reference DFS (pre)order, to compare with the product's flattened BVH.
 * kind: 0 interior, 1 singleton leaf, 2 multi-leaf, 3 null.  For leaves prim ids are appended to
 * prim_out in stored order.  Returns number of nodes (call with NULLs to size). */
int oracle_tree_dump(const oscene *s, int32_t *kind, float *boxes /*n*6*/, int32_t *first, int32_t *count,
                     int32_t *prim_out, int32_t *n_prim_out);

#ifdef __cplusplus
}
#endif
#endif
