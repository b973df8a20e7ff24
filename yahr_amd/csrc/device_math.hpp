// device_math.hpp -- gfx950 device functions of the yahr hot path (binary32, operation order literal to
// the reference so that results are bit-identical to a non-fused CPU evaluation; the only inexact
// piece is powf in the Blinn term).  Compiled with -ffp-contract=off and without fast-math:
// HIP's f32 divide and sqrt are correctly rounded by default.
#pragma once

#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace yd {

struct V3 { float x, y, z; };

#define YD __device__ __forceinline__

// ---- Vectors.hs -------------------------------------------------------------------------------
YD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
YD V3 vof(float l) { return v3(l, l, l); }                                           // :20-21
YD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }          // :24
YD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }          // :25
YD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }          // :26 component-wise
YD V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }                                    // :27
YD float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }             // (.*) :33, + infixl 6
YD float lensq(V3 v) { return dot(v, v); }                                           // :36-37
YD float len(V3 v) { return sqrtf(lensq(v)); }                                       // :39-40
YD V3 scale(float f, V3 v) { return v3(f * v.x, f * v.y, f * v.z); }                 // (@*) :42-43
YD V3 norm(V3 v) { return scale(1.0f / len(v), v); }                                 // :46-47 reciprocal, 3 muls
YD V3 cross(V3 a, V3 b) {                                                            // :49-53
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Haskell's class-default Ord min/max: NaN-asymmetric (not fminf/fmaxf)
YD float hmax(float x, float y) { return (x <= y) ? y : x; }
YD float hmin(float x, float y) { return (x <= y) ? x : y; }

YD V3 xyz(float4 v) { return v3(v.x, v.y, v.z); }

struct Ray { V3 o, d; float tmax; };                                                  // Rays.hs:11

// ---- AABBs.hs:29-40 bbRayIntersection ---------------------------------------------------------
// inv = 1 / u per axis (the reference recomputes it per box; the value is the same)
YD bool slab_hit(V3 lo, V3 hi, V3 o, V3 inv, float tmax) {
  float tnear = 0.0f, tfar = tmax;
  {
    float t0 = (lo.x - o.x) * inv.x, t1 = (hi.x - o.x) * inv.x;
    tnear = hmax(tnear, hmin(t0, t1)); tfar = hmin(tfar, hmax(t0, t1));
  }
  {
    float t0 = (lo.y - o.y) * inv.y, t1 = (hi.y - o.y) * inv.y;
    tnear = hmax(tnear, hmin(t0, t1)); tfar = hmin(tfar, hmax(t0, t1));
  }
  {
    float t0 = (lo.z - o.z) * inv.z, t1 = (hi.z - o.z) * inv.z;
    tnear = hmax(tnear, hmin(t0, t1)); tfar = hmin(tfar, hmax(t0, t1));
  }
  return !(tnear > tfar);
}

// The same predicate with hardware min/max (v_min_f32 / v_max_f32 / v_min3 / v_max3), valid when NO t0/t1
// can be NaN.  Then hmin/hmax equal the mathematical min/max (up to the sign of a zero, which no comparison
// sees), so the Boolean is identical to slab_hit's.  A NaN needs 0*inf or inf-inf: impossible when the box is
// finite, the origin is finite and every reciprocal direction component is finite and non-zero -- the caller
// checks exactly that (slab_safe) for every active lane before choosing this form.  A NaN tMax is harmless:
// hmin(tMax, x) = x = v_min(tMax, x).
YD bool slab_hit_fast(V3 lo, V3 hi, V3 o, V3 inv, float tmax) {
  const float t0x = (lo.x - o.x) * inv.x, t1x = (hi.x - o.x) * inv.x;
  const float t0y = (lo.y - o.y) * inv.y, t1y = (hi.y - o.y) * inv.y;
  const float t0z = (lo.z - o.z) * inv.z, t1z = (hi.z - o.z) * inv.z;
  const float tnear = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
  const float tfar = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
  return !(tnear > tfar);
}
// The same test as a 64-bit lane mask (v_cmp straight into an SGPR pair / vcc), for the packet walker: the mask is
// both the wave's "any lane hit" (mask != 0: the vccz branch of the compare itself) and, through inverse_ballot, the
// lane's own Boolean -- without the v_cndmask 0/1 + v_cmp_ne pair a ballot of a bool costs (2 of 30 VALU per node).
YD unsigned long long slab_mask_fast(V3 lo, V3 hi, V3 o, V3 inv, float tmax) {
  const float t0x = (lo.x - o.x) * inv.x, t1x = (hi.x - o.x) * inv.x;
  const float t0y = (lo.y - o.y) * inv.y, t1y = (hi.y - o.y) * inv.y;
  const float t0z = (lo.z - o.z) * inv.z, t1z = (hi.z - o.z) * inv.z;
  const float tnear = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
  const float tfar = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
  return __builtin_amdgcn_fcmpf(tnear, tfar, 13);                       // FCMP_ULE: !(tnear > tfar)
}
// slab_hit_fast that also hands out the entry distance (near-first ordering of the opt-in ordered walker)
YD bool slab_near_fast(V3 lo, V3 hi, V3 o, V3 inv, float tmax, float &tnear) {
  const float t0x = (lo.x - o.x) * inv.x, t1x = (hi.x - o.x) * inv.x;
  const float t0y = (lo.y - o.y) * inv.y, t1y = (hi.y - o.y) * inv.y;
  const float t0z = (lo.z - o.z) * inv.z, t1z = (hi.z - o.z) * inv.z;
  tnear = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
  const float tfar = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
  return !(tnear > tfar);
}
// AABBs.join (AABBs.hs:24-27) of two finite boxes: component-wise min / max -- no rounding, so a parent's box is
// recovered EXACTLY from its children's (Culling.hs:39 builds it as that join); the sign of a zero may differ, which
// no comparison of the slab test sees
YD V3 vmin(V3 a, V3 b) { return v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
YD V3 vmax(V3 a, V3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
YD bool finite_nonzero(float x) { const float a = fabsf(x); return (a > 0.0f) & (a < __builtin_huge_valf()); }
YD bool finite_f(float x) { return fabsf(x) < __builtin_huge_valf(); }
YD bool slab_safe(V3 o, V3 inv) {
  return finite_f(o.x) & finite_f(o.y) & finite_f(o.z) &
         finite_nonzero(inv.x) & finite_nonzero(inv.y) & finite_nonzero(inv.z);
}

// ---- Shapes.hs:36-59 collideTriangle, geometric part ---------------------------------------------
// Returns the conjunction of every acceptance test that does not need the normals; the caller then
// applies `u .* ns < 0` (Shapes.hs:57).  Short-circuit order does not matter: the tests are pure.
YD bool tri_candidate(V3 p0, V3 e1, V3 e2, V3 o, V3 u, float tmax, float &t, float &b1, float &b2) {
  V3 s = o - p0;
  V3 s1 = cross(u, e2);
  V3 s2 = cross(s, e1);
  float inv = 1.0f / dot(s1, e1);
  t = dot(s2, e2) * inv;
  b1 = dot(s1, s) * inv;
  b2 = dot(s2, u) * inv;
  float b0 = (1.0f - b1) - b2;
  return (b0 >= 0.0f) & (b0 <= 1.0f) & (b1 >= 0.0f) & (b1 <= 1.0f) & (b2 >= 0.0f) & (b2 <= 1.0f) &
         (t > 0.0f) & (t <= tmax);
}
// the same without the `t <= tMax` term (the ordered walker applies its own tie rule to it)
YD bool tri_candidate_open(V3 p0, V3 e1, V3 e2, V3 o, V3 u, float &t, float &b1, float &b2) {
  V3 s = o - p0;
  V3 s1 = cross(u, e2);
  V3 s2 = cross(s, e1);
  float inv = 1.0f / dot(s1, e1);
  t = dot(s2, e2) * inv;
  b1 = dot(s1, s) * inv;
  b2 = dot(s2, u) * inv;
  float b0 = (1.0f - b1) - b2;
  return (b0 >= 0.0f) & (b0 <= 1.0f) & (b1 >= 0.0f) & (b1 <= 1.0f) & (b2 >= 0.0f) & (b2 <= 1.0f) & (t > 0.0f);
}
// The same test cut in two for the packet walker (identical operations and values, only regrouped): stage 1
// needs s, s1, the determinant and its reciprocal and yields b1 with its range test; when no lane of the
// wavefront passes it the caller skips stage 2 (s2, t, b2, b0 and their tests) with a wave-uniform branch.
//
// Stage 0 stops before the division: det = s1.e1 and q = s1.s.  `tri_far(q, det)` -- |q| > 1.5 |det| -- PROVES
// that b1 = fl(q * fl(1/det)) fails its range test, whatever the operands:
//   * det = +-0: 1/det = +-inf and q != 0 (|q| > 0), so b1 = +-inf: not in [0,1];
//   * det, q finite, det != 0: fl(1/det) is correctly rounded (relative error <= 2^-24, <= 2^-20 when 1/det is
//     subnormal; +-inf when det is subnormal and 1/det overflows), so |q * fl(1/det)| >= 1.3 even after the
//     subnormal rounding of 1.5|det| (worst case 5/3), and one more rounding leaves |b1| > 1.29: b1 > 1 or b1 < 0;
//   * q = +-inf: b1 = +-inf;  q or det NaN, or det = +-inf: the comparison is false, nothing is claimed.
// So when EVERY live lane of the wavefront is "far", the ten-instruction IEEE divide and the rest of stage 1 can
// be skipped without changing any lane's result; otherwise stage 1 runs in full for all lanes, as before.
YD void tri_stage0(V3 p0, V3 e1, V3 e2, V3 o, V3 u, V3 &s, float &det, float &q) {
  s = o - p0;
  const V3 s1 = cross(u, e2);
  det = dot(s1, e1);
  q = dot(s1, s);
}
YD bool tri_far(float q, float det) { return fabsf(q) > 1.5f * fabsf(det); }

// tri_far for a whole PACKET of rays against one triangle, decided from the packet's bounding box in ray space:
// every ray r of the packet has ol <= o_r <= oh and ul <= u_r <= uh component-wise (all finite).  The function
// evaluates tri_stage0's expression tree -- the same binary32 operations in the same order -- in interval
// arithmetic with ROUND-TO-NEAREST end points and returns true only if then |q_r| > 1.5 |det_r| holds for the
// q_r, det_r that tri_stage0 COMPUTES for every such ray, i.e. every lane of the packet would report tri_far.
// Why the enclosure holds for the computed (rounded) values, not just the real ones: rounding fl() is monotone
// non-decreasing, and x - y, x + y (monotone in each argument) and x * y (bilinear: extrema over a box at its
// corners) take their real-number extrema at corners of the operand box; so for x in [xl, xh], y in [yl, yh]
//     fl(x - y) in [fl(xl - yh), fl(xh - yl)],    fl(x + y) in [fl(xl + yl), fl(xh + yh)],
//     fl(x * y) in [min fl(corner products), max fl(corner products)]
// and by induction over the expression tree every intermediate a lane computes lies in the interval computed
// here.  Same argument as slab_hit_fast's, and like it only valid while no operation meets an infinity or a NaN
// (inf - inf, 0 * inf; v_min / v_max would silently drop the NaN): the CALLER guarantees
//     |o| <= 2^32, |u| <= 2^20 (checked on the packet's bounds), |p0|, |e1|, |e2| <= 2^32 (checked per scene on
//     the host: DScene.boxes_finite bit 1)   =>   |s| <= 2^33, |s1| <= 2^53, |det|, |q| <= 2^88, 1.5 |det| <= 2^89,
// far below binary32's 2^128, so nothing overflows and nothing is a NaN.  Subnormal results round (or flush) the
// same way here and in tri_stage0 -- one kernel, one float mode -- and both are monotone.
// Final step: fl(1.5 |det_r|) <= fl(1.5 max|det|) by monotonicity, min|q| <= |q_r|, so
//     min|q| > fl(1.5 max|det|)   =>   tri_far(q_r, det_r) for every ray of the packet.
// The packet walker then skips the primitive step for that triangle: by tri_far's own proof (above) no lane can
// pass the b1 range test, so nothing a lane computes or stores changes -- only the skipped work.
YD void ival_mul_point(float xl, float xh, float y, float &lo, float &hi) {
  const float a = xl * y, b = xh * y;
  lo = fminf(a, b); hi = fmaxf(a, b);
}
YD void ival_mul(float xl, float xh, float yl, float yh, float &lo, float &hi) {
  const float a = xl * yl, b = xl * yh, c = xh * yl, d = xh * yh;
  lo = fminf(fminf(a, b), fminf(c, d)); hi = fmaxf(fmaxf(a, b), fmaxf(c, d));
}
YD bool tri_far_packet(V3 p0, V3 e1, V3 e2, V3 ol, V3 oh, V3 ul, V3 uh) {
  const V3 sl = ol - p0, sh = oh - p0;                                 // s = o - p0
  // s1 = cross(u, e2) = (u.y e2.z - u.z e2.y, u.z e2.x - u.x e2.z, u.x e2.y - u.y e2.x)
  float al, ah, bl, bh;
  ival_mul_point(ul.y, uh.y, e2.z, al, ah); ival_mul_point(ul.z, uh.z, e2.y, bl, bh);
  const float s1xl = al - bh, s1xh = ah - bl;
  ival_mul_point(ul.z, uh.z, e2.x, al, ah); ival_mul_point(ul.x, uh.x, e2.z, bl, bh);
  const float s1yl = al - bh, s1yh = ah - bl;
  ival_mul_point(ul.x, uh.x, e2.y, al, ah); ival_mul_point(ul.y, uh.y, e2.x, bl, bh);
  const float s1zl = al - bh, s1zh = ah - bl;
  // det = (s1.x e1.x + s1.y e1.y) + s1.z e1.z
  float xl, xh, yl, yh, zl, zh;
  ival_mul_point(s1xl, s1xh, e1.x, xl, xh); ival_mul_point(s1yl, s1yh, e1.y, yl, yh); ival_mul_point(s1zl, s1zh, e1.z, zl, zh);
  const float dl = (xl + yl) + zl, dh = (xh + yh) + zh;
  // q = (s1.x s.x + s1.y s.y) + s1.z s.z
  ival_mul(s1xl, s1xh, sl.x, sh.x, xl, xh); ival_mul(s1yl, s1yh, sl.y, sh.y, yl, yh); ival_mul(s1zl, s1zh, sl.z, sh.z, zl, zh);
  const float ql = (xl + yl) + zl, qh = (xh + yh) + zh;
  const float min_abs_q = fmaxf(fmaxf(ql, -qh), 0.0f);                 // 0 when the interval holds 0
  const float max_abs_det = fmaxf(fabsf(dl), fabsf(dh));
  return min_abs_q > 1.5f * max_abs_det;
}
// magnitude limits under which tri_far_packet cannot overflow (see above)
#define YD_CULL_MAX_COORD 4294967296.0f      /* 2^32: ray origins, triangle vertices and edges */
#define YD_CULL_MAX_DIR 1048576.0f           /* 2^20: ray directions */
YD bool tri_stage1(float det, float q, float &inv, float &b1) {
  inv = 1.0f / det;
  b1 = q * inv;
  // bitwise & on purpose (here and below): the comparisons are pure, and short-circuit && makes the compiler
  // build exec-mask branches around every term
  return (b1 >= 0.0f) & (b1 <= 1.0f);
}
YD bool tri_stage2(V3 e1, V3 e2, V3 s, V3 u, float inv, float b1, float tmax, float &t, float &b2) {
  const V3 s2 = cross(s, e1);
  t = dot(s2, e2) * inv;
  b2 = dot(s2, u) * inv;
  const float b0 = (1.0f - b1) - b2;
  return (b0 >= 0.0f) & (b0 <= 1.0f) & (b2 >= 0.0f) & (b2 <= 1.0f) & (t > 0.0f) & (t <= tmax);
}
YD V3 tri_shading_normal(float b1, float b2, V3 n0, V3 n1, V3 n2) {                  // Shapes.hs:47-48
  float b0 = (1.0f - b1) - b2;
  return (scale(b0, n0) + scale(b1, n1)) + scale(b2, n2);
}

// ---- Shapes.hs:13-29 collideSphere -----------------------------------------------------------------
YD bool sphere_hit(V3 c, float r, V3 o, V3 u, float tmax, float &t) {
  V3 d = o - c;
  float a = dot(u, u);
  float b = dot(vof(2.0f) * d, u);
  float cc = dot(d, d) - r * r;
  float delta = b * b - (4.0f * a) * cc;
  if (delta < 0.0f) return false;
  float sq = sqrtf(delta);
  float t0 = ((-b) - sq) / (2.0f * a);
  float t1 = ((-b) + sq) / (2.0f * a);
  if (t0 > 0.0f && t0 <= tmax) { t = t0; return true; }
  if (t1 > 0.0f && t1 <= tmax) { t = t1; return true; }
  return false;
}

// Acceptance of a hit distance when primitives are NOT visited in the reference's order (ordered walker).  In the
// reference's fixed order `t <= tMax` makes the LATER primitive win an exact tie (Rays.hs:39-46); visited in any other
// order the same winner needs: strictly nearer, or equally near and later in that order (`idx` = leaf-order record
// index = position in the reference's order; best = -1 while tMax is still the ray's own limit).
YD bool accept_ordered(float t, float tmax, int idx, int best) { return (t < tmax) | ((t == tmax) & (idx > best)); }
YD bool sphere_hit_ordered(V3 c, float r, V3 o, V3 u, float tmax, int idx, int best, float &t) {
  V3 d = o - c;
  float a = dot(u, u);
  float b = dot(vof(2.0f) * d, u);
  float cc = dot(d, d) - r * r;
  float delta = b * b - (4.0f * a) * cc;
  if (delta < 0.0f) return false;
  float sq = sqrtf(delta);
  float t0 = ((-b) - sq) / (2.0f * a);
  float t1 = ((-b) + sq) / (2.0f * a);
  // the reference takes the first root inside (0, tMax]; a root beyond the current best distance is "outside" there
  // too, so the same root is chosen and then put to the tie rule
  if (t0 > 0.0f && t0 <= tmax) { t = t0; return accept_ordered(t0, tmax, idx, best); }
  if (t1 > 0.0f && t1 <= tmax) { t = t1; return accept_ordered(t1, tmax, idx, best); }
  return false;
}

// ---- BSDF.hs:12-46 + Shaders.hs:12-14 --------------------------------------------------------------
#define YD_PI 3.14159274101257324f   /* pi :: Float */

YD V3 loc_lambertian(V3 i, V3 o) {                                                   // BSDF.hs:15-17
  return (i.z > 0.0f && o.z > 0.0f) ? vof(1.0f / YD_PI) : vof(0.0f);
}
YD V3 loc_blinn(float e, V3 i, V3 o) {                                               // BSDF.hs:18-30
  if (!(i.z > 0.0f && o.z > 0.0f)) return vof(0.0f);
  V3 h = norm(i + o);
  float cos_o = fabsf(o.z), cos_i = fabsf(i.z);
  float cos_h = dot(i, h);
  float o_dot_h = dot(o, h);
  float d = ((e + 2.0f) * powf(fabsf(h.z), e)) / (2.0f * YD_PI);
  float g = hmin(1.0f, hmin(((2.0f * cos_h) * cos_o) / o_dot_h, ((2.0f * cos_h) * cos_i) / o_dot_h));
  float f = 1.0f;                                                                    // fresnel, BSDF.hs:34
  return vof(((d * g) * f) / ((4.0f * cos_i) * cos_o));
}

struct Frame { V3 nn, sn, tn; };                                                      // BSDF.hs:36-38
YD Frame bsdf_frame(V3 normal, V3 dpdu) {
  Frame f; f.nn = normal; f.sn = norm(dpdu); f.tn = cross(f.nn, f.sn); return f;
}
YD V3 world_to_loc(const Frame &f, V3 v) { return v3(dot(v, f.sn), dot(v, f.tn), dot(v, f.nn)); }  // :40

// Composite [Scaled diffuse Lambertian, Scaled specular (Blinn shininess)]: sum from 0 in list order
YD V3 blinn_phong_at(V3 diffuse, V3 specular, float shininess, const Frame &f, V3 wi, V3 wo) {
  V3 i = world_to_loc(f, wi), o = world_to_loc(f, wo);
  V3 acc = vof(0.0f);
  acc = acc + diffuse * loc_lambertian(i, o);
  acc = acc + specular * loc_blinn(shininess, i, o);
  return acc;
}

// ---- Cameras.hs:40-44,83-86 -------------------------------------------------------------------------
YD V3 transform_point(const float *m, V3 p) {       // (tf !* V4 x y z 1), normalizePoint = (1/w) *^
  float tv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float acc = 0.0f;
    acc = acc + m[i * 4 + 0] * p.x;
    acc = acc + m[i * 4 + 1] * p.y;
    acc = acc + m[i * 4 + 2] * p.z;
    acc = acc + m[i * 4 + 3] * 1.0f;
    tv[i] = acc;
  }
  float iw = 1.0f / tv[3];
  return v3(iw * tv[0], iw * tv[1], iw * tv[2]);
}

// ---- sampler (build extension for spp > 1; identical integer hash in oracle/yahr_oracle.c) ----------
YD unsigned lowbias32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
YD void sample_offset(int u, int v, int k, int n_strata, unsigned seed, float &du, float &dv) {
  int i = (n_strata > 1) ? (k % n_strata) : 0, j = (n_strata > 1) ? ((k / n_strata) % n_strata) : 0;
  unsigned h = lowbias32((unsigned)u + lowbias32((unsigned)v + lowbias32((unsigned)k + lowbias32(seed))));
  unsigned h2 = lowbias32(h ^ 0x68bc21ebU);
  float xi1 = (float)(h >> 8) * (1.0f / 16777216.0f);
  float xi2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
  du = ((float)i + xi1) / (float)n_strata;
  dv = ((float)j + xi2) / (float)n_strata;
}

}  // namespace yd
