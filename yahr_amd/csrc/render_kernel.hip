// render_kernel.hip -- the render hot path of yahr as hand-written HIP for gfx950 (MI355X, wave64).
//
// One wavefront per pixel tile (main.hs:75-96: data parallelism over pixel tiles).  Per pixel:
//   Cameras.computeInitialRay -> Integrators.radiance (vcast/vhit chain) with
//   Culling/AABBs/Shapes traversal for eye, reflection and shadow rays, BSDF/Shaders/Lights shading.
//
// BVH traversal.  The reference's closure tree visits "box, left subtree, cut ray, right subtree"
// (Culling.hs:24-25,38; Rays.hs:39-46), i.e. a FIXED depth-first order.  The host stores the nodes in
// exactly that order with a skip link per interior node (include/yahr_hip.h), so traversal is a
// monotone walk over the node array with no stack at all:
//     hit interior -> i+1 ; missed interior -> skip ; leaf -> test primitives in stored order, i+1.
// Two walkers implement it, both visiting per ray exactly the reference's nodes in its order:
//   * lane walker:   every lane owns an index i and gathers its own 32-byte node (vector loads)
//   * packet walker: the wavefront owns ONE scalar index i; node and primitive records are fetched
//                    with scalar loads (wave-uniform address) into SGPRs, every lane tests its own ray,
//                    `v_cmp` + vcc!=0 decides "descend" for the wave, and a per-lane `wake` index
//                    (the skip link of the subtree the lane missed) masks lanes that are not inside
//                    the current subtree.  Preorder indices only grow, so lane is active iff wake<=i.
// Equal-t ties, t<=tMax acceptance, back-face culling and NaN behaviour therefore match the oracle
// bit for bit; shadow rays stop at the first accepted primitive (same Bool as Rays.hs:49-54).
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "render_kernel.hpp"

#pragma clang fp contract(off)

using namespace yd;

namespace {

struct Hit { int prim; float t, b1, b2; };   // prim = leaf-order record index, -1 = Nothing

struct Counters {
  unsigned primary, reflection, shadow, shaded, bad_mat;
  unsigned box, tri, sph, wave_nodes, wave_prims;
};

YD V3 ld3(const float4 *p) { float4 v = *p; return v3(v.x, v.y, v.z); }

// accept test for one primitive record against one ray.  Returns true and updates (hit, tmax) on accept.
template <bool STATS>
YD bool test_prim(const DScene &sc, int idx, float4 r0, float4 r1, float4 r2, V3 o, V3 u, float &tmax,
                  Hit &hit, Counters &c) {
  int kind = __float_as_int(r1.w);
  if (kind == 0) {
    if (STATS) c.tri++;
    float t, b1, b2;
    if (!tri_candidate(xyz(r0), xyz(r1), xyz(r2), o, u, tmax, t, b1, b2)) return false;
    const float4 *nr = sc.normals + 3 * (size_t)idx;
    V3 ns = tri_shading_normal(b1, b2, ld3(nr), ld3(nr + 1), ld3(nr + 2));
    if (!(dot(u, ns) < 0.0f)) return false;                        // back-face cull, Shapes.hs:57
    hit.prim = idx; hit.t = t; hit.b1 = b1; hit.b2 = b2; tmax = t;
    return true;
  } else {
    if (STATS) c.sph++;
    float t;
    if (!sphere_hit(xyz(r0), r1.x, o, u, tmax, t)) return false;
    hit.prim = idx; hit.t = t; hit.b1 = 0.0f; hit.b2 = 0.0f; tmax = t;
    return true;
  }
}

// ---- lane walker --------------------------------------------------------------------------------
template <bool ANY, bool STATS>
YD void traverse_lane(const DScene &sc, bool active, const Ray &ray, Hit &hit, Counters &c) {
  const V3 o = ray.o, u = ray.d;
  const V3 inv = v3(1.0f / u.x, 1.0f / u.y, 1.0f / u.z);
  float tmax = ray.tmax;
  hit.prim = -1; hit.t = 0.0f; hit.b1 = 0.0f; hit.b2 = 0.0f;
  const int n = sc.n_nodes;
  int i = active ? 0 : n;
  while (i < n) {
    const float4 lo = sc.nodes[2 * (size_t)i], hi = sc.nodes[2 * (size_t)i + 1];
    const int a = __float_as_int(lo.w), b = __float_as_int(hi.w);
    if (STATS) c.box++;
    const bool h = slab_hit(xyz(lo), xyz(hi), o, inv, tmax);
    int next = (b >= 0 || h) ? i + 1 : a;
    if (h && b >= 0) {
      for (int k = 0; k < a; ++k) {
        const float4 *pr = sc.prims + 3 * (size_t)(b + k);
        if (test_prim<STATS>(sc, b + k, pr[0], pr[1], pr[2], o, u, tmax, hit, c)) {
          if (ANY) { next = n; break; }
        }
      }
    }
    i = next;
  }
}

// ---- packet walker ------------------------------------------------------------------------------
#define YD_DONE 0x7fffffff

YD int wave_min_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { int w = __shfl_xor(v, off, 64); v = w < v ? w : v; }
  return v;
}

template <bool ANY, bool STATS>
YD void traverse_packet(const DScene &sc, bool active, const Ray &ray, Hit &hit, Counters &c) {
  const V3 o = ray.o, u = ray.d;
  const V3 inv = v3(1.0f / u.x, 1.0f / u.y, 1.0f / u.z);
  float tmax = ray.tmax;
  hit.prim = -1; hit.t = 0.0f; hit.b1 = 0.0f; hit.b2 = 0.0f;
  const int n = sc.n_nodes;
  int wake = active ? 0 : YD_DONE;
  if (!__any(active)) return;
  int i = 0;                                           // wave-uniform: lives in an SGPR
  while (i < n) {
    const float4 lo = sc.nodes[2 * (size_t)i], hi = sc.nodes[2 * (size_t)i + 1];   // scalar loads
    const int a = __float_as_int(lo.w), b = __float_as_int(hi.w);
    const bool on = wake <= i;
    if (STATS) { c.box += on ? 1u : 0u; c.wave_nodes++; }
    const bool h = on && slab_hit(xyz(lo), xyz(hi), o, inv, tmax);
    const bool anyh = __any(h);
    if (b >= 0) {                                      // leaf (uniform branch)
      if (anyh) {
        bool got = false;
        for (int k = 0; k < a; ++k) {
          const float4 *pr = sc.prims + 3 * (size_t)(b + k);                       // scalar loads
          const float4 r0 = pr[0], r1 = pr[1], r2 = pr[2];
          if (STATS) c.wave_prims++;
          if (h && !(ANY && got)) got |= test_prim<STATS>(sc, b + k, r0, r1, r2, o, u, tmax, hit, c);
        }
        if (ANY && got) wake = YD_DONE;
        if (ANY && __any(got)) {
          const int m = wave_min_i32(wake);            // next node any lane still needs
          if (m == YD_DONE) break;
          i = __builtin_amdgcn_readfirstlane(m > i + 1 ? m : i + 1);
          continue;
        }
      }
      i = i + 1;
    } else {
      if (on && !h) wake = a;
      i = anyh ? i + 1 : a;
    }
  }
}

template <int MODE, bool ANY, bool STATS>
YD void traverse(const DScene &sc, bool active, const Ray &ray, Hit &hit, Counters &c, int ray_class) {
  // ray_class: 0 primary, 1 shadow, 2 reflection
  if (MODE == TRAV_LANE) traverse_lane<ANY, STATS>(sc, active, ray, hit, c);
  else if (MODE == TRAV_PACKET) traverse_packet<ANY, STATS>(sc, active, ray, hit, c);
  else {  // hybrid: coherent ray classes as packets, reflection rays per lane
    if (ray_class == 2) traverse_lane<ANY, STATS>(sc, active, ray, hit, c);
    else traverse_packet<ANY, STATS>(sc, active, ray, hit, c);
  }
}

// ---- shading --------------------------------------------------------------------------------------
struct Surface { V3 x, n, dpdu; V3 diffuse, specular; float shininess; };

// DifferentialGeometry of the accepted hit (Shapes.hs:23-28, 49-55) + its material (main.hs:55-60)
YD Surface surface_of(const DScene &sc, const Ray &ray, const Hit &hit, Counters &c) {
  Surface s;
  const float4 *pr = sc.prims + 3 * (size_t)hit.prim;
  const float4 r0 = pr[0], r1 = pr[1], r2 = pr[2];
  const int mat = __float_as_int(r0.w);
  s.x = ray.o + scale(hit.t, ray.d);                                  // dgPoint / rayAt
  if (__float_as_int(r1.w) == 0) {
    const float4 *nr = sc.normals + 3 * (size_t)hit.prim;
    V3 ns = tri_shading_normal(hit.b1, hit.b2, ld3(nr), ld3(nr + 1), ld3(nr + 2));
    V3 ss = norm(xyz(r2));                                            // norm e2
    V3 ts = norm(cross(ss, ns));
    s.n = ns;                                                         // NOT normalised (Shapes.hs:53)
    s.dpdu = cross(ts, ns);
  } else {
    V3 nn = norm(s.x - xyz(r0));
    s.n = nn;
    s.dpdu = cross(nn, v3(0.0f, 0.0f, 1.0f));
  }
  if (mat >= 0) {
    const float4 m0 = sc.mats[2 * (size_t)mat], m1 = sc.mats[2 * (size_t)mat + 1];
    s.diffuse = xyz(m0); s.shininess = m0.w; s.specular = xyz(m1);
  } else {
    c.bad_mat++;
    s.diffuse = vof(0.0f); s.specular = vof(0.0f); s.shininess = 0.0f;
  }
  return s;
}

// Integrators.radiance for one primary ray per lane (vcast/vhit, Integrators.hs:22-43).
// The recursion is linear, so it runs as a loop; per level k the lane stores w_k = (n.r) @* f r and
// D_k = directIllumination in LDS and the result is folded back-to-front: v = w_k * v + D_k, which is
// exactly `(n .* r) @* f r * rs + direct` evaluated in the reference's order.
template <int MODE, bool STATS>
YD V3 radiance(const DScene &sc, bool valid, Ray ray, float *lds, int lane, Counters &c) {
  bool alive = valid;
  int levels = 0;
  for (int k = 0; k < sc.depth; ++k) {
    if (!__any(alive)) break;
    Hit hit;
    if (alive) { if (k == 0) c.primary++; else c.reflection++; }
    traverse<MODE, false, STATS>(sc, alive, ray, hit, c, k == 0 ? 0 : 2);
    alive = alive && hit.prim >= 0;
    Surface s; Frame fr; V3 wo = vneg(ray.d), r = vof(0.0f), direct = vof(0.0f);
    if (alive) {
      c.shaded++;
      s = surface_of(sc, ray, hit, c);
      fr = bsdf_frame(s.n, s.dpdu);
      r = ray.d - scale(2.0f * dot(ray.d, s.n), s.n);                 // reflectionDir, Integrators.hs:46-47
    }
    // directIllumination (Integrators.hs:50-61): sum = foldl (+) 0 over the lights, in order
    for (int li = 0; li < sc.n_lights; ++li) {
      const float4 l0 = sc.lights[2 * (size_t)li], l1 = sc.lights[2 * (size_t)li + 1];
      V3 kf = vof(0.0f), ldir = vof(0.0f), intensity = vof(0.0f);
      bool need = false;
      Ray probe; probe.o = vof(0.0f); probe.d = v3(0.0f, 0.0f, 1.0f); probe.tmax = 0.0f;
      if (alive) {
        V3 p2l = xyz(l0) - s.x;                                       // Lights.hs:17-19
        ldir = norm(p2l);
        intensity = scale(1.0f / lensq(p2l), xyz(l1));
        kf = blinn_phong_at(s.diffuse, s.specular, s.shininess, fr, ldir, wo);
        need = lensq(kf) > 0.0f;                                      // && is lazy: probe only if k != 0
        if (need) {
          V3 p0 = s.x + scale(0.001f, ldir);                          // reachable, Rays.hs:49-54
          V3 dd = xyz(l0) - p0;
          probe.o = p0; probe.d = norm(dd); probe.tmax = len(dd);
          c.shadow++;
        }
      }
      Hit sh;
      traverse<MODE, true, STATS>(sc, need, probe, sh, c, 1);
      V3 contrib = vof(0.0f);
      if (need && sh.prim < 0) contrib = scale(fabsf(dot(ldir, s.n)), kf) * intensity;
      if (alive) direct = direct + contrib;
    }
    if (alive) {
      V3 f_r = blinn_phong_at(s.diffuse, s.specular, s.shininess, fr, r, wo);
      V3 w = scale(dot(s.n, r), f_r);
      float *slot = lds + (size_t)k * 6 * 64 + lane;
      slot[0] = w.x; slot[64] = w.y; slot[128] = w.z;
      slot[192] = direct.x; slot[256] = direct.y; slot[320] = direct.z;
      levels = k + 1;
      ray.o = s.x + scale(0.001f, r); ray.d = r; ray.tmax = 1e6f;     // Integrators.hs:41-43
    }
  }
  V3 v = vof(0.0f);                                                   // vcast 0 _ = 0 / miss = 0
  for (int k = levels - 1; k >= 0; --k) {
    const float *slot = lds + (size_t)k * 6 * 64 + lane;
    V3 w = v3(slot[0], slot[64], slot[128]), d = v3(slot[192], slot[256], slot[320]);
    v = (w * v) + d;
  }
  return v;
}

YD Ray camera_ray(const DScene &sc, float fu, float fv) {              // Cameras.hs:83-86
  Ray r;
  V3 origin = v3(sc.origin[0], sc.origin[1], sc.origin[2]);
  V3 direction = transform_point(sc.vtf, v3(fu, fv, sc.focal)) - origin;
  r.o = origin; r.d = norm(direction); r.tmax = 1e6f;
  return r;
}

template <int MODE, bool STATS>
YD V3 render_pixel(const DScene &sc, const RParams &rp, bool valid, int u, int gv, float *lds, int lane, Counters &c) {
  if (rp.spp <= 1)                                                     // main.hs:73: the integer pixel corner
    return radiance<MODE, STATS>(sc, valid, camera_ray(sc, (float)u, (float)gv), lds, lane, c);
  V3 acc = vof(0.0f);
  for (int k = 0; k < rp.spp; ++k) {
    float du, dv;
    sample_offset(u, gv, k, rp.n_strata, rp.seed, du, dv);
    acc = acc + radiance<MODE, STATS>(sc, valid, camera_ray(sc, (float)u + du, (float)gv + dv), lds, lane, c);
  }
  return scale(1.0f / (float)rp.spp, acc);
}

YD unsigned wave_sum_u32(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

YD void flush_counters(const Counters &c, unsigned long long *ctr, int lane) {
  unsigned vals[CTR_COUNT] = {c.primary, c.reflection, c.shadow, c.shaded, c.bad_mat,
                              c.box, c.tri, c.sph, c.wave_nodes, c.wave_prims};
#pragma unroll
  for (int i = 0; i < CTR_COUNT; ++i) {
    unsigned s = (i == CTR_WAVE_NODES || i == CTR_WAVE_PRIMS) ? __builtin_amdgcn_readfirstlane(vals[i]) : wave_sum_u32(vals[i]);
    if (lane == 0 && s) atomicAdd(&ctr[i], (unsigned long long)s);
  }
}

}  // namespace

// One wavefront (= one 64-thread workgroup) per tile.
template <int MODE, bool STATS>
__global__ __launch_bounds__(64) void k_render(DScene sc, RParams rp, float *__restrict__ out,
                                               unsigned long long *__restrict__ ctr) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  Counters c = {};
  if (rp.tile_mode == 0) {
    // 8x8-pixel tile per wavefront; consecutive blocks go to different XCDs (round-robin dispatch), so
    // the tile order is permuted to give each XCD a compact band of tile rows (L2 locality).
    int b = blockIdx.x;
    int tile = b;
    if (rp.xcd_swizzle) {
      const int per = (rp.n_tiles + 7) >> 3;
      tile = (b & 7) * per + (b >> 3);
    }
    if (tile < rp.n_tiles) {
      const int ty = tile / rp.tiles_x, tx = tile - ty * rp.tiles_x;
      const int u = tx * 8 + (lane & 7);
      const int lv = ty * 8 + (lane >> 3);
      // local row -> global image row (row sharding, include/yahr_hip.h yahr_render_opts)
      const int strip = lv / rp.strip_rows;
      const int gv = rp.row0 + strip * rp.strip_rows * rp.n_shards + (lv - strip * rp.strip_rows);
      const bool valid = u < rp.width && lv < rp.n_local_rows && gv < rp.height;
      V3 v = render_pixel<MODE, STATS>(sc, rp, valid, u, gv, lds, lane, c);
      if (valid) {
        float *px = out + ((size_t)lv * rp.width + u) * 3;              // samplesToImage: w*v+u (main.hs:100)
        px[0] = v.x; px[1] = v.y; px[2] = v.z;
      }
    }
  } else {
    // the reference's squareBatches window for this batch (Sampling.hs:5-21), u-major inside the window
    const int4 win = rp.windows[blockIdx.x];
    const int ww = win.z - win.x, hh = win.w - win.y;
    const int npx = ww * hh;
    for (int p0 = 0; p0 < npx; p0 += 64) {
      const int p = p0 + lane;
      const bool valid = p < npx;
      const int u = valid ? win.x + p / hh : 0;
      const int gv = valid ? win.y + p % hh : 0;
      V3 v = render_pixel<MODE, STATS>(sc, rp, valid, u, gv, lds, lane, c);
      if (valid) {
        float *px = out + ((size_t)gv * rp.width + u) * 3;
        px[0] = v.x; px[1] = v.y; px[2] = v.z;
      }
    }
  }
  flush_counters(c, ctr, lane);
}

// ---- diagnostic kernels: device functions on arbitrary rays (KAT checks on the GPU) -----------------
template <int MODE>
__global__ __launch_bounds__(64) void k_trace(DScene sc, int n, const float *__restrict__ rays, int any_hit,
                                              int *__restrict__ prim_out, float *__restrict__ t_out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const bool valid = i < n;
  Ray r; r.o = vof(0.0f); r.d = v3(0, 0, 1); r.tmax = 0.0f;
  if (valid) { const float *p = rays + 7 * (size_t)i; r.o = v3(p[0], p[1], p[2]); r.d = v3(p[3], p[4], p[5]); r.tmax = p[6]; }
  Hit h; Counters c = {};
  if (any_hit) traverse<MODE, true, false>(sc, valid, r, h, c, 1);
  else traverse<MODE, false, false>(sc, valid, r, h, c, 0);
  if (valid) {
    prim_out[i] = h.prim >= 0 ? __float_as_int(sc.prims[3 * (size_t)h.prim + 2].w) : -1;
    if (t_out) t_out[i] = h.t;
  }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_radiance(DScene sc, int n, const float *__restrict__ rays,
                                                 float *__restrict__ rgb, unsigned long long *__restrict__ ctr) {
  extern __shared__ float lds[];
  const int i = blockIdx.x * 64 + threadIdx.x;
  const bool valid = i < n;
  Ray r; r.o = vof(0.0f); r.d = v3(0, 0, 1); r.tmax = 0.0f;
  if (valid) { const float *p = rays + 7 * (size_t)i; r.o = v3(p[0], p[1], p[2]); r.d = v3(p[3], p[4], p[5]); r.tmax = p[6]; }
  Counters c = {};
  V3 v = radiance<MODE, false>(sc, valid, r, lds, threadIdx.x, c);
  if (valid) { rgb[3 * (size_t)i] = v.x; rgb[3 * (size_t)i + 1] = v.y; rgb[3 * (size_t)i + 2] = v.z; }
  flush_counters(c, ctr, threadIdx.x);
}

__global__ void k_camera_rays(DScene sc, int n, const float *__restrict__ uv, float *__restrict__ rays) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ray r = camera_ray(sc, uv[2 * (size_t)i], uv[2 * (size_t)i + 1]);
  float *p = rays + 7 * (size_t)i;
  p[0] = r.o.x; p[1] = r.o.y; p[2] = r.o.z; p[3] = r.d.x; p[4] = r.d.y; p[5] = r.d.z; p[6] = r.tmax;
}

// ---- launchers (called from abi_device.cpp) -------------------------------------------------------
namespace yahr_dev {

template <int MODE>
static hipError_t launch_render_mode(const DScene &sc, const RParams &rp, int n_blocks, size_t lds_bytes,
                                     float *out, unsigned long long *ctr, bool stats, hipStream_t st) {
  if (stats) hipLaunchKernelGGL((k_render<MODE, true>), dim3(n_blocks), dim3(64), lds_bytes, st, sc, rp, out, ctr);
  else hipLaunchKernelGGL((k_render<MODE, false>), dim3(n_blocks), dim3(64), lds_bytes, st, sc, rp, out, ctr);
  return hipGetLastError();
}

hipError_t launch_render(const DScene &sc, const RParams &rp, int mode, int n_blocks, float *out,
                         unsigned long long *ctr, bool stats, hipStream_t st) {
  const size_t lds_bytes = (size_t)(sc.depth > 0 ? sc.depth : 1) * 6 * 64 * sizeof(float);
  switch (mode) {
    case TRAV_LANE: return launch_render_mode<TRAV_LANE>(sc, rp, n_blocks, lds_bytes, out, ctr, stats, st);
    case TRAV_PACKET: return launch_render_mode<TRAV_PACKET>(sc, rp, n_blocks, lds_bytes, out, ctr, stats, st);
    default: return launch_render_mode<TRAV_HYBRID>(sc, rp, n_blocks, lds_bytes, out, ctr, stats, st);
  }
}

hipError_t launch_trace(const DScene &sc, int mode, int n, const float *rays, int any_hit, int *prim_out,
                        float *t_out, hipStream_t st) {
  const int nb = (n + 63) / 64;
  if (mode == TRAV_LANE) hipLaunchKernelGGL((k_trace<TRAV_LANE>), dim3(nb), dim3(64), 0, st, sc, n, rays, any_hit, prim_out, t_out);
  else hipLaunchKernelGGL((k_trace<TRAV_PACKET>), dim3(nb), dim3(64), 0, st, sc, n, rays, any_hit, prim_out, t_out);
  return hipGetLastError();
}

hipError_t launch_radiance(const DScene &sc, int mode, int n, const float *rays, float *rgb,
                           unsigned long long *ctr, hipStream_t st) {
  const int nb = (n + 63) / 64;
  const size_t lds_bytes = (size_t)(sc.depth > 0 ? sc.depth : 1) * 6 * 64 * sizeof(float);
  if (mode == TRAV_LANE) hipLaunchKernelGGL((k_radiance<TRAV_LANE>), dim3(nb), dim3(64), lds_bytes, st, sc, n, rays, rgb, ctr);
  else if (mode == TRAV_PACKET) hipLaunchKernelGGL((k_radiance<TRAV_PACKET>), dim3(nb), dim3(64), lds_bytes, st, sc, n, rays, rgb, ctr);
  else hipLaunchKernelGGL((k_radiance<TRAV_HYBRID>), dim3(nb), dim3(64), lds_bytes, st, sc, n, rays, rgb, ctr);
  return hipGetLastError();
}

hipError_t launch_camera_rays(const DScene &sc, int n, const float *uv, float *rays, hipStream_t st) {
  hipLaunchKernelGGL(k_camera_rays, dim3((n + 255) / 256), dim3(256), 0, st, sc, n, uv, rays);
  return hipGetLastError();
}

}  // namespace yahr_dev
