// render_kernel.hpp -- kernel argument blocks and launchers shared by render_kernel.hip and abi_device.cpp
#pragma once

#include <hip/hip_runtime.h>

// Scene fields that are only touched between traversals (camera, lights, materials).  They live in device
// memory behind ONE pointer instead of in the kernel-argument SGPRs: the traversal loop is SGPR-hungry
// (node and primitive records are scalar), and 25 pinned argument registers were being spilled to VGPR lanes.
struct DCold {
  const float4 *mats;      // 2 per yahr_material
  const float4 *lights;    // 2 per yahr_light
  float origin[3];         // transformPoint tf (Vec3 0 0 0)   (Cameras.hs:84)
  float focal;
  float vtf[16];           // Cameras.hs:81
};

// device view of a yahr_scene (include/yahr_hip.h); every record array is read as float4 words.
// nodes and prims carry ONE zeroed padding record at the end so that "next record" prefetches never need a
// bounds clamp.
struct DScene {
  const float4 *nodes;     // 2 per yahr_node
  const float4 *prims;     // 3 per yahr_prim
  const float4 *normals;   // 3 per yahr_prim_normals
  const DCold *cold;
  int n_nodes, n_prims, n_mats, n_lights;
  int depth;
  int boxes_finite;        // every node box has finite lo <= hi: enables the NaN-free slab form
};

struct RParams {
  int width, height;       // full image
  int n_local_rows;        // rows this launch renders (compacted)
  int row0;                // first global row of strip 0 of this shard
  int strip_rows, n_shards;
  int tiles_x, n_tiles;
  int spp, n_strata; unsigned seed;
  int tile_mode;           // 0: 8x8 wave tiles, 1: reference squareBatches windows
  int xcd_swizzle;
  // samples in lanes (tile_mode 0, spp a power of two >= 2): a wavefront holds 2^sil_log2 samples of each of
  // 64 >> sil_log2 pixels (a 2^pw_log2 wide block) instead of one sample of 64 pixels; 0 = one pixel per lane
  int sil_log2, pw_log2;
  const int4 *windows;     // tile_mode 1: {x0,y0,x1,y1}, {p0,batch,0,0} per 64-pixel chunk
  // frame batch: one launch renders `frames` frames; block b -> (tile b / frames, frame b % frames), so the
  // same tile of all frames is adjacent in dispatch order.  Frame f writes at out + f * frame_stride floats
  // and takes its camera from cams[f] (NULL: the scene camera for every frame).
  int frames;
  size_t frame_stride;
  const DCold *cams;
  const int *row_order;    // tile_mode 0: dispatch position -> tile row (centre rows first), or NULL = natural
};

// Wavefront ("stream") form of the same path, for scenes whose secondary rays are incoherent.  The radiance
// recursion (Integrators.hs:22-43) is cut into separate launches at every point where the set of live rays changes,
// and between two launches the live rays are COMPACTED (ballot + prefix count + one atomic per wavefront) into a
// queue in HBM, so that wavefronts start with 64 live rays instead of the ~36 % that hit something:
//   k_stream_primary    camera rays (packet walker), closest hit, DifferentialGeometry -> path queue 0
//   per recursion level k, for the queued paths:
//     k_stream_light    Lights.illuminationAtPoint of light l: contribution cv and the shadow ray -> ray queue
//     k_stream_trace    the shadow rays (first hit)                                          -> res[path]
//     k_stream_shade    directIllumination sum, (w_k, D_k) -> level arrays, reflection ray   -> ray queue
//     k_stream_trace    the reflection rays (closest hit)                                    -> res[path]
//     k_stream_surface  DifferentialGeometry of the hits                                     -> path queue k+1
//   k_stream_resolve    per pixel: fold the levels back to front per sample, add the samples in sample order
// k_stream_trace is RESUMABLE: the skip-linked preorder walk has no stack, so a ray's whole traversal state is
// (node index, tMax, best hit).  A trace runs in a few passes; in every pass but the last a wavefront stops as soon as
// at most `exit_lanes` of its 64 rays are still walking, parks those rays (with their state) in a continuation queue
// and the next pass picks them up compacted again -- rays of very different lengths no longer hold a wavefront at
// ~37 % lane utilisation (1 M triangles), and at most half of a pass's rays reach the next pass.
// Per ray exactly the fused kernel's arithmetic and tests in the same order, so images and counters are
// bit-identical to it.
struct SParams {
  float4 *qa[2], *qb[2], *qc[2], *qd[2];   // path queues (ping-pong by level parity), 64 B per path:
                                           //   a = {x, sample id}  b = {n, material}  c = {dpdu, rd.x}  d = {rd.y, rd.z, -, -}
  unsigned *count;                         // [depth + 1] path queue lengths, then [passes + 1] ray queue lengths
  float4 *lev;                             // [(2 k + {0: w_k, 1: D_k}) * capacity + sample id]
  unsigned char *levels;                   // [capacity] number of levels whose ray hit something, per sample
  unsigned capacity;                       // samples of this pass = local rows x width x spp
  const DCold *cam;                        // camera of this pass
  int level, last;                         // recursion level k of the launch, and whether it is the last one
  // rays of the traversal in flight, 48 B per ray: a = {x0, tMax now}  b = {u, path index}  c = {node index, best
  // primitive, b1, b2} (the best hit's t IS tMax); ping-pong by pass parity, the second half as large as the first
  float4 *ca[2], *cb[2], *cc[2];
  unsigned *rcount;                        // count + depth + 1: [pass] rays entering that pass
  float4 *res;                             // [path index] {best primitive, t, b1, b2} of the finished traversal
  float4 *cv;                              // [path index] {cv, active}: contribution of the light being probed
  float4 *direct;                          // [path index] directIllumination summed over the lights done so far
  int light, n_lights;                     // k_stream_light: which light
  int pass, exit_lanes, min_steps;         // k_stream_trace
};

enum { TRAV_LANE = 1, TRAV_PACKET = 2, TRAV_ADAPTIVE = 3, TRAV_STREAM = 4 };

enum { CTR_PRIMARY = 0, CTR_REFLECTION, CTR_SHADOW, CTR_SHADED, CTR_BAD_MAT,
       CTR_BOX, CTR_TRI, CTR_SPH, CTR_WAVE_NODES, CTR_WAVE_PRIMS, CTR_COUNT,
       // appended after the summed counters: per-wave shader-clock statistics (collect_stats builds)
       CTR_MAX_WAVE_CYCLES = CTR_COUNT, CTR_SUM_WAVE_CYCLES, CTR_TOTAL };

namespace yahr_dev {
hipError_t launch_render(const DScene &sc, const RParams &rp, int mode, bool regroup, int n_blocks, float *out,
                         unsigned long long *ctr, bool stats, hipStream_t st);
hipError_t launch_trace(const DScene &sc, int mode, int n, const float *rays, int any_hit, int *prim_out,
                        float *t_out, hipStream_t st);
hipError_t launch_radiance(const DScene &sc, int mode, int n, const float *rays, float *rgb,
                           unsigned long long *ctr, hipStream_t st);
hipError_t launch_camera_rays(const DScene &sc, int n, const float *uv, float *rays, hipStream_t st);
// stream form: one pass = primary + depth level launches + resolve, all on `st`; sp.count must be zeroed by the caller
hipError_t launch_stream_pass(const DScene &sc, const RParams &rp, SParams sp, int n_trace_passes, int n_blocks, float *out,
                              unsigned long long *ctr, bool stats, hipStream_t st);
}  // namespace yahr_dev
