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
  int boxes_finite;        // bit 0: every node box has finite lo <= hi (enables the NaN-free slab form);
                           // bit 1: every triangle record's p0, e1, e2 within +-2^32 (enables tri_far_packet, device_math.hpp)
                           // bit 2: every child's box inside its parent's (enables the octant run, render_kernel.hip)
};

// Child-pair records for the stack walkers (render_kernel.hip: stack_loop), handed to the kernels that may run them (kept
// out of DScene: the packet kernels' register allocation is sensitive to the size of their argument block).  Interior
// node of rank k (k-th interior node in DFS order; the root is 0) owns 64 B = the records of its LEFT and RIGHT child side
// by side, 4 float4 words.  A child record is the child's yahr_node with other links: leaf {lo, count, hi, first primitive}
// as in `nodes`; interior {lo, rank of the child, hi, -1 - DFS index of the child}.  pairs == NULL when the scene has none
// (root is a leaf, a box is not finite, or a node's box is not exactly the join of its children's).  stack_levels = depth
// of the deepest node = entries of the per-lane LDS stack of pending right siblings.
struct DPairs {
  const float4 *pairs;
  int stack_levels;
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
  // (w_k, D_k) stack of the radiance recursion: the first lds_levels levels in LDS, the rest in lvl_spill (global,
  // [blocks of this launch][depth - lds_levels][6][threads per block]); block0 = grid position of this launch's first
  // block when a grid is cut into several launches so that the spill stays under a budget
  float *lvl_spill;
  int lds_levels;
  unsigned block0;
  // the usual launch (8x8 wave tiles, one wave per workgroup, natural order, no level spill) is a 3-D grid
  // (frames, tiles_x, tile_rows): blockIdx is (frame, tile column, tile row) and nothing has to be divided
  int grid3d, tile_rows;
};
// Scenes of recursion depth <= YD_LDS_DEPTH keep the whole radiance stack in LDS (1.5 KiB per level and wavefront, behind
// the 2 KiB parking area of radiance(): depth 2 = 5120 bytes = four of gfx950's 1280-byte LDS granules, room for 32
// wavefronts per CU where the packet kernel asks for 28; a fifth granule would leave 25); deeper ones keep YD_LDS_LEVELS levels there and the rest in the spill (a level in global memory costs next
// to nothing: pseudo-bunny 3840x2160 x 64 spp, depth 4, measured in round 3 with three levels in LDS + one spilled
// 92.9 ms per frame, all four in LDS -- fewer wavefronts per CU -- 93.8).
#ifdef YD_LDS_DEPTH_OVERRIDE
#define YD_LDS_DEPTH YD_LDS_DEPTH_OVERRIDE
#else
#define YD_LDS_DEPTH 2
#endif
#define YD_LDS_LEVELS 2

// Wavefront ("stream") form of the same path, for scenes whose secondary rays are incoherent: the radiance
// recursion (Integrators.hs:22-43) is cut at its level boundaries into separate launches, and between two launches
// the surviving paths are COMPACTED (ballot + prefix count + one atomic per wavefront) into a queue in HBM, so that
// every wavefront of the next launch starts with 64 live rays instead of the ~36 % that hit something.
//   k_stream_primary   camera rays (packet walker), closest hit, DifferentialGeometry -> queue 0
//   k_stream_level k   per queued path: shadow probes of every light, (w_k, D_k) -> level arrays, reflection ray,
//                      closest hit, DifferentialGeometry -> queue k+1            (one launch per recursion level)
//   k_stream_resolve   per pixel: fold the levels back to front per sample, add the samples in sample order
// Per ray exactly the fused kernel's arithmetic and tests, so images and counters are bit-identical to it.
struct SParams {
  float4 *qa[2], *qb[2], *qc[2], *qd[2];   // path queues (ping-pong by level parity), 64 B per path:
                                           //   a = {x, sample id}  b = {n, material}  c = {dpdu, rd.x}  d = {rd.y, rd.z, -, -}
  unsigned *count;                         // [depth + 1] queue lengths
  float4 *lev;                             // [(2 k + {0: w_k, 1: D_k}) * capacity + sample id]
  unsigned short *levels;                  // [capacity] number of levels whose ray hit something, per sample (depth <= 65535)
  unsigned capacity;                       // samples of this pass = band rows x width x spp
  int tile0, lv0, band_rows;               // the pass covers the wave tiles from tile0 on = local rows [lv0, lv0 + band_rows)
  const DCold *cam;                        // camera of this pass
  int level, last;                         // k_stream_level: recursion level k, and whether it is the last one
  // order of the level launch: the queued paths sorted by the Morton cell of their hit point (counting sort, below)
  uint2 *cell;                             // [capacity] {Morton cell, rank inside the cell} of queue entry e (of the queue being filled)
  unsigned *perm;                          // [capacity] queue entry handled by lane e of the level launch
  unsigned *hist;                          // [depth + 1][YD_SORT_CELLS] entries per cell of queue k; zeroed by the caller with the queue lengths
  unsigned *start;                         // [YD_SORT_CELLS] first slot of a cell (exclusive scan of hist[k])
};
#ifndef YD_SORT_BITS
#define YD_SORT_BITS 18                    // 6 bits per axis: 64^3 cells of the root box
#endif
#define YD_SORT_CELLS (1u << YD_SORT_BITS)

enum { TRAV_LANE = 1, TRAV_PACKET = 2, TRAV_ADAPTIVE = 3, TRAV_STREAM = 4,
       // walkers of the stream form's level launches (and of the diagnostic ray queries) on the child-pair records:
       TRAV_STACK = 5,          // every lane on its own with an LDS stack of pending right siblings: reference order
       TRAV_ADAPT_STACK = 6,    // packet while coherent, then TRAV_STACK
       TRAV_ORDERED = 7 };      // TRAV_STACK visiting the nearer child first (opt-in: YAHR_TRAVERSE_ORDERED)

enum { CTR_PRIMARY = 0, CTR_REFLECTION, CTR_SHADOW, CTR_SHADED, CTR_BAD_MAT,
       CTR_BOX, CTR_TRI, CTR_SPH, CTR_WAVE_NODES, CTR_WAVE_PRIMS, CTR_WAVE_STAGE1, CTR_WAVE_STAGE2, CTR_TRI_CULLED, CTR_WAVE_CULLS, CTR_CULL_VIOL, CTR_COUNT,
       // appended after the summed counters: per-wave shader-clock statistics (collect_stats builds)
       CTR_MAX_WAVE_CYCLES = CTR_COUNT, CTR_SUM_WAVE_CYCLES, CTR_TOTAL };

namespace yahr_dev {
hipError_t launch_render(const DScene &sc, const RParams &rp, int mode, bool regroup, int n_blocks, float *out,
                         unsigned long long *ctr, bool stats, hipStream_t st);
hipError_t launch_trace(const DScene &sc, const DPairs &dp, int mode, int n, const float *rays, int any_hit, int *prim_out,
                        float *t_out, unsigned long long *ctr, hipStream_t st);
hipError_t launch_radiance(const DScene &sc, int mode, int n, const float *rays, float *rgb,
                           unsigned long long *ctr, float *lvl_spill, int lds_levels, hipStream_t st);
hipError_t launch_warmup(hipStream_t st);          // an empty kernel: makes the runtime load this library's code object
hipError_t launch_camera_rays(const DScene &sc, int n, const float *uv, float *rays, hipStream_t st);
hipError_t launch_debug_powf(int n, const float *x, const float *y, float *out, hipStream_t st);   // test hook (device_math.hpp: glibc_powf)
void debug_powf_host(int n, const float *x, const float *y, float *out);                              // ... compiled for the host
// stream form: one pass = primary + depth level launches + resolve, all on `st`; sp.count must be zeroed by the caller
hipError_t launch_stream_pass(const DScene &sc, const DPairs &dp, const RParams &rp, SParams sp, int secondary_mode, int n_blocks, float *out,
                              unsigned long long *ctr, bool stats, hipStream_t st);
}  // namespace yahr_dev
