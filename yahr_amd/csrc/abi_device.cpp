// abi_device.cpp -- device part of the C ABI (include/yahr_hip.h sections 3 and 4): a scene resident in
// one GPU's HBM, render launches on a caller-supplied HIP stream, diagnostic ray queries, and the
// one-shot host-buffer conveniences.  There is NO CPU fallback: without a usable HIP device every entry
// point here fails with YAHR_E_NODEVICE / YAHR_E_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.hpp"
#include "render_kernel.hpp"

using yahr::set_error;

#define YD_PASS_STREAMS_MAX 4              /* internal streams (each with its own queue set) the passes of the wavefront form may be dealt to */
#ifndef YD_PASS_STREAMS
#define YD_PASS_STREAMS 2
#endif

void yahr_internal_counters_to_stats(const unsigned long long *h, yahr_stats *out);

struct yahr_ctx {
  int device = 0;
  DScene ds{};
  DPairs dp{};                            // child-pair records of the stack walkers, or {NULL, 0}
  int width = 0, height = 0;
  void *d_nodes = nullptr, *d_prims = nullptr, *d_normals = nullptr, *d_mats = nullptr, *d_lights = nullptr;
  void *d_cold = nullptr;
  void *d_pairs = nullptr;                // child-pair records of the stack walkers (render_kernel.hpp: DPairs), or NULL
  bool pairs_tried = false;               // ... built on the first use of a stack walker (ensure_pairs)
  DCold cold_host{};
  DCold *d_cams = nullptr; int n_cams = 0;
  int *d_row_order = nullptr; int row_order_n = 0;
  int auto_mode = TRAV_PACKET;            // AUTO: the fused kernel's walker ...
  size_t auto_stream_min = (size_t)-1;    // ... and from how many samples per launch on the wavefront form instead
  double probe_lanes = -1.0;              // lane utilisation of the packet walker measured by the probe of yahr_ctx_create (-1: not probed)
  bool deep_thin = false;                 // more than 300 k nodes and at most 4 primitives per leaf on average
  int stream_secondary = TRAV_ADAPTIVE;   // walker of the secondary rays in the wavefront form (auto_policy_for)
  unsigned long long *d_ctr = nullptr;
  int4 *d_windows = nullptr; int n_windows = 0, n_chunks = 0, windows_w = 0, windows_h = 0;
  // wavefront ("stream") form: path queues, per-level results, queue lengths (allocated on first use, grown on demand)
  // ... in TWO sets: the passes of a launch (frames of a batch, bands of a frame) are dealt to two internal streams, each
  // with its own queues, so that the tail of one pass -- a few long wavefronts on a mostly idle chip -- overlaps the next
  void *d_stream[YD_PASS_STREAMS_MAX] = {}; size_t stream_bytes[YD_PASS_STREAMS_MAX] = {};
  hipStream_t pass_stream[YD_PASS_STREAMS_MAX] = {};
  hipEvent_t ev_fork = nullptr, ev_join[YD_PASS_STREAMS_MAX] = {};
  // fused kernels, scenes deeper than YD_LDS_LEVELS: the upper levels of the radiance stack (allocated on first use)
  void *d_spill = nullptr; size_t spill_bytes = 0;
  unsigned *d_qcount[YD_PASS_STREAMS_MAX] = {}; int qcount_n[YD_PASS_STREAMS_MAX] = {};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  double upload_ms = 0;
};

namespace {

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                        \
      return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? YAHR_E_NODEVICE                \
             : e_ == hipErrorOutOfMemory ? YAHR_E_NOMEM : YAHR_E_HIP;                                  \
    }                                                                                      \
  } while (0)

// ---- ownership helpers: every early return (HIP_TRY) must release what the function acquired ----
// Device allocations of this file go through dev_alloc so that tests can make the k-th one fail
// (yahr_debug_fail_alloc) and check with yahr_device_mem_info that no error path leaks.
std::atomic<int> g_fail_alloc_countdown{0};
hipError_t dev_alloc(void **p, size_t bytes) {
  *p = nullptr;
  int k = g_fail_alloc_countdown.load();
  while (k > 0 && !g_fail_alloc_countdown.compare_exchange_weak(k, k - 1)) {}
  if (k == 1) return hipErrorOutOfMemory;                 // the injected failure
  return hipMalloc(p, bytes);
}
struct DevMem {                                           // one device allocation, freed on scope exit
  void *p = nullptr;
  DevMem() = default;
  DevMem(const DevMem &) = delete;
  DevMem &operator=(const DevMem &) = delete;
  ~DevMem() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return dev_alloc(&p, bytes); }
  template <class T> T *as() const { return static_cast<T *>(p); }
};
struct ScopedEvent {
  hipEvent_t e = nullptr;
  ~ScopedEvent() { if (e) (void)hipEventDestroy(e); }
};
struct CtxOwner {                                         // destroys a half-built context unless released
  yahr_ctx *c;
  explicit CtxOwner(yahr_ctx *c_) : c(c_) {}
  ~CtxOwner() { if (c) yahr_ctx_destroy(c); }
  yahr_ctx *release() { yahr_ctx *r = c; c = nullptr; return r; }
};

int check_scene(const yahr_scene *s) {
  if (!s) { set_error("null scene"); return YAHR_E_ARG; }
  if (s->abi_version != YAHR_ABI_VERSION) { set_error("yahr_scene.abi_version mismatch"); return YAHR_E_ARG; }
  if (s->width < 0 || s->height < 0 || s->depth < 0 || s->n_nodes < 0 || s->n_prims < 0 || s->n_mats < 0 || s->n_lights < 0) {
    set_error("negative count in yahr_scene"); return YAHR_E_ARG;
  }
  if ((s->n_nodes && !s->nodes) || (s->n_prims && (!s->prims || !s->normals)) || (s->n_mats && !s->mats) || (s->n_lights && !s->lights)) {
    set_error("null array in yahr_scene"); return YAHR_E_ARG;
  }
  // the kernel addresses records with 32-bit byte offsets
  if ((uint64_t)(s->n_nodes + 1) * sizeof(yahr_node) > 0xffffffffULL || (uint64_t)(s->n_prims + 1) * sizeof(yahr_prim) > 0xffffffffULL) {
    set_error("scene too large: node or primitive array exceeds 4 GiB"); return YAHR_E_ARG;
  }
  // The kernel trusts the topology: validate it here so a malformed scene cannot fault the GPU.
  for (int32_t i = 0; i < s->n_nodes; ++i) {
    const yahr_node &nd = s->nodes[i];
    if (nd.b >= 0) {
      if (nd.a < 0 || (int64_t)nd.b + nd.a > s->n_prims) { set_error("leaf primitive range out of bounds at node " + std::to_string(i)); return YAHR_E_ARG; }
    } else if (nd.b == -1) {
      if (nd.a <= i || nd.a > s->n_nodes) { set_error("skip link not forward at node " + std::to_string(i)); return YAHR_E_ARG; }
    } else { set_error("bad node tag at node " + std::to_string(i)); return YAHR_E_ARG; }
  }
  for (int32_t i = 0; i < s->n_prims; ++i) {
    const yahr_prim &p = s->prims[i];
    if ((p.kind != 0 && p.kind != 1) || p.mat < -1 || p.mat >= s->n_mats) { set_error("bad primitive record " + std::to_string(i)); return YAHR_E_ARG; }
  }
  return YAHR_OK;
}

// copies `bytes` and appends `pad` zero bytes (one padding record: the kernel prefetches "the next record")
int upload(void **dst, const void *src, size_t bytes, size_t pad = 64) {
  // *dst belongs to the context being built: on failure its owner (CtxOwner) frees it
  HIP_TRY(dev_alloc(dst, bytes + pad));
  HIP_TRY(hipMemset((char *)*dst + bytes, 0, pad));
  if (src && bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return YAHR_OK;
}

// origin = transformPoint tf (Vec3 0 0 0)  (Cameras.hs:84), same left fold as the device's transform_point
void camera_origin(const float tf[16], float origin[3]) {
  float tv[4];
  for (int i = 0; i < 4; ++i) {
    float acc = 0.0f;
    acc = acc + tf[i * 4 + 0] * 0.0f; acc = acc + tf[i * 4 + 1] * 0.0f;
    acc = acc + tf[i * 4 + 2] * 0.0f; acc = acc + tf[i * 4 + 3] * 1.0f;
    tv[i] = acc;
  }
  const float iw = 1.0f / tv[3];
  origin[0] = iw * tv[0]; origin[1] = iw * tv[1]; origin[2] = iw * tv[2];
}

int pick_mode(int traverse, int auto_mode = TRAV_PACKET) {
  switch (traverse) {
    case YAHR_TRAVERSE_LANE: return TRAV_LANE;
    case YAHR_TRAVERSE_PACKET: return TRAV_PACKET;
    case YAHR_TRAVERSE_HYBRID: return TRAV_ADAPTIVE;
    case YAHR_TRAVERSE_STREAM: return TRAV_STREAM;
    // the wavefront form with a named walker for its secondary rays
    case YAHR_TRAVERSE_STACK: case YAHR_TRAVERSE_HYBRID_STACK: case YAHR_TRAVERSE_ORDERED: return TRAV_STREAM;
    default: return auto_mode;       // YAHR_TRAVERSE_AUTO: chosen per scene in yahr_ctx_create
  }
}

// Child-pair records of the stack walkers (render_kernel.hpp: DPairs) from the flattened tree: two 32-byte child
// records per interior node, in DFS order of the interior nodes.  Returns false -- the scene then simply has no pair
// records and every walker uses the node array -- unless the tree is a proper binary tree in DFS order whose every
// interior box is EXACTLY the join of its children's boxes (Culling.hs:39; a foreign host may hand over looser boxes)
// and every box is finite with lo <= hi (the stack walker re-derives a popped node's box from its children's and
// relies on the monotonicity of the NaN-free slab test).
bool build_pairs(const yahr_scene *s, std::vector<yahr_node> &halves, int &levels) {
  const int32_t n = s->n_nodes;
  if (n < 3 || s->nodes[0].b >= 0) return false;
  std::vector<int32_t> rank((size_t)n, -1), dep((size_t)n, -1);
  int32_t n_int = 0;
  for (int32_t i = 0; i < n; ++i) if (s->nodes[i].b < 0) rank[(size_t)i] = n_int++;
  halves.assign((size_t)n_int * 2, yahr_node{});
  dep[0] = 0;
  int32_t deepest = 0;
  auto skip_of = [&](int32_t i) { return s->nodes[i].b >= 0 ? i + 1 : s->nodes[i].a; };
  auto child = [&](int32_t c) {
    yahr_node h = s->nodes[c];
    if (h.b < 0) { h.a = rank[(size_t)c]; h.b = -1 - c; }
    return h;
  };
  for (int32_t i = 0; i < n; ++i) {
    const yahr_node &nd = s->nodes[i];
    if (dep[(size_t)i] < 0) return false;                    // not reached from the root
    for (int k = 0; k < 3; ++k)
      if (!std::isfinite(nd.lo[k]) || !std::isfinite(nd.hi[k]) || !(nd.lo[k] <= nd.hi[k])) return false;
    if (nd.b >= 0) continue;
    const int32_t l = i + 1;
    if (l >= n) return false;
    const int32_t r = skip_of(l);
    if (r <= l || r >= n || skip_of(r) != nd.a) return false;
    const yahr_node &a = s->nodes[l], &b = s->nodes[r];
    for (int k = 0; k < 3; ++k)
      if (!(nd.lo[k] == std::fmin(a.lo[k], b.lo[k])) || !(nd.hi[k] == std::fmax(a.hi[k], b.hi[k]))) return false;
    dep[(size_t)l] = dep[(size_t)r] = dep[(size_t)i] + 1;
    if (dep[(size_t)l] > deepest) deepest = dep[(size_t)l];
    halves[2 * (size_t)rank[(size_t)i]] = child(l);
    halves[2 * (size_t)rank[(size_t)i] + 1] = child(r);
  }
  levels = deepest > 0 ? deepest : 1;                      // a node at depth d has at most d pending right siblings
  // 256 B of LDS per level and wavefront: beyond 48 levels (12 KiB) the stack costs more occupancy than it saves fetches
  return levels <= 48 && (uint64_t)n_int * 64 <= 0xffffffffULL;
}

// AUTO policy, from a sweep, not from the benchmark scenes: random triangles 50 k .. 4 M x BVH depth 16 / 20 x 1 / 4 / 16 /
// 64 spp at 1080p in every form (tools/auto_sweep.py, table in profiles/r03_auto_sweep.csv; ms per frame).  What it says:
//   * below ~150 k primitives the packet walker wins at every sample count (50 k: 0.73 / 1.47 / 4.13 / 14.2 ms against
//     1.80 / 1.85 / 4.09 / 17.2 in the wavefront form): tiny trees, the union of 64 paths stays small;
//   * from 200 k primitives on the wavefront ("stream") form wins as soon as a launch holds enough paths to fill its level
//     launches -- 4 spp at 1080p (8 M samples) everywhere (200 k: 4.6 against 5.2; 1 M: 14.3 against 20.6 .. 46.7; 4 M:
//     46 .. 142 against 68 .. 223) -- whatever the tree depth and the leaf size;
//   * at 1 spp (2 M samples) it still wins where leaves are FAT (more than 16 primitives: 1 M in a depth-16 tree 17.0
//     against 23.9, 4 M 63.7 against 99.8), not where they are thin (1 M in a depth-20 tree: fused adaptive 7.0, stream
//     9.5, packet 16.1; 200 k, depth 20: adaptive 2.1, packet 2.9, stream 3.9): there the fused kernel runs, the adaptive
//     walker for a big tree (more than 300 k nodes), the packet walker otherwise (200 k, depth 16: 2.8 against 3.4);
//   * secondary rays of the wavefront form: on fat leaves the stack walker after the packet start is 5-10 % ahead of the
//     skip-link lane walker (1 M, depth 16: 17.0 / 29.4 / 64.5 / 162 against 18.3 / 31.6 / 67.0 / 165), on thin leaves
//     2-5 % behind (1 M, depth 20, 16 spp: 35.7 against 34.0).
struct AutoPolicy { int fused; size_t stream_min_samples; int secondary; bool deep_thin; };
AutoPolicy auto_policy_for(const yahr_scene *s) {
  AutoPolicy p{TRAV_PACKET, (size_t)-1, TRAV_ADAPTIVE, false};           // (size_t)-1: never the wavefront form
  // (a second sweep filled the gaps -- 100 k / 300 k / 500 k primitives x depth 14 / 16 / 20 x 1 / 2 / 4 spp,
  // profiles/r03_auto_sweep_gaps.csv: at 100 k the packet walker still wins up to 8 M samples (1.5-2.9 ms against 3.5),
  // at 300 k the wavefront form from 4 M samples on but not yet at 2 M even on fat leaves (7.5 against 8.2), at 500 k it
  // does: 13.0 against 15.8)
  if (s->n_prims < 150000) return p;
  int32_t biggest = 0, n_leaves = 0;
  for (int32_t i = 0; i < s->n_nodes; ++i)
    if (s->nodes[i].b >= 0) { ++n_leaves; if (s->nodes[i].a > biggest) biggest = s->nodes[i].a; }
  const bool fat = biggest > 16;
  // a big tree whose TYPICAL leaf is thin (a mesh in a deep tree keeps one fat leaf -- the floor's -- and is thin otherwise)
  p.deep_thin = s->n_nodes > 300000 && (int64_t)s->n_prims <= 4 * (int64_t)n_leaves;
  p.fused = (!fat && s->n_nodes > 300000) ? TRAV_ADAPTIVE : TRAV_PACKET;
  // fat leaves and 400 k primitives or more: one 1080p frame at 1 spp is enough
  p.stream_min_samples = (fat && s->n_prims >= 400000) ? ((size_t)2 << 20) - 65536 : (size_t)3 << 20;
  p.secondary = fat ? TRAV_ADAPT_STACK : TRAV_ADAPTIVE;
  return p;
}

// The pair records are built the first time a stack walker is asked for (they cost a pass over the tree and 64 B per
// interior node, and the default walkers do not use them): from the node array as it sits in device memory.
int ensure_pairs(yahr_ctx *c) {
  if (c->pairs_tried) return YAHR_OK;
  c->pairs_tried = true;
  if (std::getenv("YAHR_NO_PAIRS") || c->ds.n_nodes < 3) return YAHR_OK;
  std::vector<yahr_node> nodes((size_t)c->ds.n_nodes), halves;
  HIP_TRY(hipMemcpy(nodes.data(), c->d_nodes, sizeof(yahr_node) * nodes.size(), hipMemcpyDeviceToHost));
  yahr_scene view{};
  view.n_nodes = c->ds.n_nodes; view.nodes = nodes.data();
  int levels = 0;
  if (!build_pairs(&view, halves, levels)) return YAHR_OK;          // this scene has none: the walkers use the node array
  int rc = upload(&c->d_pairs, halves.data(), sizeof(yahr_node) * halves.size());
  if (rc) { c->pairs_tried = false; return rc; }                     // (out of memory now: try again next time)
  c->dp.pairs = (const float4 *)c->d_pairs; c->dp.stack_levels = levels;
  return YAHR_OK;
}

}  // namespace

extern "C" {

void yahr_default_opts(yahr_render_opts *o) {
  if (!o) return;
  std::memset(o, 0, sizeof *o);
  o->spp = 1; o->traverse = YAHR_TRAVERSE_AUTO; o->tiles = YAHR_TILES_WAVE; o->n_shards = 1; o->strip_rows = 8;
}

int yahr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int yahr_device_warmup(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); set_error("no HIP device available (libyahr_hip has no CPU fallback)"); return YAHR_E_NODEVICE; }
  if (device < 0 || device >= ndev) { set_error("device index out of range"); return YAHR_E_NODEVICE; }
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipFree(nullptr));                              // creates the context
  HIP_TRY(yahr_dev::launch_warmup(nullptr));              // loads the kernel image
  HIP_TRY(hipDeviceSynchronize());
  return YAHR_OK;
}

int yahr_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device available (libyahr_hip has no CPU fallback)"); return YAHR_E_NODEVICE; }
  if (device < 0 || device >= ndev) { set_error("device index out of range"); return YAHR_E_NODEVICE; }
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipDeviceSynchronize());
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return YAHR_OK;
}

// Test hook (host only, no device needed): the child-pair records the stack walkers would get for this scene.
// DScene.boxes_finite for a scene (render_kernel.hpp).  The node and primitive arrays are the caller's, so what the packet
// walker's shortcuts rely on is checked here, once per context:
//   bit 0  every node box finite with lo <= hi                     -> the NaN-free slab form (device_math.hpp: slab_hit_fast)
//   bit 2  ... and every child's box inside its parent's (it is in every tree Culling.hs:39 or this library builds: a
//          parent's box IS the join of its children's)            -> the octant run: a ray that failed a node's box fails
//          every box below it, so lanes need no memory of the subtrees they left (render_kernel.hip: oct_run)
//   bit 1  every triangle record's p0, e1, e2 within +-2^32       -> the packet-level tri_far proof (tri_far_packet)
static int scene_box_flags(const yahr_scene *s) {
  int flags = 1;
  for (int32_t i = 0; i < s->n_nodes && flags; ++i)
    for (int k = 0; k < 3; ++k)
      if (!std::isfinite(s->nodes[i].lo[k]) || !std::isfinite(s->nodes[i].hi[k]) || !(s->nodes[i].lo[k] <= s->nodes[i].hi[k]))
        flags = 0;
  if (flags) {
    bool nested = true;
    for (int32_t i = 0; i + 2 < s->n_nodes && nested; ++i) {
      const yahr_node &p = s->nodes[i];
      if (p.b >= 0) continue;                                             // leaf
      const int32_t l = i + 1, r = s->nodes[l].b < 0 ? s->nodes[l].a : l + 1;
      if (r <= l || r >= s->n_nodes) { nested = false; break; }
      for (int k = 0; k < 3; ++k)
        if (!(s->nodes[l].lo[k] >= p.lo[k] && s->nodes[l].hi[k] <= p.hi[k] && s->nodes[r].lo[k] >= p.lo[k] && s->nodes[r].hi[k] <= p.hi[k]))
          nested = false;
    }
    if (nested) flags |= 4;
  }
  bool bounded = true;
  for (int32_t i = 0; i < s->n_prims && bounded; ++i) {
    const yahr_prim &p = s->prims[i];
    if (p.kind != 0) continue;
    for (int k = 0; k < 3; ++k)
      if (!(std::fabs(p.p0[k]) <= 4294967296.0f) || !(std::fabs(p.e1[k]) <= 4294967296.0f) || !(std::fabs(p.e2[k]) <= 4294967296.0f))
        bounded = false;
  }
  if (bounded) flags |= 2;
  return flags;
}
int32_t yahr_debug_box_flags(const yahr_scene *s) {
  if (!s || (s->n_nodes && !s->nodes) || (s->n_prims && !s->prims)) { set_error("bad arguments"); return YAHR_E_ARG; }
  return scene_box_flags(s);
}

int32_t yahr_debug_pair_records(const yahr_scene *s, yahr_node *out, int32_t cap, int32_t *levels) {
  if (levels) *levels = 0;
  if (!s || s->n_nodes < 0 || (s->n_nodes && !s->nodes)) return 0;
  std::vector<yahr_node> halves; int lv = 0;
  if (!build_pairs(s, halves, lv)) return 0;
  if (levels) *levels = lv;
  if (out) std::memcpy(out, halves.data(), sizeof(yahr_node) * std::min((size_t)(cap > 0 ? cap : 0), halves.size()));
  return (int32_t)halves.size();
}

void yahr_debug_fail_alloc(int32_t nth) { g_fail_alloc_countdown.store(nth > 0 ? nth : 0); }

int32_t yahr_shard_rows(int32_t height, int32_t n_shards, int32_t shard, int32_t strip_rows) {
  if (n_shards <= 1) return height;
  if (strip_rows <= 0 || shard < 0 || shard >= n_shards) return 0;
  int32_t rows = 0;
  for (int32_t s = shard; (int64_t)s * strip_rows < height; s += n_shards) {
    int32_t r0 = s * strip_rows, r1 = r0 + strip_rows;
    rows += (r1 > height ? height : r1) - r0;
  }
  return rows;
}
int32_t yahr_shard_global_row(int32_t local, int32_t n_shards, int32_t shard, int32_t strip_rows) {
  if (n_shards <= 1) return local;
  int32_t k = local / strip_rows;
  return (k * n_shards + shard) * strip_rows + (local - k * strip_rows);
}

// samples_hint: the largest launch (rows x width x spp of one frame) this context will ever be asked for under AUTO, when the
// caller knows it (the one-shot calls do); SIZE_MAX otherwise.  It only decides whether the coherence probe below is worth
// running: the probe settles a choice that AUTO makes from `auto_stream_min` samples on.
static int ctx_create_impl(const yahr_scene *s, int device, yahr_ctx **out, size_t samples_hint) {
  if (!out) { set_error("null argument"); return YAHR_E_ARG; }
  *out = nullptr;
  // YAHR_TIMING=1: the phases of this call on stderr (wall clock; diagnostics of the command line's start-up cost)
  const bool timing = std::getenv("YAHR_TIMING") && std::atoi(std::getenv("YAHR_TIMING")) != 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
  auto t_phase = now();
  int rc = check_scene(s);
  if (rc != YAHR_OK) return rc;
  const double ms_check = ms_since(t_phase);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) { set_error("no HIP device available (libyahr_hip has no CPU fallback)"); return YAHR_E_NODEVICE; }
  if (device < 0 || device >= ndev) { set_error("device index out of range"); return YAHR_E_NODEVICE; }
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error(std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only"); return YAHR_E_NODEVICE;
  }
  CtxOwner own(new yahr_ctx());                           // every return below destroys it, except the last
  yahr_ctx *c = own.c;
  c->device = device; c->width = s->width; c->height = s->height;
  const double ms_device = ms_since(t_phase) - ms_check;
  t_phase = now();
  ScopedEvent u0, u1;
  HIP_TRY(hipEventCreate(&u0.e)); HIP_TRY(hipEventCreate(&u1.e));
  HIP_TRY(hipEventRecord(u0.e, nullptr));
  if ((rc = upload(&c->d_nodes, s->nodes, sizeof(yahr_node) * (size_t)s->n_nodes)) ||
      (rc = upload(&c->d_prims, s->prims, sizeof(yahr_prim) * (size_t)s->n_prims)) ||
      (rc = upload(&c->d_normals, s->normals, sizeof(yahr_prim_normals) * (size_t)s->n_prims)) ||
      (rc = upload(&c->d_mats, s->mats, sizeof(yahr_material) * (size_t)s->n_mats)) ||
      (rc = upload(&c->d_lights, s->lights, sizeof(yahr_light) * (size_t)s->n_lights)) ||
      (rc = upload((void **)&c->d_ctr, nullptr, sizeof(unsigned long long) * CTR_TOTAL)))
    return rc;
  HIP_TRY(hipMemset(c->d_ctr, 0, sizeof(unsigned long long) * CTR_TOTAL));
  HIP_TRY(hipEventRecord(u1.e, nullptr));
  HIP_TRY(hipEventSynchronize(u1.e));
  float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, u0.e, u1.e)); c->upload_ms = ms;
  HIP_TRY(hipEventCreate(&c->ev0)); HIP_TRY(hipEventCreate(&c->ev1));
  const double ms_upload = ms_since(t_phase);
  t_phase = now();
  DCold cold{};
  cold.mats = (const float4 *)c->d_mats; cold.lights = (const float4 *)c->d_lights;
  camera_origin(s->cam_tf, cold.origin);
  std::memcpy(cold.vtf, s->cam_vtf, sizeof cold.vtf);
  cold.focal = s->focal;
  c->cold_host = cold;
  if ((rc = upload(&c->d_cold, &cold, sizeof cold))) return rc;
  DScene &d = c->ds;
  d.nodes = (const float4 *)c->d_nodes; d.prims = (const float4 *)c->d_prims; d.normals = (const float4 *)c->d_normals;
  d.cold = (const DCold *)c->d_cold;
  d.n_nodes = s->n_nodes; d.n_prims = s->n_prims; d.n_mats = s->n_mats; d.n_lights = s->n_lights;
  d.depth = s->depth;
  { const AutoPolicy ap = auto_policy_for(s); c->auto_mode = ap.fused; c->auto_stream_min = ap.stream_min_samples; c->stream_secondary = ap.secondary; c->deep_thin = ap.deep_thin; }
  // what the packet walker's shortcuts may assume about this scene (scene_box_flags), minus what the environment switches off
  d.boxes_finite = scene_box_flags(s);
  // YAHR_NO_NESTED=1 / YAHR_NO_LEAF_CULL=1: A/B measurements and tests (results are identical either way)
  if (std::getenv("YAHR_NO_NESTED") && std::atoi(std::getenv("YAHR_NO_NESTED")) != 0) d.boxes_finite &= ~4;
  if (std::getenv("YAHR_NO_LEAF_CULL") && std::atoi(std::getenv("YAHR_NO_LEAF_CULL")) != 0) d.boxes_finite &= ~2;
  // (test hook: claim nestedness unchecked -- tests/test_gpu_parity.py shows with it that the check is what keeps the
  // octant run off trees it would walk wrongly)
  if ((d.boxes_finite & 1) && std::getenv("YAHR_DEBUG_ASSUME_NESTED") && std::atoi(std::getenv("YAHR_DEBUG_ASSUME_NESTED")) != 0) d.boxes_finite |= 4;
  // Is the wavefront form a candidate at all?  Then ask the scene how coherent its rays are: a probe render (four strips
  // of 8 rows of the scene camera's frame, 1 spp, packet walker, counted build) yields the lane utilisation of the packet
  // walker's node steps = box tests / (64 x wavefront node steps).  Triangle soups -- where compaction between bounces
  // pays -- measure 0.05-0.20, tessellated meshes 0.45-0.80 (tools/coherence_probe.py; 16 rows or the whole frame give the
  // same figure), and on meshes the packet walker beats the wavefront form 2-3x at every size and sample count tried
  // (profiles/r03_mesh_sweep.csv: 2.2 M triangles x 16 spp 6.1 ms against 16.1), so AUTO keeps the wavefront form for
  // scenes below 0.3.  Costs one small launch and a counter read-back here, where the scene is uploaded anyway.
  const double ms_policy = ms_since(t_phase);
  t_phase = now();
  if (c->auto_stream_min != (size_t)-1 && samples_hint >= c->auto_stream_min && c->width > 0 && c->height > 0 &&
      !std::getenv("YAHR_NO_COHERENCE_PROBE")) {
    const int strip = std::min(8, c->height);
    DevMem probe;
    HIP_TRY(probe.alloc(sizeof(float) * 3 * (size_t)strip * (size_t)c->width));
    yahr_render_opts po; yahr_default_opts(&po);
    po.traverse = YAHR_TRAVERSE_PACKET; po.collect_stats = 1;
    for (int k = 0; k < 4; ++k) {
      const int r0 = std::min(c->height - strip, (int)((int64_t)c->height * (2 * k + 1) / 8) / 8 * 8);
      if ((rc = yahr_ctx_render(c, r0 < 0 ? 0 : r0, (r0 < 0 ? 0 : r0) + strip, &po, probe.as<float>(), nullptr))) return rc;
    }
    yahr_stats ps;
    rc = yahr_ctx_stats(c, &ps);                                // (synchronises; an unknown material met by the probe is not an error yet)
    if (rc && rc != YAHR_E_MATERIAL) return rc;
    c->timed = false;
    c->probe_lanes = ps.wave_node_steps ? (double)ps.box_tests / (64.0 * (double)ps.wave_node_steps) : 1.0;
    if (c->probe_lanes >= 0.3) { c->auto_stream_min = (size_t)-1; c->auto_mode = TRAV_PACKET; }
  }
  if (timing)
    std::fprintf(stderr, "yahr-hip timing: yahr_ctx_create: validate %.1f ms, device query %.1f ms, allocate + upload %.1f ms (%.1f MB), "
                         "policy scan %.1f ms, coherence probe %.1f ms\n", ms_check, ms_device, ms_upload,
                 ((double)s->n_nodes * sizeof(yahr_node) + (double)s->n_prims * (sizeof(yahr_prim) + sizeof(yahr_prim_normals))) / 1e6,
                 ms_policy, ms_since(t_phase));
  *out = own.release();
  return YAHR_OK;
}

int yahr_ctx_create(const yahr_scene *s, int device, yahr_ctx **out) { return ctx_create_impl(s, device, out, (size_t)-1); }

void yahr_ctx_destroy(yahr_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipFree(c->d_nodes); (void)hipFree(c->d_prims); (void)hipFree(c->d_normals); (void)hipFree(c->d_mats);
  (void)hipFree(c->d_lights); (void)hipFree(c->d_ctr); (void)hipFree(c->d_windows); (void)hipFree(c->d_cold); (void)hipFree(c->d_pairs);
  (void)hipFree(c->d_cams); (void)hipFree(c->d_row_order); (void)hipFree(c->d_spill);
  for (int k = 0; k < YD_PASS_STREAMS_MAX; ++k) {
    (void)hipFree(c->d_stream[k]); (void)hipFree(c->d_qcount[k]);
    if (c->pass_stream[k]) (void)hipStreamDestroy(c->pass_stream[k]);
    if (c->ev_join[k]) (void)hipEventDestroy(c->ev_join[k]);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  delete c;
}

// Dispatch position -> tile row, centre rows first (small table, built on first use of a row count; like
// ensure_windows it allocates, so warm a configuration up once before capturing it into a graph).
static int ensure_row_order(yahr_ctx *c, int n_rows) {
  if (c->d_row_order && c->row_order_n == n_rows) return YAHR_OK;
  std::vector<int> order((size_t)n_rows);
  for (int r = 0; r < n_rows; ++r) order[(size_t)r] = r;
  const double mid = 0.5 * (n_rows - 1);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::fabs(a - mid) < std::fabs(b - mid); });
  if (c->d_row_order) { HIP_TRY(hipFree(c->d_row_order)); c->d_row_order = nullptr; }
  HIP_TRY(dev_alloc((void **)&c->d_row_order, sizeof(int) * (size_t)(n_rows + 1)));
  HIP_TRY(hipMemcpy(c->d_row_order, order.data(), sizeof(int) * (size_t)n_rows, hipMemcpyHostToDevice));
  c->row_order_n = n_rows;
  return YAHR_OK;
}

// Prepare the reference's squareBatches windows on the device (outside any capture).
static int ensure_windows(yahr_ctx *c, int n_batches) {
  if (c->d_windows && c->n_windows == n_batches && c->windows_w == c->width && c->windows_h == c->height) return YAHR_OK;
  // one wavefront per 64-pixel chunk of a window: {x0,y0,x1,y1}, {first pixel, batch, 0, 0}
  std::vector<int4> win;
  for (int b = 0; b < n_batches; ++b) {
    int32_t w4[4]; yahr_batch_window(c->width, c->height, b, n_batches, w4);
    const int64_t npx = (int64_t)(w4[2] - w4[0]) * (w4[3] - w4[1]);
    for (int64_t p0 = 0; p0 < npx; p0 += 64) {
      win.push_back(make_int4(w4[0], w4[1], w4[2], w4[3]));
      win.push_back(make_int4((int)p0, b, 0, 0));
    }
  }
  if (c->d_windows) { HIP_TRY(hipFree(c->d_windows)); c->d_windows = nullptr; }
  HIP_TRY(dev_alloc((void **)&c->d_windows, sizeof(int4) * (win.size() + 2)));
  if (!win.empty()) HIP_TRY(hipMemcpy(c->d_windows, win.data(), sizeof(int4) * win.size(), hipMemcpyHostToDevice));
  c->n_windows = n_batches; c->n_chunks = (int)(win.size() / 2); c->windows_w = c->width; c->windows_h = c->height;
  return YAHR_OK;
}

// ---- wavefront ("stream") form ------------------------------------------------------------------------
// bytes per sample of a pass: two ping-pong path queues of 64 B, (w_k, D_k) of every level, the level count, and the
// Morton cell + sorted position of a queue entry (4 B each); the cell histogram of the sort comes on top once
static size_t stream_bytes_per_sample(int depth) { return 2 * 64 + (size_t)depth * 32 + 2 + 12; }
static size_t stream_fixed_bytes(int depth) { return sizeof(unsigned) * YD_SORT_CELLS * (size_t)(depth + 2) + 4096; }

// The image is processed in BANDS of whole tile rows so that the queues of one pass stay below a budget: an eighth of
// the device memory that is free (never more than 16 GiB; what this context already holds for the purpose counts as
// free), at least one tile row.  Returns the rows per band (a multiple of tile_h), 0 if even one tile row does not fit.
static int stream_band_rows(yahr_ctx *c, int n_local_rows, int tile_h, int spp, int depth) {
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) return 0;
  size_t held = 0;
  for (int k = 0; k < YD_PASS_STREAMS_MAX; ++k) held += c->stream_bytes[k];
  size_t budget = (fr + held) / 8;
  const size_t cap = (size_t)16 << 30;
  if (budget > cap) budget = cap;
  if (budget < c->stream_bytes[0]) budget = c->stream_bytes[0];
  if (const char *e = std::getenv("YAHR_STREAM_BUDGET_KB")) {               // tests: force several bands on a small image
    const long v = std::atol(e);
    if (v > 0) budget = (size_t)v << 10;
  }
  const size_t per_tile_row = (size_t)tile_h * (size_t)c->width * (size_t)spp * stream_bytes_per_sample(depth);
  if (per_tile_row == 0 || per_tile_row > budget) return 0;              // (the cell counts, stream_fixed_bytes, come on top)
  size_t k = budget / per_tile_row;
  const size_t max_samples = 0x7fffffffULL / ((size_t)tile_h * (size_t)c->width * (size_t)spp);      // sample ids are 31-bit
  if (k > max_samples) k = max_samples;
  if (k == 0) return 0;
  const size_t total = ((size_t)n_local_rows + (size_t)tile_h - 1) / (size_t)tile_h;
  if (k > total) k = total;
  return (int)(k * (size_t)tile_h);
}

// like ensure_windows: allocates on the first use of a configuration (warm it up once before a graph capture)
static int ensure_stream(yahr_ctx *c, int set, size_t n_samples, int depth, SParams &sp) {
  const size_t need = n_samples * stream_bytes_per_sample(depth) + stream_fixed_bytes(depth);
  if (need > c->stream_bytes[set]) {
    if (c->d_stream[set]) { HIP_TRY(hipFree(c->d_stream[set])); c->d_stream[set] = nullptr; c->stream_bytes[set] = 0; }
    HIP_TRY(dev_alloc(&c->d_stream[set], need));
    c->stream_bytes[set] = need;
  }
  if (depth + 1 > c->qcount_n[set]) {
    if (c->d_qcount[set]) { HIP_TRY(hipFree(c->d_qcount[set])); c->d_qcount[set] = nullptr; c->qcount_n[set] = 0; }
    HIP_TRY(dev_alloc((void **)&c->d_qcount[set], sizeof(unsigned) * (size_t)(depth + 1)));
    c->qcount_n[set] = depth + 1;
  }
  char *p = (char *)c->d_stream[set];
  for (int s = 0; s < 2; ++s) {
    sp.qa[s] = (float4 *)p; p += 16 * n_samples;
    sp.qb[s] = (float4 *)p; p += 16 * n_samples;
    sp.qc[s] = (float4 *)p; p += 16 * n_samples;
    sp.qd[s] = (float4 *)p; p += 16 * n_samples;
  }
  sp.lev = (float4 *)p; p += 32 * (size_t)depth * n_samples;
  sp.cell = (uint2 *)p; p += 8 * n_samples;
  sp.perm = (unsigned *)p; p += 4 * n_samples;
  sp.hist = (unsigned *)p; p += sizeof(unsigned) * YD_SORT_CELLS * (size_t)(depth + 1);
  sp.start = (unsigned *)p; p += sizeof(unsigned) * YD_SORT_CELLS;
  sp.levels = (unsigned short *)p;
  sp.count = c->d_qcount[set];
  sp.capacity = (unsigned)n_samples;
  return YAHR_OK;
}

// the two internal streams of the wavefront form and the events that fork them off the caller's stream and join them back
static int ensure_pass_streams(yahr_ctx *c, int n) {
  for (int k = 0; k < n; ++k) {
    if (!c->pass_stream[k]) HIP_TRY(hipStreamCreateWithFlags(&c->pass_stream[k], hipStreamNonBlocking));
    if (!c->ev_join[k]) HIP_TRY(hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming));
  }
  if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  return YAHR_OK;
}

int yahr_ctx_render(yahr_ctx *c, int32_t row0, int32_t row1, const yahr_render_opts *opts_in, float *d_rgb, void *hip_stream) {
  if (!c) { set_error("null argument"); return YAHR_E_ARG; }
  yahr_render_opts o; if (opts_in) o = *opts_in; else yahr_default_opts(&o);
  if (o.spp < 1) { set_error("spp must be >= 1"); return YAHR_E_ARG; }
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(hipSetDevice(c->device));
  RParams rp{};
  rp.width = c->width; rp.height = c->height;
  rp.spp = o.spp; rp.seed = o.seed;
  { int n = (int)std::floor(std::sqrt((double)o.spp) + 0.5); rp.n_strata = (n * n == o.spp) ? n : 1; }
  // Measured on the pseudo-bunny: giving each XCD a contiguous band of tile rows loses 7-20 % (blocks go to
  // XCDs round-robin, statically, and image regions differ in cost by orders of magnitude), so the natural
  // tile order -- neighbouring tiles on different XCDs, perfectly balanced -- is the default.
  rp.xcd_swizzle = (o.tuning & 1) ? 1 : 0;
  // regroup rays between traversals (workgroups of 4 waves).  Diagnostics: bit-identical output, but measured
  // SLOWER everywhere (1M triangles adaptive 9.4 -> 11.6 ms, pseudo-bunny packet 0.51 -> 0.56): the walkers are
  // bound by gather latency, not by idle lanes, and the workgroup barriers make 4 waves wait for the slowest
  const bool regroup = (o.tuning & YAHR_TUNE_REGROUP) != 0;
  int n_blocks = 0;
  if (o.tiles == YAHR_TILES_REFERENCE) {
    if (o.n_shards > 1 || row0 != 0 || row1 != c->height) { set_error("reference tiles render the whole image only"); return YAHR_E_ARG; }
    int nb = o.n_batches > 0 ? o.n_batches : yahr_n_batches(1, c->width, c->height);
    int rc = ensure_windows(c, nb); if (rc) return rc;
    rp.tile_mode = 1; rp.windows = c->d_windows; rp.n_local_rows = c->height; rp.strip_rows = c->height > 0 ? c->height : 1; rp.n_shards = 1;
    rp.n_tiles = c->n_chunks;
    n_blocks = regroup ? (c->n_chunks + 3) / 4 : c->n_chunks;
  } else {
    if (o.n_shards > 1) {
      if (o.shard < 0 || o.shard >= o.n_shards || o.strip_rows <= 0) { set_error("bad shard spec"); return YAHR_E_ARG; }
      rp.n_shards = o.n_shards; rp.strip_rows = o.strip_rows; rp.row0 = o.shard * o.strip_rows;
      rp.n_local_rows = yahr_shard_rows(c->height, o.n_shards, o.shard, o.strip_rows);
    } else {
      if (row0 < 0 || row1 > c->height || row0 > row1) { set_error("bad row range"); return YAHR_E_ARG; }
      rp.n_shards = 1; rp.row0 = row0; rp.n_local_rows = row1 - row0;
      rp.strip_rows = rp.n_local_rows > 0 ? rp.n_local_rows : 1;
    }
    rp.tile_mode = 0;
    // samples in lanes: with spp a power of two the 64 lanes of a wave hold min(spp, 64) samples of each of
    // 64 / min(spp, 64) neighbouring pixels (same pixels, same samples, same sum order -- only which lane traces what)
    rp.sil_log2 = 0; rp.pw_log2 = 3;
    if (o.spp >= 2 && (o.spp & (o.spp - 1)) == 0 && !regroup && (o.tuning & YAHR_TUNE_ONE_PIXEL_PER_LANE) == 0) {
      int sw = o.spp < 64 ? o.spp : 64;
      while ((1 << rp.sil_log2) < sw) ++rp.sil_log2;
      const int p_log2 = 6 - rp.sil_log2;              // pixels per wave = 2^p_log2, block 2^pw wide, 2^(p-pw) high
      rp.pw_log2 = (p_log2 + 1) / 2;
    }
    const int tile_w = 1 << rp.pw_log2, tile_h = (64 >> rp.sil_log2) >> rp.pw_log2;
    rp.tiles_x = (c->width + tile_w - 1) / tile_w;
    const int tile_rows = (rp.n_local_rows + tile_h - 1) / tile_h;
    rp.n_tiles = rp.tiles_x * tile_rows;
    n_blocks = ((rp.n_tiles + 7) / 8) * 8;     // the XCD swizzle needs a multiple of 8 blocks
    if (regroup) n_blocks = ((rp.tiles_x + 1) / 2) * ((tile_rows + 1) / 2);     // a workgroup owns 2x2 tiles
    // centre-rows-first dispatch measured 9 % SLOWER than natural order on the pseudo-bunny (the long chains
    // all start together and nothing cheap is left to pack around them): diagnostics only, tuning bit 1
    if ((o.tuning & 3) == 2 && tile_rows > 2) {
      int rc = ensure_row_order(c, tile_rows); if (rc) return rc;
      rp.row_order = c->d_row_order;
    }
  }
  if (n_blocks == 0 || c->width == 0 || rp.n_local_rows == 0) return YAHR_OK;   // nothing to render
  if (!d_rgb) { set_error("null output buffer"); return YAHR_E_ARG; }
  rp.frames = o.frames > 1 ? o.frames : 1;
  if (o.frame_rows != 0 && o.frame_rows < rp.n_local_rows) { set_error("frame_rows smaller than the rows rendered"); return YAHR_E_ARG; }
  rp.frame_stride = (size_t)(o.frame_rows ? o.frame_rows : rp.n_local_rows) * (size_t)c->width * 3;
  if (rp.frames > 1) {
    if (c->n_cams != 0 && c->n_cams < rp.frames) { set_error("fewer cameras set than frames requested"); return YAHR_E_ARG; }
    rp.cams = c->n_cams ? c->d_cams : nullptr;
  }
  const int per_frame = n_blocks;                                            // workgroups of one frame; a launch holds whole frames
  int mode = pick_mode(o.traverse, c->auto_mode);
  // AUTO on a coherent scene (the probe of yahr_ctx_create): always the packet walker.  (One sample per pixel through a
  // deep tree of thin leaves is the one case where the adaptive walker is ahead -- 1.1 M-triangle mesh, depth 20: 1.8 ms
  // against 2.1; 2.2 M: 2.5 against 3.4 -- but its downside is not worth that: the same mesh inside a cloud of loose
  // triangles takes 45 ms adaptive against 9.4 packet, the lane walker crawling through every fat leaf it meets.)
  // AUTO: the wavefront form once a launch holds enough paths to fill its level launches (auto_policy_for)
  if (o.traverse == YAHR_TRAVERSE_AUTO &&
      (size_t)rp.n_local_rows * (size_t)c->width * (size_t)(o.spp > 0 ? o.spp : 1) >= c->auto_stream_min)   // (per frame: a batch is rendered frame by frame in that form)
    mode = TRAV_STREAM;
  int band_rows = 0, tile_h = 1;
  if (mode == TRAV_STREAM) {
    // per frame of the batch and per band of rows: camera-ray launch, one launch per recursion level, resolve
    tile_h = (64 >> rp.sil_log2) >> rp.pw_log2;
    // (the per-sample level count is 16 bits wide)
    const bool can = rp.tile_mode == 0 && !regroup && rp.row_order == nullptr && !rp.xcd_swizzle && c->ds.depth <= 65535;
    if (can) band_rows = stream_band_rows(c, rp.n_local_rows, tile_h, o.spp, c->ds.depth);
    if (band_rows == 0) {
      if (o.traverse == YAHR_TRAVERSE_STREAM || o.traverse >= YAHR_TRAVERSE_STACK) {
        set_error("stream traversal needs wave tiles, no tuning bits, and device memory for the queues of one tile row");
        return YAHR_E_ARG;
      }
      mode = c->auto_mode;                                                 // AUTO: the fused kernel
    }
  }
  // walker of the stream form's secondary rays: the context's choice, or the one the caller names
  int secondary = c->stream_secondary;
  if (o.traverse == YAHR_TRAVERSE_STACK) secondary = TRAV_STACK;
  else if (o.traverse == YAHR_TRAVERSE_HYBRID_STACK) secondary = TRAV_ADAPT_STACK;
  else if (o.traverse == YAHR_TRAVERSE_ORDERED) secondary = TRAV_ORDERED;
  else if (o.stream_walker == YAHR_TRAVERSE_LANE) secondary = TRAV_LANE;
  else if (o.stream_walker == YAHR_TRAVERSE_PACKET) secondary = TRAV_PACKET;
  else if (o.stream_walker == YAHR_TRAVERSE_HYBRID) secondary = TRAV_ADAPTIVE;
  else if (o.stream_walker == YAHR_TRAVERSE_STACK) secondary = TRAV_STACK;
  else if (o.stream_walker == YAHR_TRAVERSE_HYBRID_STACK) secondary = TRAV_ADAPT_STACK;
  else if (o.stream_walker == YAHR_TRAVERSE_ORDERED) secondary = TRAV_ORDERED;
  if (mode == TRAV_STREAM && (secondary == TRAV_STACK || secondary == TRAV_ADAPT_STACK || secondary == TRAV_ORDERED)) {
    int rc = ensure_pairs(c); if (rc) return rc;           // (first use: allocates -- warm up before a graph capture)
  }
  HIP_TRY(hipEventRecord(c->ev0, st));
  if (mode == TRAV_STREAM) {
    const int frames = rp.frames;
    const size_t stride = rp.frame_stride;
    const DCold *cams = rp.cams;
    if (c->ds.depth <= 0) {                                                // vcast 0 _ = 0: no ray is cast at all
      for (int f = 0; f < frames; ++f)
        HIP_TRY(hipMemsetAsync(d_rgb + (size_t)f * stride, 0, sizeof(float) * 3 * (size_t)rp.n_local_rows * (size_t)c->width, st));
    } else {
      const size_t band_samples = (size_t)band_rows * (size_t)c->width * (size_t)o.spp;
      const int n_bands = (rp.n_local_rows + band_rows - 1) / band_rows;
      // A pass (one band of one frame: camera-ray launch, level launches, resolve) ends in a tail of a few long wavefronts
      // on a mostly idle chip.  With more than one pass in the launch they are dealt to TWO internal streams, each with
      // queues of its own, forked off the caller's stream and joined back into it (events: legal inside a graph capture):
      // 1 M triangles x 16 spp 33.4 -> 32.2 ms per frame of a batch, 200 k triangles 9.2 -> 8.4.  Passes write disjoint
      // rows / frames and only add to the counters, so which stream runs which cannot change a result.
      int ns = YD_PASS_STREAMS;
      if (const char *e = std::getenv("YAHR_PASS_STREAMS")) ns = std::max(1, std::min(YD_PASS_STREAMS_MAX, std::atoi(e)));
      ns = std::min(ns, frames * n_bands);
      const bool two = ns >= 2;
      SParams sps[YD_PASS_STREAMS_MAX] = {};
      int rc = ensure_stream(c, 0, band_samples, c->ds.depth, sps[0]); if (rc) return rc;
      hipStream_t ps[YD_PASS_STREAMS_MAX] = {st, st, st, st};
      if (two) {
        for (int k = 1; k < ns; ++k) if ((rc = ensure_stream(c, k, band_samples, c->ds.depth, sps[k]))) return rc;
        if ((rc = ensure_pass_streams(c, ns))) return rc;
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        for (int k = 0; k < ns; ++k) { ps[k] = c->pass_stream[k]; HIP_TRY(hipStreamWaitEvent(ps[k], c->ev_fork, 0)); }
      }
      rp.frames = 1; rp.cams = nullptr; rp.frame_stride = 0;
      int pass = 0, err = YAHR_OK;
      for (int f = 0; f < frames && !err; ++f) {
        for (int lv0 = 0; lv0 < rp.n_local_rows && !err; lv0 += band_rows, ++pass) {
          const int k = two ? (pass % ns) : 0;
          SParams &q = sps[k];
          q.cam = cams ? cams + f : c->ds.cold;
          q.lv0 = lv0;
          q.band_rows = std::min(band_rows, rp.n_local_rows - lv0);
          q.tile0 = (lv0 / tile_h) * rp.tiles_x;
          const int band_tiles = ((q.band_rows + tile_h - 1) / tile_h) * rp.tiles_x;
          if (hipMemsetAsync(c->d_qcount[k], 0, sizeof(unsigned) * (size_t)(c->ds.depth + 1), ps[k]) != hipSuccess ||
              hipMemsetAsync(q.hist, 0, sizeof(unsigned) * YD_SORT_CELLS * (size_t)(c->ds.depth + 1), ps[k]) != hipSuccess ||
              yahr_dev::launch_stream_pass(c->ds, c->dp, rp, q, secondary, band_tiles,
                                           d_rgb + (size_t)f * stride + (size_t)lv0 * (size_t)c->width * 3, c->d_ctr,
                                           o.collect_stats != 0, ps[k]) != hipSuccess) {
            set_error(std::string("launching a pass of the wavefront form failed: ") + hipGetErrorString(hipGetLastError()));
            err = YAHR_E_HIP;
          }
        }
      }
      if (two)                                                             // join, also after an error: nothing stays forked
        for (int k = 0; k < ns; ++k)
          if (hipEventRecord(c->ev_join[k], ps[k]) != hipSuccess || hipStreamWaitEvent(st, c->ev_join[k], 0) != hipSuccess) {
            if (!err) { set_error("joining the pass streams failed"); err = YAHR_E_HIP; }
          }
      if (err) return err;
    }
  } else {
    // recursionDepth is an unbounded Int upstream (Integrators.hs:18-20): the first YD_LDS_LEVELS levels of the (w_k, D_k)
    // stack live in LDS, the levels above them in a global array with one slice per workgroup of the launch.  A grid whose
    // slices would not fit the budget (an eighth of the free memory, at most 8 GiB) is cut into several launches.
    // (the ray-regrouping diagnostics build has no spill path: it keeps up to three levels in LDS, as it always did)
    rp.lds_levels = (c->ds.depth <= YD_LDS_DEPTH || (regroup && c->ds.depth <= 3)) ? c->ds.depth : YD_LDS_LEVELS;
    const size_t extra = (size_t)(c->ds.depth - rp.lds_levels);
    if (extra > 0 && regroup) { set_error("the ray-regrouping diagnostics build keeps its level stack in LDS: recursion depth <= 3 only"); return YAHR_E_ARG; }
    // the usual launch as a 3-D grid (frame, tile column, tile row): no integer division on the way to the pixel
    rp.grid3d = 0;
    if (rp.tile_mode == 0 && !regroup && !rp.xcd_swizzle && rp.row_order == nullptr && extra == 0 && rp.tiles_x > 0) {
      rp.tile_rows = rp.n_tiles / rp.tiles_x;
      if (rp.tile_rows <= 65535 && rp.tiles_x <= 65535) rp.grid3d = 1;
    }
    // A 1-D grid holds fewer than 2^32 work-items (the dispatch packet's grid size is 32 bits per dimension; the runtime
    // does not refuse a larger one, it runs modulo 2^32): 2^26 - 8 wavefront-sized workgroups, a quarter of that in the
    // regrouping build.  The 3-D grid is far from that in every dimension.  Every other launch holds as many WHOLE FRAMES
    // of the batch as fit (one 8192 x 8192 x 64 spp frame is 2^26 workgroups and refused), the chunks of a deep scene
    // are cut to it as well.
    int max_blocks = ((1 << 26) - 8) / (regroup ? 4 : 1);
    if (const char *e = std::getenv("YAHR_MAX_GRID_BLOCKS")) { const int v = std::atoi(e); if (v > 0) max_blocks = std::min(max_blocks, v); }   // tests
    const int frames = rp.frames;
    int fpl = frames;                                                        // frames per launch
    if (!rp.grid3d) {
      // (a deep scene's launches are cut into chunks below: whole frames only have to keep the block index in 31 bits)
      const int cap = extra > 0 ? 0x7fffffff : max_blocks;
      if (per_frame > cap) { set_error("image too large for one launch: more than 2^32 samples in a frame"); return YAHR_E_ARG; }
      fpl = std::min(frames, cap / per_frame);
    }
    int chunk = (int)std::min<int64_t>((int64_t)per_frame * fpl, 0x7fffffff);          // workgroups of the largest launch
    if (extra > 0) {
      const size_t per_block = extra * 6 * 64 * sizeof(float);
      size_t fr = 0, tot = 0;
      HIP_TRY(hipMemGetInfo(&fr, &tot));
      size_t budget = std::min((fr + c->spill_bytes) / 8, (size_t)8 << 30);
      if (const char *e = std::getenv("YAHR_SPILL_BUDGET_KB")) { const long v = std::atol(e); if (v > 0) budget = (size_t)v << 10; }   // tests
      budget = std::max(budget, c->spill_bytes);
      size_t fit = budget / per_block;
      if (fit == 0) { set_error("recursion depth too large: the level stack of one workgroup exceeds the device memory budget"); return YAHR_E_NOMEM; }
      if (fit < (size_t)chunk) chunk = (int)fit;
      chunk = std::min(chunk, max_blocks);
      const size_t need = (size_t)chunk * per_block;
      if (need > c->spill_bytes) {
        if (c->d_spill) { HIP_TRY(hipFree(c->d_spill)); c->d_spill = nullptr; c->spill_bytes = 0; }
        HIP_TRY(dev_alloc(&c->d_spill, need));
        c->spill_bytes = need;
      }
      rp.lvl_spill = (float *)c->d_spill;
    }
    const DCold *cams = rp.cams;
    for (int f0 = 0; f0 < frames; f0 += fpl) {
      rp.frames = std::min(fpl, frames - f0);
      rp.cams = cams ? cams + f0 : nullptr;
      float *out = d_rgb + (size_t)f0 * rp.frame_stride;
      const int n_launch = (int)std::min<int64_t>((int64_t)per_frame * rp.frames, 0x7fffffff);      // (3-D grid: not used)
      if (extra == 0) {
        HIP_TRY(yahr_dev::launch_render(c->ds, rp, mode, regroup, n_launch, out, c->d_ctr, o.collect_stats != 0, st));
      } else {
        for (int64_t b0 = 0; b0 < n_launch; b0 += chunk) {                // (64-bit: n_launch may be close to 2^31)
          rp.block0 = (unsigned)b0;
          HIP_TRY(yahr_dev::launch_render(c->ds, rp, mode, regroup, (int)std::min<int64_t>(chunk, n_launch - b0), out, c->d_ctr, o.collect_stats != 0, st));
        }
      }
    }
  }
  HIP_TRY(hipEventRecord(c->ev1, st));
  c->timed = true;
  return YAHR_OK;
}

int yahr_ctx_set_cameras(yahr_ctx *c, int32_t n, const float *cams) {
  if (!c || n < 0 || (n && !cams)) { set_error("bad camera arguments"); return YAHR_E_ARG; }
  HIP_TRY(hipSetDevice(c->device));
  if (c->d_cams) { HIP_TRY(hipFree(c->d_cams)); c->d_cams = nullptr; }
  c->n_cams = 0;
  if (n == 0) return YAHR_OK;
  std::vector<DCold> blocks((size_t)n, c->cold_host);
  for (int32_t i = 0; i < n; ++i) {
    const float *p = cams + 12 * (size_t)i;
    yahr::Camera cam;
    cam.imW = p[0]; cam.imH = p[1]; cam.focal = p[2];
    cam.look.x = p[3]; cam.look.y = p[4]; cam.look.z = p[5];
    cam.up.x = p[6]; cam.up.y = p[7]; cam.up.z = p[8];
    cam.pos.x = p[9]; cam.pos.y = p[10]; cam.pos.z = p[11];
    if ((int)std::floor(cam.imW) != c->width || (int)std::floor(cam.imH) != c->height) {
      set_error("camera " + std::to_string(i) + ": image size differs from the scene's"); return YAHR_E_ARG;
    }
    float tf[16];
    yahr::camera_matrices(cam, tf, blocks[(size_t)i].vtf);
    camera_origin(tf, blocks[(size_t)i].origin);
    blocks[(size_t)i].focal = cam.focal;
  }
  HIP_TRY(dev_alloc((void **)&c->d_cams, sizeof(DCold) * (size_t)n));
  HIP_TRY(hipMemcpy(c->d_cams, blocks.data(), sizeof(DCold) * (size_t)n, hipMemcpyHostToDevice));
  c->n_cams = n;
  return YAHR_OK;
}

int yahr_ctx_stats(yahr_ctx *c, yahr_stats *out) {
  if (!c || !out) { set_error("null argument"); return YAHR_E_ARG; }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long h[CTR_TOTAL];
  HIP_TRY(hipMemcpy(h, c->d_ctr, sizeof h, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(c->d_ctr, 0, sizeof h));
  yahr_internal_counters_to_stats(h, out);
  if (c->timed) { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1)); out->kernel_ms = ms; }
  out->upload_ms = c->upload_ms;
  if (out->bad_material) {
    set_error("Map.!: given key is not an element in the map (an object with an unknown materialId was shaded)");
    return YAHR_E_MATERIAL;
  }
  return YAHR_OK;
}

// ---- internal entry points for abi_multi.cpp (not part of the C ABI) ---------------------------------------------
}  // extern "C"
hipError_t yahr_internal_dev_alloc(void **p, size_t bytes) { return dev_alloc(p, bytes); }
int yahr_internal_counter_words(void) { return CTR_TOTAL; }
// counters of the launches since the last collection -> pinned host memory, then reset; both ordered on `hip_stream`
// (no host synchronisation: the caller waits for an event it records behind this)
int yahr_internal_ctx_counters_async(yahr_ctx *c, unsigned long long *host_pinned, void *hip_stream) {
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(host_pinned, c->d_ctr, sizeof(unsigned long long) * CTR_TOTAL, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemsetAsync(c->d_ctr, 0, sizeof(unsigned long long) * CTR_TOTAL, st));
  return YAHR_OK;
}
void yahr_internal_counters_to_stats(const unsigned long long *h, yahr_stats *out) {
  std::memset(out, 0, sizeof *out);
  out->primary_rays = h[CTR_PRIMARY]; out->reflection_rays = h[CTR_REFLECTION]; out->shadow_rays = h[CTR_SHADOW];
  out->shaded_hits = h[CTR_SHADED]; out->bad_material = h[CTR_BAD_MAT];
  out->box_tests = h[CTR_BOX]; out->tri_tests = h[CTR_TRI]; out->sphere_tests = h[CTR_SPH];
  out->wave_node_steps = h[CTR_WAVE_NODES]; out->wave_prim_steps = h[CTR_WAVE_PRIMS];
  out->wave_stage1_steps = h[CTR_WAVE_STAGE1]; out->wave_stage2_steps = h[CTR_WAVE_STAGE2];
  out->tri_tests_culled = h[CTR_TRI_CULLED]; out->wave_cull_steps = h[CTR_WAVE_CULLS]; out->tri_cull_violations = h[CTR_CULL_VIOL];
  out->max_wave_cycles = h[CTR_MAX_WAVE_CYCLES]; out->sum_wave_cycles = h[CTR_SUM_WAVE_CYCLES];
}
extern "C" {

static int ray_query(yahr_ctx *c, int32_t n, const float *rays, int which, int traverse, int32_t *iout, float *fout) {
  if (!c || n < 0 || (n && !rays)) { set_error("bad ray query arguments"); return YAHR_E_ARG; }
  if (n == 0) return YAHR_OK;
  HIP_TRY(hipSetDevice(c->device));
  DevMem d_rays, d_i, d_f;                               // freed on every return
  HIP_TRY(d_rays.alloc(sizeof(float) * 7 * (size_t)n));
  HIP_TRY(hipMemcpy(d_rays.p, rays, sizeof(float) * 7 * (size_t)n, hipMemcpyHostToDevice));
  int mode = pick_mode(traverse, c->auto_mode);
  if (mode == TRAV_STREAM) mode = TRAV_ADAPTIVE;            // single rays: the stream form is a whole-image pipeline
  const int walker = traverse == YAHR_TRAVERSE_STACK ? TRAV_STACK : traverse == YAHR_TRAVERSE_HYBRID_STACK ? TRAV_ADAPT_STACK
                     : traverse == YAHR_TRAVERSE_ORDERED ? TRAV_ORDERED : 0;    // closest / first hit on the pair records
  if (which == 2) {
    HIP_TRY(d_f.alloc(sizeof(float) * 3 * (size_t)n));
    DevMem d_lvl;                                           // levels of the radiance stack beyond the LDS part
    const int lds_levels = std::min(c->ds.depth, YD_LDS_LEVELS);
    if (c->ds.depth > lds_levels)
      HIP_TRY(d_lvl.alloc((size_t)((n + 63) / 64) * (size_t)(c->ds.depth - lds_levels) * 6 * 64 * sizeof(float)));
    HIP_TRY(yahr_dev::launch_radiance(c->ds, mode, n, d_rays.as<float>(), d_f.as<float>(), c->d_ctr, d_lvl.as<float>(), lds_levels, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(fout, d_f.p, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
  } else {
    if (mode == TRAV_ADAPTIVE && traverse != YAHR_TRAVERSE_HYBRID) mode = TRAV_PACKET;
    if (walker) { mode = walker; int rc = ensure_pairs(c); if (rc) return rc; }
    HIP_TRY(d_i.alloc(sizeof(int) * (size_t)n));
    HIP_TRY(d_f.alloc(sizeof(float) * (size_t)n));
    HIP_TRY(yahr_dev::launch_trace(c->ds, c->dp, mode, n, d_rays.as<float>(), which == 1, d_i.as<int>(), d_f.as<float>(), c->d_ctr, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(iout, d_i.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    if (fout) HIP_TRY(hipMemcpy(fout, d_f.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  }
  return YAHR_OK;
}

int yahr_ctx_trace(yahr_ctx *c, int32_t n, const float *rays, int32_t traverse, int32_t *prim_orig, float *t) {
  if (n && !prim_orig) { set_error("null output"); return YAHR_E_ARG; }
  return ray_query(c, n, rays, 0, traverse, prim_orig, t);
}
int yahr_ctx_occluded(yahr_ctx *c, int32_t n, const float *rays, int32_t traverse, int32_t *blocked) {
  if (n && !blocked) { set_error("null output"); return YAHR_E_ARG; }
  int rc = ray_query(c, n, rays, 1, traverse, blocked, nullptr);
  if (rc == YAHR_OK) for (int32_t i = 0; i < n; ++i) blocked[i] = blocked[i] >= 0 ? 1 : 0;
  return rc;
}
int yahr_ctx_radiance(yahr_ctx *c, int32_t n, const float *rays, int32_t traverse, float *rgb) {
  if (n && !rgb) { set_error("null output"); return YAHR_E_ARG; }
  return ray_query(c, n, rays, 2, traverse, nullptr, rgb);
}
int yahr_ctx_camera_rays(yahr_ctx *c, int32_t n, const float *uv, float *rays) {
  if (!c || n < 0 || (n && (!uv || !rays))) { set_error("bad arguments"); return YAHR_E_ARG; }
  if (n == 0) return YAHR_OK;
  HIP_TRY(hipSetDevice(c->device));
  DevMem d_uv, d_rays;
  HIP_TRY(d_uv.alloc(sizeof(float) * 2 * (size_t)n));
  HIP_TRY(d_rays.alloc(sizeof(float) * 7 * (size_t)n));
  HIP_TRY(hipMemcpy(d_uv.p, uv, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(yahr_dev::launch_camera_rays(c->ds, n, d_uv.as<float>(), d_rays.as<float>(), nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(rays, d_rays.p, sizeof(float) * 7 * (size_t)n, hipMemcpyDeviceToHost));
  return YAHR_OK;
}

int yahr_debug_powf(int device, int32_t n, const float *x, const float *y, float *out) {
  if (n < 0 || (n && (!x || !y || !out))) { set_error("bad arguments"); return YAHR_E_ARG; }
  if (n == 0) return YAHR_OK;
  HIP_TRY(hipSetDevice(device));
  DevMem d_x, d_y, d_o;
  HIP_TRY(d_x.alloc(sizeof(float) * (size_t)n));
  HIP_TRY(d_y.alloc(sizeof(float) * (size_t)n));
  HIP_TRY(d_o.alloc(sizeof(float) * (size_t)n));
  HIP_TRY(hipMemcpy(d_x.p, x, sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_y.p, y, sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(yahr_dev::launch_debug_powf(n, d_x.as<float>(), d_y.as<float>(), d_o.as<float>(), nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, d_o.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  return YAHR_OK;
}

int yahr_debug_powf_host(int32_t n, const float *x, const float *y, float *out) {
  if (n < 0 || (n && (!x || !y || !out))) { set_error("bad arguments"); return YAHR_E_ARG; }
  yahr_dev::debug_powf_host(n, x, y, out);
  return YAHR_OK;
}

int yahr_render_rows(const yahr_scene *scene, int32_t row0, int32_t row1, int device,
                     const yahr_render_opts *opts, float *rgb_out, yahr_stats *stats) {
  if (!rgb_out) { set_error("null output"); return YAHR_E_ARG; }
  if (!scene) { set_error("null scene"); return YAHR_E_ARG; }
  yahr_ctx *c = nullptr;
  yahr_render_opts o; if (opts) o = *opts; else yahr_default_opts(&o);
  int32_t rows = (o.n_shards > 1) ? yahr_shard_rows(scene->height, o.n_shards, o.shard, o.strip_rows) : (row1 - row0);
  // one-shot: this is the only launch the context will ever see, so the coherence probe (a counted launch: ~20 ms, most of
  // it the scratch arena such a build needs) runs only if AUTO could take the wavefront form for it at all
  const size_t samples = (o.traverse == YAHR_TRAVERSE_AUTO && rows > 0 && scene->width > 0)
                             ? (size_t)rows * (size_t)scene->width * (size_t)(o.spp > 0 ? o.spp : 1) : 0;
  int rc = ctx_create_impl(scene, device, &c, samples);
  if (rc) return rc;
  size_t bytes = sizeof(float) * 3 * (size_t)(rows > 0 ? rows : 0) * (size_t)scene->width;
  float *d_rgb = nullptr;
  auto fail = [&](int code) { if (d_rgb) (void)hipFree(d_rgb); yahr_ctx_destroy(c); return code; };
  if (bytes) {
    if (dev_alloc((void **)&d_rgb, bytes) != hipSuccess) { set_error("hipMalloc(image) failed"); return fail(YAHR_E_NOMEM); }
    if (hipMemset(d_rgb, 0, bytes) != hipSuccess) { set_error("hipMemset failed"); return fail(YAHR_E_HIP); }
    rc = yahr_ctx_render(c, row0, row1, &o, d_rgb, nullptr);
    if (rc) return fail(rc);
  }
  yahr_stats st; rc = yahr_ctx_stats(c, &st);
  if (stats) *stats = st;
  if (rc) return fail(rc);
  if (bytes && hipMemcpy(rgb_out, d_rgb, bytes, hipMemcpyDeviceToHost) != hipSuccess) { set_error("hipMemcpy(image) failed"); return fail(YAHR_E_HIP); }
  return fail(YAHR_OK);
}

}  // extern "C"
