// bvh_build.hip -- the reference's midpoint BVH (Culling.hs:27-60) built on the GPU, level by level, with the IDENTICAL
// result: the node array (DFS order, skip links, boxes) and the leaf order equal host_scene.cpp's bit for bit.
//
// Why it can be identical.  buildTree is a top-down recursion whose every decision is a pure function of the node's
// item list: bounds of the centroids (min / max folds: exact, order matters only for the sign of a zero, which neither
// `maxDimension` nor `centroid <= mid` sees), split axis, midpoint, a STABLE partition.  A stable partition of a
// contiguous range is a prefix sum over its flags, so all nodes of one tree level are split at once:
//   per level:  k_bounds    segmented min / max of the centroids of every node that can still be split (order-preserving
//                           integer keys + atomics, reduced per workgroup / wavefront first when those hold one node)
//               k_decide    maxDimension (Vectors.hs:55-66), mid = centroid of the bounds (AABBs.hs:48-49)
//               k_scan_*    exclusive prefix sum of the flags `centroid_d <= mid` over ALL items (three small kernels)
//               k_children  nl = flags inside the node; nl == n -> multiLeaf (Culling.hs:36-37), else two child nodes
//               k_scatter   every item to its place in the next level's order (left part first, both parts in order)
//   afterwards: leaf boxes (join of the primitives' boxes in leaf order, Haskell's min / max: on a tie the FIRST wins a
//               min and the LAST a max -- that is what decides the sign of a zero, and it is reproduced), bottom-up
//               subtree sizes and interior boxes (join left right, Culling.hs:39), top-down DFS indices, emit.
// Inputs must be NaN-free and finite (the host checks and otherwise builds on the CPU).  A partition that puts nothing on
// the left (only possible when halving a subnormal rounds) aborts the device build, and the host builds instead.
//
// The surface-area heuristic (sahSplit, Culling.hs:62-112; round 4) fits the same scheme: it only replaces the FLAG of a
// node with more than four items -- `bucketId <= split` instead of `centroid <= mid`:
//   k_decide       also hands the node a bucket slot and keeps centMin / centMax of the split axis
//   k_sah_init     16 x (count, box) per slot: 0 and AABBs.empty
//   k_sah_buckets  every item into its bucket: count + 1, box joined (atomics on order-preserving keys; a workgroup that
//                  holds one node collects in LDS first)
//   k_sah_pick     wrappedArea, the 15 costs from prefix / suffix joins in the reference's fold order, minimumBy
//                  (with its NaN behaviour: compare is GT for unordered operands) -- one thread per node
// Bucket boxes come out of atomics instead of an ordered fold, so they may differ from the host's in the SIGN OF A ZERO
// (Haskell's min keeps the first of two equal values, max the last).  No decision can see that: a zero extent times a
// count is a zero of either sign, sums and 0.125 + x absorb it, 0 / wrappedArea is a NaN or a zero whatever the sign,
// and wrappedArea itself is a sum that starts at +0.  The boxes that END UP in the tree are formed afterwards, in leaf
// order, exactly as for the midpoint tree.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.hpp"

#pragma clang fp contract(off)

namespace {

#define BT 1024                      /* threads per workgroup of the item kernels */

struct BNode {                       // a node while the tree grows, in creation (level) order
  int first, count, left, level;     // left = -1: leaf; else children are left, left + 1
  int cb[6];                         // bounds of the centroids as orderable keys: min x y z, max x y z
  int dim; float mid; int nl;
  float lo[3], hi[3];                // box (leaves: from the primitives; interior: join of the children)
  int size, idx;                     // nodes of the subtree; DFS index
  int slot, best; float cmin, cmax;  // surface-area heuristic: bucket slot (-1: midpoint split), chosen split, centMin / centMax
};

__device__ __forceinline__ int fkey(float f) { int k = __float_as_int(f); return k ^ ((k >> 31) & 0x7fffffff); }
__device__ __forceinline__ float fkey_inv(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
__device__ __forceinline__ float hmin(float x, float y) { return (x <= y) ? x : y; }      // Haskell's class-default min / max
__device__ __forceinline__ float hmax(float x, float y) { return (x <= y) ? y : x; }

__global__ void k_items(int n, const float *__restrict__ boxes, float4 *__restrict__ items, int *__restrict__ seg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *b = boxes + 6 * (size_t)i;
  // AABBs.centroid: 0.5 @* min + 0.5 @* max  (AABBs.hs:48-49)
  items[i] = make_float4(0.5f * b[0] + 0.5f * b[3], 0.5f * b[1] + 0.5f * b[4], 0.5f * b[2] + 0.5f * b[5], __int_as_float(i));
  seg[i] = 0;
}

__device__ __forceinline__ void node_reset(BNode &nd, int first, int count, int level) {
  nd.first = first; nd.count = count; nd.left = -1; nd.level = level; nd.dim = -1; nd.mid = 0.0f; nd.nl = 0;
  nd.cb[0] = nd.cb[1] = nd.cb[2] = 0x7fffffff; nd.cb[3] = nd.cb[4] = nd.cb[5] = (int)0x80000000;
  nd.size = 1; nd.idx = 0;
  nd.slot = -1; nd.best = 0; nd.cmin = nd.cmax = 0.0f;
}
__global__ void k_root(BNode *nodes, int n, int *n_nodes) { node_reset(nodes[0], 0, n, 0); *n_nodes = 1; }

// is this node split at `level`?  (Culling.hs:31-35: n == 0 / n == 1 / depth 0 are leaves)
__device__ __forceinline__ bool splittable(const BNode &nd, int level) { return nd.level == level && nd.count >= 2; }

__global__ __launch_bounds__(BT) void k_bounds(int n, int level, const float4 *__restrict__ items, const int *__restrict__ seg,
                                               BNode *nodes) {
  __shared__ int sh[16][6];
  __shared__ int sh_same;
  const int i = blockIdx.x * BT + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool valid = i < n;
  const int nd = valid ? seg[i] : -1;
  const bool act = valid && splittable(nodes[nd], level);
  int k[6];
  if (act) {
    const float4 c = items[i];
    k[0] = k[3] = fkey(c.x); k[1] = k[4] = fkey(c.y); k[2] = k[5] = fkey(c.z);
  } else {
    k[0] = k[1] = k[2] = 0x7fffffff; k[3] = k[4] = k[5] = (int)0x80000000;
  }
  // one node per wavefront?  (segments are contiguous item ranges: near the root nearly always)
  const int nd0 = __shfl(nd, 0, 64);
  const bool wave_same = __all(nd == nd0 || !valid) != 0;
  if (threadIdx.x == 0) sh_same = 1;
  __syncthreads();
  const int block_nd = seg[min((int)(blockIdx.x * BT), n - 1)];
  if (!(wave_same && nd0 == block_nd) && lane == 0) sh_same = 0;       // (benign race: every writer writes 0)
  if (wave_same) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int o = __shfl_xor(k[j], off, 64);
        k[j] = j < 3 ? min(k[j], o) : max(k[j], o);
      }
    if (lane == 0)
#pragma unroll
      for (int j = 0; j < 6; ++j) sh[wave][j] = k[j];
  }
  __syncthreads();
  if (sh_same) {                                                        // the whole workgroup holds one node: 6 atomics
    if (threadIdx.x < 6 && splittable(nodes[block_nd], level)) {
      const int j = threadIdx.x;
      int v = sh[0][j];
      for (int w = 1; w < BT / 64; ++w) v = j < 3 ? min(v, sh[w][j]) : max(v, sh[w][j]);
      if (j < 3) atomicMin(&nodes[block_nd].cb[j], v); else atomicMax(&nodes[block_nd].cb[j], v);
    }
  } else if (wave_same) {
    if (lane < 6 && nd0 >= 0 && splittable(nodes[nd0], level)) {
      if (lane < 3) atomicMin(&nodes[nd0].cb[lane], sh[wave][lane]); else atomicMax(&nodes[nd0].cb[lane], sh[wave][lane]);
    }
  } else if (act) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { atomicMin(&nodes[nd].cb[j], k[j]); atomicMax(&nodes[nd].cb[j + 3], k[j + 3]); }
  }
}

__global__ void k_decide(int n0, int n1, int level, BNode *nodes, int sah, int *n_slots) {
  const int j = n0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n1) return;
  BNode &nd = nodes[j];
  if (!splittable(nd, level)) return;
  const float lo[3] = {fkey_inv(nd.cb[0]), fkey_inv(nd.cb[1]), fkey_inv(nd.cb[2])};
  const float hi[3] = {fkey_inv(nd.cb[3]), fkey_inv(nd.cb[4]), fkey_inv(nd.cb[5])};
  const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
  const int d = (ex > ey && ex > ez) ? 0 : (ey > ez ? 1 : 2);            // maxDimension, Vectors.hs:55-66
  nd.dim = d;
  nd.mid = 0.5f * lo[d] + 0.5f * hi[d];                                   // dim d (centroid bounds)
  if (sah && nd.count > 4) {                                              // `length xs > 4` (Culling.hs:110), else midpointSplit
    nd.cmin = lo[d]; nd.cmax = hi[d];
    nd.slot = atomicAdd(n_slots, 1);
  }
}

// ---- sahSplit (Culling.hs:62-112) ----------------------------------------------------------------------------------
#define NBK 16
#define SLOT_WORDS (NBK * 7)         /* per bucket: count, box as keys (min x y z, max x y z) */
__device__ __forceinline__ int bucket_id(float c, float cmin, float cmax) {          // Culling.hs:69-79
  const float length = cmax - cmin;
  const float dd = c - cmin;
  const float frac = dd / length;
  const float v = 16.0f * frac;
  // GHC's floor :: Float -> Int of a NaN or an infinity is minBound, then `max 0` -> 0
  int fl = (v != v || fabsf(v) == __builtin_huge_valf()) ? 0 : (int)fminf(fmaxf(floorf(v), -1.0f), 16.0f);
  if (fl < 0) fl = 0;
  if (fl > NBK - 1) fl = NBK - 1;
  return fl;
}
__device__ __forceinline__ float comp(float4 c, int d) { return d == 0 ? c.x : (d == 1 ? c.y : c.z); }

__global__ void k_sah_init(const int *__restrict__ n_slots, int *__restrict__ slots) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)*n_slots * SLOT_WORDS) return;
  const int w = (int)(t % 7);
  slots[t] = w == 0 ? 0 : (w <= 3 ? fkey(__builtin_huge_valf()) : fkey(-__builtin_huge_valf()));   // count 0, AABBs.empty
}

__global__ __launch_bounds__(BT) void k_sah_buckets(int n, int level, const float4 *__restrict__ items, const int *__restrict__ seg,
                                                    const BNode *__restrict__ nodes, const float *__restrict__ boxes,
                                                    int *__restrict__ slots) {
  __shared__ int sh[SLOT_WORDS];
  __shared__ int sh_same;
  const int i = blockIdx.x * BT + threadIdx.x;
  const bool valid = i < n;
  const int j = valid ? seg[i] : -1;
  const bool act = valid && splittable(nodes[j], level) && nodes[j].slot >= 0;
  // a workgroup whose items all belong to ONE node (near the root: nearly always) collects in LDS and flushes once
  const int block_nd = seg[min((int)(blockIdx.x * BT), n - 1)];
  if (threadIdx.x == 0) sh_same = 1;
  for (int w = threadIdx.x; w < SLOT_WORDS; w += BT) { const int k = w % 7; sh[w] = k == 0 ? 0 : (k <= 3 ? fkey(__builtin_huge_valf()) : fkey(-__builtin_huge_valf())); }
  __syncthreads();
  if (valid && j != block_nd) sh_same = 0;                                // (benign race: every writer writes 0)
  __syncthreads();
  const bool same = sh_same != 0;
  if (act) {
    const BNode &nd = nodes[j];
    const float4 c = items[i];
    const int b = bucket_id(comp(c, nd.dim), nd.cmin, nd.cmax);
    const float *bx = boxes + 6 * (size_t)__float_as_int(c.w);
    int *dst = same ? sh + 7 * b : slots + (size_t)nd.slot * SLOT_WORDS + 7 * b;
    atomicAdd(dst, 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) { atomicMin(dst + 1 + k, fkey(bx[k])); atomicMax(dst + 4 + k, fkey(bx[3 + k])); }
  }
  if (same) {
    __syncthreads();
    const BNode &nd = nodes[block_nd];
    if (splittable(nd, level) && nd.slot >= 0)
      for (int w = threadIdx.x; w < SLOT_WORDS; w += BT) {
        const int k = w % 7, v = sh[w];
        int *dst = slots + (size_t)nd.slot * SLOT_WORDS + w;
        if (k == 0) { if (v) atomicAdd(dst, v); } else if (k <= 3) atomicMin(dst, v); else atomicMax(dst, v);
      }
  }
}

struct BBox { float lo[3], hi[3]; };
__device__ __forceinline__ BBox bb_empty() { const float inf = __builtin_huge_valf(); return BBox{{inf, inf, inf}, {-inf, -inf, -inf}}; }
__device__ __forceinline__ BBox bb_join(BBox a, BBox b) {                 // AABBs.hs:25-27
  BBox r;
#pragma unroll
  for (int k = 0; k < 3; ++k) { r.lo[k] = hmin(a.lo[k], b.lo[k]); r.hi[k] = hmax(a.hi[k], b.hi[k]); }
  return r;
}
__device__ __forceinline__ float bb_surf(BBox b) {                         // AABBs.hs:51-53
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return 2.0f * ((dx * dy + dx * dz) + dy * dz);
}
__global__ void k_sah_pick(int n0, int n1, int level, BNode *nodes, const int *__restrict__ slots) {
  const int j = n0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n1) return;
  BNode &nd = nodes[j];
  if (!splittable(nd, level) || nd.slot < 0) return;
  const int *sl = slots + (size_t)nd.slot * SLOT_WORDS;
  int cnt[NBK]; BBox bbs[NBK];
  for (int b = 0; b < NBK; ++b) {
    cnt[b] = sl[7 * b];
    for (int k = 0; k < 3; ++k) { bbs[b].lo[k] = fkey_inv(sl[7 * b + 1 + k]); bbs[b].hi[k] = fkey_inv(sl[7 * b + 4 + k]); }
  }
  float wrapped = 0.0f;                                                    // foldl (\x (_, bb) -> x + surf bb) 0 buckets
  for (int b = 0; b < NBK; ++b) wrapped = wrapped + bb_surf(bbs[b]);
  int best_i = 0; float best_c = 0.0f;
  for (int i = 0; i <= NBK - 2; ++i) {                                     // costOfSplitAt i, splitIds = [0 .. nBuckets - 2]
    BBox b0 = bb_empty(), b1 = bb_empty(); int c0 = 0, c1 = 0;
    for (int k = 0; k <= i; ++k) { b0 = bb_join(b0, bbs[k]); c0 += cnt[k]; }
    for (int k = i + 1; k < NBK; ++k) { b1 = bb_join(b1, bbs[k]); c1 += cnt[k]; }
    const float cost = 0.125f + ((float)c0 * bb_surf(b0) + (float)c1 * bb_surf(b1)) / wrapped;
    // minimumBy (compare `on` snd): keep x unless compare x y == GT, and compare is GT when x > y or the two are unordered
    if (i == 0) { best_i = 0; best_c = cost; }
    else if (!(best_c < cost) && !(best_c == cost)) { best_i = i; best_c = cost; }
  }
  nd.best = best_i;
}

__device__ __forceinline__ unsigned flag_of(const BNode &nd, int level, float4 c) {
  if (!splittable(nd, level)) return 0u;
  const float v = nd.dim == 0 ? c.x : (nd.dim == 1 ? c.y : c.z);
  if (nd.slot >= 0) return bucket_id(v, nd.cmin, nd.cmax) <= nd.best ? 1u : 0u;      // partition ((<= split) . bucketId), Culling.hs:111
  return v <= nd.mid ? 1u : 0u;                                           // midpointSplit: `<=` goes left (Culling.hs:56-60)
}

// exclusive prefix sum of the flags, first step: inside every workgroup, and the workgroup's total
__global__ __launch_bounds__(BT) void k_scan_local(int n, int level, const float4 *__restrict__ items, const int *__restrict__ seg,
                                                   const BNode *__restrict__ nodes, unsigned *__restrict__ local,
                                                   unsigned *__restrict__ totals) {
  __shared__ unsigned wsum[BT / 64];
  const int i = blockIdx.x * BT + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned f = i < n ? flag_of(nodes[seg[i]], level, items[i]) : 0u;
  const unsigned long long m = __ballot(f != 0u);
  const unsigned before = (unsigned)__popcll(m & ((1ULL << lane) - 1ULL));
  if (lane == 0) wsum[wave] = (unsigned)__popcll(m);
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < BT / 64; ++w) { base += w < wave ? wsum[w] : 0u; tot += wsum[w]; }
  if (i < n) local[i] = base + before;
  if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}
// second step: exclusive sums of the workgroup totals (one workgroup; in place); totals[nb] = grand total
__global__ __launch_bounds__(BT) void k_scan_tops(int nb, unsigned *totals) {
  __shared__ unsigned wsum[BT / 64];
  __shared__ unsigned carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += BT) {
    const int b = b0 + threadIdx.x;
    const unsigned v = b < nb ? totals[b] : 0u;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned o = (unsigned)__shfl_up((int)inc, off, 64); if (lane >= off) inc += o; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned base = carry;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) base += w < wave ? wsum[w] : 0u;
    if (b < nb) totals[b] = base + inc - v;
    __syncthreads();
    if (threadIdx.x == BT - 1) carry = base + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[nb] = carry;
}
__device__ __forceinline__ unsigned scan_at(int i, int n, const unsigned *local, const unsigned *totals, int nb) {
  return i >= n ? totals[nb] : local[i] + totals[i / BT];
}

__global__ void k_children(int n0, int n1, int level, int n, int nb, BNode *nodes, const unsigned *__restrict__ local,
                           const unsigned *__restrict__ totals, int *n_nodes, int *failed) {
  const int j = n0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n1) return;
  BNode &nd = nodes[j];
  if (!splittable(nd, level)) return;
  const int nl = (int)(scan_at(nd.first + nd.count, n, local, totals, nb) - scan_at(nd.first, n, local, totals, nb));
  nd.nl = nl;
  if (nl == nd.count) return;                                            // null xs2: multiLeaf (Culling.hs:36-37)
  if (nl == 0) { *failed = 1; return; }                                  // cannot happen with exact halves; the host decides
  const int c = atomicAdd(n_nodes, 2);
  nd.left = c;
  node_reset(nodes[c], nd.first, nl, level + 1);
  node_reset(nodes[c + 1], nd.first + nl, nd.count - nl, level + 1);
}

__global__ __launch_bounds__(BT) void k_scatter(int n, int nb, int level, const float4 *__restrict__ items, const int *__restrict__ seg,
                                                const BNode *__restrict__ nodes, const unsigned *__restrict__ local,
                                                const unsigned *__restrict__ totals, float4 *__restrict__ items_out,
                                                int *__restrict__ seg_out) {
  const int i = blockIdx.x * BT + threadIdx.x;
  if (i >= n) return;
  const int j = seg[i];
  const BNode &nd = nodes[j];
  const float4 c = items[i];
  int pos = i, s = j;
  if (splittable(nd, level) && nd.left >= 0) {
    const unsigned f = flag_of(nd, level, c);
    const int r = (int)(scan_at(i, n, local, totals, nb) - scan_at(nd.first, n, local, totals, nb));    // left items before i
    pos = f ? nd.first + r : nd.first + nd.nl + (i - nd.first - r);
    s = f ? nd.left : nd.left + 1;
  }
  items_out[pos] = c; seg_out[pos] = s;
}

// box of a leaf: foldl join empty over its primitives' boxes in leaf order (Culling.hs:52-54, AABBs.hs:24-27)
__global__ void k_leaf_boxes(int n_nodes, BNode *nodes, const float4 *__restrict__ items, const float *__restrict__ boxes) {
  // one wavefront per node: lanes fold contiguous runs in order, then the runs are joined in lane order (the join of two
  // folds is the fold of the concatenation: min keeps its left operand on a tie, max its right one).  A grid-stride loop:
  // one wavefront per node in one launch would be 64 work-items per node, and a grid holds fewer than 2^32 of them (the
  // trees of 34 M primitives and more have over 2^26 nodes).
  // (64-bit counter: up to 2^31 nodes plus one stride of up to 2^22 does not fit an int)
  const int lane = threadIdx.x & 63;
  const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long jj = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); jj < (long long)n_nodes; jj += stride) {      // (wave-uniform)
    const int j = (int)jj;
    BNode &nd = nodes[j];
    if (nd.left >= 0) continue;
    const float inf = __builtin_huge_valf();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    const int per = (nd.count + 63) / 64;
    for (int t = lane * per, e = min(nd.count, (lane + 1) * per); t < e; ++t) {
      const int id = __float_as_int(items[nd.first + t].w);
      const float *b = boxes + 6 * (size_t)id;
#pragma unroll
      for (int k = 0; k < 3; ++k) { lo[k] = hmin(lo[k], b[k]); hi[k] = hmax(hi[k], b[3 + k]); }
    }
    if (nd.count > 1) {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float ol = __shfl_down(lo[k], off, 64), oh = __shfl_down(hi[k], off, 64);
          if (lane + off < 64) { lo[k] = hmin(lo[k], ol); hi[k] = hmax(hi[k], oh); }
        }
    }
    if (lane == 0)
#pragma unroll
      for (int k = 0; k < 3; ++k) { nd.lo[k] = lo[k]; nd.hi[k] = hi[k]; }
  }
}

__global__ void k_bottom_up(int n0, int n1, BNode *nodes) {
  const int j = n0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n1) return;
  BNode &nd = nodes[j];
  if (nd.left < 0) { nd.size = 1; return; }
  const BNode &a = nodes[nd.left], &b = nodes[nd.left + 1];
  nd.size = 1 + a.size + b.size;
#pragma unroll
  for (int k = 0; k < 3; ++k) { nd.lo[k] = hmin(a.lo[k], b.lo[k]); nd.hi[k] = hmax(a.hi[k], b.hi[k]); }   // join b1 b2, Culling.hs:39
}
__global__ void k_top_down(int n0, int n1, BNode *nodes) {
  const int j = n0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n1) return;
  const BNode &nd = nodes[j];
  if (nd.left < 0) return;
  nodes[nd.left].idx = nd.idx + 1;                                       // the left child follows its parent
  nodes[nd.left + 1].idx = nd.idx + 1 + nodes[nd.left].size;             // the right one follows the left subtree
}
__global__ void k_emit(int n_nodes, const BNode *__restrict__ nodes, yahr_node *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_nodes) return;
  const BNode &nd = nodes[j];
  yahr_node o;
  o.lo[0] = nd.lo[0]; o.lo[1] = nd.lo[1]; o.lo[2] = nd.lo[2];
  o.hi[0] = nd.hi[0]; o.hi[1] = nd.hi[1]; o.hi[2] = nd.hi[2];
  if (nd.left < 0) { o.a = nd.count; o.b = nd.first; }                   // leaf: count, first primitive (leaf order)
  else { o.a = nd.idx + nd.size; o.b = -1; }                             // interior: skip link
  out[nd.idx] = o;
}
__global__ void k_order(int n, const float4 *__restrict__ items, int *__restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) order[i] = __float_as_int(items[i].w);
}

struct Dev {                                     // device allocations of one build, freed on scope exit
  std::vector<void *> p;
  ~Dev() { for (void *q : p) (void)hipFree(q); }
  template <class T> bool alloc(T **out, size_t count) {
    void *q = nullptr;
    if (hipMalloc(&q, sizeof(T) * (count ? count : 1)) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return false; }
    p.push_back(q); *out = (T *)q; return true;
  }
};

}  // namespace

namespace yahr {

// boxes: n x {lo xyz, hi xyz}, finite.  On success fills `nodes` (DFS order) and `order` (leaf order: primitive ids) and
// returns true; returns false (with the reason in *why) when the device build cannot be used -- nothing is written then.
bool device_build_bvh(const float *boxes, int32_t n, int max_depth, int split, std::vector<yahr_node> &nodes, std::vector<int32_t> &order,
                      double *device_ms, std::string *why) {
  auto fail = [&](const std::string &m) { if (why) *why = m; return false; };
  if (n < 2 || max_depth < 0) return fail("nothing to split");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return fail("no HIP device"); }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail("hipGetDevice failed");
  Dev mem;
  float *d_boxes; float4 *it[2]; int *sg[2]; BNode *d_nodes; unsigned *d_local, *d_tot; int *d_cnt; yahr_node *d_out; int *d_order;
  int *d_slots = nullptr;
  const bool sah = split == YAHR_SPLIT_SAH;
  const size_t cap_slots = (size_t)n / 5 + 1;                            // the nodes of one level with more than four items
  const int nb = (n + BT - 1) / BT;
  const size_t cap = 2 * (size_t)n;                                      // every split makes two non-empty children: < 2 n nodes
  if (!mem.alloc(&d_boxes, 6 * (size_t)n) || !mem.alloc(&it[0], (size_t)n) || !mem.alloc(&it[1], (size_t)n) ||
      !mem.alloc(&sg[0], (size_t)n) || !mem.alloc(&sg[1], (size_t)n) || !mem.alloc(&d_nodes, cap) ||
      !mem.alloc(&d_local, (size_t)n) || !mem.alloc(&d_tot, (size_t)nb + 1) || !mem.alloc(&d_cnt, 3) ||
      !mem.alloc(&d_out, cap) || !mem.alloc(&d_order, (size_t)n) || (sah && !mem.alloc(&d_slots, cap_slots * SLOT_WORDS)))
    return fail("out of device memory");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct Ev { hipEvent_t &a, &b; ~Ev() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } evs{e0, e1};
#define BV(expr) do { if ((expr) != hipSuccess) { (void)hipGetLastError(); return fail(std::string(#expr) + " failed"); } } while (0)
  // (every launch is checked by itself: a later successful launch must not be able to hide a refused one)
#define LAUNCH(...) do { hipLaunchKernelGGL(__VA_ARGS__); BV(hipGetLastError()); } while (0)
  BV(hipEventCreate(&e0)); BV(hipEventCreate(&e1));
  BV(hipMemcpy(d_boxes, boxes, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice));
  BV(hipMemset(d_cnt, 0, sizeof(int) * 3));
  BV(hipEventRecord(e0, nullptr));
  LAUNCH(k_items, dim3(nb), dim3(BT), 0, nullptr, n, d_boxes, it[0], sg[0]);
  LAUNCH(k_root, dim3(1), dim3(1), 0, nullptr, d_nodes, n, d_cnt);
  std::vector<int> lvl_start{0, 1};                                      // node ids of level l: [lvl_start[l], lvl_start[l + 1])
  int cur = 0;
  for (int level = 0; level < max_depth; ++level) {
    const int n0 = lvl_start[(size_t)level], n1 = lvl_start[(size_t)level + 1];
    if (n1 == n0) break;
    const int gb = (n1 - n0 + 255) / 256;
    LAUNCH(k_bounds, dim3(nb), dim3(BT), 0, nullptr, n, level, it[cur], sg[cur], d_nodes);
    if (sah) BV(hipMemsetAsync(d_cnt + 2, 0, sizeof(int), nullptr));
    LAUNCH(k_decide, dim3(gb), dim3(256), 0, nullptr, n0, n1, level, d_nodes, sah ? 1 : 0, d_cnt + 2);
    if (sah) {
      // at most min(nodes of the level, n / 5 + 1) slots; the kernels read the count on the device
      const size_t max_slots = std::min<size_t>((size_t)(n1 - n0), cap_slots);
      LAUNCH(k_sah_init, dim3((unsigned)((max_slots * SLOT_WORDS + 255) / 256)), dim3(256), 0, nullptr, d_cnt + 2, d_slots);
      LAUNCH(k_sah_buckets, dim3(nb), dim3(BT), 0, nullptr, n, level, it[cur], sg[cur], d_nodes, d_boxes, d_slots);
      LAUNCH(k_sah_pick, dim3(gb), dim3(256), 0, nullptr, n0, n1, level, d_nodes, d_slots);
    }
    LAUNCH(k_scan_local, dim3(nb), dim3(BT), 0, nullptr, n, level, it[cur], sg[cur], d_nodes, d_local, d_tot);
    LAUNCH(k_scan_tops, dim3(1), dim3(BT), 0, nullptr, nb, d_tot);
    LAUNCH(k_children, dim3(gb), dim3(256), 0, nullptr, n0, n1, level, n, nb, d_nodes, d_local, d_tot, d_cnt, d_cnt + 1);
    LAUNCH(k_scatter, dim3(nb), dim3(BT), 0, nullptr, n, nb, level, it[cur], sg[cur], d_nodes, d_local, d_tot, it[cur ^ 1], sg[cur ^ 1]);
    cur ^= 1;
    int h[2];
    BV(hipMemcpy(h, d_cnt, sizeof h, hipMemcpyDeviceToHost));            // nodes so far; did a partition misbehave?
    if (h[1]) return fail("a partition put nothing on the left (subnormal midpoint): host build");
    if ((size_t)h[0] > cap) return fail("node capacity exceeded");
    lvl_start.push_back(h[0]);
    if (h[0] == n1) break;                                               // nothing was split on this level
  }
  const int total = lvl_start.back();
  size_t leaf_blocks = std::min<size_t>(((size_t)total + 3) / 4, (size_t)1 << 20);      // 4 wavefronts per workgroup, grid-stride beyond 4 M nodes
  if (const char *e = std::getenv("YAHR_BVH_LEAF_BLOCKS")) { const long v = std::atol(e); if (v > 0) leaf_blocks = std::min(leaf_blocks, (size_t)v); }   // tests: the stride loop on small trees
  LAUNCH(k_leaf_boxes, dim3((unsigned)leaf_blocks), dim3(256), 0, nullptr, total, d_nodes, it[cur], d_boxes);
  for (int l = (int)lvl_start.size() - 2; l >= 0; --l) {
    const int n0 = lvl_start[(size_t)l], n1 = lvl_start[(size_t)l + 1];
    if (n1 > n0) LAUNCH(k_bottom_up, dim3((n1 - n0 + 255) / 256), dim3(256), 0, nullptr, n0, n1, d_nodes);
  }
  for (size_t l = 0; l + 1 < lvl_start.size(); ++l) {
    const int n0 = lvl_start[l], n1 = lvl_start[l + 1];
    if (n1 > n0) LAUNCH(k_top_down, dim3((n1 - n0 + 255) / 256), dim3(256), 0, nullptr, n0, n1, d_nodes);
  }
  LAUNCH(k_emit, dim3((total + 255) / 256), dim3(256), 0, nullptr, total, d_nodes, d_out);
  LAUNCH(k_order, dim3(nb), dim3(BT), 0, nullptr, n, it[cur], d_order);
  BV(hipGetLastError());
  BV(hipEventRecord(e1, nullptr));
  std::vector<yahr_node> out_nodes((size_t)total);
  std::vector<int32_t> out_order((size_t)n);
  BV(hipMemcpy(out_nodes.data(), d_out, sizeof(yahr_node) * (size_t)total, hipMemcpyDeviceToHost));
  BV(hipMemcpy(out_order.data(), d_order, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
  float ms = 0;
  if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && device_ms) *device_ms = ms;
#undef LAUNCH
#undef BV
  nodes.swap(out_nodes);
  order.swap(out_order);
  return true;
}

}  // namespace yahr
