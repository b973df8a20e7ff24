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
//                    the current subtree.  Preorder indices only grow, so a lane is active iff wake<=i.
//   * stack walker:  (wavefront form only, opt-in / AUTO on fat-leaf trees) every lane on its own over child-PAIR
//                    records with an LDS stack of pending right siblings; still the reference's order -- and
//                    an opt-in variant that visits the nearer child first (stack_loop)
// Equal-t ties, t<=tMax acceptance, back-face culling and NaN behaviour therefore match the oracle
// bit for bit; shadow rays stop at the first accepted primitive (same Bool as Rays.hs:49-54).
//
// Kernels: k_render<MODE, STATS, WPB, DEEP> (fused: one wavefront renders its tile's pixels to the end; DEEP = recursion
// depth > 2, upper levels of the radiance stack in global memory), and the wavefront ("stream") form k_stream_primary /
// k_stream_level / k_stream_resolve with the counting sort k_sort_* in between (paths compacted between bounces).
//
// What bounds the fused kernel (rocprofv3 PMC, profiles/): not bytes -- the whole scene sits in L2/MALL and a
// frame moves ~43 MB over HBM, 25 of them the image -- but vector-instruction issue and, for a lone frame, the serial
// chains of the heaviest wavefronts: a tile over dense geometry walks thousands of dependent node/primitive steps, each
// of which waits for a scalar-load round trip.  So the code is shaped for FEW INSTRUCTIONS PER STEP:
//   * packets whose rays share a direction octant walk from node to node in hand-written assembly (oct_run: 21 VALU per
//     step instead of 29), fat multiLeafs are settled per packet (leaf_culled), and the state of the radiance recursion
//     waits in LDS, not in registers, across the traversals (radiance) -- round 4;
//   * each walker is instantiated once per kernel (radiance is a wave-uniform state machine with a single
//     traversal call site; first-hit mode is a run-time flag; one pixel loop for all tilings / sample counts);
//   * the fast build carries no counters at all (ray and test counts come from the collect_stats build);
//   * scene fields used only between traversals live behind one pointer (DCold), not in argument SGPRs;
//   * record arrays are padded by one record, so prefetching "the next record" needs no clamp;
//   * the slab test runs branch-free for all lanes and is masked afterwards;
//   * the triangle test is cut in two stages with a wave-uniform exit between them.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_math.hpp"
#include "render_kernel.hpp"

#pragma clang fp contract(off)

using namespace yd;

// Tuning knobs (compile time).  YAHR_WAVES: waves per SIMD the register allocator must make room for.
// The lane walker (and the adaptive walker that ends in it) waits on dependent per-lane gathers, a third of which
// miss the L2: it is latency bound and wants every wave slot (1M triangles, adaptive: 9.3 ms at 5 waves/SIMD,
// 8.2 at 6, 7.2 at 8 -- the 64-VGPR budget spills ~30 values, all outside the traversal loops).
#ifndef YAHR_WAVES_LANE
#define YAHR_WAVES_LANE 8
#endif
#ifndef YAHR_WAVES
// packet walker, 8-frame batches of the pseudo-bunny: 4/5/6 waves per SIMD are the same code (79 VGPRs) 0.432 ms
// per frame; 7 (72 VGPRs, three spills in the kernel prologue/epilogue) 0.415; 8 (64 VGPRs, spills in the loops) 0.526
#define YAHR_WAVES 7
#endif
// waves per workgroup when rays are regrouped between traversals (yahr_render_opts.tuning bit 2)
#define YAHR_REGROUP_WAVES 4

namespace {

struct Hit { int prim; float t, b1, b2; };   // prim = leaf-order record index, -1 = Nothing

// Counters exist in collect_stats builds only.  Ray counters are wave-uniform SGPR accumulators, updated
// at wave-uniform program points with the popcount of a lane mask; the per-test counters are per lane
// because they are bumped inside divergent control flow.
struct Counters {
  unsigned primary, reflection, shadow, shaded, wave_nodes, wave_prims, wave_stage1, wave_stage2, wave_culls;   // uniform
  unsigned box, tri, sph, tri_culled, cull_viol;                          // per lane
};
YD unsigned lanes(bool p) { return (unsigned)__popcll(__ballot(p)); }

YD V3 ld3(const float4 *p) { float4 v = *p; return v3(v.x, v.y, v.z); }

struct NodeRec { float4 lo, hi; };            // yahr_node as two 16-byte words
struct PrimRec { float4 r0, r1, r2; };        // yahr_prim as three 16-byte words
// record fetches use 32-bit BYTE offsets (one shift or multiply instead of 64-bit index arithmetic; the host
// refuses scenes whose node or primitive array would not fit in 4 GiB)
#ifndef YAHR_OFF32
#define YAHR_OFF32 1
#endif
// The record arrays are read through CONSTANT-address-space pointers (addrspace 4: memory that nothing writes while the
// kernel runs -- true of the scene): a load with a wave-uniform address is then a scalar load (s_load_dwordx8 / x4 into
// SGPRs) BY CONSTRUCTION.  As plain global loads they are scalar only while the compiler can prove that nothing ahead
// of them may have written memory -- a store, an atomic, a barrier or an inline-asm statement in front silently turned
// the packet walker's record fetches into per-lane gathers (2x slower, still correct; `const __restrict__` kernel
// arguments cured the first three, the octant run's assembly needed this).  With a per-lane address (lane walker) the
// same code is an ordinary vector load.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f4v *cf4p;
YD float4 ld4(const float4 *base, unsigned byte_off, int k) {
  const f4v v = ((cf4p)(unsigned long long)((const char *)base + byte_off))[k];
  return make_float4(v.x, v.y, v.z, v.w);
}
#else
YD float4 ld4(const float4 *base, unsigned byte_off, int k) { return ((const float4 *)((const char *)base + byte_off))[k]; }   // (host pass: never runs)
#endif
YD NodeRec load_node(const DScene &sc, int i) {
  NodeRec n; n.lo = ld4(sc.nodes, (unsigned)i << 5, 0); n.hi = ld4(sc.nodes, (unsigned)i << 5, 1); return n;
}
YD PrimRec load_prim(const DScene &sc, int i) {
  PrimRec p; p.r0 = ld4(sc.prims, (unsigned)i * 48u, 0); p.r1 = ld4(sc.prims, (unsigned)i * 48u, 1); p.r2 = ld4(sc.prims, (unsigned)i * 48u, 2);
  return p;
}
// wave-uniform "any lane": compare the ballot itself (HIP's __any() goes through an integer VGPR)
#if YAHR_OFF32
YD bool wany(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ULL; }
#else
YD bool wany(bool p) { return __any(p); }
#endif

// Cold scene block.  (Do NOT launder this pointer through an `asm volatile`: the statement counts as a
// possible clobber and the compiler then refuses scalar loads for every scene array -- measured.)
YD const DCold *cold_of(const DScene &sc) { return sc.cold; }

// ---- accept tests ---------------------------------------------------------------------------------
// second half of collideTriangle for a lane that passed every geometric test: interpolated normal and
// back-face cull (Shapes.hs:47-48,57); updates (hit, tmax) on accept
YD bool tri_accept(const DScene &sc, int idx, V3 u, float t, float b1, float b2, float &tmax, Hit &hit) {
  const float4 *nr = (const float4 *)((const char *)sc.normals + (unsigned)idx * 48u);
  const V3 ns = tri_shading_normal(b1, b2, ld3(nr), ld3(nr + 1), ld3(nr + 2));
  if (!(dot(u, ns) < 0.0f)) return false;
  hit.prim = idx; hit.t = t; hit.b1 = b1; hit.b2 = b2; tmax = t;
  return true;
}

// one primitive against one ray, lane walker form.  ORDERED: primitives arrive in another order than the reference's,
// so an exact tie goes to the later record instead of to the later visit (device_math.hpp: accept_ordered)
template <bool STATS, bool ORDERED = false>
YD bool test_prim_lane(const DScene &sc, int idx, const PrimRec &p, V3 o, V3 u, float &tmax, Hit &hit, Counters &c) {
  if (__float_as_int(p.r1.w) == 0) {
    if (STATS) c.tri++;
    float t, b1, b2;
    if (ORDERED) {
      if (!tri_candidate_open(xyz(p.r0), xyz(p.r1), xyz(p.r2), o, u, t, b1, b2)) return false;
      if (!accept_ordered(t, tmax, idx, hit.prim)) return false;
    } else if (!tri_candidate(xyz(p.r0), xyz(p.r1), xyz(p.r2), o, u, tmax, t, b1, b2)) return false;
    return tri_accept(sc, idx, u, t, b1, b2, tmax, hit);
  }
  if (STATS) c.sph++;
  float t;
  if (ORDERED ? !sphere_hit_ordered(xyz(p.r0), p.r1.x, o, u, tmax, idx, hit.prim, t)
              : !sphere_hit(xyz(p.r0), p.r1.x, o, u, tmax, t)) return false;
  hit.prim = idx; hit.t = t; hit.b1 = 0.0f; hit.b2 = 0.0f; tmax = t;
  return true;
}

// one (wave-uniform) primitive against the rays of the lanes with `live` set, packet walker form.
// Stage 1 runs branch-free on every lane; stage 2 only if some live lane passed stage 1.
template <bool STATS>
YD bool test_prim_packet(const DScene &sc, int idx, const PrimRec &p, bool live, V3 o, V3 u, float &tmax, Hit &hit,
                         Counters &c) {
  bool got = false;
  if (__float_as_int(p.r1.w) == 0) {                                  // wave-uniform branch
    if (STATS) c.tri += live ? 1u : 0u;
    V3 s; float det, q;
    tri_stage0(xyz(p.r0), xyz(p.r1), xyz(p.r2), o, u, s, det, q);
    // lanes that are not live count as "far" (q = inf): one select instead of AND-ing lane masks with scalar
    // instructions, and the branch is the vccz of the compare itself
    if (wany(!tri_far(live ? q : __builtin_huge_valf(), det))) {   // else: b1 provably out of range on every live lane
      float inv, b1;
      // (a division-free b2 reject ahead of the divide -- |s2.u| > 1.5 |det|, same proof as tri_far -- was measured
      // with a counter before being written: it would remove 5 % of these steps, 0.3 % of the frame's VALU work)
      if (STATS) c.wave_stage1++;
      const bool p1 = tri_stage1(det, q, inv, b1) & live;
      if (wany(p1)) {
        if (STATS) c.wave_stage2++;
        float t, b2;
        const bool p2 = tri_stage2(xyz(p.r1), xyz(p.r2), s, u, inv, b1, tmax, t, b2) & p1;
        if (p2) got = tri_accept(sc, idx, u, t, b1, b2, tmax, hit);
      }
    }
  } else {
    if (STATS) c.sph += live ? 1u : 0u;
    if (live) {
      float t;
      if (sphere_hit(xyz(p.r0), p.r1.x, o, u, tmax, t)) {
        hit.prim = idx; hit.t = t; hit.b1 = 0.0f; hit.b2 = 0.0f; tmax = t; got = true;
      }
    }
  }
  return got;
}

// ---- lane walker --------------------------------------------------------------------------------
#ifndef YD_LEAF_SEARCHERS
#define YD_LEAF_SEARCHERS 16
#endif
// continues a traversal from per-lane node index i (>= n: nothing left) with the lane's current tmax / hit
template <bool STATS, bool FAST>
YD void lane_loop(const DScene &sc, int i, bool any_hit, V3 o, V3 u, V3 inv, float tmax, Hit &hit, Counters &c) {
  const int n = sc.n_nodes;
  // "while-while": a lane whose box test hits a leaf parks the leaf (pb, pa) and idles while the other lanes
  // keep searching; the wave scans the parked leaves together once fewer than YD_LEAF_SEARCHERS lanes are
  // still searching.  Per lane the order of tests is untouched (a parked lane does nothing until its leaf has
  // been scanned, so every later box test sees the shrunken tmax).  Scanning a leaf the moment one lane finds
  // it put a second, dependent gather (the primitive record) into ~85 % of the wave's iterations.
  int pb = -1, pa = 0;
  for (;;) {
    // lane masks straight from the compares (v_cmp into SGPR pairs; a ballot of a bool costs two more VALU)
    const unsigned long long m_parked = __builtin_amdgcn_sicmp(pb, 0, 39);                       // pb >= 0
    const unsigned long long m_search = __builtin_amdgcn_sicmp(i, n, 40) & ~m_parked;            // i < n && pb < 0
    const bool searching = __builtin_amdgcn_inverse_ballot_w64(m_search);
    const int ns = __builtin_popcountll(m_search);
    const bool parked = m_parked != 0ULL;
    if (ns == 0 && !parked) break;
    if (ns >= YD_LEAF_SEARCHERS || !parked) {                   // node step (wave-uniform choice)
      if (searching) {
        const NodeRec nd = load_node(sc, i);
        const int a = __float_as_int(nd.lo.w), b = __float_as_int(nd.hi.w);
        if (STATS) c.box++;
        const bool h = FAST ? slab_hit_fast(xyz(nd.lo), xyz(nd.hi), o, inv, tmax)
                            : slab_hit(xyz(nd.lo), xyz(nd.hi), o, inv, tmax);
        if (h && b >= 0) { pb = b; pa = a; }
        i = (b >= 0 || h) ? i + 1 : a;
      }
    } else if (pb >= 0) {                                       // leaf step
      for (int k = 0; k < pa; ++k) {
        const PrimRec p = load_prim(sc, pb + k);
        if (test_prim_lane<STATS>(sc, pb + k, p, o, u, tmax, hit, c)) {
          if (any_hit) { i = n; break; }
        }
      }
      pb = -1;
    }
  }
}

template <bool STATS>
YD void traverse_lane(const DScene &sc, bool active, bool any_hit, V3 o, V3 u, float tmax, Hit &hit, Counters &c) {
  const V3 inv = v3(1.0f / u.x, 1.0f / u.y, 1.0f / u.z);
  hit.prim = -1; hit.t = 0.0f; hit.b1 = 0.0f; hit.b2 = 0.0f;
  // NaN-free slab form when the whole wave qualifies (see traverse_packet / device_math.hpp)
  const bool fast = (sc.boxes_finite & 1) != 0 && !wany(active & !slab_safe(o, inv));
  if (fast) lane_loop<STATS, true>(sc, active ? 0 : sc.n_nodes, any_hit, o, u, inv, tmax, hit, c);
  else lane_loop<STATS, false>(sc, active ? 0 : sc.n_nodes, any_hit, o, u, inv, tmax, hit, c);
}

// ---- stack walker: lane walker on child-pair records with an LDS stack of pending right siblings ----------------
// The lane walker above fetches ONE node per dependent round trip and learns only from that node where to go next.
// Here an interior node's record holds BOTH children (render_kernel.hpp: DPairs), so one round trip yields two
// box tests, and a right child that must wait for the left subtree (Culling.hs:24-25: box, left, cut ray, right) waits
// as a 4-byte entry on a per-lane stack in LDS ("LDS-staged node stacks") instead of being found again by a skip link:
//   entry >= 0        an interior child: the rank of its own pair record
//   entry <  0        a leaf child: ~(2 * rank of the parent + side) = the half record holding its box, count, first primitive
//   YD_DEAD           counting builds only: a right child whose box test is known to fail (the reference still counts it)
// EXACTNESS.  The reference tests the right child's box when it gets there, with the tMax left behind by the left
// subtree.  In the NaN-free slab form (the only one this walker runs in) that test is `near <= far_box && near <= tMax`
// with min / max exact, so it is MONOTONE in tMax: a box that fails now fails for every smaller tMax -- such a child is
// never pushed -- and a box that passes now passes later as long as tMax has not changed.  If tMax did change in
// between (a hit was accepted: the per-lane watermark `stale` = stack height at the last change; everything below it
// was pushed before), the popped child is tested again with the tMax of that moment, exactly the reference's test: a
// leaf child from its half record, an interior child from the join of its two children's boxes, which IS its box
// (Culling.hs:39; min / max do not round; the host verifies it for every node before it hands out pair records).
// So every ray performs the reference's box and primitive tests with the reference's operands in the reference's
// order; only the fetches are regrouped.  Counting builds count a box test when the reference performs it: the left
// child's at the pair, the right child's (and the root's) at the pop.
#define YD_DEAD ((int)0x80000000)
#define YD_NONE ((int)0x80000001)          /* ~0x7ffffffe: no half record has that index */
template <bool STATS, bool ORDERED>
YD void stack_loop(const DScene &sc, const float4 *pairs, int *stk, int first, int sp, bool any_hit, V3 o, V3 u, V3 inv, float tmax, Hit &hit,
                   Counters &c) {
  // stk = this lane's column of the wavefront's stack (64 ints per level); entries [0, sp) are pending and count as
  // pushed before the last change of tMax.  `first` (YD_NONE: nothing) is the entry on top of them, handed over in a
  // register so that the stack needs one level less: the node a traversal starts at -- the root (entry 0), or the node
  // at which the packet walker handed the lane over; it tests its own box like a popped stale entry.
  int stale = sp, cur = -1;                            // cur >= 0: pair to process next without a pop
  int pb = -1, pa = 0;
  for (;;) {
    const unsigned long long m_parked = __builtin_amdgcn_sicmp(pb, 0, 39);                       // pb >= 0
    int todo = cur > sp - 1 ? cur : sp - 1;                                                      // cur >= 0 || sp > 0
    todo = first != YD_NONE ? 0 : todo;                                                          // || a first entry
    const unsigned long long m_search = __builtin_amdgcn_sicmp(todo, 0, 39) & ~m_parked;
    const bool searching = __builtin_amdgcn_inverse_ballot_w64(m_search);
    const int ns = __builtin_popcountll(m_search);
    const bool parked = m_parked != 0ULL;
    if (ns == 0 && !parked) break;
    if (ns >= YD_LEAF_SEARCHERS || !parked) {                   // node step (wave-uniform choice)
      if (searching) {
        int e = cur; bool self = false;
        if (first != YD_NONE) {                                 // the node the traversal starts at
          e = first; first = YD_NONE; self = true;
          if (STATS && !ORDERED) c.box++;
        } else if (e < 0) {                                     // pop
          --sp; e = stk[sp * 64];
          self = sp < stale; stale = self ? sp : stale;
          if (STATS && !ORDERED) c.box++;                       // the reference tests this child's box now
        }
        cur = -1;
        if (!STATS || ORDERED || e != YD_DEAD) {
          const bool is_pair = e >= 0;
          const float4 *q = (const float4 *)((const char *)pairs + (is_pair ? ((unsigned)e << 6) : ((unsigned)(~e) << 5)));
          const float4 a0 = q[0], a1 = q[1];                    // left child, or the popped leaf child itself
          float4 a2 = a0, a3 = a1;
          if (is_pair) { a2 = q[2]; a3 = q[3]; }
          bool go = true;
          if (is_pair & self) {                                 // the popped node's own box = join of its children's
            if (STATS && ORDERED) c.box++;
            go = slab_hit_fast(vmin(xyz(a0), xyz(a2)), vmax(xyz(a1), xyz(a3)), o, inv, tmax);
          }
          if (go) {
            float n0, n1;
            bool h0 = slab_near_fast(xyz(a0), xyz(a1), o, inv, tmax, n0);
            bool h1 = is_pair & slab_near_fast(xyz(a2), xyz(a3), o, inv, tmax, n1);
            int c0a = __float_as_int(a0.w), c0b = __float_as_int(a1.w), c1a = __float_as_int(a2.w), c1b = __float_as_int(a3.w);
            if (STATS) c.box += ORDERED ? (is_pair ? 2u : 1u) : (is_pair ? 1u : 0u);
            if (is_pair) {
              int s1 = 1;                                       // side of the child that may have to wait
              // ORDERED: the nearer child first (the right one only if strictly nearer).  Reference order: when the
              // left child is missed the right one is next, tested with this very tMax -- no detour over the stack
              if (ORDERED ? (h1 & (!h0 | (n1 < n0))) : !h0) {
                if (STATS && !ORDERED) c.box++;
                const bool th = h0; h0 = h1; h1 = th;
                int ti = c0a; c0a = c1a; c1a = ti; ti = c0b; c0b = c1b; c1b = ti;
                s1 = 0;
              }
              if (h1) stk[64 * sp++] = c1b >= 0 ? ~(2 * e + s1) : c1a;
              else if (STATS && !ORDERED && s1 == 1) stk[64 * sp++] = YD_DEAD;
            }
            if (h0) { if (c0b >= 0) { pb = c0b; pa = c0a; } else cur = c0a; }
          }
        }
      }
    } else if (pb >= 0) {                                       // leaf step
      for (int k = 0; k < pa; ++k) {
        const PrimRec p = load_prim(sc, pb + k);
        if (test_prim_lane<STATS, ORDERED>(sc, pb + k, p, o, u, tmax, hit, c)) {
          if (any_hit) { sp = 0; cur = -1; break; }
          stale = sp;                                           // tMax changed: what is pending was pushed before
        }
      }
      pb = -1;
    }
  }
}

// Stack of the pending right siblings of DFS node j (all of them "stale": tested on the pop) and, in `first`, the entry of
// node j itself: what a lane needs to continue at node j a traversal that the packet walker began.  Walks down from the
// root through the pair records: the interior child record carries -1 - its DFS index, a leaf right child is the last
// node of its parent's subtree.  At most one trip per tree level.
YD int stack_descend(const DScene &sc, const float4 *pairs, int *stk, int j, int &first) {
  const int n = sc.n_nodes;
  first = YD_NONE;
  if (j >= n) return 0;
  if (j == 0) { first = 0; return 0; }
  int sp = 0, k = 0, p = 0, end = n;
  for (;;) {
    const float4 *q = (const float4 *)((const char *)pairs + ((unsigned)k << 6));
    const int la = __float_as_int(q[0].w), lb = __float_as_int(q[1].w), ra = __float_as_int(q[2].w), rb = __float_as_int(q[3].w);
    const int l_idx = p + 1, r_idx = rb >= 0 ? end - 1 : -1 - rb;
    const int l_ent = lb >= 0 ? ~(2 * k) : la, r_ent = rb >= 0 ? ~(2 * k + 1) : ra;
    if (j < r_idx) {
      stk[64 * sp++] = r_ent;
      if (j <= l_idx || lb >= 0) { first = l_ent; break; }
      k = la; p = l_idx; end = r_idx;
    } else {
      if (j == r_idx || rb >= 0) { first = r_ent; break; }
      k = ra; p = r_idx;
    }
  }
  return sp;
}

// can this wavefront run the stack walker?  (pair records exist, NaN-free slab form legal for every active lane)
YD bool stack_ok(const DScene &sc, const float4 *pairs, bool active, V3 o, V3 inv) {
  return pairs != nullptr && (sc.boxes_finite & 1) != 0 && !wany(active & !slab_safe(o, inv));
}

template <bool STATS, bool ORDERED>
YD void traverse_stack(const DScene &sc, const float4 *pairs, int *stk, bool active, bool any_hit, V3 o, V3 u, float tmax, Hit &hit,
                       Counters &c) {
  const V3 inv = v3(1.0f / u.x, 1.0f / u.y, 1.0f / u.z);
  hit.prim = -1; hit.t = 0.0f; hit.b1 = 0.0f; hit.b2 = 0.0f;
  if (stack_ok(sc, pairs, active, o, inv)) {
    stack_loop<STATS, ORDERED>(sc, pairs, stk, active ? 0 : YD_NONE, 0, any_hit, o, u, inv, tmax, hit, c);   // from the root's pair
  } else {
    lane_loop<STATS, false>(sc, active ? 0 : sc.n_nodes, any_hit, o, u, inv, tmax, hit, c);
  }
}

// ---- packet walker ------------------------------------------------------------------------------
#define YD_DONE 0x7fffffff
// adaptive hand-off: every YD_WINDOW node steps the wave looks at how many lane-steps were useful; below
// YD_MIN_ACTIVE (of 64) on average the rays have diverged and each lane finishes on its own (lane walker).
// Pays on thin-leaf trees (1M random triangles: packet 26.5 ms, lane 11.1, adaptive 10.6 per frame); on trees
// with fat multi-leaves the lane walker is hopeless (70 gathers per lane per leaf) and the hand-off only
// loses (pseudo-bunny 0.94 -> 1.2-1.6 ms whatever the window/threshold), so AUTO chooses per scene on the
// host: yahr_ctx_create, by the largest leaf.
#ifndef YD_WINDOW
#define YD_WINDOW 32
#endif
#ifndef YD_MIN_ACTIVE
#define YD_MIN_ACTIVE 14
#endif

// Wave-wide minimum through the DPP cross-lane path (row shifts, then the two row broadcasts of gfx9): six VALU
// instructions and one v_readlane instead of six LDS round trips (ds_bpermute).  Needs the whole wave executing,
// which holds at every call site (traversal control flow is wave-uniform and no lane ever leaves a kernel early).
YD int wave_min_i32(int v) {
  constexpr int id = 0x7fffffff;
#define YD_MIN_STEP(ctrl, rows)                                                             \
  { const int w = __builtin_amdgcn_update_dpp(id, v, ctrl, rows, 0xf, false); v = w < v ? w : v; }
  YD_MIN_STEP(0x111, 0xf)                              // row_shr:1
  YD_MIN_STEP(0x112, 0xf)                              // row_shr:2
  YD_MIN_STEP(0x114, 0xf)                              // row_shr:4
  YD_MIN_STEP(0x118, 0xf)                              // row_shr:8   -> lane 15 of every row holds its row's min
  YD_MIN_STEP(0x142, 0xa)                              // row_bcast:15 into rows 1 and 3
  YD_MIN_STEP(0x143, 0xc)                              // row_bcast:31 into rows 2 and 3 -> lane 63 holds the min
#undef YD_MIN_STEP
  return __builtin_amdgcn_readlane(v, 63);
}

// Wave-wide minimum (maximum) of six floats at once: the DPP ladder of wave_min_i32 (row shifts 1, 2, 4, 8,
// then the two row broadcasts; lane 63 ends up with the result), written as v_min_f32_dpp / v_max_f32_dpp so that the
// cross-lane move rides on the arithmetic instruction (the compiler emits v_mov_b32_dpp + s_nop + a canonicalising
// v_max + the v_min for every step of the builtin form: 24 instructions per reduction instead of 6).  A lane whose DPP
// source does not exist (first lanes of a row for a shift, rows outside the row mask of a broadcast) keeps its value:
// with bound_ctrl off such a lane's write is disabled.  The six ladders are interleaved, so every instruction reads
// a register written six instructions earlier -- beyond the two wait states a DPP read of a fresh VALU result
// needs; inline asm gets no automatic s_nop, so the statement opens with its own: the compiler may well have written
// the first operand in the instruction right before it.  v_min / v_max ignore NaNs: callers pass NaN-free values (FAST
// traversals only).  Needs the whole wave executing, like wave_min_i32.
// (two ladders of six: with all twelve in one statement the packet kernel needs six more VGPRs at its peak)
#define YD_LADDER6(op, ctl)                                                                      \
  op " %0, %0, %0 " ctl "\n\t" op " %1, %1, %1 " ctl "\n\t" op " %2, %2, %2 " ctl "\n\t"        \
  op " %3, %3, %3 " ctl "\n\t" op " %4, %4, %4 " ctl "\n\t" op " %5, %5, %5 " ctl "\n\t"
#define YD_WAVE_REDUCE6(op)                                                                      \
  asm("s_nop 1\n\t"                                                                              \
      YD_LADDER6(op, "row_shr:1 row_mask:0xf bank_mask:0xf") YD_LADDER6(op, "row_shr:2 row_mask:0xf bank_mask:0xf")      \
      YD_LADDER6(op, "row_shr:4 row_mask:0xf bank_mask:0xf") YD_LADDER6(op, "row_shr:8 row_mask:0xf bank_mask:0xf")      \
      YD_LADDER6(op, "row_bcast:15 row_mask:0xa bank_mask:0xf") YD_LADDER6(op, "row_bcast:31 row_mask:0xc bank_mask:0xf") \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5))
YD void wave_min6(float &a0, float &a1, float &a2, float &a3, float &a4, float &a5) { YD_WAVE_REDUCE6("v_min_f32_dpp"); }
YD void wave_max6(float &a0, float &a1, float &a2, float &a3, float &a4, float &a5) { YD_WAVE_REDUCE6("v_max_f32_dpp"); }
#undef YD_WAVE_REDUCE6
#undef YD_LADDER6
YD float lane63(float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63)); }

// ---- fat multiLeafs: most primitive steps proven away per PACKET ------------------------------------------------
// A multiLeaf (Culling.hs:52-54) has no boxes inside: collideAll tests every primitive against every ray that enters
// the joint box.  The reference's midpoint tree leaves fat ones wherever it runs out of depth -- the pseudo-bunny's
// 35-triangle floor leaf (joint box = the whole floor slab) is entered by three rays in four and took 90 % of the
// frame's triangle tests, each a full primitive step (28 VALU) that ends in "every live lane is tri_far".
// Here the wavefront first settles that outcome for up to 64 triangles AT ONCE, one triangle per lane: the rays that
// entered the leaf are bounded by a box in ray space (12 wave reductions), every lane runs tri_far_packet
// (device_math.hpp: tri_stage0 in interval arithmetic, ~85 VALU) on its own triangle, and a ballot yields the
// triangles that are NOT proven far for the whole packet.  Only those take the usual primitive step, in stored order.
// A skipped step is one in which the existing code would have found every live lane far and done nothing (the
// enclosure covers every ray that entered the leaf, live or not), so hits, t, tMax and every per-ray count are
// untouched; counting builds add the skipped tests to tri_tests (the reference performs them) and to tri_culled.
#ifndef YD_CULL_MIN
#define YD_CULL_MIN 12                                 // primitives in a leaf from which the packet test pays
#endif
template <bool STATS, bool ANY>
YD bool leaf_culled(const DScene &sc, int a, int b, bool h, V3 o, V3 u, float &tmax, Hit &hit, Counters &c) {
  const float inf = __builtin_huge_valf();
  // box of the rays that entered the leaf, in ray space (lanes outside contribute the identities)
  float l0 = h ? o.x : inf, l1 = h ? o.y : inf, l2 = h ? o.z : inf, l3 = h ? u.x : inf, l4 = h ? u.y : inf, l5 = h ? u.z : inf;
  wave_min6(l0, l1, l2, l3, l4, l5);
  const V3 ol = v3(lane63(l0), lane63(l1), lane63(l2)), ul = v3(lane63(l3), lane63(l4), lane63(l5));
  float g0 = h ? o.x : -inf, g1 = h ? o.y : -inf, g2 = h ? o.z : -inf, g3 = h ? u.x : -inf, g4 = h ? u.y : -inf, g5 = h ? u.z : -inf;
  wave_max6(g0, g1, g2, g3, g4, g5);
  const V3 oh = v3(lane63(g0), lane63(g1), lane63(g2)), uh = v3(lane63(g3), lane63(g4), lane63(g5));
  // magnitudes under which the interval evaluation cannot overflow (wave-uniform values)
  const float mo = fmaxf(fmaxf(fmaxf(fabsf(ol.x), fabsf(oh.x)), fmaxf(fabsf(ol.y), fabsf(oh.y))), fmaxf(fabsf(ol.z), fabsf(oh.z)));
  const float mu = fmaxf(fmaxf(fmaxf(fabsf(ul.x), fabsf(uh.x)), fmaxf(fabsf(ul.y), fabsf(uh.y))), fmaxf(fabsf(ul.z), fabsf(uh.z)));
  const bool bounded = (mo <= YD_CULL_MAX_COORD) & (mu <= YD_CULL_MAX_DIR);
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  bool got = false;
  for (int base = 0; base < a; base += 64) {                           // wave-uniform
    const int n_here = a - base < 64 ? a - base : 64;
    bool need = false;
    if (lane < n_here) {
      const PrimRec p = load_prim(sc, b + base + lane);                // per-lane record
      const bool far_all = tri_far_packet(xyz(p.r0), xyz(p.r1), xyz(p.r2), ol, oh, ul, uh);
      const bool is_tri = __float_as_int(p.r1.w) == 0;
      need = !(is_tri & bounded & far_all);
    }
    unsigned long long m = __builtin_amdgcn_ballot_w64(need);
    if (STATS) {
      c.wave_culls++;
      // counting builds CHECK the proof: every skipped test is evaluated per lane after all, and a lane that entered the
      // leaf and is not `far` is a violation (yahr_stats.tri_cull_violations: zero, or the enclosure argument is broken)
      unsigned long long skipped = ~m & (n_here < 64 ? (1ULL << n_here) - 1ULL : ~0ULL);
      while (skipped != 0ULL) {
        const int j = __builtin_ctzll(skipped);
        skipped &= skipped - 1ULL;
        const PrimRec p = load_prim(sc, b + base + j);
        V3 s; float det, q;
        tri_stage0(xyz(p.r0), xyz(p.r1), xyz(p.r2), o, u, s, det, q);
        c.cull_viol += (h & !tri_far(q, det)) ? 1u : 0u;
      }
    }
    int prev = -1;
    while (m != 0ULL) {                                                // wave-uniform
      const int j = __builtin_ctzll(m);
      m &= m - 1ULL;
      const PrimRec p = load_prim(sc, b + base + j);                   // wave-uniform index: scalar load
      const bool live = h & !(ANY & got);
      if (STATS) { const unsigned skipped = live ? (unsigned)(j - prev - 1) : 0u; c.tri += skipped; c.tri_culled += skipped; c.wave_prims++; }
      got |= test_prim_packet<STATS>(sc, b + base + j, p, live, o, u, tmax, hit, c);
      prev = j;
    }
    if (STATS) { const unsigned skipped = (h & !(ANY & got)) ? (unsigned)(n_here - prev - 1) : 0u; c.tri += skipped; c.tri_culled += skipped; }
  }
  return got;
}

// The node index i is wave-uniform, so node and primitive records arrive through SCALAR loads into SGPRs.
// The successor in array order (i+1: always the next node after a leaf, and the left child of a hit
// interior node) and a leaf's first primitive are requested before the slab test of the current node, so
// that round trip overlaps the VALU work; only a skip needs a dependent load.
// FAST selects the hardware-min/max slab test (device_math.hpp: slab_hit_fast), legal for this traversal
// when no lane can produce a NaN in it.
// Returns n when the traversal is complete, or (ADAPT only) the node index at which the wave gave up being a
// packet: lane l then continues at max(i, wake_l) with its tmax / hit (outputs), see traverse_packet.
// Box test of the packet walker for the lanes that are inside the current subtree (`on`): h = hit & on.
// NaN-free form: a lane that is not `on` gets tMax = -1, so its tfar <= -1 < 0 <= tnear and the ONE float compare
// already says "miss" -- the lane mask never has to be AND-ed in with scalar instructions, and "any lane" is the
// vccz branch of that compare (a scalar compare on a VALU-written mask stalls the wave for the write-back).
// Needs every lane NaN-free, also the ones that never were active: traverse_packet gives those a harmless ray.
template <bool FAST>
YD void packet_box(const NodeRec &nd, bool on, V3 o, V3 inv, float tmax, bool &h, bool &anyh) {
  if (FAST) {
    const unsigned long long hm = slab_mask_fast(xyz(nd.lo), xyz(nd.hi), o, inv, on ? tmax : -1.0f);
    h = __builtin_amdgcn_inverse_ballot_w64(hm);
    anyh = hm != 0ULL;
  } else {
    const bool hs = slab_hit(xyz(nd.lo), xyz(nd.hi), o, inv, tmax);
    h = hs & on;
    anyh = (__builtin_amdgcn_ballot_w64(hs) & __builtin_amdgcn_ballot_w64(on)) != 0ULL;
  }
}

// The primitives of a leaf some lane entered (h = the lanes that did): collideAll over the multiLeaf in stored order
// (Culling.hs:52-54, Rays.hs:39-46).  p0 = the record of the leaf's first primitive, requested by the caller before
// its slab test.  Returns got = the lanes that accepted a hit here.
template <bool STATS, bool FAST, bool ANY, bool CULL>
YD bool packet_leaf(const DScene &sc, int a, int b, PrimRec p0, bool h, V3 o, V3 u, float &tmax, Hit &hit, Counters &c, int cull_min) {
  constexpr bool any_hit = ANY;
  bool got = false;
  // fat leaf (FAST loops only: every lane NaN-free): most of its primitive steps are settled per packet.
  // cull_min = YD_CULL_MIN, or INT_MAX when the scene's triangles are not bounded (DScene.boxes_finite bit 1)
  if (CULL && FAST && a >= cull_min) return leaf_culled<STATS, any_hit>(sc, a, b, h, o, u, tmax, hit, c);
#ifdef YAHR_PRIO_FAT
  // a wavefront inside a fat multiLeaf is on a long serial chain: let it issue ahead of its SIMD's other waves
  if (a >= YAHR_PRIO_FAT) __builtin_amdgcn_s_setprio(3);
#endif
  // two primitives per trip, the record buffers swapped: the prefetch of the next record (padded array) never
  // has to be copied into place (12 SGPR moves per primitive otherwise)
  PrimRec p1;
  for (int k = 0; k < a; k += 2) {
    p1 = load_prim(sc, b + k + 1);
    if (STATS) c.wave_prims++;
    got |= test_prim_packet<STATS>(sc, b + k, p0, h & !(any_hit & got), o, u, tmax, hit, c);
    if (k + 1 >= a) break;
    p0 = load_prim(sc, b + k + 2);
    if (STATS) c.wave_prims++;
    got |= test_prim_packet<STATS>(sc, b + k + 1, p1, h & !(any_hit & got), o, u, tmax, hit, c);
  }
#ifdef YAHR_PRIO_FAT
  if (a >= YAHR_PRIO_FAT) __builtin_amdgcn_s_setprio(0);
#endif
  return got;
}

// One node step of the packet walker: `cur` is node i, the record of node i+1 is requested into `nxt` (padded
// array: in bounds even for the last node).  Returns the index of the next node (>= n: traversal complete).
template <bool STATS, bool FAST, bool ADAPT, bool ANY, bool CULL>
YD int packet_step(const DScene &sc, const NodeRec &cur, NodeRec &nxt, int i, V3 o, V3 u, V3 inv, float &tmax, Hit &hit,
                   int &wake, Counters &c, int &useful, int cull_min) {
  constexpr bool any_hit = ANY;                        // first-hit (shadow) mode is a compile-time property here
  const int n = sc.n_nodes;
  const int a = __float_as_int(cur.lo.w), b = __float_as_int(cur.hi.w);
  nxt = load_node(sc, i + 1);
  const bool on = wake <= i;
  if (ADAPT) useful += (int)lanes(on);
  if (STATS) { c.box += on ? 1u : 0u; c.wave_nodes++; }
  int next = i + 1;
  // ONE wave-uniform branch per node; the slab test is written out in both arms on purpose (testing the
  // leaf flag twice made the compiler carry it as a 64-bit lane mask: ~8 extra scalar instructions per step)
  if (b >= 0) {                                        // leaf
    const PrimRec p0 = load_prim(sc, b);               // first primitive, requested before the slab test
    bool h, anyh;
    packet_box<FAST>(cur, on, o, inv, tmax, h, anyh);
    if (anyh) {
      const bool got = packet_leaf<STATS, FAST, ANY, CULL>(sc, a, b, p0, h, o, u, tmax, hit, c, cull_min);
      if (any_hit) {
        if (got) wake = YD_DONE;
        if (wany(got)) {
          // next node any lane still needs; readfirstlane tells the compiler the minimum is wave-uniform
          // (otherwise the loop exit looks divergent and every node load degrades to a per-lane gather)
          const int m = __builtin_amdgcn_readfirstlane(wave_min_i32(wake));
          if (m == YD_DONE) return n;
          if (m > next) next = m;
        }
      }
    }
  } else {                                             // interior
    bool h, anyh;
    packet_box<FAST>(cur, on, o, inv, tmax, h, anyh);
    // a lane that is in this subtree and misses sleeps until the skip target.  Written as max(): a lane that
    // is NOT in the subtree went to sleep at an enclosing node j, and skip(i) <= skip(j) = its wake
    wake = h ? wake : (wake > a ? wake : a);
    next = anyh ? i + 1 : a;
  }
  return next;
}

// ---- packets whose rays share ONE direction octant: node steps without the min / max network ------------------
// bbRayIntersection (AABBs.hs:29-40) takes, per axis, hmin / hmax of the two plane distances t0 = (lo - o) * inv and
// t1 = (hi - o) * inv.  In the NaN-free regime of the FAST loops (finite lo <= hi, finite o, finite non-zero inv) IEEE
// subtraction and multiplication are monotone, so for inv > 0:  lo - o <= hi - o  and  t0 <= t1, i.e. hmin(t0, t1) IS
// t0 and hmax(t0, t1) IS t1 as values (equal values can differ in the sign of a zero only, which neither v_max3 /
// v_min3 nor the final `tnear > tfar` can see); for inv < 0 the other way round.  96.6 % of the pseudo-bunny's node
// steps (99 % of the camera and shadow rays', 95 % of the reflection rays'; tools/oct_probe.py) belong to traversals
// whose active lanes all look into the same octant: then WHICH plane is the near one is a property of the wavefront,
// the six per-axis v_min / v_max (and the canonicalisation of tMax') disappear -- 29 -> 22 VALU per node step, 17 without
// the sleep index, see below -- and the choice costs no instruction at all if the walk is assembled once per octant: the near / far operands are just other
// SGPRs of the node record.
// To keep the eight copies small they hold ONLY the walk from node to node: the run returns when some lane enters a
// leaf, the leaf's primitives are handled by code that exists once (packet_leaf), and the run is entered again.
// Mixed-octant wavefronts, the exact (non-FAST) slab form, the adaptive walker and the counting builds keep packet_loop.
//
// The run is hand-written gfx950 assembly.  The same walk in C++ (eight instantiations of a templated loop that leaves
// through its leaf hits) compiled to the 22 VALU but around them the compiler (a) sank the request for the next node
// below the slab arithmetic, so that every step waited out a scalar-load round trip, and (b) funnelled the loop's exits
// through flag registers, a dozen scalar instructions per step: 0.360 ms per pseudo-bunny frame against 0.330 without
// the octant path.  In assembly the request goes out first, and a step is 17 VALU + 6-8 scalar instructions:
//     s_waitcnt lgkmcnt(0)                 current record (requested one step earlier) is here
//     s_add / s_lshl / s_load_dwordx8      request node i + 1 into the OTHER record buffer
//     6 v_sub, 6 v_mul, v_max (0), v_min (tMax), v_max3, v_min3, v_cmp_ngt -> vcc = the lanes that enter the box
//     leaf:      vcc != 0 -> leave (index, count, first primitive, mask);  else on to i + 1
//     interior:  vcc != 0 -> i + 1 (already requested);  else request `skip`
// NO SLEEP INDEX.  packet_loop keeps, per lane, the index at which the lane wakes up again (`wake`: the skip link of the
// subtree it failed), hands sleeping lanes tMax' = -1 before every test and updates the index after it: 4 of 21 VALU in the
// first form of this run.  It is redundant when every child's box lies inside its parent's: with lo_c >= lo_p, hi_c <= hi_p,
// the same o and inv, monotone rounding gives per axis tnear_c >= tnear_p and tfar_c <= tfar_p (inv < 0: the planes swap
// roles), tMax only shrinks during a traversal (a NaN tMax = "no limit" can only be replaced by a number), so a lane with
// tnear_p > tfar_p has tnear_c >= tnear_p > tfar_p >= tfar_c: a ray that fails a node's box fails every box below it by
// itself, and fails a leaf's joint box before any of its primitives is asked.  Nestedness holds in every tree Culling.hs:39
// or this library builds (a parent's box is the join of its children's: no rounding) and is CHECKED per scene when a
// context is created, because the node array is the caller's (abi_device.cpp: DScene.boxes_finite bit 2); trees that fail
// keep packet_loop.  Lanes that take no part (inactive, or -- first-hit traversals -- finished) carry tMax = -1, which
// fails every box.  What the sleep index also gave first-hit traversals -- a jump to the next node some lane still needs
// when a lane finishes -- is gone: the nodes in between fail for every lane and leave by their skip links (~6 % more node
// steps on the pseudo-bunny, at 17 instead of 21 VALU).
// -- on the way from a node to its successor: s_waitcnt, s_add, s_lshl, s_load, s_cmp, two conditional branches that fall
// through (and one s_branch every second step).
// Two record buffers, the body written twice with their roles swapped (like packet_loop).  No bounds checks: the node
// array ends in one zeroed padding record, which reads as a leaf without primitives -- walking onto it (i = n) either
// leaves the run with i = n or steps to n + 1 = skip-free end, so the run ALSO stops at any index >= n (checked on every
// path that can reach n: after a leaf miss and after a skip).  A skip DRAINS the (now useless) request for i + 1 before it
// requests its target into the first buffer: in body B that request is in flight into the very same registers, scalar
// loads return out of order, and the stale record could land last (without the wait the walk ran off the array: a GPU
// memory fault in round 4).  The records live in fixed registers s[48:55] / s[56:63]
// (low enough to be allocatable in every kernel: at 8 waves per SIMD the top of the SGPR file is reserved)
// (an asm operand cannot name the components of a register tuple); the current record is copied in on entry, and the
// statement drains its last request before it ends, so nothing lands in a register the compiler has taken back.
// tMax needs no canonicalisation here: traverse_packet hands over a canonical tMax (a quiet NaN stays "no limit":
// v_min_f32 returns the other operand).
// IDX = the register that holds the index of the node in this body, NXI = the one that receives IDX + 1: body A walks
// node %[i] and puts i + 1 into %[nx], body B walks node %[nx] and puts nx + 1 into %[i] -- no register move on the way
// from a node to its successor, and body A FALLS THROUGH into body B.  On every exit the index of the last node is in %[i].
#define YD_OCT_STEP(L, IDX, NXI, NX, FX, NY, FY, NZ, FZ, RA, RB, NXT)                                       \
  ".Loct%=_" L ":\n\t"                                                                                       \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                 \
  "s_add_i32 " NXI ", " IDX ", 1\n\t"                                                                        \
  "s_lshl_b32 %[off], " NXI ", 5\n\t"                                                                        \
  "s_load_dwordx8 " NXT ", %[base], %[off]\n\t"                                                              \
  "v_sub_f32_e32 %[t0], " NX ", %[ox]\n\t"                                                                   \
  "v_sub_f32_e32 %[t2], " NY ", %[oy]\n\t"                                                                   \
  "v_sub_f32_e32 %[t4], " NZ ", %[oz]\n\t"                                                                   \
  "v_sub_f32_e32 %[t1], " FX ", %[ox]\n\t"                                                                   \
  "v_mul_f32_e32 %[t0], %[t0], %[ix]\n\t"                                                                    \
  "v_mul_f32_e32 %[t2], %[t2], %[iy]\n\t"                                                                    \
  "v_mul_f32_e32 %[t4], %[t4], %[iz]\n\t"                                                                    \
  "v_mul_f32_e32 %[t1], %[t1], %[ix]\n\t"                                                                    \
  "v_max_f32_e32 %[t4], 0, %[t4]\n\t"                                                                        \
  "v_max3_f32 %[t0], %[t0], %[t2], %[t4]\n\t"                                                                \
  "v_sub_f32_e32 %[t2], " FY ", %[oy]\n\t"                                                                   \
  "v_sub_f32_e32 %[t4], " FZ ", %[oz]\n\t"                                                                   \
  "v_mul_f32_e32 %[t2], %[t2], %[iy]\n\t"                                                                    \
  "v_mul_f32_e32 %[t4], %[t4], %[iz]\n\t"                                                                    \
  "v_min_f32_e32 %[t4], %[t4], %[tmax]\n\t"                                                                  \
  "v_min3_f32 %[t1], %[t1], %[t2], %[t4]\n\t"                                                                \
  "v_cmp_ngt_f32_e32 vcc, %[t0], %[t1]\n\t"                                                                  \
  "s_cmp_lt_i32 " RB ", 0\n\t"                                                                               \
  "s_cbranch_scc0 .Loct%=_" L "_leaf\n\t"                                                                    \
  "s_cbranch_vccz .Loct%=_" L "_skip\n\t"
// the rarer ways out of a step, placed behind the two bodies
#define YD_OCT_TAIL(L, IDX, NXI, RA, RB, OTHER)                                                              \
  ".Loct%=_" L "_skip:\n\t"                        /* no lane entered the interior node: on to its skip target */  \
  "s_mov_b32 %[i], " RA "\n\t"                                                                               \
  "s_cmp_ge_i32 %[i], %[n]\n\t"                                                                              \
  "s_cbranch_scc1 .Loct%=_end\n\t"                                                                           \
  "s_lshl_b32 %[off], %[i], 5\n\t"                                                                           \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                 \
  "s_load_dwordx8 s[48:55], %[base], %[off]\n\t"                                                             \
  "s_branch .Loct%=_A\n"                                                                                     \
  ".Loct%=_" L "_leaf:\n\t"                                                                                  \
  "s_cbranch_vccnz .Loct%=_" L "_hit\n\t"          /* a leaf no lane entered: on to the next node */             \
  "s_cmp_ge_i32 " NXI ", %[n]\n\t"                                                                           \
  "s_cbranch_scc0 .Loct%=_" OTHER "\n\t"                                                                     \
  "s_mov_b32 %[i], " NXI "\n\t"                                                                              \
  "s_branch .Loct%=_end\n"                                                                                   \
  ".Loct%=_" L "_hit:\n\t"                                                                                   \
  "s_mov_b32 %[la], " RA "\n\t"                                                                              \
  "s_mov_b32 %[lb], " RB "\n\t"                                                                              \
  "s_mov_b64 %[hm], vcc\n\t"                                                                                 \
  "s_mov_b32 %[i], " IDX "\n\t"                                                                              \
  "s_branch .Loct%=_end\n"
// record in s[48:55] / s[56:63]:  lo.x lo.y lo.z a hi.x hi.y hi.z b
#define YD_OCT_RUN(NXA, FXA, NYA, FYA, NZA, FZA, NXB, FXB, NYB, FYB, NZB, FZB)                                \
  asm("s_mov_b32 s48, %[r0]\n\ts_mov_b32 s49, %[r1]\n\ts_mov_b32 s50, %[r2]\n\ts_mov_b32 s51, %[r3]\n\t"     \
      "s_mov_b32 s52, %[r4]\n\ts_mov_b32 s53, %[r5]\n\ts_mov_b32 s54, %[r6]\n\ts_mov_b32 s55, %[r7]\n"       \
      YD_OCT_STEP("A", "%[i]", "%[nx]", NXA, FXA, NYA, FYA, NZA, FZA, "s51", "s55", "s[56:63]")              \
      YD_OCT_STEP("B", "%[nx]", "%[i]", NXB, FXB, NYB, FYB, NZB, FZB, "s59", "s63", "s[48:55]")              \
      "s_branch .Loct%=_A\n"                                                                                 \
      YD_OCT_TAIL("A", "%[i]", "%[nx]", "s51", "s55", "B")                                                   \
      YD_OCT_TAIL("B", "%[nx]", "%[i]", "s59", "s63", "A")                                                   \
      ".Loct%=_end:\n\t"                                                                                     \
      "s_waitcnt lgkmcnt(0)\n"                                                                               \
      : [i] "+s"(i), [la] "+s"(la), [lb] "+s"(lb), [hm] "+s"(hm), [nx] "=&s"(nx), [off] "=&s"(off),                        \
        [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t4] "=&v"(t4)                                                    \
      : [base] "s"(base), [n] "s"(n), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [ix] "v"(inv.x), [iy] "v"(inv.y),      \
        [iz] "v"(inv.z), [tmax] "v"(tmax), [r0] "s"(A.lo.x), [r1] "s"(A.lo.y), [r2] "s"(A.lo.z), [r3] "s"(A.lo.w),        \
        [r4] "s"(A.hi.x), [r5] "s"(A.hi.y), [r6] "s"(A.hi.z), [r7] "s"(A.hi.w)                                            \
      : "vcc", "scc", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61",   \
        "s62", "s63")

// Walks from node i (record `A`, arrived) until some lane enters a LEAF: returns the leaf's index, with its count and
// first primitive in la / lb and the lanes that entered it in hm -- or an index >= n when the traversal is complete.
// tMax cannot change in here (no primitive is tested).  OCT bit k set: inv component k is negative, the hi
// plane of that axis is the near one.
template <int OCT>
YD int oct_run(const float4 *base, int n, int i, const NodeRec &A, V3 o, V3 inv, float tmax, int &la, int &lb,
               unsigned long long &hm) {
  int nx, off;
  float t0, t1, t2, t4;
#define YD_N(bit, lo, hi) ((OCT & bit) ? hi : lo)
  // (the operand names must be string literals: one statement per octant)
  if (OCT == 0) YD_OCT_RUN("s48", "s52", "s49", "s53", "s50", "s54", "s56", "s60", "s57", "s61", "s58", "s62");
  if (OCT == 1) YD_OCT_RUN("s52", "s48", "s49", "s53", "s50", "s54", "s60", "s56", "s57", "s61", "s58", "s62");
  if (OCT == 2) YD_OCT_RUN("s48", "s52", "s53", "s49", "s50", "s54", "s56", "s60", "s61", "s57", "s58", "s62");
  if (OCT == 3) YD_OCT_RUN("s52", "s48", "s53", "s49", "s50", "s54", "s60", "s56", "s61", "s57", "s58", "s62");
  if (OCT == 4) YD_OCT_RUN("s48", "s52", "s49", "s53", "s54", "s50", "s56", "s60", "s57", "s61", "s62", "s58");
  if (OCT == 5) YD_OCT_RUN("s52", "s48", "s49", "s53", "s54", "s50", "s60", "s56", "s57", "s61", "s62", "s58");
  if (OCT == 6) YD_OCT_RUN("s48", "s52", "s53", "s49", "s54", "s50", "s56", "s60", "s61", "s57", "s62", "s58");
  if (OCT == 7) YD_OCT_RUN("s52", "s48", "s53", "s49", "s54", "s50", "s60", "s56", "s61", "s57", "s62", "s58");
#undef YD_N
  // (asm outputs: tell the compiler they are wave-uniform -- it takes them for per-lane values otherwise and builds the
  // whole leaf code with exec-masked loops and per-lane gathers of the primitive records.  Here, next to the statement,
  // a readfirstlane of a scalar register folds away; behind the switch over the octants it cost ten VALU per leaf visit)
  la = __builtin_amdgcn_readfirstlane(la);
  lb = __builtin_amdgcn_readfirstlane(lb);
  hm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(hm >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)hm);
  return __builtin_amdgcn_readfirstlane(i);
}

template <bool ANY, bool CULL>
YD int packet_loop_oct(const DScene &sc, int oct, bool active, V3 o, V3 u, V3 inv, float &tmax, Hit &hit, Counters &c) {
  constexpr bool any_hit = ANY;
  const int n = sc.n_nodes;
  // a lane that takes no part (any more) misses every box: tFar = min(.., -1) < 0 <= tNear
  if (!active) tmax = -1.0f;
  const int cull_min = (sc.boxes_finite & 2) ? YD_CULL_MIN : 0x7fffffff;   // wave-uniform (leaf_culled)
  int i = 0;
#ifdef YD_PRIO_VISITS
  int visits = 0;
#ifndef YD_PRIO_LEVEL
#define YD_PRIO_LEVEL 3
#endif
#endif
  NodeRec A = load_node(sc, 0);
  for (;;) {
    int a = 0, b = 0;
    unsigned long long hm = 0ULL;
    switch (oct) {                                                        // wave-uniform: one copy of the run per octant
      case 0: i = oct_run<0>(sc.nodes, n, i, A, o, inv, tmax, a, b, hm); break;
      case 1: i = oct_run<1>(sc.nodes, n, i, A, o, inv, tmax, a, b, hm); break;
      case 2: i = oct_run<2>(sc.nodes, n, i, A, o, inv, tmax, a, b, hm); break;
      case 3: i = oct_run<3>(sc.nodes, n, i, A, o, inv, tmax, a, b, hm); break;
      case 4: i = oct_run<4>(sc.nodes, n, i, A, o, inv, tmax, a, b, hm); break;
      case 5: i = oct_run<5>(sc.nodes, n, i, A, o, inv, tmax, a, b, hm); break;
      case 6: i = oct_run<6>(sc.nodes, n, i, A, o, inv, tmax, a, b, hm); break;
      default: i = oct_run<7>(sc.nodes, n, i, A, o, inv, tmax, a, b, hm); break;
    }
    if (i >= n) break;
#ifdef YD_PRIO_VISITS
    // a traversal that has visited many leaves belongs to one of the few long wavefronts that decide when a LONE frame is
    // done; while the chip is full they crawl at a seventh of a SIMD -- let them issue ahead of their neighbours
    if (++visits == YD_PRIO_VISITS) __builtin_amdgcn_s_setprio(YD_PRIO_LEVEL);
#endif
    A = load_node(sc, i + 1);                                             // the node after the leaf, requested ahead of its primitives
    const PrimRec p0 = load_prim(sc, b);
    const bool h = __builtin_amdgcn_inverse_ballot_w64(hm);
    const bool got = packet_leaf<false, true, ANY, CULL>(sc, a, b, p0, h, o, u, tmax, hit, c, cull_min);
    if (any_hit) {
      // a lane with its first hit is finished; the walk goes on from i + 1 for the others (no jump to the next node some
      // lane still needs, as packet_loop makes: the nodes in between fail for every lane and leave by their skip links)
      if (got) tmax = -1.0f;
      if (wany(got) && !wany(!(tmax < 0.0f))) break;                      // (a lane whose tMax was negative from the start cannot hit anything)
    }
    if (i + 1 >= n) break;
    i = i + 1;
  }
#ifdef YD_PRIO_VISITS
  if (visits >= YD_PRIO_VISITS) __builtin_amdgcn_s_setprio(0);
#endif
  return n;
}

// The loop body exists twice with the two node buffers swapped: a step to the next node in array order (every
// leaf, every interior node some lane hit) finds its record already in the other buffer, without copying eight
// SGPRs and without making the prefetch wait for the last use of the current node.
template <bool STATS, bool FAST, bool ADAPT, bool ANY, bool CULL>
YD int packet_loop(const DScene &sc, bool active, V3 o, V3 u, V3 inv, float &tmax, Hit &hit, int &wake, Counters &c) {
  const int n = sc.n_nodes;
  wake = active ? 0 : YD_DONE;
  int i = 0;                                           // wave-uniform: lives in an SGPR
  int window = YD_WINDOW, useful = 0;                  // wave-uniform
  const int cull_min = (sc.boxes_finite & 2) ? YD_CULL_MIN : 0x7fffffff;   // wave-uniform (leaf_culled)
  NodeRec A = load_node(sc, 0), B;
  for (;;) {
    if (ADAPT) {
      if (window == 0) { if (useful < YD_WINDOW * YD_MIN_ACTIVE) return i; window = YD_WINDOW; useful = 0; }
      --window;
    }
    int next = packet_step<STATS, FAST, ADAPT, ANY, CULL>(sc, A, B, i, o, u, inv, tmax, hit, wake, c, useful, cull_min);
    if (next >= n) return n;
    if (next != i + 1) { A = load_node(sc, next); i = next; continue; }
    i = next;
    if (ADAPT) {
      if (window == 0) { if (useful < YD_WINDOW * YD_MIN_ACTIVE) return i; window = YD_WINDOW; useful = 0; }
      --window;
    }
    next = packet_step<STATS, FAST, ADAPT, ANY, CULL>(sc, B, A, i, o, u, inv, tmax, hit, wake, c, useful, cull_min);
    if (next >= n) return n;
    if (next != i + 1) A = load_node(sc, next);
    i = next;
  }
}

// ADAPT: packet while the rays stay coherent, then every lane on its own.  Both halves perform, per ray, the
// reference's tests in the reference's order, so where the hand-off happens cannot change a result.
// STK (ADAPT only): the lanes finish on the stack walker when the wavefront may run it (`stk` = the lane's LDS stack)
template <bool STATS, bool ADAPT, bool STK = false, bool CULL = true, bool OCT = CULL>
YD void traverse_packet(const DScene &sc, bool active, bool any_hit, V3 o, V3 u, float tmax, Hit &hit, Counters &c,
                        int *stk = nullptr, const float4 *pairs = nullptr) {
  V3 inv = v3(1.0f / u.x, 1.0f / u.y, 1.0f / u.z);
  hit.prim = -1; hit.t = 0.0f; hit.b1 = 0.0f; hit.b2 = 0.0f;
  const int n = sc.n_nodes;
  if (n == 0 || !wany(active)) return;
  // every node box finite (host check) and no active lane able to form 0*inf or inf-inf in the slab test?
  const bool fast = (sc.boxes_finite & 1) != 0 && !wany(active & !slab_safe(o, inv));
  if (fast && !active) { o = vof(0.0f); inv = vof(1.0f); }     // inactive lanes: NaN-free too (packet_box)
  // tMax in canonical form (a signalling NaN becomes a quiet one: no comparison and no hmin can tell them apart).  The
  // octant run's v_min_f32 needs it -- in IEEE mode the instruction answers a signalling NaN with a NaN instead of the
  // other operand -- and from here on tMax only ever takes computed, i.e. canonical, values.
  tmax = __builtin_canonicalizef(tmax);
  int wake, i;
  // four copies of the loop (fast/exact slab x closest/first hit), selected once per traversal: run-time flags
  // inside the loop cost scalar instructions on every step
  // ... and, for packets whose active lanes share one direction octant, the loops without the min / max network
  bool oct_ok = false; int oct = 0;
#ifndef YAHR_NO_OCTANT
  // (not for tiny trees: entering the run costs ~20 scalar instructions per leaf visit -- record copy-in, octant dispatch,
  // a second request for the node after the leaf -- which only a walk of some length pays back; the 8-sphere scene of
  // BASELINE config 2, 21 nodes, lost 6 % with it)
#ifndef YD_OCT_MIN_NODES
#define YD_OCT_MIN_NODES 64
#endif
  if (OCT && !STATS && !ADAPT && fast && (sc.boxes_finite & 4) && n >= YD_OCT_MIN_NODES) {
    const unsigned long long act = __builtin_amdgcn_ballot_w64(active);
    const unsigned long long mx = __builtin_amdgcn_ballot_w64(active & (inv.x < 0.0f)), my = __builtin_amdgcn_ballot_w64(active & (inv.y < 0.0f)),
                             mz = __builtin_amdgcn_ballot_w64(active & (inv.z < 0.0f));
    oct_ok = ((mx == 0ULL) | (mx == act)) & ((my == 0ULL) | (my == act)) & ((mz == 0ULL) | (mz == act));
    oct = (mx != 0ULL ? 1 : 0) | (my != 0ULL ? 2 : 0) | (mz != 0ULL ? 4 : 0);
  }
#endif
  if (OCT && !STATS && !ADAPT && fast && oct_ok) {
    if (any_hit) i = packet_loop_oct<true, CULL>(sc, oct, active, o, u, inv, tmax, hit, c);
    else i = packet_loop_oct<false, CULL>(sc, oct, active, o, u, inv, tmax, hit, c);
  } else if (fast) {
    if (any_hit) i = packet_loop<STATS, true, ADAPT, true, CULL>(sc, active, o, u, inv, tmax, hit, wake, c);
    else i = packet_loop<STATS, true, ADAPT, false, CULL>(sc, active, o, u, inv, tmax, hit, wake, c);
  } else {
    if (any_hit) i = packet_loop<STATS, false, ADAPT, true, CULL>(sc, active, o, u, inv, tmax, hit, wake, c);
    else i = packet_loop<STATS, false, ADAPT, false, CULL>(sc, active, o, u, inv, tmax, hit, wake, c);
  }
  if (ADAPT && i < n) {
    // a lane that already has its first hit in any-hit mode is parked at YD_DONE >= n
    const int j = wake > i ? wake : i;
    if (STK && fast && pairs != nullptr) {
      // the lane's pending right siblings and node j itself, rebuilt from the pair records; all of them are tested on
      // the pop with the tMax of that moment, like every box the lane has not seen yet
      int first;
      const int sp = stack_descend(sc, pairs, stk, j, first);
      stack_loop<STATS, false>(sc, pairs, stk, first, sp, any_hit, o, u, inv, tmax, hit, c);
    } else if (fast) lane_loop<STATS, true>(sc, j, any_hit, o, u, inv, tmax, hit, c);
    else lane_loop<STATS, false>(sc, j, any_hit, o, u, inv, tmax, hit, c);
  }
}

// `coherent` (wave-uniform): the rays are camera rays.  The adaptive walker then stays a packet to the end: neighbouring
// camera rays share almost their whole path, and the hand-off to the lane walker only costs there (1 M triangles, 16 spp,
// camera rays alone: adaptive 15.6 ms, packet 11.1 ms).  Which walker runs never changes a result.
// CULL: fat multiLeafs settled per packet (leaf_culled) and the octant run for packets that share a direction octant
// (oct_run).  Off in the level launches of the wavefront form: those walk the secondary rays of triangle SOUPS (AUTO
// takes that form for incoherent scenes only), whose packets are too wide for the proof to succeed and share an octant
// in a quarter of their node steps, and at 64 VGPRs the extra code costs the kernel two dozen spilled registers.
template <int MODE, bool STATS, bool CULL = true, bool OCT = CULL>
YD void traverse(const DScene &sc, bool active, bool any_hit, bool coherent, V3 o, V3 u, float tmax, Hit &hit, Counters &c,
                 int *stk = nullptr, const float4 *pairs = nullptr) {
  if (MODE == TRAV_LANE) traverse_lane<STATS>(sc, active, any_hit, o, u, tmax, hit, c);
  else if (MODE == TRAV_STACK) traverse_stack<STATS, false>(sc, pairs, stk, active, any_hit, o, u, tmax, hit, c);
  else if (MODE == TRAV_ORDERED) traverse_stack<STATS, true>(sc, pairs, stk, active, any_hit, o, u, tmax, hit, c);
  else if (MODE == TRAV_PACKET) traverse_packet<STATS, false, false, CULL, OCT>(sc, active, any_hit, o, u, tmax, hit, c);
  else if (coherent) traverse_packet<STATS, false, false, CULL, OCT>(sc, active, any_hit, o, u, tmax, hit, c);
  else if (MODE == TRAV_ADAPT_STACK) traverse_packet<STATS, true, true, CULL>(sc, active, any_hit, o, u, tmax, hit, c, stk, pairs);
  else traverse_packet<STATS, true, false, CULL>(sc, active, any_hit, o, u, tmax, hit, c);     // TRAV_ADAPTIVE
}

// The three record arrays reach every kernel as `const T *__restrict__` KERNEL ARGUMENTS (and are bound into the
// kernel's private DScene copy): for such pointers the AMDGPU back end knows the memory is never written during
// the kernel, so a wave-uniform load from them stays a scalar load (s_load_dwordx8 into SGPRs) whatever stores,
// atomics or barriers come before it.  Reached through the pointer fields of the by-value DScene block instead,
// the same loads silently became per-lane gathers as soon as anything that may write memory preceded them.
#define YD_SCENE_PARAMS const float4 *__restrict__ a_nodes, const float4 *__restrict__ a_prims, \
                        const float4 *__restrict__ a_normals
#define YD_SCENE_BIND DScene sc = sc_in; sc.nodes = a_nodes; sc.prims = a_prims; sc.normals = a_normals;
// kernels that may run the stack walkers take the pair records the same way (render_kernel.hpp: DPairs)
#define YD_SCENE_PARAMS_PAIRS YD_SCENE_PARAMS, const float4 *__restrict__ a_pairs, int stack_levels

// ---- ray regrouping inside a workgroup of WPB waves ----------------------------------------------
// Barrier whose fences cover LDS only: a full __syncthreads() fences global memory too, which makes every later
// scene load "clobberable" and costs the packet walker its scalar loads (tests/test_isa.py).
YD void block_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int WPB>
YD bool block_any(bool p, float *xch) {
  if (WPB == 1) return wany(p);
  int *flag = (int *)(xch + 12 * 64 * WPB) + WPB;      // behind ray[7], hit[4], src[1] and the wave counts
  const bool w = wany(p);                              // (a traversal's barriers separate two calls)
  if ((threadIdx.x & 63) == 0) flag[threadIdx.x >> 6] = w ? 1 : 0;
  block_sync();
  bool r = false;
#pragma unroll
  for (int k = 0; k < WPB; ++k) r |= flag[k] != 0;
  return r;
}

// Secondary rays leave holes: only pixels whose primary ray hit something cast shadow and reflection rays (38 %
// on the 1M-triangle scene), scattered over the tile.  With WPB > 1 the workgroup's active rays are packed, in
// thread order, into the fewest waves before every traversal and the hits are handed back afterwards through
// LDS; a wave left without rays skips the traversal.  Per ray nothing changes (same tests, same order).
template <int MODE, bool STATS, int WPB>
YD void traverse_block(const DScene &sc, bool active, bool any_hit, bool coherent, V3 o, V3 u, float tmax, Hit &hit,
                       Counters &c, float *xch) {
  if (WPB == 1) { traverse<MODE, STATS>(sc, active, any_hit, coherent, o, u, tmax, hit, c); return; }
  constexpr int B = 64 * WPB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float *ray = xch, *hits = xch + 7 * B;
  int *src = (int *)(xch + 11 * B), *cnt = src + B;
  const unsigned long long m = __ballot(active);
  if (lane == 0) cnt[wave] = (int)__popcll(m);
  if (active) {
    ray[tid] = o.x; ray[B + tid] = o.y; ray[2 * B + tid] = o.z;
    ray[3 * B + tid] = u.x; ray[4 * B + tid] = u.y; ray[5 * B + tid] = u.z; ray[6 * B + tid] = tmax;
  }
  block_sync();
  int base = 0, total = 0;
#pragma unroll
  for (int k = 0; k < WPB; ++k) { const int ck = cnt[k]; base += k < wave ? ck : 0; total += ck; }
  if (active) src[base + (int)__popcll(m & ((1ULL << lane) - 1ULL))] = tid;
  block_sync();
  const bool mine = tid < total;
  int s = 0;
  V3 o2 = vof(0.0f), u2 = v3(0.0f, 0.0f, 1.0f); float t2 = 0.0f;
  if (mine) {
    s = src[tid];
    o2 = v3(ray[s], ray[B + s], ray[2 * B + s]);
    u2 = v3(ray[3 * B + s], ray[4 * B + s], ray[5 * B + s]); t2 = ray[6 * B + s];
  }
  Hit h2;
  traverse<MODE, STATS>(sc, mine, any_hit, false, o2, u2, t2, h2, c);     // packed rays of 4 tiles: not coherent
  if (mine) {
    hits[s] = __int_as_float(h2.prim); hits[B + s] = h2.t; hits[2 * B + s] = h2.b1; hits[3 * B + s] = h2.b2;
  }
  block_sync();
  hit.prim = -1; hit.t = 0.0f; hit.b1 = 0.0f; hit.b2 = 0.0f;
  if (active) {
    hit.prim = __float_as_int(hits[tid]); hit.t = hits[B + tid]; hit.b1 = hits[2 * B + tid]; hit.b2 = hits[3 * B + tid];
  }
}

// ---- shading --------------------------------------------------------------------------------------
// DifferentialGeometry of the accepted hit (Shapes.hs:23-28, 49-55); returns the material index.  `dpdu` comes back
// NORMALISED (see the end of the function).
YD int surface_of(const DScene &sc, V3 ro, V3 rd, const Hit &hit, V3 &x, V3 &n, V3 &dpdu) {
  const PrimRec p = load_prim(sc, hit.prim);
  x = ro + scale(hit.t, rd);                                          // dgPoint / rayAt
  if (__float_as_int(p.r1.w) == 0) {
    const float4 *nr = (const float4 *)((const char *)sc.normals + (unsigned)hit.prim * 48u);
    V3 ns = tri_shading_normal(hit.b1, hit.b2, ld3(nr), ld3(nr + 1), ld3(nr + 2));
    V3 ss = norm(xyz(p.r2));                                          // norm e2
    V3 ts = norm(cross(ss, ns));
    n = ns;                                                           // NOT normalised (Shapes.hs:53)
    dpdu = cross(ts, ns);
  } else {
    V3 nn = norm(x - xyz(p.r0));
    n = nn;
    dpdu = cross(nn, v3(0.0f, 0.0f, 1.0f));
  }
  // What travels on from here is sn = norm dgDPDU, the only thing the BSDF frame takes from dpdu (BSDF.hs:36-38): every
  // BSDF.at of this hit -- one per light and one for the reflection -- would form it again from the same three floats.
  dpdu = norm(dpdu);
  return __float_as_int(p.r0.w);
}

// BSDF.at (blinnPhong material) dg wi wo   (Shaders.hs:12-14, BSDF.hs:12-46); mat < 0: unknown material.
// dpdu = norm dgDPDU, as surface_of hands it out
YD V3 material_at(const DScene &sc, int mat, V3 n, V3 dpdu, V3 wi, V3 wo) {
  V3 diffuse = vof(0.0f), specular = vof(0.0f); float shininess = 0.0f;
  if (mat >= 0) {
    const float4 *mats = cold_of(sc)->mats;
    const float4 m0 = mats[2 * (size_t)mat], m1 = mats[2 * (size_t)mat + 1];
    diffuse = xyz(m0); shininess = m0.w; specular = xyz(m1);
  }
  Frame fr; fr.nn = n; fr.sn = dpdu; fr.tn = cross(fr.nn, fr.sn);      // bsdf_frame with sn = norm dpdu already formed (surface_of)
  return blinn_phong_at(diffuse, specular, shininess, fr, wi, wo);
}

// Integrators.radiance for one primary ray per lane (vcast/vhit, Integrators.hs:22-43).
// The recursion is linear, so it runs as a loop; per level k the lane stores w_k = (n.r) @* f r and
// D_k = directIllumination in LDS and the result is folded back-to-front: v = w_k * v + D_k, which is
// exactly `(n .* r) @* f r * rs + direct` evaluated in the reference's order.
//
// Wave-uniform state machine with ONE traversal call site.  step = k*(L+1) + phase:
//   phase 0      closest hit of the level-k ray, then DifferentialGeometry of the hit
//   phase 1..L   shadow probe towards light phase-1 (Lights.hs:15-24, Rays.hs:49-54); its contribution
//                cv = |lightDir.n| @* k * intensity is formed BEFORE the probe so that only cv stays live
//   after phase L: w_k, D_k to LDS, reflection ray (Integrators.hs:40-47)
// DEEP: levels 0 .. lds_levels-1 of the (w_k, D_k) stack live in LDS, the levels above them in `spill`, this workgroup's
// slice of a global array laid out the same way (recursionDepth is an unbounded Int upstream, Integrators.hs:18-20; LDS
// is not).  Same values, same fold order: where a level is kept cannot change a result.  Without DEEP every level fits
// LDS and no code for the spill exists.
// Level slot k (six floats per thread: LDS, or -- DEEP, k >= lds_levels -- this workgroup's slice of the global spill).
// `in_lds` is wave-uniform; written out per address space so that every access is a plain ds_ / global_ instruction.
// (LDS pointers carry their address space in the TYPE here: a generic pointer that has been laundered through an asm
// statement is a FLAT pointer to the compiler -- flat_load / flat_store instead of ds_read / ds_write, three times the
// vector-memory instructions of the 8-sphere scene before this was noticed)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) float lfloat;
#else
typedef float lfloat;
#endif
template <int B>
YD void slot_put(lfloat *lvl, float *spill, bool in_lds, int k, int lds_levels, int tid, int part, V3 v) {
  if (in_lds) { lfloat *q = lvl + ((size_t)k * 6 + 3 * part) * B + tid; q[0] = v.x; q[B] = v.y; q[2 * B] = v.z; }
  else { float *q = spill + ((size_t)(k - lds_levels) * 6 + 3 * part) * B + tid; q[0] = v.x; q[B] = v.y; q[2 * B] = v.z; }
}
template <int B>
YD V3 slot_get(const lfloat *lvl, const float *spill, bool in_lds, int k, int lds_levels, int tid, int part) {
  if (in_lds) {
    const lfloat *q = lvl + ((size_t)k * 6 + 3 * part) * B + tid;
    asm("" : "+v"(q));                                  // (no forwarding of the stored registers: see radiance)
    return v3(q[0], q[B], q[2 * B]);
  }
  const float *q = spill + ((size_t)(k - lds_levels) * 6 + 3 * part) * B + tid;
  return v3(q[0], q[B], q[2 * B]);
}

template <int MODE, bool STATS, int WPB, bool DEEP = false>
YD V3 radiance(const DScene &sc, bool valid, V3 ro_in, V3 rd_in, float *lds, float *xch, Counters &c, bool &bad_mat,
               float *__restrict__ spill = nullptr, int lds_levels = 0) {
  constexpr int B = 64 * WPB;                                         // threads per workgroup
  const int tid = threadIdx.x;
  bool alive = valid;
  int levels = 0;
  const int L = sc.n_lights, per = L + 1, n_steps = sc.depth * per;
  V3 cv = vof(0.0f);
  int mat = -1;
  // REGISTERS ACROSS A TRAVERSAL.  The traversal loops are where the time goes, and the packet kernel's budget is 72
  // VGPRs (7 waves per SIMD); whatever the integrator keeps alive across a traversal comes out of that budget or goes to
  // scratch memory.  So the state of the recursion lives in LDS between its uses, not in registers:
  //   * the DifferentialGeometry of the level's hit (x, n, dpdu) in a parking area of eight floats per thread (the first
  //     8 * B floats of `lds`; the (w_k, D_k) stack follows): written after the level's closest-hit traversal, read
  //     before every shadow probe and after the last one;
  //   * the level's ray and its running sum of direct light in the level's own slot, whose contents change over the
  //     level:  [rd | ro] when the ray is cast (the slot of level k + 1 is where level k leaves the reflection ray)
  //          -> [rd | direct] while the lights are probed -> [w_k | D_k] at the end, which is what the fold below reads.
  // Every value is a block-local variable loaded where it is used; only cv (the light's contribution, formed before the
  // probe), the material index and three flags cross a traversal in registers.  Addresses are laundered through an
  // empty asm at every read so that the compiler cannot forward the stored registers to the loads (which would keep
  // them alive after all).  Same values, same operations: where a value waits cannot change a result.
  // LDS is handed out in granules of 1280 bytes on gfx950: 20 floats per thread are 5120 bytes per wavefront = 4 granules,
  // room for 32 wavefronts per CU; the 21st float would cost a fifth granule and leave 25 where the packet kernel wants
  // 28.  So x.x stays in a register (`x_x`) and eight floats are parked.
  lfloat *park = (lfloat *)lds + tid;
  lfloat *lvl = (lfloat *)lds + 8 * B;
  float x_x = 0.0f;
#define YD_PARK_STORE(x, n, dpdu) { x_x = x.x; park[0] = x.y; park[B] = x.z; park[2 * B] = n.x; park[3 * B] = n.y; \
    park[4 * B] = n.z; park[5 * B] = dpdu.x; park[6 * B] = dpdu.y; park[7 * B] = dpdu.z; }
#define YD_PARK_LOAD(x, n, dpdu) { const lfloat *pk = park; asm("" : "+v"(pk)); x = v3(x_x, pk[0], pk[B]); \
    n = v3(pk[2 * B], pk[3 * B], pk[4 * B]); dpdu = v3(pk[5 * B], pk[6 * B], pk[7 * B]); }
  if (n_steps > 0) {
    const bool in0 = !DEEP || 0 < lds_levels;
    slot_put<B>(lvl, spill, in0, 0, lds_levels, tid, 0, rd_in);
    slot_put<B>(lvl, spill, in0, 0, lds_levels, tid, 1, ro_in);
  }
  for (int step = 0; step < n_steps; ++step) {
    const int k = step / per, phase = step - k * per;
    const bool in_lds = !DEEP || k < lds_levels;                      // wave-uniform: where slot k lives
    // (lanes that cast nothing get a harmless ray)
    V3 qo = vof(0.0f), qd = v3(0.0f, 0.0f, 1.0f); float qt = 1e6f; bool active;
    if (phase == 0) {
      if (!block_any<WPB>(alive, xch)) break;
      if (STATS) { if (k == 0) c.primary += lanes(alive); else c.reflection += lanes(alive); }
      active = alive;
      // (only the lanes that cast: a dead lane's slot holds whatever LDS held, and its direction still reaches the
      // branch-free first stage of every primitive step -- a NaN there would send whole wavefronts into the IEEE divide
      // of stage 1 for nothing)
      if (alive) {
        qd = slot_get<B>(lvl, spill, in_lds, k, lds_levels, tid, 0);
        qo = slot_get<B>(lvl, spill, in_lds, k, lds_levels, tid, 1);
      }
    } else {
      active = false;
      if (alive) {
        V3 x, n, dpdu;
        YD_PARK_LOAD(x, n, dpdu)
        const V3 rd = slot_get<B>(lvl, spill, in_lds, k, lds_levels, tid, 0);
        const float4 *lights = cold_of(sc)->lights;
        const float4 l0 = lights[2 * (size_t)(phase - 1)], l1 = lights[2 * (size_t)(phase - 1) + 1];
        const V3 p2l = xyz(l0) - x;                                   // Lights.hs:17-19
        const V3 ldir = norm(p2l);
        const V3 intensity = scale(1.0f / lensq(p2l), xyz(l1));
        const V3 kf = material_at(sc, mat, n, dpdu, ldir, vneg(rd));  // Integrators.hs:58
        active = lensq(kf) > 0.0f;                                    // && is lazy: probe only if k /= 0
        if (active) {
          const V3 p0 = x + scale(0.001f, ldir);                      // reachable, Rays.hs:49-54
          const V3 dd = xyz(l0) - p0;
          qo = p0; qd = norm(dd); qt = len(dd);
          cv = scale(fabsf(dot(ldir, n)), kf) * intensity;            // Integrators.hs:60
        }
      }
      if (STATS) c.shadow += lanes(active);
    }
    Hit hit;
    traverse_block<MODE, STATS, WPB>(sc, active, phase != 0, step == 0, qo, qd, qt, hit, c, xch);
    if (phase == 0) {
      alive = alive && hit.prim >= 0;
      if (STATS) c.shaded += lanes(alive);
      if (alive) {
        V3 x, n, dpdu;
        const V3 rd = slot_get<B>(lvl, spill, in_lds, k, lds_levels, tid, 0), ro = slot_get<B>(lvl, spill, in_lds, k, lds_levels, tid, 1);
        mat = surface_of(sc, ro, rd, hit, x, n, dpdu);
        YD_PARK_STORE(x, n, dpdu)
        // Map.! on a missing key (main.hs:55).  Only a lane flag here: a GLOBAL store inside this loop would make
        // every later scene load "clobberable" and ban the scalar-load path of the packet walker.
        bad_mat = bad_mat | (mat < 0);
      }
      slot_put<B>(lvl, spill, in_lds, k, lds_levels, tid, 1, vof(0.0f));          // direct = sum from 0
    } else if (alive) {
      const V3 direct = slot_get<B>(lvl, spill, in_lds, k, lds_levels, tid, 1);
      slot_put<B>(lvl, spill, in_lds, k, lds_levels, tid, 1,
                  direct + ((active && hit.prim < 0) ? cv : vof(0.0f)));          // sum = foldl (+) 0, light order
    }
    if (phase == L && alive) {
      V3 x, n, dpdu;
      YD_PARK_LOAD(x, n, dpdu)
      const V3 rd = slot_get<B>(lvl, spill, in_lds, k, lds_levels, tid, 0);
      const V3 r = rd - scale(2.0f * dot(rd, n), n);                  // reflectionDir, Integrators.hs:46-47
      const V3 f_r = material_at(sc, mat, n, dpdu, r, vneg(rd));
      const V3 w = scale(dot(n, r), f_r);
      slot_put<B>(lvl, spill, in_lds, k, lds_levels, tid, 0, w);      // slot k is now (w_k, D_k)
      levels = k + 1;
      if (k + 1 < sc.depth) {                                         // the reflection ray waits in slot k + 1 (Integrators.hs:41-43)
        const bool in1 = !DEEP || k + 1 < lds_levels;
        slot_put<B>(lvl, spill, in1, k + 1, lds_levels, tid, 0, r);
        slot_put<B>(lvl, spill, in1, k + 1, lds_levels, tid, 1, x + scale(0.001f, r));
      }
    }
  }
  V3 v = vof(0.0f);                                                   // vcast 0 _ = 0 / miss = 0
  if (DEEP)
    for (int k = levels - 1; k >= lds_levels; --k) {                  // (per-lane trip count: the lane's own levels)
      const float *slot = spill + (size_t)(k - lds_levels) * 6 * B + tid;
      V3 w = v3(slot[0], slot[B], slot[2 * B]), d = v3(slot[3 * B], slot[4 * B], slot[5 * B]);
      v = (w * v) + d;
    }
  for (int k = (DEEP && levels > lds_levels ? lds_levels : levels) - 1; k >= 0; --k) {
    const lfloat *slot = lvl + (size_t)k * 6 * B + tid;
    V3 w = v3(slot[0], slot[B], slot[2 * B]), d = v3(slot[3 * B], slot[4 * B], slot[5 * B]);
    v = (w * v) + d;
  }
#undef YD_PARK_STORE
#undef YD_PARK_LOAD
  return v;
}

YD void camera_ray(const DCold *cold, float fu, float fv, V3 &ro, V3 &rd) {   // Cameras.hs:83-86
  const V3 origin = v3(cold->origin[0], cold->origin[1], cold->origin[2]);
  const V3 direction = transform_point(cold->vtf, v3(fu, fv, cold->focal)) - origin;
  ro = origin; rd = norm(direction);
}

YD unsigned wave_sum_u32(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

YD void flush_counters(const Counters &c, unsigned long long *ctr, int lane) {
  const unsigned box = wave_sum_u32(c.box), tri = wave_sum_u32(c.tri), sph = wave_sum_u32(c.sph), culled = wave_sum_u32(c.tri_culled), viol = wave_sum_u32(c.cull_viol);
  const unsigned vals[CTR_COUNT] = {c.primary, c.reflection, c.shadow, c.shaded, 0u,
                                    box, tri, sph, c.wave_nodes, c.wave_prims, c.wave_stage1, c.wave_stage2, culled, c.wave_culls, viol};
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < CTR_COUNT; ++i)
      if (vals[i]) atomicAdd(&ctr[i], (unsigned long long)vals[i]);
  }
}

}  // namespace

// lane -> pixel for the 8x8 wave tiles (tile_mode 0), with samples in lanes and row sharding:
// lane -> (pixel of the tile, sample of the round): q-th pixel of a (64 >> sil) pixel block, 2^pw wide;
// local row -> global image row (row sharding, include/yahr_hip.h yahr_render_opts)
YD void tile_pixel(const RParams &rp, int tx, int ty, bool tile_ok, int lane, int &u, int &gv, size_t &row, bool &valid) {
  const int q = lane >> rp.sil_log2;
  const int pw = 1 << rp.pw_log2, ph = (64 >> rp.sil_log2) >> rp.pw_log2;
  u = tx * pw + (q & (pw - 1));
  const int lv = ty * ph + (q >> rp.pw_log2);
  // (launch parameters that steer branches or divide are made opaque: as invariants of the sample loop the compiler
  // hoisted the conditions and the reciprocals out of it and kept them -- spilled -- across the whole radiance
  // computation; see pixel_of)
  int n_shards = rp.n_shards;
  asm("" : "+s"(n_shards));
  gv = rp.row0 + lv;                                   // one shard: rows [row0, row0 + n_local_rows)
  if (n_shards > 1) {                                  // wave-uniform: a per-lane division only when rows are dealt out
    int strip_rows = rp.strip_rows;
    asm("" : "+s"(strip_rows));
    const int strip = lv / strip_rows;
    gv = rp.row0 + strip * strip_rows * n_shards + (lv - strip * strip_rows);
  }
  valid = tile_ok && u < rp.width && lv < rp.n_local_rows && gv < rp.height;
  row = (size_t)lv;
}
YD void wave_tile_pixel(const RParams &rp, int tile, int lane, int &u, int &gv, size_t &row, bool &valid) {
  int tiles_x = rp.tiles_x;
  asm("" : "+s"(tiles_x));
  int ty = tile / tiles_x;
  const int tx = tile - ty * tiles_x;
  // optional permutation of the tile rows (diagnostics; centre-rows-first measured slower than natural order)
  if (rp.row_order && tile < rp.n_tiles) ty = rp.row_order[ty];
  tile_pixel(rp, tx, ty, tile < rp.n_tiles, lane, u, gv, row, valid);
}

// Which pixel does this lane render?  frame of the batch, pixel (u, gv), output row, and whether there is one.  The ids are
// laundered through empty asm statements so that two calls are two computations (see k_render).  Returns the lane id.
template <int WPB, bool DEEP>
YD int pixel_of(const RParams &rp, int &frame, int &u, int &gv, size_t &row, bool &valid) {
  unsigned blk = blockIdx.x; int tid = threadIdx.x;
  asm("" : "+s"(blk)); asm("" : "+v"(tid));
  // ... and so are the launch parameters the branches below test: as loop invariants of the sample loop the compiler
  // hoisted those conditions and kept them in VGPRs (0 / 1) across the whole radiance computation -- spilled, again
  int n_frames = rp.frames, tile_mode = rp.tile_mode, swz = rp.xcd_swizzle, grid3d = rp.grid3d;
  asm("" : "+s"(n_frames)); asm("" : "+s"(tile_mode)); asm("" : "+s"(swz)); asm("" : "+s"(grid3d));
  const int lane = tid & 63, wave = tid >> 6;
  if (WPB == 1 && !DEEP && grid3d) {
    // the usual launch: a 3-D grid (frame, tile column, tile row) -- dispatched x fastest, so the same tile of every frame
    // of a batch still sits next to each other -- and no integer division at all on the way to the pixel
    unsigned by = blockIdx.y, bz = blockIdx.z;
    asm("" : "+s"(by)); asm("" : "+s"(bz));
    frame = (int)blk;
    tile_pixel(rp, (int)by, (int)bz, true, lane, u, gv, row, valid);
    return lane;
  }
  // frame batch: the same tile of every frame of the batch sits next to each other in dispatch order
  const unsigned bid = DEEP ? blk + rp.block0 : blk;   // (DEEP: a launch may be one chunk of a grid)
  frame = n_frames > 1 ? (int)(bid % (unsigned)n_frames) : 0;
  const int unit = n_frames > 1 ? (int)(bid / (unsigned)n_frames) : (int)bid;
  int tile = unit;
  if (WPB == 1) {
    if (tile_mode == 0 && swz) {                       // diagnostics: contiguous band of tiles per XCD
      const int per = (rp.n_tiles + 7) >> 3;
      tile = (tile & 7) * per + (tile >> 3);
    }
  } else if (tile_mode == 0) {                         // 2x2 tiles per workgroup
    const int ux_n = (rp.tiles_x + 1) >> 1;
    const int uy = unit / ux_n, ux = unit - uy * ux_n;
    const int tx = ux * 2 + (wave & 1), ty = uy * 2 + (wave >> 1);
    tile = tx < rp.tiles_x ? ty * rp.tiles_x + tx : rp.n_tiles;
  } else {
    tile = unit * WPB + wave;
  }
  if (tile_mode == 0) {
    wave_tile_pixel(rp, tile, lane, u, gv, row, valid);
  } else {
    // 64-pixel chunk of a reference window: {x0,y0,x1,y1}, {first pixel of the chunk, ...}; u-major order
    const int tl = tile < rp.n_tiles ? tile : rp.n_tiles - 1;          // n_tiles = number of chunks here
    const int4 win = rp.windows[2 * (size_t)tl], q = rp.windows[2 * (size_t)tl + 1];
    const int hh = win.w - win.y, npx = (win.z - win.x) * hh;
    const int p = q.x + lane;
    valid = tile < rp.n_tiles && p < npx;
    u = valid ? win.x + p / hh : 0;
    gv = valid ? win.y + p % hh : 0;
    row = (size_t)gv;
  }
  return lane;
}

// One wavefront per tile: an 8x8-pixel tile, or one of the reference's squareBatches windows (Sampling.hs:5-21)
// walked 64 pixels at a time, u-major like the reference.  WPB = 1: one wave per workgroup.  WPB = 4: a
// workgroup owns 2x2 tiles (or 4 consecutive window chunks) and regroups its rays before every traversal
// (traverse_block).
template <int MODE, bool STATS, int WPB, bool DEEP = false>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(MODE == TRAV_PACKET ? YAHR_WAVES : YAHR_WAVES_LANE, 8)))
void k_render(DScene sc_in, YD_SCENE_PARAMS, RParams rp, float *__restrict__ out, unsigned long long *__restrict__ ctr) {
  extern __shared__ float lds[];
  YD_SCENE_BIND
  float *xch = lds + (size_t)(8 + (sc.depth > 0 ? sc.depth : 1) * 6) * 64 * WPB;   // exchange area (WPB > 1 only; never DEEP)
  // DEEP: this workgroup's slice of the level spill (levels beyond the LDS part; only ever touched by such levels)
  float *spill = DEEP ? rp.lvl_spill + (size_t)blockIdx.x * (size_t)(sc.depth - rp.lds_levels) * (6 * 64 * WPB) : nullptr;
  Counters c = {};
  bool bad_mat = false;
  const unsigned long long t_start = STATS ? clock64() : 0ULL;
#ifdef YD_TILE_CLOCK
  const unsigned long long t_clock0 = wall_clock64();
#endif
  // NOTE: the ONLY global stores of this kernel come after the last traversal.  A store (or atomic) ahead of
  // a scene load makes that load "clobberable" for the compiler, which then refuses the scalar-load path and
  // silently turns the packet walker's node fetches into per-lane gathers (tests/test_isa.py guards this).
  V3 acc = vof(0.0f);
  // One sample loop for both lane layouts (ONE radiance call site: it is the whole kernel, inlined).
  //  * sil_log2 == 0: a lane is a pixel and takes its samples one after the other;
  //  * samples in lanes: the wave's rays go through a few neighbouring pixels (ONE pixel at 64 spp), so the packet
  //    walks hardly more than a single ray's path.  The pixel is still the sum of its samples IN SAMPLE ORDER
  //    (left fold from 0): the samples of a round are parked in LDS and the pixel's first lane adds them up.
  // Which pixel a lane renders is RECOMPUTED where it is needed -- at the top of every sample round and for the final
  // store -- from block and thread ids made opaque to common-subexpression elimination: carried through the radiance
  // computation instead, these values (pixel, row, frame, output pointer) were the registers the packet kernel spilled
  // to scratch (six dwords per lane: 290 MB of scratch writes per 8-frame launch).
  const int sw = 1 << rp.sil_log2;
  for (int s0 = 0; s0 < rp.spp; s0 += sw) {
    int frame, u, gv; size_t row; bool valid;
    const int lane = pixel_of<WPB, DEEP>(rp, frame, u, gv, row, valid);
    const int j0 = lane & (sw - 1);
    const DCold *cam = rp.cams ? rp.cams + frame : sc.cold;
    float du = 0.0f, dv = 0.0f;                        // spp == 1: the integer pixel corner (main.hs:73)
    int spp = rp.spp, n_strata = rp.n_strata;          // (opaque, like the divisors of pixel_of: no hoisted reciprocals)
    asm("" : "+s"(spp)); asm("" : "+s"(n_strata));
    if (spp > 1) sample_offset(u, gv, s0 + j0, n_strata, rp.seed, du, dv);
    V3 ro, rd;
    camera_ray(cam, (float)u + du, (float)gv + dv, ro, rd);
    const V3 v = radiance<MODE, STATS, WPB, DEEP>(sc, valid, ro, rd, lds, xch, c, bad_mat, spill, rp.lds_levels);
    if (rp.sil_log2 == 0) {
      acc = (rp.spp > 1) ? acc + v : v;
    } else {
      const int ln = threadIdx.x & 63;
      float *park = lds;                               // the radiance stack of this wave is free again
      park[ln] = v.x; park[64 + ln] = v.y; park[128 + ln] = v.z;
      // the pixel's first lane reads what the OTHER lanes of its wavefront stored: order the LDS accesses explicitly
      // (the DS unit serves one wavefront's operations in order; this keeps the compiler from moving them, too)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      if ((ln & (sw - 1)) == 0)
        for (int k = 0; k < sw; ++k) acc = acc + v3(park[ln + k], park[64 + ln + k], park[128 + ln + k]);
    }
  }
  int frame, u, gv; size_t row; bool valid;
  const int lane = pixel_of<WPB, DEEP>(rp, frame, u, gv, row, valid);
  const int j0 = lane & (sw - 1);
  out += (size_t)frame * rp.frame_stride;
  if (rp.sil_log2 != 0) valid = valid && j0 == 0;      // the pixel's first lane stores it
  if (rp.spp > 1) acc = scale(1.0f / (float)rp.spp, acc);
#ifdef YD_TILE_CLOCK
  // diagnostics build (tools/tile_clock.py): lane 0 of every wavefront reports when it ran instead of its pixel --
  // duration and start in ticks of the constant 100 MHz clock, and the hardware id of the wave slot
  if (lane == 0) {
    const unsigned long long t1 = wall_clock64();
    acc = v3((float)(t1 - t_clock0), (float)(t_clock0 & 0xffffffULL), (float)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));
  }
#endif
  if (valid) {
    float *px = out + (row * rp.width + u) * 3;          // samplesToImage: w*v+u (main.hs:100)
    px[0] = acc.x; px[1] = acc.y; px[2] = acc.z;
  }
  const unsigned n_bad = lanes(bad_mat);
  if (n_bad && lane == 0) atomicAdd(&ctr[CTR_BAD_MAT], (unsigned long long)n_bad);
  if (STATS) {
    flush_counters(c, ctr, lane);
    if (lane == 0) {                                   // critical-path diagnostics: longest wave, all waves
      const unsigned long long dt = clock64() - t_start;
      atomicMax(&ctr[CTR_MAX_WAVE_CYCLES], dt);
      atomicAdd(&ctr[CTR_SUM_WAVE_CYCLES], dt);
    }
  }
}

// ---- wavefront ("stream") form: render_kernel.hpp SParams ---------------------------------------------
// Morton cell of a hit point inside the root box (order of the level launches, see k_sort_* below)
YD unsigned part1by2(unsigned v) { v = (v | (v << 8)) & 0x300f00fu; v = (v | (v << 4)) & 0x30c30c3u; v = (v | (v << 2)) & 0x9249249u; return v; }
YD unsigned sort_cell(const DScene &sc, V3 p) {
  const float4 lo = sc.nodes[0], hi = sc.nodes[1];                     // the root box
  constexpr float n = (float)(1u << (YD_SORT_BITS / 3));
  const float fx = (p.x - lo.x) / (hi.x - lo.x) * n, fy = (p.y - lo.y) / (hi.y - lo.y) * n, fz = (p.z - lo.z) / (hi.z - lo.z) * n;
  // (a NaN or out-of-range coordinate lands in cell 0 or the last one: any cell will do)
  const unsigned ix = (unsigned)fminf(fmaxf(fx, 0.0f), n - 1.0f), iy = (unsigned)fminf(fmaxf(fy, 0.0f), n - 1.0f),
                 iz = (unsigned)fminf(fmaxf(fz, 0.0f), n - 1.0f);
  return (part1by2(ix) | (part1by2(iy) << 1) | (part1by2(iz) << 2)) & (YD_SORT_CELLS - 1u);
}

#ifndef YD_CELL_GROUPS
#define YD_CELL_GROUPS 4
#endif
// append the live paths of this wavefront to queue `level_out`: ballot, one atomic per wavefront, prefix count; and count
// and rank them inside the Morton cells of their hit points (first step of the counting sort that orders the next level
// launch -- here its scattered atomics hide behind the traversal work of the producing launch)
YD void q_append(const DScene &sc, const SParams &sp, int level_out, bool alive, int sid, V3 rd, V3 x, V3 n, V3 dpdu, int mat) {
  const unsigned long long m = __ballot(alive);
  if (m == 0ULL) return;                                               // wave-uniform
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ULL << lane) - 1ULL;
  const int leader = __ffsll((long long)m) - 1;
  unsigned base = 0;
  if (lane == leader) base = atomicAdd(&sp.count[level_out], (unsigned)__popcll(m));
  // rank inside the cell: the lanes of a wavefront that fall into the same cell (camera rays of one 2x2-pixel tile:
  // nearly all 64) share ONE atomic -- 64 atomics on one word from one instruction are served one after the other
  // (returning atomics, one per lane: camera-ray launch 10.7 -> 12.6 ms).  Up to YD_CELL_GROUPS cells are grouped
  // (wave-uniform loop), the lanes left over go on their own; all atomics are in flight together.
  const unsigned cell = sort_cell(sc, x);
  int owner = lane; unsigned group = 1u, before = 0u;
  unsigned long long todo = m;
  for (int it = 0; it < YD_CELL_GROUPS && todo != 0ULL; ++it) {
    const int l = __ffsll((long long)todo) - 1;
    const unsigned c0 = (unsigned)__builtin_amdgcn_readlane((int)cell, l);
    const unsigned long long same = __ballot(alive && cell == c0) & todo;
    if ((same >> lane) & 1ULL) { owner = l; group = (unsigned)__popcll(same); before = (unsigned)__popcll(same & below); }
    todo &= ~same;
  }
  unsigned rank0 = 0;
  if (alive && owner == lane) rank0 = atomicAdd(&sp.hist[(size_t)level_out * YD_SORT_CELLS + cell], group);
  base = (unsigned)__shfl((int)base, leader, 64);
  rank0 = (unsigned)__shfl((int)rank0, owner, 64);
  if (alive) {
    const unsigned i = base + (unsigned)__popcll(m & below);
    const int s = level_out & 1;
    sp.qa[s][i] = make_float4(x.x, x.y, x.z, __int_as_float(sid));
    sp.qb[s][i] = make_float4(n.x, n.y, n.z, __int_as_float(mat));
    sp.qc[s][i] = make_float4(dpdu.x, dpdu.y, dpdu.z, rd.x);
    sp.qd[s][i] = make_float4(rd.y, rd.z, 0.0f, 0.0f);
    sp.cell[i] = make_uint2(cell, rank0 + before);                     // {cell, rank inside it}
  }
}

// camera rays: one wavefront per wave tile (same lane -> pixel/sample layout as k_render), closest hit with the
// packet walker (camera rays are coherent), DifferentialGeometry of the hits -> queue 0
// the camera-ray launch of the wavefront form carries no radiance state: 64 VGPRs hold it, and all 8 wave slots of a SIMD
// pay (1 M triangles, 16 spp: 33.4 ms per frame at 8 waves, 34.0 at 7, 34.8 at 6)
#ifndef YAHR_WAVES_PRIMARY
#define YAHR_WAVES_PRIMARY 8
#endif
template <bool STATS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(YAHR_WAVES_PRIMARY, 8)))
void k_stream_primary(DScene sc_in, YD_SCENE_PARAMS, RParams rp, SParams sp, unsigned long long *__restrict__ ctr) {
  YD_SCENE_BIND
  const int lane = threadIdx.x & 63;
  Counters c = {};
  bool bad_mat = false;
  int u, gv; size_t row; bool valid;
  wave_tile_pixel(rp, sp.tile0 + (int)blockIdx.x, lane, u, gv, row, valid);
  valid = valid && row < (size_t)(sp.lv0 + sp.band_rows);
  const int sw = 1 << rp.sil_log2, j0 = lane & (sw - 1);
  for (int s0 = 0; s0 < rp.spp; s0 += sw) {
    float du = 0.0f, dv = 0.0f;                        // spp == 1: the integer pixel corner (main.hs:73)
    if (rp.spp > 1) sample_offset(u, gv, s0 + j0, rp.n_strata, rp.seed, du, dv);
    V3 ro, rd;
    camera_ray(sp.cam, (float)u + du, (float)gv + dv, ro, rd);
    const int sid = (int)(((row - (size_t)sp.lv0) * (size_t)rp.width + (size_t)u) * (size_t)rp.spp) + s0 + j0;   // band-local
    if (STATS) c.primary += lanes(valid);
    Hit hit;
    traverse<TRAV_PACKET, STATS>(sc, valid, false, true, ro, rd, 1e6f, hit, c);
    const bool alive = valid && hit.prim >= 0;
    if (STATS) c.shaded += lanes(alive);
    V3 x = vof(0.0f), n = vof(0.0f), dpdu = vof(0.0f);
    int mat = -1;
    if (alive) { mat = surface_of(sc, ro, rd, hit, x, n, dpdu); bad_mat = bad_mat | (mat < 0); }
    if (valid) sp.levels[sid] = 0;
    q_append(sc, sp, 0, alive, sid, rd, x, n, dpdu, mat);
  }
  const unsigned n_bad = lanes(bad_mat);
  if (n_bad && lane == 0) atomicAdd(&ctr[CTR_BAD_MAT], (unsigned long long)n_bad);
  if (STATS) flush_counters(c, ctr, lane);
}

// one recursion level (vhit, Integrators.hs:32-43) for 64 queued paths per wavefront: directIllumination with a
// shadow probe per light, (w_k, D_k), then the reflection ray's closest hit -> queue k+1
// ---- order of a level launch: counting sort of the queued paths by the Morton cell of their hit point ------
// Which 64 paths share a wavefront does not matter much to a lane walker; which paths run AT THE SAME TIME on the chip
// does: they decide what the L2s hold.  In queue order (the order in which the producing wavefronts finished, roughly
// image order) concurrent paths lie along whole camera rays; sorted by the 3-D cell of their hit point they start in one
// corner of the scene and sweep through it, and shadow and reflection rays of neighbouring cells walk neighbouring
// subtrees (1 M triangles, 16 spp: 46.5 -> 36.3 ms per frame; read through a random permutation instead: 76.5 ms; 6 bits
// per axis are as good as 10, 4 lose half of the gain).  Three small launches whose work follows the queue's LENGTH
// (a library radix sort would have to take its capacity): count per cell and rank inside the cell (ONE returning
// atomic per path, issued by the producing launch as it appends -- q_append -- where its latency hides behind the
// traversals of the other wavefronts; 12 M scattered atomics in a launch of their own take 0.8-1.1 ms), exclusive scan
// of the cells, scatter (plain stores).  The order inside a cell is whatever the atomics give; no result depends on it.
// exclusive prefix sum over the cells, hist[level] -> start: workgroup b owns cells [1024 b, 1024 b + 1024) and adds
// up everything in front of them itself (on average half of the 1 MB of counts, from L2), so one launch does it
__global__ __launch_bounds__(1024) void k_sort_scan(SParams sp) {
  __shared__ unsigned w_pre[16], w_own[16];
  const unsigned t = threadIdx.x, b = blockIdx.x, lane = t & 63u, w = t >> 6;
  const unsigned *hist = sp.hist + (size_t)sp.level * YD_SORT_CELLS;
  unsigned pre = 0;
  for (unsigned j = t; j < b * 1024u; j += 1024u) pre += hist[j];
  const unsigned own = hist[b * 1024u + t];
  unsigned inc = own;                                                  // inclusive scan inside the wavefront
  for (int off = 1; off < 64; off <<= 1) { const unsigned v = (unsigned)__shfl_up((int)inc, off, 64); if ((int)lane >= off) inc += v; }
  for (int off = 32; off > 0; off >>= 1) pre += (unsigned)__shfl_xor((int)pre, off, 64);
  if (lane == 63u) w_own[w] = inc;
  if (lane == 0u) w_pre[w] = pre;
  __syncthreads();
  unsigned base = 0;
  for (unsigned k = 0; k < 16u; ++k) base += w_pre[k] + (k < w ? w_own[k] : 0u);
  sp.start[b * 1024u + t] = base + inc - own;
}
__global__ __launch_bounds__(1024) void k_sort_scatter(SParams sp) {
  const unsigned e = blockIdx.x * 1024u + threadIdx.x;
  if (e >= sp.count[sp.level]) return;
  const uint2 cr = sp.cell[e];
  sp.perm[sp.start[cr.x] + cr.y] = e;
}

// The grid covers the queue's CAPACITY (its length is known on the device only); wavefronts beyond the length exit at
// once.  Dispatching those empty 64-thread workgroups costs ~0.8 ms per 518 k of them, but everything tried against it
// lost more (1 M triangles, 16 spp, ms per frame: one wavefront per workgroup 47.9; four independent wavefronts per
// workgroup 51.1 -- a workgroup's slots are only released together; a persistent grid of 8192 wavefronts taking chunks
// in grid-stride order or from an atomic cursor 70+ -- it concentrates the queue-append atomics, and one counter word
// takes ~90 of them per microsecond).
#define YD_LEVEL_WAVES 1
// the shadow rays of a level launch walk as packets (Morton order: neighbours in space, one light): their packets share a
// direction octant, so they take the octant run (oct_run); the fat-leaf culling stays off in these launches
#ifndef YD_LEVEL_OCT
#define YD_LEVEL_OCT 1
#endif
template <int MODE, bool STATS>
__global__ __launch_bounds__(64 * YD_LEVEL_WAVES) __attribute__((amdgpu_waves_per_eu(YAHR_WAVES_LANE, 8)))
void k_stream_level(DScene sc_in, YD_SCENE_PARAMS_PAIRS, SParams sp, unsigned long long *__restrict__ ctr) {
  extern __shared__ int trav_stack[];                                  // stack walkers: stack_levels x 64 entries per wavefront
  YD_SCENE_BIND
  const int lane = threadIdx.x & 63;
  int *stk = trav_stack + (size_t)(threadIdx.x >> 6) * 64 * (size_t)stack_levels + lane;
  const int k = sp.level, in = k & 1;
  const unsigned n_in = sp.count[k];
  const unsigned chunk = blockIdx.x * YD_LEVEL_WAVES + (threadIdx.x >> 6);
  if (chunk * 64u >= n_in) return;                                     // wave-uniform: the grid covers the capacity
  const unsigned e = chunk * 64u + (unsigned)lane;
  const bool valid = e < n_in;
  const unsigned ei = valid ? sp.perm[e] : 0u;                          // Morton order of the hit points (k_sort_*)
  const float4 qa = sp.qa[in][ei], qb = sp.qb[in][ei], qc = sp.qc[in][ei], qd = sp.qd[in][ei];
  const V3 x = xyz(qa), n = xyz(qb), dpdu = xyz(qc), rd = v3(qc.w, qd.x, qd.y);
  const int sid = __float_as_int(qa.w), mat = __float_as_int(qb.w);
  Counters c = {};
  bool bad_mat = false;
  V3 direct = vof(0.0f);
  const int L = sc.n_lights;
  for (int l = 0; l < L; ++l) {                                        // directIllumination, Integrators.hs:50-61
    bool active = false;
    V3 qo = x, qdir = rd, cv = vof(0.0f); float qt = 1e6f;
    if (valid) {
      const float4 *lights = cold_of(sc)->lights;
      const float4 l0 = lights[2 * (size_t)l], l1 = lights[2 * (size_t)l + 1];
      const V3 p2l = xyz(l0) - x;                                      // Lights.hs:17-19
      const V3 ldir = norm(p2l);
      const V3 intensity = scale(1.0f / lensq(p2l), xyz(l1));
      const V3 kf = material_at(sc, mat, n, dpdu, ldir, vneg(rd));     // Integrators.hs:58
      active = lensq(kf) > 0.0f;                                       // && is lazy: probe only if k /= 0
      if (active) {
        const V3 p0 = x + scale(0.001f, ldir);                         // reachable, Rays.hs:49-54
        const V3 dd = xyz(l0) - p0;
        qo = p0; qdir = norm(dd); qt = len(dd);
        cv = scale(fabsf(dot(ldir, n)), kf) * intensity;               // Integrators.hs:60
      }
    }
    if (STATS) c.shadow += lanes(active);
    Hit hit;
    // the paths of a wavefront are neighbours in space (Morton order) and a shadow ray's direction follows from its
    // origin: coherent rays, so the adaptive walker stays a packet for them (1 M triangles: 38.3 -> 37.2 ms per frame)
    traverse<MODE, STATS, false, YD_LEVEL_OCT != 0>(sc, active, true, true, qo, qdir, qt, hit, c, stk, a_pairs);
    if (valid) direct = direct + ((active && hit.prim < 0) ? cv : vof(0.0f));   // sum = foldl (+) 0, light order
  }
  const V3 r = rd - scale(2.0f * dot(rd, n), n);                       // reflectionDir, Integrators.hs:46-47
  const V3 f_r = material_at(sc, mat, n, dpdu, r, vneg(rd));
  const V3 w = scale(dot(n, r), f_r);
  if (valid) {
    sp.lev[(size_t)(2 * k) * sp.capacity + (unsigned)sid] = make_float4(w.x, w.y, w.z, 0.0f);
    sp.lev[(size_t)(2 * k + 1) * sp.capacity + (unsigned)sid] = make_float4(direct.x, direct.y, direct.z, 0.0f);
    sp.levels[sid] = (unsigned short)(k + 1);
  }
  if (!sp.last) {
    const V3 ro2 = x + scale(0.001f, r);                               // Integrators.hs:41-43
    if (STATS) c.reflection += lanes(valid);
    Hit hit;
    traverse<MODE, STATS, false>(sc, valid, false, false, ro2, r, 1e6f, hit, c, stk, a_pairs);
    const bool alive = valid && hit.prim >= 0;
    if (STATS) c.shaded += lanes(alive);
    V3 x2 = vof(0.0f), n2 = vof(0.0f), dpdu2 = vof(0.0f);
    int mat2 = -1;
    if (alive) { mat2 = surface_of(sc, ro2, r, hit, x2, n2, dpdu2); bad_mat = bad_mat | (mat2 < 0); }
    q_append(sc, sp, k + 1, alive, sid, r, x2, n2, dpdu2, mat2);
  }
  const unsigned n_bad = lanes(bad_mat);
  if (n_bad && lane == 0) atomicAdd(&ctr[CTR_BAD_MAT], (unsigned long long)n_bad);
  if (STATS) flush_counters(c, ctr, lane);
}

// per pixel: every sample's levels folded back to front (v = w_k * v + D_k: `(n .* r) @* f r * rs + direct`,
// Integrators.hs:37), the samples added in sample order from 0, times 1/spp -- the fused kernel's arithmetic
__global__ __launch_bounds__(256) void k_stream_resolve(RParams rp, SParams sp, int depth, float *__restrict__ out) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;                 // pixel of the band; `out` points at the band
  if (p >= (size_t)sp.band_rows * (size_t)rp.width) return;
  V3 acc = vof(0.0f);
  for (int s = 0; s < rp.spp; ++s) {
    const size_t sid = p * (size_t)rp.spp + (size_t)s;
    const int nl = sp.levels[sid];
    V3 v = vof(0.0f);                                                  // vcast 0 _ = 0 / miss = 0
    for (int k = nl - 1; k >= 0; --k) {
      const float4 w = sp.lev[(size_t)(2 * k) * sp.capacity + sid], d = sp.lev[(size_t)(2 * k + 1) * sp.capacity + sid];
      v = (xyz(w) * v) + xyz(d);
    }
    acc = (rp.spp > 1) ? acc + v : v;
  }
  if (rp.spp > 1) acc = scale(1.0f / (float)rp.spp, acc);
  float *px = out + p * 3;                                             // samplesToImage: w*v+u (main.hs:100)
  px[0] = acc.x; px[1] = acc.y; px[2] = acc.z;
}

// The same with one SAMPLE per lane, for spp = 2 .. 64 a power of two: the level arrays are indexed by sample id, so a
// pixel per thread reads them at a stride of spp records (every 16-byte load its own sector: 4.4 GB fetched per
// 1080p x 16 spp frame, 0.64 ms); a sample per lane reads them contiguously (0.6 GB).  The pixel's first lane then adds
// the samples up in sample order (shuffles), from 0, exactly as above.
__global__ __launch_bounds__(256) void k_stream_resolve_lanes(RParams rp, SParams sp, int depth, int spp_log2, float *__restrict__ out) {
  const size_t sid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n_samples = ((size_t)sp.band_rows * (size_t)rp.width) << spp_log2;
  const bool valid = sid < n_samples;
  V3 v = vof(0.0f);                                                    // vcast 0 _ = 0 / miss = 0
  if (valid) {
    const int nl = sp.levels[sid];
    for (int k = nl - 1; k >= 0; --k) {
      const float4 w = sp.lev[(size_t)(2 * k) * sp.capacity + sid], d = sp.lev[(size_t)(2 * k + 1) * sp.capacity + sid];
      v = (xyz(w) * v) + xyz(d);
    }
  }
  const int lane = threadIdx.x & 63, spp = 1 << spp_log2, first = lane & ~(spp - 1);
  V3 acc = vof(0.0f);
  for (int j = 0; j < spp; ++j)                                        // wave-uniform trip count
    acc = acc + v3(__shfl(v.x, first + j, 64), __shfl(v.y, first + j, 64), __shfl(v.z, first + j, 64));
  if (valid && lane == first) {
    acc = scale(1.0f / (float)spp, acc);
    float *px = out + (sid >> spp_log2) * 3;                           // samplesToImage: w*v+u (main.hs:100)
    px[0] = acc.x; px[1] = acc.y; px[2] = acc.z;
  }
}

// ---- diagnostic kernels: device functions on arbitrary rays (KAT checks on the GPU) -----------------
// (a COUNTING build: the diagnostic ray queries add their box / primitive tests to the context's counters, so a test
// can see what a box test decided -- a ray that enters a leaf's box costs a primitive test, one that misses it does not)
template <int MODE>
__global__ __launch_bounds__(64) void k_trace(DScene sc_in, YD_SCENE_PARAMS_PAIRS, int n, const float *__restrict__ rays,
                                              int any_hit, int *__restrict__ prim_out, float *__restrict__ t_out,
                                              unsigned long long *__restrict__ ctr) {
  extern __shared__ int trav_stack[];
  YD_SCENE_BIND
  const int i = blockIdx.x * 64 + threadIdx.x;
  const bool valid = i < n;
  V3 o = vof(0.0f), d = v3(0, 0, 1); float tmax = 0.0f;
  if (valid) { const float *p = rays + 7 * (size_t)i; o = v3(p[0], p[1], p[2]); d = v3(p[3], p[4], p[5]); tmax = p[6]; }
  Hit h; Counters c = {};
  traverse<MODE, true>(sc, valid, any_hit != 0, false, o, d, tmax, h, c, trav_stack + threadIdx.x, a_pairs);
  if (valid) {
    prim_out[i] = h.prim >= 0 ? __float_as_int(sc.prims[3 * (size_t)h.prim + 2].w) : -1;
    if (t_out) t_out[i] = h.t;
  }
  flush_counters(c, ctr, threadIdx.x & 63);
}

// radiance of arbitrary rays; tMax of the given ray is ignored (the integrator casts with 1e6, like vcast)
template <int MODE>
__global__ __launch_bounds__(64) void k_radiance(DScene sc_in, YD_SCENE_PARAMS, int n, const float *__restrict__ rays,
                                                 float *__restrict__ rgb, unsigned long long *__restrict__ ctr,
                                                 float *__restrict__ lvl_spill, int lds_levels) {
  extern __shared__ float lds[];
  YD_SCENE_BIND
  const int i = blockIdx.x * 64 + threadIdx.x;
  const bool valid = i < n;
  V3 ro = vof(0.0f), rd = v3(0, 0, 1);
  if (valid) { const float *p = rays + 7 * (size_t)i; ro = v3(p[0], p[1], p[2]); rd = v3(p[3], p[4], p[5]); }
  Counters c = {};
  bool bad_mat = false;
  V3 v = radiance<MODE, false, 1, true>(sc, valid, ro, rd, lds, lds, c, bad_mat,
                                        lvl_spill + (size_t)blockIdx.x * (size_t)(sc.depth - lds_levels) * (6 * 64), lds_levels);
  if (valid) { rgb[3 * (size_t)i] = v.x; rgb[3 * (size_t)i + 1] = v.y; rgb[3 * (size_t)i + 2] = v.z; }
  const unsigned n_bad = lanes(bad_mat);
  if (n_bad && threadIdx.x == 0) atomicAdd(&ctr[CTR_BAD_MAT], (unsigned long long)n_bad);
}

__global__ void k_camera_rays(DScene sc, int n, const float *__restrict__ uv, float *__restrict__ rays) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 ro, rd;
  camera_ray(sc.cold, uv[2 * (size_t)i], uv[2 * (size_t)i + 1], ro, rd);
  float *p = rays + 7 * (size_t)i;
  p[0] = ro.x; p[1] = ro.y; p[2] = ro.z; p[3] = rd.x; p[4] = rd.y; p[5] = rd.z; p[6] = 1e6f;
}

// ---- launchers (called from abi_device.cpp) -------------------------------------------------------
namespace yahr_dev {

template <int MODE, int WPB>
static hipError_t launch_render_mode(const DScene &sc, const RParams &rp, int n_blocks, float *out,
                                     unsigned long long *ctr, bool stats, hipStream_t st) {
  // per-level (w_k, D_k) stack of every thread, then (WPB > 1) the exchange area of traverse_block / block_any
  // ... in front of it the parking area of radiance() (8 floats per thread)
  const size_t lds_bytes = ((size_t)(8 + (rp.lds_levels > 0 ? rp.lds_levels : 1) * 6) * 64 * WPB +
                            (WPB > 1 ? 12 * 64 * WPB + 2 * WPB : 0)) * sizeof(float);
  if (WPB == 1 && rp.lds_levels < sc.depth) {            // deep scene: upper levels of the stack in rp.lvl_spill
    if (stats) hipLaunchKernelGGL((k_render<MODE, true, 1, true>), dim3(n_blocks), dim3(64), lds_bytes, st, sc, sc.nodes, sc.prims, sc.normals, rp, out, ctr);
    else hipLaunchKernelGGL((k_render<MODE, false, 1, true>), dim3(n_blocks), dim3(64), lds_bytes, st, sc, sc.nodes, sc.prims, sc.normals, rp, out, ctr);
    return hipGetLastError();
  }
  const dim3 grid = (WPB == 1 && rp.grid3d) ? dim3((unsigned)rp.frames, (unsigned)rp.tiles_x, (unsigned)rp.tile_rows) : dim3((unsigned)n_blocks);
  if (stats) hipLaunchKernelGGL((k_render<MODE, true, WPB>), grid, dim3(64 * WPB), lds_bytes, st, sc, sc.nodes, sc.prims, sc.normals, rp, out, ctr);
  else hipLaunchKernelGGL((k_render<MODE, false, WPB>), grid, dim3(64 * WPB), lds_bytes, st, sc, sc.nodes, sc.prims, sc.normals, rp, out, ctr);
  return hipGetLastError();
}

hipError_t launch_render(const DScene &sc, const RParams &rp, int mode, bool regroup, int n_blocks, float *out,
                         unsigned long long *ctr, bool stats, hipStream_t st) {
  if (regroup) {
    switch (mode) {
      case TRAV_LANE: return launch_render_mode<TRAV_LANE, YAHR_REGROUP_WAVES>(sc, rp, n_blocks, out, ctr, stats, st);
      case TRAV_PACKET: return launch_render_mode<TRAV_PACKET, YAHR_REGROUP_WAVES>(sc, rp, n_blocks, out, ctr, stats, st);
      default: return launch_render_mode<TRAV_ADAPTIVE, YAHR_REGROUP_WAVES>(sc, rp, n_blocks, out, ctr, stats, st);
    }
  }
  switch (mode) {
    case TRAV_LANE: return launch_render_mode<TRAV_LANE, 1>(sc, rp, n_blocks, out, ctr, stats, st);
    case TRAV_PACKET: return launch_render_mode<TRAV_PACKET, 1>(sc, rp, n_blocks, out, ctr, stats, st);
    default: return launch_render_mode<TRAV_ADAPTIVE, 1>(sc, rp, n_blocks, out, ctr, stats, st);
  }
}

template <int MODE>
static void launch_stream_levels(const DScene &sc, const DPairs &dp, SParams sp, int n_level_blocks, unsigned long long *ctr, bool stats,
                                 hipStream_t st) {
  const unsigned sort_blocks = (sp.capacity + 1023u) / 1024u;
  // the stack walkers keep a per-lane stack of pending right siblings in LDS: stack_levels x 64 entries per wavefront
  size_t lds = (MODE == TRAV_STACK || MODE == TRAV_ADAPT_STACK || MODE == TRAV_ORDERED)
                         ? (size_t)dp.stack_levels * 64 * YD_LEVEL_WAVES * sizeof(int) : 0;
  // occupancy experiments: MORE dynamic LDS than the walker needs (never less: the stack walkers index it by tree level)
  if (const char *e = getenv("YAHR_DEBUG_LDS")) { const size_t v = (size_t)atoi(e); if (v > lds) lds = v; }
  for (int k = 0; k < sc.depth; ++k) {
    sp.level = k; sp.last = (k == sc.depth - 1) ? 1 : 0;
    hipLaunchKernelGGL(k_sort_scan, dim3(YD_SORT_CELLS / 1024u), dim3(1024), 0, st, sp);
    hipLaunchKernelGGL(k_sort_scatter, dim3(sort_blocks), dim3(1024), 0, st, sp);
    if (stats) hipLaunchKernelGGL((k_stream_level<MODE, true>), dim3(n_level_blocks), dim3(64 * YD_LEVEL_WAVES), lds, st, sc, sc.nodes, sc.prims, sc.normals, dp.pairs, dp.stack_levels, sp, ctr);
    else hipLaunchKernelGGL((k_stream_level<MODE, false>), dim3(n_level_blocks), dim3(64 * YD_LEVEL_WAVES), lds, st, sc, sc.nodes, sc.prims, sc.normals, dp.pairs, dp.stack_levels, sp, ctr);
  }
}

hipError_t launch_stream_pass(const DScene &sc, const DPairs &dp, const RParams &rp, SParams sp, int secondary_mode, int n_blocks, float *out,
                              unsigned long long *ctr, bool stats, hipStream_t st) {
  if (stats) hipLaunchKernelGGL((k_stream_primary<true>), dim3(n_blocks), dim3(64), 0, st, sc, sc.nodes, sc.prims, sc.normals, rp, sp, ctr);
  else hipLaunchKernelGGL((k_stream_primary<false>), dim3(n_blocks), dim3(64), 0, st, sc, sc.nodes, sc.prims, sc.normals, rp, sp, ctr);
  // the level launches cover the queue's capacity; wavefronts beyond the queue's length exit at once
  const int n_level_blocks = (int)((sp.capacity + 64u * YD_LEVEL_WAVES - 1u) / (64u * YD_LEVEL_WAVES));
  // (a scene without pair records runs the stack modes too: they fall back to the skip-link lane walker per wavefront)
  switch (secondary_mode) {
    case TRAV_LANE: launch_stream_levels<TRAV_LANE>(sc, dp, sp, n_level_blocks, ctr, stats, st); break;
    case TRAV_PACKET: launch_stream_levels<TRAV_PACKET>(sc, dp, sp, n_level_blocks, ctr, stats, st); break;
    case TRAV_STACK: launch_stream_levels<TRAV_STACK>(sc, dp, sp, n_level_blocks, ctr, stats, st); break;
    case TRAV_ADAPT_STACK: launch_stream_levels<TRAV_ADAPT_STACK>(sc, dp, sp, n_level_blocks, ctr, stats, st); break;
    case TRAV_ORDERED: launch_stream_levels<TRAV_ORDERED>(sc, dp, sp, n_level_blocks, ctr, stats, st); break;
    default: launch_stream_levels<TRAV_ADAPTIVE>(sc, dp, sp, n_level_blocks, ctr, stats, st); break;
  }
  const size_t n_px = (size_t)sp.band_rows * (size_t)rp.width;
  int spp_log2 = 0;
  while ((1 << spp_log2) < rp.spp) ++spp_log2;
  if (rp.spp >= 2 && rp.spp <= 64 && (1 << spp_log2) == rp.spp)
    hipLaunchKernelGGL(k_stream_resolve_lanes, dim3((unsigned)(((n_px << spp_log2) + 255) / 256)), dim3(256), 0, st, rp, sp, sc.depth,
                       spp_log2, out);
  else
    hipLaunchKernelGGL(k_stream_resolve, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, st, rp, sp, sc.depth, out);
  return hipGetLastError();
}

hipError_t launch_trace(const DScene &sc, const DPairs &dp, int mode, int n, const float *rays, int any_hit, int *prim_out,
                        float *t_out, unsigned long long *ctr, hipStream_t st) {
  const int nb = (n + 63) / 64;
  const size_t lds = (size_t)dp.stack_levels * 64 * sizeof(int);
#define YD_TRACE(M, L) hipLaunchKernelGGL((k_trace<M>), dim3(nb), dim3(64), L, st, sc, sc.nodes, sc.prims, sc.normals, dp.pairs, dp.stack_levels, n, rays, any_hit, prim_out, t_out, ctr)
  switch (mode) {
    case TRAV_LANE: YD_TRACE(TRAV_LANE, 0); break;
    case TRAV_STACK: YD_TRACE(TRAV_STACK, lds); break;
    case TRAV_ADAPT_STACK: YD_TRACE(TRAV_ADAPT_STACK, lds); break;
    case TRAV_ORDERED: YD_TRACE(TRAV_ORDERED, lds); break;
    case TRAV_ADAPTIVE: YD_TRACE(TRAV_ADAPTIVE, 0); break;
    default: YD_TRACE(TRAV_PACKET, 0); break;
  }
#undef YD_TRACE
  return hipGetLastError();
}

hipError_t launch_radiance(const DScene &sc, int mode, int n, const float *rays, float *rgb,
                           unsigned long long *ctr, float *lvl_spill, int lds_levels, hipStream_t st) {
  const int nb = (n + 63) / 64;
  const size_t lds_bytes = (size_t)(8 + (lds_levels > 0 ? lds_levels : 1) * 6) * 64 * sizeof(float);
  if (mode == TRAV_LANE) hipLaunchKernelGGL((k_radiance<TRAV_LANE>), dim3(nb), dim3(64), lds_bytes, st, sc, sc.nodes, sc.prims, sc.normals, n, rays, rgb, ctr, lvl_spill, lds_levels);
  else if (mode == TRAV_PACKET) hipLaunchKernelGGL((k_radiance<TRAV_PACKET>), dim3(nb), dim3(64), lds_bytes, st, sc, sc.nodes, sc.prims, sc.normals, n, rays, rgb, ctr, lvl_spill, lds_levels);
  else hipLaunchKernelGGL((k_radiance<TRAV_ADAPTIVE>), dim3(nb), dim3(64), lds_bytes, st, sc, sc.nodes, sc.prims, sc.normals, n, rays, rgb, ctr, lvl_spill, lds_levels);
  return hipGetLastError();
}

__global__ void k_warmup() {}
hipError_t launch_warmup(hipStream_t st) {
  hipLaunchKernelGGL(k_warmup, dim3(1), dim3(64), 0, st);
  return hipGetLastError();
}

// test hook: device_math.hpp's glibc_powf on arrays (tests/test_gpu_parity.py compares it with the host's powf)
__global__ void k_debug_powf(int n, const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = glibc_powf(x[i], y[i]);
}
void debug_powf_host(int n, const float *x, const float *y, float *out) {       // the same source, compiled for the host
  for (int i = 0; i < n; ++i) out[i] = glibc_powf(x[i], y[i]);
}
hipError_t launch_debug_powf(int n, const float *x, const float *y, float *out, hipStream_t st) {
  hipLaunchKernelGGL(k_debug_powf, dim3((n + 255) / 256), dim3(256), 0, st, n, x, y, out);
  return hipGetLastError();
}

hipError_t launch_camera_rays(const DScene &sc, int n, const float *uv, float *rays, hipStream_t st) {
  hipLaunchKernelGGL(k_camera_rays, dim3((n + 255) / 256), dim3(256), 0, st, sc, n, uv, rays);
  return hipGetLastError();
}

}  // namespace yahr_dev
