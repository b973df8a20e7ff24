// host_scene.cpp -- expand, collision model, BVH build and flattening (host side, runs once per scene).
//
// Restates, array-based and with identical split decisions:
//   Scene.hs:61-86    expand (TriangleMesh, Subsampled)
//   main.hs:41-60     buildCollisionModel (primitive order = `objects >>= expand`, material map)
//   Cameras.hs:58-81  cameraToWorld, rasterToCamera, vtf
//   Culling.hs:21-112 cull / bvh / buildTree / multiLeaf / midpointSplit / sahSplit
//   AABBs.hs:10-27,45-53 box algebra;  Shapes.hs:32-33,62-63 primitive bounds
// and emits the tree in the reference's DFS order as a skip-linked node array (include/yahr_hip.h).
#include "host_scene.hpp"

#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <numeric>
#include <thread>
#include <stdexcept>

namespace yahr {

namespace {

// ---- Vectors.hs (host copies; same operation order as the device code) ----
inline Vec3 v3(float x, float y, float z) { Vec3 r; r.x = x; r.y = y; r.z = z; return r; }
inline Vec3 add(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 sub(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 neg(Vec3 a) { return v3(-a.x, -a.y, -a.z); }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }     // Vectors.hs:33
inline Vec3 scale(float f, Vec3 v) { return v3(f * v.x, f * v.y, f * v.z); }           // (@*) :42-43
inline Vec3 norm(Vec3 v) { return scale(1.0f / std::sqrt(dot(v, v)), v); }             // :46-47
inline Vec3 cross(Vec3 a, Vec3 b) {                                                     // :49-53
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float dim(int d, Vec3 v) { return d == 0 ? v.x : (d == 1 ? v.y : v.z); }
// Haskell class-default min/max (NaN-asymmetric)
inline float hmin(float x, float y) { return (x <= y) ? x : y; }
inline float hmax(float x, float y) { return (x <= y) ? y : x; }

struct Box { Vec3 lo, hi; };
const float kInf = std::numeric_limits<float>::infinity();
inline Box box_empty() { Box b; b.lo = v3(kInf, kInf, kInf); b.hi = v3(-kInf, -kInf, -kInf); return b; }   // AABBs.hs:10-11
inline Box box_from_points(Vec3 x, Vec3 y) {                                                               // :13-14
  Box b; b.lo = v3(hmin(x.x, y.x), hmin(x.y, y.y), hmin(x.z, y.z)); b.hi = v3(hmax(x.x, y.x), hmax(x.y, y.y), hmax(x.z, y.z)); return b;
}
inline Box box_include(Box b, Vec3 x) {                                                                    // :16-18
  Box r; r.lo = v3(hmin(b.lo.x, x.x), hmin(b.lo.y, x.y), hmin(b.lo.z, x.z)); r.hi = v3(hmax(b.hi.x, x.x), hmax(b.hi.y, x.y), hmax(b.hi.z, x.z)); return r;
}
inline Box box_join(Box a, Box b) {                                                                        // :25-27
  Box r; r.lo = v3(hmin(a.lo.x, b.lo.x), hmin(a.lo.y, b.lo.y), hmin(a.lo.z, b.lo.z));
  r.hi = v3(hmax(a.hi.x, b.hi.x), hmax(a.hi.y, b.hi.y), hmax(a.hi.z, b.hi.z)); return r;
}
inline Vec3 box_centroid(Box b) { return add(scale(0.5f, b.lo), scale(0.5f, b.hi)); }                      // :48-49
inline float box_surf(Box b) { Vec3 d = sub(b.hi, b.lo); return 2.0f * ((d.x * d.y + d.x * d.z) + d.y * d.z); }  // :51-53
inline int max_dimension(Vec3 v) { if (v.x > v.y && v.x > v.z) return 0; if (v.y > v.z) return 1; return 2; }   // Vectors.hs:62-66

// element-wise loops over big scenes: contiguous chunks on up to 8 threads (every element is written exactly once,
// by index, so the result does not depend on the thread count)
template <class F> void parallel_for(size_t n, F fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t nt = (n < 65536 || hw < 2) ? 1 : std::min<size_t>(8, hw);
  if (nt == 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; ++t)
    th.emplace_back([=] { for (size_t i = n * t / nt, e = n * (t + 1) / nt; i < e; ++i) fn(i); });
  for (auto &x : th) x.join();
}

// fn(t) for t = 0 .. T-1, each on a thread of its own (the caller's thread takes t = 0)
template <class F> void run_threads(int T, F fn) {
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back([=] { fn(t); });
  fn(0);
  for (auto &x : th) x.join();
}

// ---- BVH builder -------------------------------------------------------------------------------
struct Builder {
  // the permutation being partitioned carries each primitive's centroid with it: bounds and partitions then
  // stream through memory instead of chasing an index into a 12 MB centroid table (1 M primitives: 450 -> ~150 ms)
  struct Item { Vec3 c; int32_t id; };
  const std::vector<Box> &boxes;        // per primitive (expand order)
  std::vector<Item> order;              // permutation (+ centroid of the primitive's box); finally the leaf order
  std::vector<Item> scratch;
  std::vector<yahr_node> &nodes;
  int split_mode;

  // The passes over a big node's items (bounds of the centroids, SAH buckets, the stable partition) are cut into
  // contiguous chunks for several threads when NO centroid is a NaN: the reference's folds use Haskell's min/max, whose
  // results depend on the order of the operands only through NaNs and through which of two equal values (+0, -0) wins
  // -- the first for min, the last for max -- and combining the chunks' results in chunk order keeps exactly that.  A
  // NaN centroid (a NaN or an empty box) sends the whole build down the sequential passes.
  bool clean = true;
  int32_t chunked_min = 131072;          // items from which a pass is chunked (YAHR_BUILD_CHUNK_MIN: tests)

  Builder(const std::vector<Box> &b, std::vector<yahr_node> &n, int split) : boxes(b), nodes(n), split_mode(split) {
    order.resize(b.size());
    std::atomic<bool> nan_seen{false};
    parallel_for(b.size(), [&](size_t i) {
      const Vec3 c = box_centroid(b[i]);
      order[i] = Item{c, (int32_t)i};
      if (c.x != c.x || c.y != c.y || c.z != c.z) nan_seen.store(true, std::memory_order_relaxed);
    });
    clean = !nan_seen.load();
    scratch.resize(b.size());
    if (const char *e = std::getenv("YAHR_BUILD_CHUNK_MIN")) chunked_min = (int32_t)std::max(2, std::atoi(e));
  }
  bool chunked(int32_t n, int threads) const { return threads > 1 && clean && n >= chunked_min && n >= 2 * threads; }

  static void put(yahr_node &nd, Box bb, int32_t a, int32_t b) {
    nd.lo[0] = bb.lo.x; nd.lo[1] = bb.lo.y; nd.lo[2] = bb.lo.z; nd.a = a;
    nd.hi[0] = bb.hi.x; nd.hi[1] = bb.hi.y; nd.hi[2] = bb.hi.z; nd.b = b;
  }

  static Box emit_leaf_node(std::vector<yahr_node> &out, Box bb, int32_t n, int32_t first) {
    yahr_node nd; put(nd, bb, n, first); out.push_back(nd);
    return bb;
  }
  Box emit_multi_leaf(std::vector<yahr_node> &out, int32_t first, int32_t n) {   // Culling.hs:52-54
    Box bb = box_empty();
    for (int32_t i = 0; i < n; ++i) bb = box_join(bb, boxes[order[first + i].id]);
    return emit_leaf_node(out, bb, n, first);
  }

  // stable partition of order[first, first+n) by pred; returns size of the left part
  template <class P> int32_t stable_split(int32_t first, int32_t n, P pred, int threads = 1) {
    if (chunked(n, threads)) {
      // count per chunk, then every chunk writes its left and right items to their final places in `scratch`
      const int T = threads;
      std::vector<int32_t> left((size_t)T + 1, 0);
      auto lo_of = [&](int t) { return (int32_t)((int64_t)n * t / T); };
      run_threads(T, [&](int t) {
        int32_t c = 0;
        for (int32_t i = lo_of(t), e = lo_of(t + 1); i < e; ++i) c += pred(order[first + i]) ? 1 : 0;
        left[(size_t)t + 1] = c;
      });
      for (int t = 0; t < T; ++t) left[(size_t)t + 1] += left[(size_t)t];          // left items in front of chunk t
      const int32_t nl_all = left[(size_t)T];
      run_threads(T, [&](int t) {
        int32_t l = left[(size_t)t], r = nl_all + (lo_of(t) - left[(size_t)t]);
        for (int32_t i = lo_of(t), e = lo_of(t + 1); i < e; ++i) {
          const Item p = order[first + i];
          if (pred(p)) scratch[first + l++] = p; else scratch[first + r++] = p;
        }
      });
      run_threads(T, [&](int t) {
        std::copy(scratch.begin() + first + lo_of(t), scratch.begin() + first + lo_of(t + 1), order.begin() + first + lo_of(t));
      });
      return nl_all;
    }
    int32_t nl = 0, nr = 0;
    for (int32_t i = 0; i < n; ++i) {
      const Item p = order[first + i];
      if (pred(p)) order[first + nl++] = p; else scratch[first + nr++] = p;
    }
    std::copy(scratch.begin() + first, scratch.begin() + first + nr, order.begin() + first + nl);
    return nl;
  }

  int bucket_id(const Item &p, int d, float cmin, float cmax) const {   // Culling.hs:69-79
    float length = cmax - cmin;
    float dd = dim(d, p.c) - cmin;
    float frac = dd / length;
    float v = 16.0f * frac;
    // GHC floor :: Float -> Int of NaN/inf = minBound, then `max 0` -> 0
    long fl = (std::isnan(v) || std::isinf(v)) ? -1L : (long)std::floor(v);
    if (fl < 0) fl = 0;
    if (fl > 15) fl = 15;
    return (int)fl;
  }

  int32_t midpoint(int32_t first, int32_t n, int d, Box cb, int threads = 1) {       // Culling.hs:56-60
    float mid = dim(d, box_centroid(cb));
    return stable_split(first, n, [&](const Item &p) { return dim(d, p.c) <= mid; }, threads);
  }

  int32_t sah(int32_t first, int32_t n, int d, Box cb, int threads = 1) {            // Culling.hs:62-112
    if (!(n > 4)) return midpoint(first, n, d, cb, threads);
    constexpr int NB = 16;
    float cmin = dim(d, cb.lo), cmax = dim(d, cb.hi);
    int cnt[NB]; Box bbs[NB];
    for (int i = 0; i < NB; ++i) { cnt[i] = 0; bbs[i] = box_empty(); }
    if (chunked(n, threads)) {
      // buckets per chunk, joined in chunk order (see `clean`)
      const int T = threads;
      struct Part { int cnt[NB]; Box bbs[NB]; };
      std::vector<Part> part((size_t)T);
      run_threads(T, [&](int t) {
        Part &q = part[(size_t)t];
        for (int i = 0; i < NB; ++i) { q.cnt[i] = 0; q.bbs[i] = box_empty(); }
        for (int32_t i = (int32_t)((int64_t)n * t / T), e = (int32_t)((int64_t)n * (t + 1) / T); i < e; ++i) {
          const Item &p = order[first + i];
          int b = bucket_id(p, d, cmin, cmax);
          q.cnt[b] += 1; q.bbs[b] = box_join(q.bbs[b], boxes[p.id]);
        }
      });
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < NB; ++i) { cnt[i] += part[(size_t)t].cnt[i]; bbs[i] = box_join(bbs[i], part[(size_t)t].bbs[i]); }
    } else
    for (int32_t i = 0; i < n; ++i) {
      const Item &p = order[first + i];
      int b = bucket_id(p, d, cmin, cmax);
      cnt[b] += 1; bbs[b] = box_join(bbs[b], boxes[p.id]);
    }
    float wrapped = 0.0f;
    for (int i = 0; i < NB; ++i) wrapped = wrapped + box_surf(bbs[i]);
    int best_i = 0; float best_c = 0.0f;
    for (int i = 0; i <= NB - 2; ++i) {
      Box b0 = box_empty(), b1 = box_empty(); int c0 = 0, c1 = 0;
      for (int k = 0; k <= i; ++k) { b0 = box_join(b0, bbs[k]); c0 += cnt[k]; }
      for (int k = i + 1; k < NB; ++k) { b1 = box_join(b1, bbs[k]); c1 += cnt[k]; }
      float cost = 0.125f + ((float)c0 * box_surf(b0) + (float)c1 * box_surf(b1)) / wrapped;
      // minimumBy (compare `on` snd): keep x unless compare x y == GT; compare is GT when x>y or unordered
      if (i == 0) { best_i = 0; best_c = cost; }
      else if (!(best_c < cost) && !(best_c == cost)) { best_i = i; best_c = cost; }
    }
    return stable_split(first, n, [&](const Item &p) { return bucket_id(p, d, cmin, cmax) <= best_i; }, threads);
  }

  // buildTree (Culling.hs:31-50); emits the subtree's nodes in DFS order at the end of `out` and returns its box.
  // Skip links are indices into `out`.  `par` > 0: the two children of a large node are built concurrently into
  // vectors of their own and spliced in afterwards (left subtree first, links rebased) -- the partitions work on
  // disjoint ranges of `order`, so the tree is the same one, node for node, as the sequential build's.
  Box build(std::vector<yahr_node> &out, int max_depth, int32_t first, int32_t n, int par) {
    if (n == 0) {                                                    // :32  (empty, nullCollider)
      yahr_node nd; put(nd, box_empty(), 0, first); out.push_back(nd);
      return box_empty();
    }
    if (n == 1) {                                                    // :33
      Box bb = boxes[order[first].id];
      yahr_node nd; put(nd, bb, 1, first); out.push_back(nd);
      return bb;
    }
    if (max_depth == 0) return emit_multi_leaf(out, first, n);       // :34
    // the passes over this node's items run on 2^par threads (16 at the root, 8 in each of its children, ...: the
    // subtrees of a level are built concurrently, below)
    const int threads = par > 0 ? std::min(1 << std::min(par, 4), (int)std::max(1u, std::thread::hardware_concurrency())) : 1;
    // bbOfCentroids = boundAllPoints (map centroid bbs)  (:49, AABBs.hs:20-23)
    Box cb;
    if (chunked(n, threads)) {
      const int T = threads;
      std::vector<Box> part((size_t)T);
      run_threads(T, [&](int t) {
        const int32_t b = (int32_t)((int64_t)n * t / T), e = (int32_t)((int64_t)n * (t + 1) / T);
        Box q; q.lo = order[first + b].c; q.hi = order[first + b].c;  // = box_from_points with the element after it
        for (int32_t i = b + 1; i < e; ++i) q = box_include(q, order[first + i].c);
        part[(size_t)t] = q;
      });
      cb = part[0];
      for (int t = 1; t < T; ++t) cb = box_join(cb, part[(size_t)t]);
    } else {
      cb = box_from_points(order[first].c, order[first + 1].c);
      for (int32_t i = 2; i < n; ++i) cb = box_include(cb, order[first + i].c);
    }
    int d = max_dimension(sub(cb.hi, cb.lo));                        // :48
    int32_t nl = (split_mode == YAHR_SPLIT_SAH) ? sah(first, n, d, cb, threads) : midpoint(first, n, d, cb, threads);
    if (nl == n) return emit_multi_leaf(out, first, n);              // :36-37  null xs2
    size_t me = out.size();
    out.push_back(yahr_node{});
    Box b1, b2;
    if (par > 0 && n >= 8192) {
      std::vector<yahr_node> left, right;
      std::thread worker([&] { b1 = build(left, max_depth - 1, first, nl, par - 1); });
      b2 = build(right, max_depth - 1, first + nl, n - nl, par - 1);
      worker.join();
      for (const std::vector<yahr_node> *part : {&left, &right}) {
        const int32_t base = (int32_t)out.size();
        for (yahr_node nd : *part) { if (nd.b < 0) nd.a += base; out.push_back(nd); }   // interior: rebase the skip link
      }
    } else {
      b1 = build(out, max_depth - 1, first, nl, 0);
      b2 = build(out, max_depth - 1, first + nl, n - nl, 0);
    }
    Box jb = box_join(b1, b2);                                       // :39
    put(out[me], jb, (int32_t)out.size(), -1);
    return jb;
  }
};

void m44_mul(const float f[16], const float g[16], float out[16]) {   // Linear (!*!): foldl' (^+^) zero
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < 4; ++k) acc = acc + f[i * 4 + k] * g[k * 4 + j];
      out[i * 4 + j] = acc;
    }
}

void put3(float d[3], Vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

}  // namespace

void camera_matrices(const Camera &c, float tf[16], float vtf[16]) {
  Vec3 forward = norm(c.look);                                       // Cameras.hs:59-65
  Vec3 left = norm(cross(forward, c.up));
  Vec3 up = cross(left, forward);
  Vec3 a = neg(left), b = up, cc = forward;
  float dir_tf[16] = {a.x, b.x, cc.x, 0, a.y, b.y, cc.y, 0, a.z, b.z, cc.z, 0, 0, 0, 0, 1};
  float tr[16] = {1, 0, 0, c.pos.x, 0, 1, 0, c.pos.y, 0, 0, 1, c.pos.z, 0, 0, 0, 1};
  m44_mul(tr, dir_tf, tf);
  float frame_w = c.imW / c.imH, frame_h = 1.0f;                     // :68-75
  float r2c[16] = {frame_w / c.imW, 0, 0, -(frame_w / 2.0f),
                   0, -(frame_h / c.imH), 0, frame_h / 2.0f,
                   0, 0, 1, 0,
                   0, 0, 0, 1};
  m44_mul(tf, r2c, vtf);                                             // :81
}

void expand_object(const SceneObject &o, const std::function<int32_t(const std::string &)> &mat_of,
                   std::vector<Primitive> &out) {
  switch (o.kind) {
    case SceneObject::TriangleMesh: {                                // Scene.hs:62-79
      size_t n = o.tris.size() / 3;
      if (o.has_smooth && o.smooth.size() < n) n = o.smooth.size();  // zipWith truncates
      int32_t mat = mat_of(o.material_id);
      // every triangle on its own (big meshes on several threads); an index out of bounds is reported for the FIRST
      // offending triangle, as a sequential walk would
      const size_t base = out.size();
      out.resize(base + n);
      std::atomic<size_t> first_bad{n};
      parallel_for(n, [&](size_t k) {
        int32_t i0 = o.tris[3 * k], i1 = o.tris[3 * k + 1], i2 = o.tris[3 * k + 2];
        auto ok = [&](int32_t i, size_t lim) { return i >= 0 && (size_t)i < lim; };
        bool sm = o.has_smooth ? (o.smooth[k] != 0) : false;        // fromMaybe (repeat False)
        const bool use_normals = o.has_normals && sm;
        if (!(ok(i0, o.points.size()) && ok(i1, o.points.size()) && ok(i2, o.points.size())) ||
            (use_normals && !(ok(i0, o.normals.size()) && ok(i1, o.normals.size()) && ok(i2, o.normals.size())))) {
          size_t cur = first_bad.load();
          while (k < cur && !first_bad.compare_exchange_weak(cur, k)) {}
          return;
        }
        Primitive t; t.kind = 0; t.mat = mat;
        t.p0 = o.points[i0]; t.p1 = o.points[i1]; t.p2 = o.points[i2];
        if (use_normals) {
          t.n0 = o.normals[i0]; t.n1 = o.normals[i1]; t.n2 = o.normals[i2];
        } else {
          Vec3 nf = norm(cross(sub(t.p2, t.p0), sub(t.p1, t.p0)));   // :78
          t.n0 = nf; t.n1 = nf; t.n2 = nf;
        }
        out[base + k] = t;
      });
      if (first_bad.load() < n) {
        const size_t k = first_bad.load();
        const int32_t i0 = o.tris[3 * k], i1 = o.tris[3 * k + 1], i2 = o.tris[3 * k + 2];
        auto ok = [&](int32_t i, size_t lim) { return i >= 0 && (size_t)i < lim; };
        const bool pts = ok(i0, o.points.size()) && ok(i1, o.points.size()) && ok(i2, o.points.size());
        out.resize(base);
        throw std::runtime_error(std::string("TriangleMesh index out of bounds (") + (pts ? "normals" : "points") + ")");
      }
      break;
    }
    case SceneObject::Subsampled: {                                  // Scene.hs:80-85
      std::vector<Primitive> all;
      for (const auto &c : o.children) expand_object(c, mat_of, all);
      long of_hundred = (long)std::ceil(o.subsample_size * 100.0f);
      size_t pos = 0, n = all.size();
      while (pos < n) {
        long take = of_hundred < 0 ? 0 : of_hundred;
        for (long i = 0; i < take && pos + (size_t)i < n; ++i) out.push_back(all[pos + (size_t)i]);
        pos += (size_t)take; if (pos > n) pos = n;
        long drop = 100 - of_hundred; if (drop < 0) drop = 0;
        if (take == 0 && drop == 0) break;
        pos += (size_t)drop;
      }
      break;
    }
    case SceneObject::Sphere: {                                      // :86  expand x = [x]
      Primitive p; p.kind = 1; p.p0 = o.position; p.radius = o.radius; p.mat = mat_of(o.material_id);
      out.push_back(p); break;
    }
    case SceneObject::Triangle: {
      Primitive p; p.kind = 0; p.p0 = o.p0; p.p1 = o.p1; p.p2 = o.p2; p.n0 = o.n0; p.n1 = o.n1; p.n2 = o.n2;
      p.mat = mat_of(o.material_id);
      out.push_back(p); break;
    }
  }
}

void HostScene::build() {
  if (built) return;
  // materials: Data.Map.fromList -- the later duplicate wins (main.hs:57)
  std::map<std::string, int32_t> mat_index;
  mat_recs.clear();
  for (const auto &m : materials) {
    yahr_material r{}; put3(r.diffuse, m.diffuse); put3(r.specular, m.specular); r.shininess = m.shininess;
    mat_index[m.id] = (int32_t)mat_recs.size();
    mat_recs.push_back(r);
  }
  light_recs.clear();
  for (const auto &l : lights) { yahr_light r{}; put3(r.pos, l.pos); put3(r.spectrum, l.spectrum); light_recs.push_back(r); }

  // sceneObjects = S.objects s >>= S.expand   (main.hs:44); unknown material id: error when first
  // shaded (Map.!, main.hs:55) -> index -1, reported by the render call as YAHR_E_MATERIAL
  auto mat_of = [&](const std::string &id) -> int32_t {
    auto it = mat_index.find(id);
    return it == mat_index.end() ? -1 : it->second;
  };
  const auto T0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) { if (std::getenv("YAHR_BUILD_TIMING")) std::fprintf(stderr, "  host build: %-10s %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count()); };
  prims.clear();
  for (const auto &o : objects) expand_object(o, mat_of, prims);
  lap("expand");
  std::vector<Box> boxes(prims.size());
  parallel_for(prims.size(), [&](size_t i) {
    const Primitive &p = prims[i];
    if (p.kind == 1) {
      Vec3 r3 = v3(p.radius, p.radius, p.radius);
      boxes[i] = box_from_points(add(p.p0, r3), sub(p.p0, r3));                             // Shapes.hs:32-33
    } else {
      boxes[i] = box_include(box_from_points(p.p0, p.p1), p.p2);                            // Shapes.hs:62-63
    }
  });

  lap("boxes");
  nodes.clear();
  std::vector<int32_t> leaf_order;                                    // leaf position -> primitive (expand order)
  // Which builder?  The GPU one (bvh_build.hip) yields the identical node array and leaf order (both split modes) for
  // finite boxes; AUTO takes it for big scenes when a device is there (YAHR_BVH_BUILDER=host|device|auto overrides).
  int want = builder;
  if (const char *e = std::getenv("YAHR_BVH_BUILDER")) want = !std::strcmp(e, "host") ? YAHR_BUILDER_HOST : !std::strcmp(e, "device") ? YAHR_BUILDER_DEVICE : YAHR_BUILDER_AUTO;
  builder_used = YAHR_BUILDER_HOST; builder_device_ms = 0; builder_note.clear();
  if (want != YAHR_BUILDER_HOST) {
    std::string why;
    bool ok = prims.size() >= 2 && bvh_depth >= 1 && prims.size() < (size_t)1 << 30;
    if (!ok) why = "the device builder splits at least two primitives";
    if (ok && want == YAHR_BUILDER_AUTO && prims.size() < 32768) { ok = false; why = "small scene"; }
    if (ok) {
      std::atomic<bool> bad{false};
      parallel_for(boxes.size(), [&](size_t i) {
        const Box &b = boxes[i];
        if (!(std::isfinite(b.lo.x) && std::isfinite(b.lo.y) && std::isfinite(b.lo.z) && std::isfinite(b.hi.x) &&
              std::isfinite(b.hi.y) && std::isfinite(b.hi.z))) bad.store(true, std::memory_order_relaxed);
      });
      if (bad.load()) { ok = false; why = "a primitive's box is not finite"; }
    }
    if (ok) ok = device_build_bvh(&boxes[0].lo.x, (int32_t)boxes.size(), bvh_depth, split, nodes, leaf_order, &builder_device_ms, &why);
    if (ok) builder_used = YAHR_BUILDER_DEVICE;
    else {
      builder_note = why;
      if (want == YAHR_BUILDER_DEVICE) throw std::runtime_error("device BVH build not possible: " + why);
    }
  }
  if (builder_used == YAHR_BUILDER_HOST) {
    nodes.clear();
    Builder bld(boxes, nodes, split);
    lap("centroids");
    // bvh _ _ [] = const Nothing (Culling.hs:28).  Up to 2^4 subtrees are built concurrently (same tree).
    int par = std::thread::hardware_concurrency() > 1 ? 4 : 0;
    if (const char *e = std::getenv("YAHR_BUILD_PAR")) par = std::atoi(e);          // 0 = sequential build (diagnostics)
    if (!prims.empty()) bld.build(nodes, bvh_depth, 0, (int32_t)prims.size(), par);
    leaf_order.resize(prims.size());
    parallel_for(prims.size(), [&](size_t i) { leaf_order[i] = bld.order[i].id; });
  }

  lap("tree");
  prim_recs.resize(prims.size()); prim_normals.resize(prims.size());
  parallel_for(prims.size(), [&](size_t i) {
    const Primitive &p = prims[leaf_order[i]];
    yahr_prim r{}; yahr_prim_normals nr{};
    r.mat = p.mat; r.kind = p.kind; r.orig = leaf_order[i];
    put3(r.p0, p.p0);
    if (p.kind == 1) { r.e1[0] = p.radius; }
    else {
      put3(r.e1, sub(p.p1, p.p0)); put3(r.e2, sub(p.p2, p.p0));     // e1, e2 of Shapes.hs:38-39
      put3(nr.n0, p.n0); put3(nr.n1, p.n1); put3(nr.n2, p.n2);
    }
    prim_recs[i] = r; prim_normals[i] = nr;
  });

  lap("records");
  std::memset(&flat, 0, sizeof flat);
  flat.abi_version = YAHR_ABI_VERSION;
  flat.width = (int32_t)std::floor(camera.imW); flat.height = (int32_t)std::floor(camera.imH);   // main.hs:122-123
  flat.depth = depth;
  camera_matrices(camera, flat.cam_tf, flat.cam_vtf);
  flat.focal = camera.focal;
  flat.n_nodes = (int32_t)nodes.size(); flat.nodes = nodes.data();
  flat.n_prims = (int32_t)prim_recs.size(); flat.prims = prim_recs.data(); flat.normals = prim_normals.data();
  flat.n_mats = (int32_t)mat_recs.size(); flat.mats = mat_recs.data();
  flat.n_lights = (int32_t)light_recs.size(); flat.lights = light_recs.data();
  built = true;
}

}  // namespace yahr
