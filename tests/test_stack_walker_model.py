"""CPU model of the stack walker's EXACTNESS ARGUMENT (yahr_amd/csrc/render_kernel.hip: stack_loop), independent of the GPU:
a numpy-binary32 walk over the child-pair records with the LDS stack of pending right siblings, the "stale" watermark and
the re-derivation of a popped node's box from its children's -- against the plain depth-first walk over the node array in the
reference's order (Culling.hs:24-25,38; Rays.hs:39-46) and against the oracle.  Closest hit and first hit: same primitive,
same t (bit for bit), same number of box and primitive tests.  TEST INFRASTRUCTURE (a model, not the product)."""
import ctypes as C

import numpy as np
import pytest

import yahr_amd
from oracle_py import OracleScene
from yahr_amd import scenes

F = np.float32
DEAD = -(1 << 31)


def slab(lo, hi, o, inv, tmax):
    """the NaN-free slab test of device_math.hpp (slab_hit_fast): min / max exact, so `near <= far_box && near <= tMax`"""
    with np.errstate(all="ignore"):
        t0 = (lo - o) * inv
        t1 = (hi - o) * inv
    near = max(F(0), np.minimum(t0, t1).max())
    far = min(F(tmax), np.maximum(t0, t1).min())
    return not (near > far)


def dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def cross(a, b):
    return np.array([F(F(a[1] * b[2]) - F(a[2] * b[1])), F(F(a[2] * b[0]) - F(a[0] * b[2])), F(F(a[0] * b[1]) - F(a[1] * b[0]))], dtype=F)


def tri(a, idx, o, u, tmax):
    """collideTriangle (Shapes.hs:36-59) as device_math.hpp: tri_candidate + tri_accept; -> t or None"""
    p0, e1, e2 = a["prim_p0"][idx], a["prim_e1"][idx], a["prim_e2"][idx]
    with np.errstate(all="ignore"):
        s = (o - p0).astype(F)
        s1 = cross(u, e2)
        s2 = cross(s, e1)
        inv = F(1) / dot(s1, e1)
        t = F(dot(s2, e2) * inv); b1 = F(dot(s1, s) * inv); b2 = F(dot(s2, u) * inv)
        b0 = F(F(F(1) - b1) - b2)
        if not (b0 >= 0 and b0 <= 1 and b1 >= 0 and b1 <= 1 and b2 >= 0 and b2 <= 1 and t > 0 and t <= tmax):
            return None
        ns = (F(b0) * a["n0"][idx] + F(b1) * a["n1"][idx]).astype(F) + (F(b2) * a["n2"][idx]).astype(F)
        return t if dot(u, ns.astype(F)) < 0 else None


def walk_reference(a, o, u, tmax, any_hit):
    """the node array in DFS order with skip links: box, left subtree, cut ray, right subtree"""
    inv = (F(1) / u).astype(F)
    n = len(a["node_a"])
    best, boxes, prims, i = -1, 0, 0, 0
    while i < n:
        boxes += 1
        h = slab(a["node_lo"][i], a["node_hi"][i], o, inv, tmax)
        na, nb = int(a["node_a"][i]), int(a["node_b"][i])
        if nb >= 0:
            if h:
                for k in range(nb, nb + na):
                    prims += 1
                    t = tri(a, k, o, u, tmax)
                    if t is not None:
                        best, tmax = k, t
                        if any_hit:
                            return best, tmax, boxes, prims
            i += 1
        else:
            i = i + 1 if h else na
    return best, tmax, boxes, prims


def walk_stack(a, halves, levels, o, u, tmax, any_hit):
    """stack_loop<STATS = true, ORDERED = false>, one lane"""
    inv = (F(1) / u).astype(F)
    lo = halves[:, 0:3].view(F); hi = halves[:, 4:7].view(F)
    ha = halves[:, 3].view(np.int32); hb = halves[:, 7].view(np.int32)
    stk, stale, cur, first = [], 0, -1, 0
    best, boxes, prims, deepest = -1, 0, 0, 0
    while first is not None or cur >= 0 or stk:
        e, self_test = cur, False
        if first is not None:
            e, first, self_test = first, None, True
            boxes += 1
        elif e < 0:
            e = stk.pop()
            self_test = len(stk) < stale
            stale = min(stale, len(stk))
            boxes += 1                                   # the reference tests this child's box now
        cur = -1
        if e == DEAD:
            continue
        if e >= 0:                                       # an interior node: its pair
            l, r = 2 * e, 2 * e + 1
            if self_test and not slab(np.minimum(lo[l], lo[r]), np.maximum(hi[l], hi[r]), o, inv, tmax):
                continue                                 # the popped node's own box = join of its children's
            boxes += 1                                   # the left child's test happens right away
            h0, h1 = slab(lo[l], hi[l], o, inv, tmax), slab(lo[r], hi[r], o, inv, tmax)
            c0, c1 = (int(ha[l]), int(hb[l])), (int(ha[r]), int(hb[r]))
            if not h0:                                   # left missed: the right child is next, with this very tMax
                boxes += 1
                h0, c0 = h1, c1
            else:
                stk.append((~(2 * e + 1) if c1[1] >= 0 else c1[0]) if h1 else DEAD)
                deepest = max(deepest, len(stk))
        else:                                            # a leaf child: its half record, always tested on the pop
            s = ~e
            h0, c0 = slab(lo[s], hi[s], o, inv, tmax), (int(ha[s]), int(hb[s]))
        if h0:
            if c0[1] >= 0:
                for k in range(c0[1], c0[1] + c0[0]):
                    prims += 1
                    t = tri(a, k, o, u, tmax)
                    if t is not None:
                        best, tmax = k, t
                        if any_hit:
                            return best, tmax, boxes, prims, deepest
                        stale = len(stk)                 # tMax changed: what is pending was pushed before
            else:
                cur = c0[0]
    return best, tmax, boxes, prims, deepest


def _pairs(scene):
    lv = C.c_int32()
    n = yahr_amd.lib.yahr_debug_pair_records(scene.flat, None, 0, C.byref(lv))
    buf = np.zeros((n, 8), dtype=np.uint32)
    yahr_amd.lib.yahr_debug_pair_records(scene.flat, buf.ctypes.data_as(C.c_void_p), n, C.byref(lv))
    return buf, lv.value


@pytest.mark.parametrize("bvh_depth", [12, 4, 20])
def test_stack_walk_is_the_reference_walk(bvh_depth):
    rng = np.random.default_rng(5 + bvh_depth)
    d = scenes.random_tris(1500, 16, 16, bvh_depth=bvh_depth, seed=77)
    quads = []                                           # coincident, axis-aligned: exact ties and zero-thickness boxes
    for i in range(40):
        x, y, z = float(rng.integers(-6, 5)), float(rng.integers(-6, 5)), float(2 + i % 3)
        for _ in range(2):
            quads.append(("triangle", (x, y, z), (x + 3, y, z), (x, y + 3, z), (0, 0, -1), (0, 0, -1), (0, 0, -1), "tri"))
    d = d.with_(objects=d.objects + quads)
    scene = d.to_scene().build()
    a = scene.arrays()
    halves, levels = _pairs(scene)
    osc = OracleScene.from_desc(d).build()
    hits = 0
    for r in range(160):
        o = rng.uniform(-9, 9, 3).astype(F)
        u = rng.normal(size=3)
        u = (u / np.linalg.norm(u)).astype(F)
        if r % 2 == 0:
            o = np.array([rng.integers(-12, 12) * 0.5, rng.integers(-12, 12) * 0.5, -1.0], dtype=F)
            u = np.array([1e-3, 2e-3, 1.0], dtype=F)    # towards the quads (no exactly zero component: the NaN-free form)
        tmax = F(rng.uniform(0.5, 8.0)) if r % 3 == 0 else F(1e6)
        for any_hit in (False, True):
            ref = walk_reference(a, o, u, tmax, any_hit)
            got = walk_stack(a, halves, levels, o, u, tmax, any_hit)
            assert got[0] == ref[0] and F(got[1]).view(np.uint32) == F(ref[1]).view(np.uint32), (r, any_hit, got, ref)
            assert got[2:4] == ref[2:4], (r, any_hit, "test counts", got, ref)
            assert got[4] <= levels
        w, t = osc.intersect(np.concatenate([o, u, [tmax]]).astype(F))
        best = walk_reference(a, o, u, tmax, False)
        assert (w < 0) == (best[0] < 0) and (w < 0 or (int(a["prim_orig"][best[0]]) == w and F(t).view(np.uint32) == F(best[1]).view(np.uint32)))
        hits += best[0] >= 0
    assert hits > 20
