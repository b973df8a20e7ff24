"""Host BVH build + flattening (yahr_amd/csrc/host_scene.cpp) against the oracle's literal closure tree
(Culling.hs:21-112).  Node order, boxes, leaf contents and skip links must agree exactly.  CPU only."""
import numpy as np
import pytest

import yahr_amd
from oracle_py import OracleScene
from yahr_amd import scenes


def check_against_oracle(desc):
    a = desc.to_scene().build().arrays()
    tr = OracleScene.from_desc(desc).build().tree()
    n = len(tr["kind"])
    assert n == len(a["node_a"])
    leaf = tr["kind"] != 0
    assert np.array_equal(leaf, a["node_b"] >= 0)
    assert np.array_equal(tr["first"][leaf], a["node_b"][leaf])
    assert np.array_equal(tr["count"][leaf], a["node_a"][leaf])
    assert np.array_equal(tr["boxes"][:, :3].view(np.uint32), a["node_lo"].view(np.uint32))
    assert np.array_equal(tr["boxes"][:, 3:].view(np.uint32), a["node_hi"].view(np.uint32))
    assert np.array_equal(tr["prim"], a["prim_orig"])
    # skip links: subtree end computed independently from the preorder kinds
    end = np.zeros(n, dtype=np.int64)

    def walk(i):
        if tr["kind"][i] != 0:
            end[i] = i + 1
            return i + 1
        j = walk(i + 1)
        j = walk(j)
        end[i] = j
        return j
    import sys
    sys.setrecursionlimit(10000)
    if n:
        assert walk(0) == n
    inner = ~leaf
    assert np.array_equal(a["node_a"][inner], end[inner])
    assert (a["node_b"][inner] == -1).all()
    return a


@pytest.mark.parametrize("split", ["Midpoint", "SurfaceAreaHeuristic"])
def test_spec_sphere_grid(split):
    a = check_against_oracle(scenes.spec_sphere_grid(split))
    assert a["node_a"][a["node_b"] >= 0].sum() == 200


@pytest.mark.parametrize("split", ["Midpoint", "SurfaceAreaHeuristic"])
def test_small_scenes(split):
    for d in (scenes.scene_fixed(16, 16), scenes.spheres8(16, 9), scenes.test_yahr_scene()):
        check_against_oracle(d.with_(split=split))


@pytest.mark.parametrize("split,depth", [("Midpoint", 16), ("SurfaceAreaHeuristic", 16), ("Midpoint", 3), ("SurfaceAreaHeuristic", 0)])
def test_pseudo_bunny(split, depth):
    a = check_against_oracle(scenes.pseudo_bunny(32, 18, bvh_depth=depth, split=split))
    assert len(a["prim_orig"]) == 69698 and sorted(a["prim_orig"]) == list(range(69698))


def test_random_triangles_depth_cap():
    a = check_against_oracle(scenes.random_tris(30000, 32, 18, bvh_depth=9))
    assert (a["node_b"] >= 0).sum() <= 2 ** 9          # depth cap => at most 2^d leaves (Culling.hs:34)


def test_degenerate_inputs():
    base = scenes.scene_fixed(8, 8)
    # all primitives share one centroid: every split sends everything left -> multiLeaf (Culling.hs:36-37)
    same = [("sphere", (1, 2, 3), float(r), "sphere") for r in (1, 2, 3, 4, 5, 6)]
    for split in ("Midpoint", "SurfaceAreaHeuristic"):
        a = check_against_oracle(base.with_(objects=same, split=split))
        assert len(a["node_a"]) == 1 and a["node_a"][0] == 6 and a["node_b"][0] == 0
    # no objects: bvh _ _ [] = const Nothing (Culling.hs:28) -> zero nodes
    a = check_against_oracle(base.with_(objects=[]))
    assert len(a["node_a"]) == 0
    # one object: singleton leaf
    a = check_against_oracle(base.with_(objects=same[:1]))
    assert len(a["node_a"]) == 1 and a["node_a"][0] == 1
    # NaN coordinates must not crash or desynchronise the two builders
    weird = same[:3] + [("sphere", (float("nan"), 0, 0), 1.0, "sphere"), ("sphere", (0, float("inf"), 0), 1.0, "sphere")]
    for split in ("Midpoint", "SurfaceAreaHeuristic"):
        check_against_oracle(base.with_(objects=weird, split=split))


def test_hypothesis_random_scenes():
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(st.integers(0, 2 ** 32 - 1), st.integers(1, 300), st.integers(0, 8), st.booleans())
    def inner(seed, n, depth, sah):
        rng = np.random.default_rng(seed)
        objs = []
        for _ in range(n):
            if rng.random() < 0.3:
                objs.append(("sphere", tuple(rng.integers(-4, 5, 3).astype(float)), float(rng.integers(1, 3)), "sphere"))
            else:
                p = rng.integers(-6, 7, (3, 3)).astype(np.float32)     # integer grid: many exact ties
                nn = (0.0, 1.0, 0.0)
                objs.append(("triangle", tuple(p[0]), tuple(p[1]), tuple(p[2]), nn, nn, nn, "bottom"))
        d = scenes.scene_fixed(8, 8).with_(objects=objs, bvh_depth=depth, split="SurfaceAreaHeuristic" if sah else "Midpoint")
        check_against_oracle(d)
    inner()


@pytest.mark.parametrize("split", ["Midpoint", "SurfaceAreaHeuristic"])
def test_concurrent_subtree_build_is_the_same_tree(split, monkeypatch):
    """Large nodes build their two subtrees concurrently and splice them (host_scene.cpp); node for node, link for
    link and primitive for primitive the result must be the sequential build's -- and the oracle's."""
    d = scenes.random_tris(40000, 64, 48, bvh_depth=18)
    d.split = split
    monkeypatch.setenv("YAHR_BUILD_PAR", "0")
    seq = d.to_scene().build().arrays()
    monkeypatch.setenv("YAHR_BUILD_PAR", "3")
    par = check_against_oracle(d)
    assert sorted(seq) == sorted(par)
    for k in seq:
        assert np.ascontiguousarray(seq[k]).tobytes() == np.ascontiguousarray(par[k]).tobytes(), k
    # the passes over a big node's items (centroid bounds, SAH buckets, stable partition) in chunks on several threads:
    # forced down to nodes of 64 items, with 2 to 16 threads per node
    monkeypatch.setenv("YAHR_BUILD_CHUNK_MIN", "64")
    for p in ("1", "2", "4"):
        monkeypatch.setenv("YAHR_BUILD_PAR", p)
        chunked = d.to_scene().build().arrays()
        for k in seq:
            assert np.ascontiguousarray(seq[k]).tobytes() == np.ascontiguousarray(chunked[k]).tobytes(), (k, p)


def _pair_records(scene):
    import ctypes as C
    lv = C.c_int32()
    n = yahr_amd.lib.yahr_debug_pair_records(scene.flat, None, 0, C.byref(lv))
    if n == 0:
        return None, 0
    buf = np.zeros((n, 8), dtype=np.uint32)
    assert yahr_amd.lib.yahr_debug_pair_records(scene.flat, buf.ctypes.data_as(C.c_void_p), n, C.byref(lv)) == n
    return buf, lv.value


@pytest.mark.parametrize("desc", [scenes.pseudo_bunny(16, 9), scenes.pseudo_bunny(16, 9, bvh_depth=3), scenes.spec_sphere_grid("SurfaceAreaHeuristic"),
                                  scenes.random_tris(5000, 16, 9, bvh_depth=20), scenes.scene_fixed(16, 16)], ids=["bunny", "bunny3", "grid", "tris", "fixed"])
def test_pair_records_of_the_stack_walkers(desc):
    """The child-pair records (render_kernel.hpp: DPairs) the stack walkers run on, built on the host from the flattened
    tree: one pair per interior node in DFS order of the interior nodes, the left child (node i + 1) and the right child
    (the node behind the left subtree) side by side; leaf children as in the node array, interior children with the rank of
    their own pair and -1 - their DFS index; every interior box exactly the join of its children's (what lets a popped
    node re-derive its own box); stack levels = depth of the deepest node."""
    scene = desc.to_scene().build()
    a = scene.arrays()
    halves, levels = _pair_records(scene)
    na, nb = a["node_a"], a["node_b"]
    inner = np.flatnonzero(nb < 0)
    assert halves is not None and len(halves) == 2 * len(inner)
    rank = {int(i): k for k, i in enumerate(inner)}
    depth = {0: 0}
    lo = halves[:, 0:3].view(np.float32); hi = halves[:, 4:7].view(np.float32)
    ha = halves[:, 3].view(np.int32); hb = halves[:, 7].view(np.int32)
    for i in map(int, inner):
        left = i + 1
        right = left + 1 if nb[left] >= 0 else int(na[left])
        assert (int(na[right]) if nb[right] < 0 else right + 1) == int(na[i])          # the right subtree ends where node i's does
        for side, c in ((0, left), (1, right)):
            h = 2 * rank[i] + side
            assert np.array_equal(lo[h].view(np.uint32), a["node_lo"][c].view(np.uint32)) and np.array_equal(hi[h].view(np.uint32), a["node_hi"][c].view(np.uint32))
            if nb[c] >= 0:
                assert ha[h] == na[c] and hb[h] == nb[c]
            else:
                assert ha[h] == rank[c] and hb[h] == -1 - c
            depth[c] = depth[i] + 1
        assert np.array_equal(a["node_lo"][i], np.minimum(a["node_lo"][left], a["node_lo"][right]))
        assert np.array_equal(a["node_hi"][i], np.maximum(a["node_hi"][left], a["node_hi"][right]))
    assert levels == max(depth.values())


def test_scenes_without_pair_records():
    """a root that is a leaf, an empty box (the null collider's) or a non-finite box: no pair records, the walkers use the node array"""
    for d in (scenes.pseudo_bunny(16, 9, bvh_depth=0), scenes.scene_fixed(8, 8).with_(objects=[]),
              scenes.scene_fixed(8, 8).with_(objects=scenes.scene_fixed(8, 8).objects + [("sphere", (0.0, 0.0, 0.0), float("inf"), "sphere")])):
        assert _pair_records(d.to_scene().build())[0] is None
