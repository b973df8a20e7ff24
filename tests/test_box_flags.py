"""What a context lets the packet walker assume about a scene's arrays (abi_device.cpp: scene_box_flags, read through the
host-only hook yahr_debug_box_flags): bit 0 -- every node box finite with lo <= hi (NaN-free slab form); bit 2 -- every
child's box inside its parent's (octant run without a sleep index: a ray that failed a node fails everything below it);
bit 1 -- triangle records within +-2^32 (packet-level tri_far).  Every tree this library builds has all three; the node
and primitive arrays are the CALLER's, so broken ones must be recognised.  No GPU needed."""
import ctypes

import numpy as np

from yahr_amd import scenes
from yahr_amd._lib import lib


def _nodes(scene):
    f = scene.flat.contents
    return np.ctypeslib.as_array(ctypes.cast(f.nodes, ctypes.POINTER(ctypes.c_float)), shape=(f.n_nodes, 8))


def _prims(scene):
    f = scene.flat.contents
    return np.ctypeslib.as_array(ctypes.cast(f.prims, ctypes.POINTER(ctypes.c_float)), shape=(f.n_prims, 12))


def _flags(scene):
    return lib.yahr_debug_box_flags(scene.flat)


def test_built_trees_have_every_property():
    for d in (scenes.scene_fixed(8, 8), scenes.spheres8(8, 8), scenes.pseudo_bunny(8, 8, n_lon=48, n_lat=25),
              scenes.pseudo_bunny(8, 8, n_lon=24, n_lat=13, bvh_depth=3), scenes.random_tris(3000, 8, 8, bvh_depth=12),
              scenes.pseudo_bunny(8, 8, n_lon=48, n_lat=25).with_(split="SurfaceAreaHeuristic")):
        assert _flags(d.to_scene().build()) == 7


def test_broken_arrays_are_recognised():
    fresh = lambda: scenes.pseudo_bunny(8, 8, n_lon=48, n_lat=25).to_scene().build()
    # an interior box shrunk under its children: still finite, no longer nested
    s = fresh(); n = _nodes(s)
    inter = np.flatnonzero(n[:, 7].view(np.int32) < 0)
    assert len(inter) > 100
    i = inter[len(inter) // 2]
    n[i, 4:7] -= np.float32(0.25) * (n[i, 4:7] - n[i, 0:3])
    assert _flags(s) == 3
    # ... one ulp is enough, on either side, of either child
    for side, step in ((slice(0, 3), np.float32(np.inf)), (slice(4, 7), np.float32(-np.inf))):
        s = fresh(); n = _nodes(s)
        n[i, side] = np.nextafter(n[i, side], step)
        assert _flags(s) == 3
    # equal boxes are nested (a parent's box is the join of its children's: one of them touches every face)
    s = fresh(); n = _nodes(s)
    n[i + 1, 0:3] = n[i, 0:3]; n[i + 1, 4:7] = n[i, 4:7]
    assert _flags(s) & 1
    # a box that is not finite, or inverted: no shortcut at all on the boxes
    for val in (np.inf, np.nan):
        s = fresh(); n = _nodes(s)
        n[i, 4] = val
        assert _flags(s) == 2
    s = fresh(); n = _nodes(s)
    n[i, 0], n[i, 4] = n[i, 4] + np.float32(1), n[i, 0]
    assert _flags(s) == 2
    # a triangle record beyond 2^32: the boxes keep their properties, the packet-level triangle proof is off
    s = fresh(); p = _prims(s)
    tri = np.flatnonzero(p[:, 7].view(np.int32) == 0)[0]
    p[tri, 4] = np.float32(5e9)
    assert _flags(s) == 5
    p[tri, 4] = np.float32(np.nan)
    assert _flags(s) == 5
    # spheres only: nothing to bound
    assert _flags(scenes.spheres8(8, 8).with_(objects=[o for o in scenes.spheres8(8, 8).objects if o[0] == "sphere"]).to_scene().build()) == 7
