"""The oracle (oracle/yahr_oracle.c) against a SECOND restatement of the reference's shading path written from the
Haskell alone (tests/second_opinion.py, numpy binary32): non-zero radiance must agree bit for bit.  This is what pins
the magnitudes of Integrators.hs:37, Lights.hs:19 and BSDF.hs:15-30, which the reference's own known-answer tests
(test/Spec.hs:265-328) only pin in sign and ordering."""
import numpy as np
import pytest

import second_opinion as so
from oracle_py import OracleScene
from yahr_amd import scenes

F = np.float32


def expand(desc):
    """`objects >>= expand` (Scene.hs:61-86) for the object kinds used here, material ids resolved like main.hs:55-60"""
    mat_idx = {m["id"]: i for i, m in enumerate(desc.materials)}
    prims = []
    for o in desc.objects:
        if o[0] == "sphere":
            prims.append(("sphere", so.vec(*o[1]), F(o[2]), mat_idx[o[3]]))
        elif o[0] == "triangle":
            prims.append(("triangle",) + tuple(so.vec(*v) for v in o[1:7]) + (mat_idx[o[7]],))
        elif o[0] == "mesh":
            _, pts, nrm, tris, smooth, mat = o
            pts = np.asarray(pts, dtype=F).reshape(-1, 3)
            n_t = len(tris) if smooth is None else min(len(tris), len(smooth))         # zipWith truncates
            for k in range(n_t):
                i0, i1, i2 = (int(t) for t in tris[k])
                p0, p1, p2 = so.vec(*pts[i0]), so.vec(*pts[i1]), so.vec(*pts[i2])
                if nrm is not None and smooth is not None and smooth[k]:
                    nn = np.asarray(nrm, dtype=F).reshape(-1, 3)
                    n0, n1, n2 = so.vec(*nn[i0]), so.vec(*nn[i1]), so.vec(*nn[i2])
                else:
                    fn = so.norm(so.cross(so.vsub(p2, p0), so.vsub(p1, p0)))           # Scene.hs:78
                    n0 = n1 = n2 = fn
                prims.append(("triangle", p0, p1, p2, n0, n1, n2, mat_idx[mat]))
        else:
            raise ValueError(o[0])
    mats = {i: (so.vec(*m["diffuse"]), so.vec(*m["specular"]), F(m["shininess"])) for i, m in enumerate(desc.materials)}
    lights = [(so.vec(*p), so.vec(*s)) for p, s in desc.lights]
    return so.Scene(prims, mats, lights, desc.depth)


def scene_desc():
    """spheres, loose triangles with their own (unnormalised) vertex normals, a flat and a smooth mesh, two lights,
    Blinn materials with different exponents, recursion depth 3 -- and `BVH 0`, so that the traversal is collideAll"""
    mats = [scenes._mat("a", (0, 0, 0), (0.7, 0.5, 0.3), (0.6, 0.6, 0.2), 10),
            scenes._mat("b", (0, 0, 0), (0.2, 0.8, 0.4), (0.9, 0.1, 0.5), 3.5),
            scenes._mat("c", (0, 0, 0), (0.9, 0.9, 0.9), (0.0, 0.0, 0.0), 1)]
    tri = lambda p0, p1, p2, n, m: ("triangle", p0, p1, p2, n[0], n[1], n[2], m)
    objs = [
        ("sphere", (-1.5, 0.2, 6.0), 1.0, "a"), ("sphere", (1.2, -0.3, 5.0), 0.8, "b"), ("sphere", (0.1, 1.6, 7.5), 0.9, "c"),
        tri((-6, -1.5, 2), (6, -1.5, 12), (6, -1.5, 2), [(0, 1, 0), (0.1, 1.3, 0), (0, 0.9, 0.2)], "b"),
        tri((-6, -1.5, 2), (-6, -1.5, 12), (6, -1.5, 12), [(0, 1, 0), (0, 1, 0), (0, 1, 0)], "c"),
        tri((-3, -1.5, 9), (3, 4, 9.5), (3, -1.5, 9), [(0, 0.2, -1), (0.3, 0, -1.2), (0, 0, -0.8)], "a"),
        ("mesh", np.array([(-3, -1.5, 9), (-3, 4, 9.5), (3, 4, 9.5), (0, 6, 9.2)], dtype=F),
         np.array([(0, 0, -1), (0.2, 0.1, -1), (-0.1, 0.3, -1), (0, 0.5, -1)], dtype=F),
         np.array([(0, 1, 2), (1, 3, 2)], dtype=np.int32), [True, False], "b"),
    ]
    cam = dict(imW=24, imH=16, focal=1.2, look=(0.05, -0.1, 1), up=(0, 1, 0), pos=(0.2, 0.8, -1.0))
    lights = [((4, 6, 1), (60, 55, 50)), ((-5, 3, 4), (20, 30, 45))]
    return scenes.SceneDesc(3, 0, "Midpoint", cam, mats, lights, objs)


def test_oracle_and_second_restatement_agree_bit_for_bit_on_non_zero_radiance():
    d = scene_desc()
    osc = OracleScene.from_desc(d).build()
    s2 = expand(d)
    n_nonzero, n_spec, worst = 0, 0, 0.0
    for v in range(16):
        for u in range(24):
            ray = osc.camera_ray(u, v)
            want = osc.radiance(ray)
            got = np.array(so.radiance(s2, (so.vec(*ray[0:3]), so.vec(*ray[3:6]), F(ray[6]))), dtype=F)
            if want.any():
                n_nonzero += 1
            # bit for bit, except that `**` may differ by an ulp between two correct libm implementations
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                rel = float(np.max(np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-30)))
                worst = max(worst, rel)
                assert rel < 3e-7, (u, v, got, want)
                n_spec += 1
    assert n_nonzero > 200                              # the scene is lit almost everywhere
    assert n_spec <= 0.02 * 24 * 16, (n_spec, worst)    # and all but a few pixels are bit-identical


def test_second_restatement_reproduces_the_spec_triangle_kat():
    """test/Spec.hs:117-145 on the second restatement itself (so that it, too, is pinned by a reference vector)"""
    p0, p1, p2, n = so.vec(0, 0, 5), so.vec(2, 0, 5), so.vec(1, 2, 5), so.vec(0, 0, -1)
    for x, y, hit in ((1, 1, True), (0.1, 0.1, True), (0.9, 0.1, True), (0, 1, False), (-0.1, 1, False), (2, 1, False)):
        h = so.collide_triangle(7, p0, p1, p2, n, n, n, (so.vec(x, y, 0), so.vec(0, 0, 1), F(1e6)))
        assert (h is not None) == hit
        if hit:
            assert abs(float(h[0]) - 5.0) < 1e-4 and h[2] == 7
            assert np.allclose(h[1]["point"], (x, y, 5), atol=1e-4) and np.allclose(h[1]["normal"], (0, 0, -1))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_scenes_oracle_vs_second_restatement(seed):
    """seeded random scenes (spheres and triangles with random, un-normalised vertex normals, 0-3 lights, depth 1-4,
    random Blinn materials): the two restatements agree bit for bit up to the ulp of `**`"""
    rng = np.random.default_rng(seed)
    mats = [scenes._mat("m%d" % i, (0, 0, 0), tuple(rng.uniform(0.1, 0.9, 3)), tuple(rng.uniform(0.0, 0.8, 3)),
                        float(rng.choice([1, 2.5, 8, 30]))) for i in range(3)]
    objs = []
    for _ in range(int(rng.integers(2, 6))):
        objs.append(("sphere", tuple(rng.uniform(-3, 3, 3) + (0, 0, 7)), float(rng.uniform(0.4, 1.4)), "m%d" % rng.integers(3)))
    for _ in range(int(rng.integers(4, 12))):
        c = rng.uniform(-3, 3, 3) + (0, 0, 7)
        p = [tuple(c + rng.uniform(-2.5, 2.5, 3)) for _ in range(3)]
        nrm = [tuple(rng.normal(size=3) * rng.uniform(0.5, 1.5)) for _ in range(3)]
        objs.append(("triangle", p[0], p[1], p[2], nrm[0], nrm[1], nrm[2], "m%d" % rng.integers(3)))
    lights = [(tuple(rng.uniform(-6, 6, 3) + (0, 5, 0)), tuple(rng.uniform(10, 120, 3))) for _ in range(int(rng.integers(0, 4)))]
    cam = dict(imW=14, imH=10, focal=float(rng.uniform(0.8, 2.0)), look=tuple(rng.uniform(-0.2, 0.2, 2)) + (1.0,),
               up=(0, 1, 0), pos=tuple(rng.uniform(-1, 1, 3)))
    d = scenes.SceneDesc(int(rng.integers(1, 5)), 0, "Midpoint", cam, mats, lights, objs)
    osc = OracleScene.from_desc(d).build()
    s2 = expand(d)
    n_nonzero = 0
    for v in range(10):
        for u in range(14):
            ray = osc.camera_ray(u, v)
            want = osc.radiance(ray)
            got = np.array(so.radiance(s2, (so.vec(*ray[0:3]), so.vec(*ray[3:6]), F(ray[6]))), dtype=F)
            n_nonzero += bool(want.any())
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                both_nan = np.isnan(got) & np.isnan(want)
                rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-30)
                assert np.all((rel < 3e-7) | both_nan), (seed, u, v, got, want)
    assert n_nonzero > 5 or len(lights) == 0
