"""Seeded random scenes for the parity tests and tools/fuzz_walkers.py (TEST INFRASTRUCTURE): spheres, loose triangles and
small meshes (smooth and flat) at random places and sizes, 0-3 lights, recursion depth 1-4, either split mode, BVH depth
0-16, random camera."""
import numpy as np

from yahr_amd import scenes


def random_scene_desc(seed):
    rng = np.random.default_rng(1000 + seed)
    mats = [scenes._mat("m%d" % i, (0, 0, 0), tuple(rng.uniform(0.1, 0.9, 3)), tuple(rng.uniform(0.0, 0.6, 3)),
                        float(rng.choice([1, 4, 10, 40, 200]))) for i in range(3)]
    objs = []
    for _ in range(int(rng.integers(3, 9))):
        objs.append(("sphere", tuple(rng.uniform(-4, 4, 3) + (0, 0, 8)), float(rng.uniform(0.3, 1.5)), "m%d" % rng.integers(3)))
    for _ in range(int(rng.integers(5, 40))):
        c = rng.uniform(-4, 4, 3) + (0, 0, 8)
        p = [tuple(c + rng.uniform(-1.5, 1.5, 3)) for _ in range(3)]
        nrm = np.cross(np.subtract(p[2], p[0]), np.subtract(p[1], p[0]))
        nrm = tuple(nrm / (np.linalg.norm(nrm) + 1e-30))
        objs.append(("triangle", p[0], p[1], p[2], nrm, nrm, nrm, "m%d" % rng.integers(3)))
    for _ in range(int(rng.integers(1, 4))):
        nv = int(rng.integers(4, 30))
        pts = (rng.uniform(-1.0, 1.0, (nv, 3)) + rng.uniform(-3, 3, 3) + (0, 0, 8)).astype(np.float32)
        tris = rng.integers(0, nv, (int(rng.integers(2, 40)), 3)).astype(np.int32)
        smooth = rng.integers(0, 2, len(tris)).astype(bool) if rng.random() < 0.7 else None
        nrm = None
        if rng.random() < 0.6:
            nrm = rng.normal(size=(nv, 3)).astype(np.float32)
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        objs.append(("mesh", pts, nrm, tris, smooth, "m%d" % rng.integers(3)))
    lights = [(tuple(rng.uniform(-8, 8, 3) + (0, 6, 0)), tuple(rng.uniform(20, 200, 3))) for _ in range(int(rng.integers(0, 4)))]
    cam = dict(imW=int(rng.integers(33, 90)), imH=int(rng.integers(17, 70)), focal=float(rng.uniform(0.7, 2.5)),
               look=tuple(rng.uniform(-0.3, 0.3, 2)) + (1.0,), up=(0, 1, 0), pos=tuple(rng.uniform(-1, 1, 3)))
    return scenes.SceneDesc(int(rng.integers(1, 5)), int(rng.integers(0, 17)), str(rng.choice(["Midpoint", "SurfaceAreaHeuristic"])),
                         cam, mats, lights, objs)
