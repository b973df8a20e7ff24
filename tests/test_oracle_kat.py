"""The oracle against the reference's own known-answer tests (test/Spec.hs), CPU only.
oracle/kat.c restates every hspec case; here the same vectors (tests/golden/spec_kat.json) are also driven
through the oracle's C API from Python, so that the ctypes route used by the GPU parity tests is pinned too."""
import ctypes as C
import json
import math
import os
import subprocess

import numpy as np

import oracle_py
from oracle_py import OracleScene, f3

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "spec_kat.json")))
PI = float(np.float32(math.pi))


def near(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))) < 1e-4


def vec(a_pi):
    a = np.float32(np.float32(a_pi) * np.float32(PI))
    return np.array([np.cos(a), 0.0, np.sin(a)], dtype=np.float32)


def test_kat_binary_all_pass():
    r = subprocess.run([oracle_py.KAT_PATH], capture_output=True, text=True)
    lines = r.stdout.strip().splitlines()
    assert lines[-1].startswith("SUMMARY"), r.stdout
    passed, total = map(int, lines[-1].split()[1].split("/"))
    fails = [l for l in lines if l.startswith("FAIL")]
    assert r.returncode == 0 and passed == total and total >= 70 and not fails, fails


def test_cross():
    L = oracle_py.lib()
    L.oracle_cross.argtypes = [C.POINTER(C.c_float)] * 3
    for c in KAT["cross"]["cases"]:
        out = (C.c_float * 3)()
        L.oracle_cross(f3(c["a"]), f3(c["b"]), out)
        assert near(list(out), c["out"])


def passes_through(p, ray):
    x0, u = ray[0:3].astype(np.float64), ray[3:6].astype(np.float64)
    d = int(np.argmax(np.abs(u)))
    t = (p[d] - x0[d]) / u[d]
    return np.linalg.norm(np.asarray(p) - x0 - t * u) < 1e-4


def camera_scene(case):
    k = KAT["camera"]
    s = OracleScene()
    s.L.oracle_scene_set_camera(s.h, k["imW"], k["imH"], k["focal"], f3(case["look"]), f3(case["up"]), f3(k["origin"]))
    return s


def test_camera_rays():
    k = KAT["camera"]
    o = np.array(k["origin"], dtype=np.float64)
    for case in k["cases"]:
        s = camera_scene(case)
        r = s.camera_ray(100, 100)
        assert near(r[0:3], o) and near(r[3:6], case["look"]) and r[6] == np.float32(1e6)
        r = s.camera_ray(0, 0)
        assert near(r[0:3], o)
        assert passes_through(o + np.array(case["corner_through_offset"]), r)


def test_triangle_and_bounds():
    L = oracle_py.lib()
    t = KAT["triangle"]
    for xy, expect in [(p, True) for p in t["hits"]] + [(p, False) for p in t["misses"]]:
        tt = C.c_float()
        pt, nn = (C.c_float * 3)(), (C.c_float * 3)()
        ok = L.oracle_collide_triangle(f3(t["p0"]), f3(t["p1"]), f3(t["p2"]), f3(t["n"]), f3(t["n"]), f3(t["n"]),
                                       f3([xy[0], xy[1], 0]), f3(t["ray_dir"]), np.float32(t["tmax"]),
                                       C.byref(tt), pt, nn, None, None)
        assert bool(ok) == expect
        if expect:
            assert near(list(pt), [xy[0], xy[1], 5]) and near(list(nn), t["n"])
    L.oracle_bound_triangle.argtypes = [C.POINTER(C.c_float)] * 5
    lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
    L.oracle_bound_triangle(f3(t["p0"]), f3(t["p1"]), f3(t["p2"]), lo, hi)
    assert near(list(lo), t["bound_min"]) and near(list(hi), t["bound_max"])
    b = KAT["bound_sphere"]
    L.oracle_bound_sphere.argtypes = [C.c_float] + [C.POINTER(C.c_float)] * 3
    L.oracle_bound_sphere(b["r"], f3(b["c"]), lo, hi)
    assert near(list(lo), b["min"]) and near(list(hi), b["max"])


def test_mesh_expand():
    m = KAT["mesh_expand"]
    s = OracleScene()
    s.L.oracle_scene_add_material(s.h, f3([1, 1, 1]), f3([0, 0, 0]), 1.0)
    s._emit(("m", np.array(m["points"], dtype=np.float32), None, np.array(m["tris"], dtype=np.int32), None, 0))
    pr, mats = s.prims()
    assert pr.shape[0] == 2
    for i, tri in enumerate(m["expected"]):
        assert pr[i, 0] == 0
        assert near(pr[i, 1:4], tri[0]) and near(pr[i, 4:7], tri[1]) and near(pr[i, 7:10], tri[2])
        # flat normal = norm ((p2 - p0) x (p1 - p0))   Scene.hs:78
        p0, p1, p2 = (np.array(v, dtype=np.float64) for v in tri)
        n = np.cross(p2 - p0, p1 - p0)
        n /= np.linalg.norm(n)
        assert near(pr[i, 10:13], n) and near(pr[i, 13:16], n) and near(pr[i, 16:19], n)


def test_aabb_ray_box():
    L = oracle_py.lib()
    rb = KAT["aabb"]["ray_box"]
    L.oracle_norm.argtypes = [C.POINTER(C.c_float)] * 2
    for dirs, expect in ((rb["miss_dirs_unnormalised"], 0), (rb["hit_dirs_unnormalised"], 1)):
        for org, d in zip(rb["origins"], dirs):
            u = (C.c_float * 3)()
            L.oracle_norm(f3(d), u)
            assert L.oracle_bb_ray(f3(rb["min"]), f3(rb["max"]), f3(org), u, np.float32(1e6), None) == expect


def test_bvh_sphere_grid_both_split_modes():
    from yahr_amd import scenes
    k = KAT["bvh"]
    for split in ("SurfaceAreaHeuristic", "Midpoint"):
        s = OracleScene.from_desc(scenes.spec_sphere_grid(split)).build()
        for xy, ijk in zip(k["rays_xy"], k["expect_ijk"]):
            w, t = s.intersect([xy[0], xy[1], 0, 0, 0, 1, 1e6])
            assert w == ((ijk[0] - 1) * 10 + (ijk[1] - 1)) * 2 + (ijk[2] - 1)


def test_blinn_bsdf():
    L = oracle_py.lib()
    b = KAT["bsdf_blinn"]

    def at(light, eye):
        out = (C.c_float * 3)()
        L.oracle_bsdf_at(1, np.float32(b["exponent"]), f3(b["normal"]), f3(b["dpdu"]), f3(vec(light)), f3(vec(eye)), out)
        return np.array(list(out))
    assert near(at(b["rear"]["light"], b["rear"]["eye"]), [0, 0, 0])
    for l in b["no_transmission"]["lights"]:
        assert near(at(l, b["no_transmission"]["eye"]), [0, 0, 0])
    fs = [float((at(l, b["monotone"]["eye"]) ** 2).sum()) for l in b["monotone"]["lights"]]
    assert fs == sorted(fs) and fs[-1] > 0


def test_integrator_cases_are_black():
    k = KAT["integrator"]
    for eye_a, light_a, ray in ((None, 0.25, [0, 0, -1, 0, 0, 1, 1e6]), (1.5, 0.25, None)):
        s = OracleScene()
        s.L.oracle_scene_set_integrator(s.h, 1)
        m = s.L.oracle_scene_add_material_raw_blinn(s.h, np.float32(k["exponent"]))
        s.L.oracle_scene_add_triangle(s.h, f3(k["p0"]), f3(k["p1"]), f3(k["p2"]), f3(k["n"]), f3(k["n"]), f3(k["n"]), m)
        lp = vec(light_a) if eye_a is None else -vec(light_a)
        s.L.oracle_scene_add_light(s.h, f3(lp), f3([1, 1, 1]))
        if ray is None:
            e = vec(eye_a)
            ray = list(-e) + list(e) + [1e6]
        assert near(s.radiance(ray), [0, 0, 0])


def test_square_batches_cover_without_overlap():
    L = oracle_py.lib()
    for w, h, n in KAT["square_batches"]["dims"]:
        cnt = np.zeros((h, w), dtype=np.int32)
        for b in range(n):
            win = (C.c_int32 * 4)()
            L.oracle_batch_window(w, h, b, n, win)
            cnt[win[1]:win[3], win[0]:win[2]] += 1
        assert (cnt == 1).all()


def test_n_batches_precedence():
    # main.hs:128-130 parses as roundUpPow2 (((max (32*numThreads) width) * height) `div` 256)
    L = oracle_py.lib()
    assert L.oracle_n_batches(1, 1920, 1080) == 8192
    assert L.oracle_n_batches(8, 256, 256) == 256
    assert L.oracle_n_batches(1, 3840, 2160) == 32768
    assert L.oracle_n_batches(64, 1920, 1080) == 16384     # 32*64 = 2048 > 1920


def test_subsample_pick():
    L = oracle_py.lib()
    idx = (C.c_int32 * 1000)()
    n = L.oracle_subsample_pick(250, np.float32(0.25), idx)      # take 25 of every 100
    got = list(idx[:n])
    assert got == list(range(0, 25)) + list(range(100, 125)) + list(range(200, 225))
    assert L.oracle_subsample_pick(10, np.float32(1.0), idx) == 10
    assert L.oracle_subsample_pick(250, np.float32(0.0), idx) == 0
    assert L.oracle_subsample_pick(130, np.float32(0.011), idx) == 4   # ceiling(1.1) = 2 of every 100


def test_oracle_render_is_deterministic_and_thread_invariant():
    from yahr_amd import scenes
    d = scenes.scene_fixed(48, 32)
    a, sa = OracleScene.from_desc(d).build().render(threads=1)
    b, sb = OracleScene.from_desc(d).build().render(threads=4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa == sb
    rows, _ = OracleScene.from_desc(d).build().render_rows(5, 17, threads=2)
    assert np.array_equal(rows.view(np.uint32), a[5:17].view(np.uint32))
    assert a.max() > 0
