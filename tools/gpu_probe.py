"""GPU probe: parity against the oracle + kernel time per traversal mode on the benchmark scenes.
Usage (on the GPU box): python tools/gpu_probe.py [scene ...]   scenes: fixed spheres bunny bunny_sah tris"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import oracle_py
import yahr_amd
from yahr_amd import scenes

NTHREADS = min(16, os.cpu_count() or 8)
MODES = {"lane": yahr_amd.TRAVERSE_LANE, "packet": yahr_amd.TRAVERSE_PACKET, "hybrid": yahr_amd.TRAVERSE_HYBRID}


def parity(img, ref):
    err = np.abs(img - ref)
    tol = 1e-4 * np.maximum(np.abs(img), np.abs(ref)) + 1e-6
    bad = int((err > tol).any(axis=2).sum())
    exact = float((img.view(np.uint32) == ref.view(np.uint32)).all(axis=2).mean())
    return bad, exact, float(err.max())


def run(name, desc, spp=1, oracle=True, reps=5):
    t0 = time.time()
    scene = desc.to_scene().build()
    t_build = time.time() - t0
    a = scene.flat.contents
    print(f"== {name}: {scene.width}x{scene.height} depth {scene.depth} nodes {a.n_nodes} prims {a.n_prims} lights {a.n_lights} "
          f"(host build {t_build:.2f}s)", flush=True)
    ref = None
    if oracle:
        t0 = time.time()
        ref, ost = oracle_py.OracleScene.from_desc(desc).build().render(spp=spp, threads=NTHREADS)
        t_or = time.time() - t0
        orays = ost["primary_rays"] + ost["reflection_rays"] + ost["shadow_rays"]
        print(f"   oracle: {t_or:.2f}s on {NTHREADS} threads, {orays} rays -> {orays / t_or / 1e6:.2f} Mrays/s; "
              f"box {ost['box_tests']} tri {ost['tri_tests']} sph {ost['sphere_tests']} shaded {ost['shaded_hits']}", flush=True)
    ctx = yahr_amd.Context(scene, 0)
    import torch
    for mname, mode in MODES.items():
        out = ctx.render(spp=spp, traverse=mode, collect_stats=True)
        torch.cuda.synchronize()
        st = ctx.stats()
        img = out.cpu().numpy()
        times = []
        for _ in range(reps):
            ctx.render(spp=spp, traverse=mode)
            torch.cuda.synchronize()
            times.append(ctx.stats()["kernel_ms"])
        rays = st["primary_rays"] + st["reflection_rays"] + st["shadow_rays"]
        ms = min(times)
        msg = (f"   {mname:6s}: {ms:8.3f} ms (median {sorted(times)[len(times) // 2]:.3f}) {rays / ms / 1e3:9.1f} Mrays/s "
               f"{scene.width * scene.height * spp / ms / 1e3:8.1f} Msamples/s | box {st['box_tests']} tri {st['tri_tests']} "
               f"sph {st['sphere_tests']} wave_nodes {st['wave_node_steps']} wave_prims {st['wave_prim_steps']}")
        if ref is not None:
            bad, exact, mx = parity(img, ref)
            same_counts = (st["box_tests"] == ost["box_tests"] and st["tri_tests"] == ost["tri_tests"]
                           and st["sphere_tests"] == ost["sphere_tests"] and rays == orays)
            msg += f" | parity: {bad} px outside 1e-4, bit-exact {exact * 100:.3f}%, max abs {mx:.2e}, counts==oracle {same_counts}"
        print(msg, flush=True)
    ctx.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["fixed", "spheres", "bunny"]
    for w in which:
        if w == "fixed":
            run("scene_fixed 256x256", scenes.scene_fixed())
        elif w == "spheres":
            run("spheres8 1080p", scenes.spheres8())
        elif w == "bunny":
            run("pseudo-bunny 1080p depth 2", scenes.pseudo_bunny())
        elif w == "bunny_sah":
            run("pseudo-bunny 1080p SAH", scenes.pseudo_bunny(split="SurfaceAreaHeuristic"))
        elif w == "bunny_small":
            run("pseudo-bunny 480x270", scenes.pseudo_bunny(480, 270))
        elif w == "tris":
            run("1M random tris 1080p", scenes.random_tris(), oracle=False, reps=3)
        elif w == "tris_small":
            run("100k random tris 480x270", scenes.random_tris(100000, 480, 270, bvh_depth=14))
