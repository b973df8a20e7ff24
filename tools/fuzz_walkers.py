"""Differential fuzz of the forms of the render path on the GPU (lane, packet, adaptive, wavefront, wavefront in bands, and
-- round 3 -- the wavefront form on the two stack walkers, the fast (uncounted) builds of all of them, recursion depths up
to 9 (level spill), and the device-built BVH against the host's)
   [round 2 text:] Differential fuzz of the four forms of the render path on the GPU (one-off, not part of the test suite): seeded random
scenes (tests/random_scenes.py) and random 'triangle soup' scenes at random sizes, recursion depths, samples per pixel,
row shards and -- for the wavefront form -- random queue budgets (several bands), compared BIT FOR BIT on the image and
on every counter with the lane walker.  usage: fuzz_walkers.py [first seed] [number of seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import yahr_amd
from random_scenes import random_scene_desc
from yahr_amd import scenes

COUNT_KEYS = ("primary_rays", "reflection_rays", "shadow_rays", "shaded_hits", "box_tests", "tri_tests", "sphere_tests")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
ran = {}
culled_total = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(50_000 + seed)
    if seed % 3 == 2:
        d = scenes.random_tris(int(rng.integers(50, 20000)), int(rng.integers(20, 200)), int(rng.integers(10, 120)),
                               depth=int(rng.integers(1, 10)), bvh_depth=int(rng.integers(0, 21)), seed=int(rng.integers(1, 1 << 30)))
    else:
        d = random_scene_desc(seed)
        d = d.with_camera(imW=int(rng.integers(9, 300)), imH=int(rng.integers(5, 160)))
    scene = d.to_scene().build()
    if seed % 3 == 2 and scene.flat.contents.n_prims >= 2 and d.bvh_depth >= 1:          # (both split modes since round 4)
        if seed % 2:
            d = d.with_(split="SurfaceAreaHeuristic")
            scene = d.to_scene().build()
        dev = d.to_scene().set_builder(yahr_amd.BUILDER_DEVICE).build()      # the GPU-built tree must be the host's
        host = d.to_scene().set_builder(yahr_amd.BUILDER_HOST).build()
        a, b = host.arrays(), dev.arrays()
        if any(not np.array_equal(np.ascontiguousarray(a[k]).view(np.uint32), np.ascontiguousarray(b[k]).view(np.uint32)) for k in a):
            bad += 1
            print(f"MISMATCH seed {seed}: device-built BVH differs from the host's", flush=True)
        ran["device-bvh"] = ran.get("device-bvh", 0) + 1
    spp = int(rng.choice([1, 1, 2, 3, 4, 8, 16, 64]))
    kw = dict(spp=spp, seed=int(rng.integers(0, 1000)))
    if rng.random() < 0.3:
        kw.update(n_shards=int(rng.integers(2, 5)), strip_rows=int(rng.integers(1, 12)))
        kw["shard"] = int(rng.integers(0, kw["n_shards"]))
    ref = None
    for mode, name in ((yahr_amd.TRAVERSE_LANE, "lane"), (yahr_amd.TRAVERSE_PACKET, "packet"),
                       (yahr_amd.TRAVERSE_HYBRID, "adaptive"), (yahr_amd.TRAVERSE_STREAM, "stream"),
                       (yahr_amd.TRAVERSE_STREAM, "stream-bands"), (yahr_amd.TRAVERSE_STACK, "stack"),
                       (yahr_amd.TRAVERSE_HYBRID_STACK, "hybrid-stack"), (yahr_amd.TRAVERSE_AUTO, "auto")):
        if name == "stream-bands":
            tile_rows_bytes = max(1, scene.width * spp * (141 + 32 * max(1, d.depth)) * 8)
            os.environ["YAHR_STREAM_BUDGET_KB"] = str(max(1, int(tile_rows_bytes * rng.uniform(1.0, 4.0)) >> 10))
        try:
            if "n_shards" in kw:
                import torch
                ctx = yahr_amd.Context(scene, 0)
                t = ctx.render(traverse=mode, collect_stats=True, **kw)
                torch.cuda.synchronize()
                img, st = t.cpu().numpy(), ctx.stats()
                ctx.close()
            else:
                img, st = yahr_amd.render_image(scene, traverse=mode, collect_stats=True, **kw)
        except yahr_amd.YahrError as e:
            if name == "stream-bands":              # budget below one tile row: an explicit STREAM request is an error
                continue
            raise
        finally:
            os.environ.pop("YAHR_STREAM_BUDGET_KB", None)
        ran[name] = ran.get(name, 0) + 1
        if ref is None:
            ref = (img, st)
            continue
        same = np.array_equal(np.ascontiguousarray(img).view(np.uint32), np.ascontiguousarray(ref[0]).view(np.uint32))
        cnt = [k for k in COUNT_KEYS if st[k] != ref[1][k]]
        if st["tri_cull_violations"]:                  # the fat-leaf culling's self-check (counting builds re-run every skipped test)
            cnt.append("tri_cull_violations")
        culled_total = culled_total + st["tri_tests_culled"]
        if not same or cnt:
            bad += 1
            print(f"MISMATCH seed {seed} {name} kw {kw} size {scene.width}x{scene.height} depth {d.depth}: image {'same' if same else 'DIFFERS'} counters {cnt}", flush=True)
    # the uncounted (fast) builds of every form: the image alone
    if "n_shards" not in kw:
        for mode, name in ((yahr_amd.TRAVERSE_PACKET, "packet-fast"), (yahr_amd.TRAVERSE_HYBRID, "adaptive-fast"),
                           (yahr_amd.TRAVERSE_STREAM, "stream-fast"), (yahr_amd.TRAVERSE_HYBRID_STACK, "hybrid-stack-fast"),
                           (yahr_amd.TRAVERSE_STACK, "stack-fast")):
            img, _ = yahr_amd.render_image(scene, traverse=mode, collect_stats=False, **kw)
            ran[name] = ran.get(name, 0) + 1
            if not np.array_equal(np.ascontiguousarray(img).view(np.uint32), np.ascontiguousarray(ref[0]).view(np.uint32)):
                bad += 1
                print(f"MISMATCH seed {seed} {name} kw {kw} size {scene.width}x{scene.height} depth {d.depth}: image DIFFERS", flush=True)
    if seed % 20 == 0:
        print(f"seed {seed} ok so far (mismatches: {bad})", flush=True)
print(f"done: {count} seeds, {bad} mismatches; renders per form: {ran}; triangle tests settled per packet: {culled_total}")
sys.exit(1 if bad else 0)
