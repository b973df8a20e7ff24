"""Differential fuzz of the LAUNCH forms of yahr_ctx_render (tools/fuzz_walkers.py fuzzes the walkers): seeded random scenes
rendered as frame batches with per-frame cameras, in wave tiles or the reference's squareBatches tiling, as row ranges or row
shards, with the diagnostics builds (tuning bits), supersampling, deep recursion (level spill in chunks), and with the 1-D grid
limit and the spill budget lowered at random so that batches are cut into several launches -- every frame compared BIT FOR BIT
with a plain one-frame packet-walker render of a scene that holds that frame's camera.
usage: fuzz_launch_forms.py [first seed] [number of seeds]   (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import yahr_amd
from random_scenes import random_scene_desc
from yahr_amd import scenes, shard_global_rows

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
ran = {}


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


for seed in range(first, first + count):
    rng = np.random.default_rng(90_000 + seed)
    if seed % 4 == 3:
        d = scenes.random_tris(int(rng.integers(50, 5000)), int(rng.integers(20, 160)), int(rng.integers(10, 100)),
                               depth=int(rng.integers(1, 8)), bvh_depth=int(rng.integers(0, 18)), seed=int(rng.integers(1, 1 << 30)))
    else:
        d = random_scene_desc(seed)
        d = d.with_camera(imW=int(rng.integers(9, 200)), imH=int(rng.integers(5, 120)))
        if rng.random() < 0.25:
            d = d.with_(depth=int(rng.integers(4, 9)))                      # the deep kernel
    scene = d.to_scene().build()
    w, h = scene.width, scene.height
    frames = int(rng.choice([1, 1, 2, 3, 5, 7]))
    spp = int(rng.choice([1, 1, 1, 2, 3, 4, 16]))
    sseed = int(rng.integers(0, 1000))
    cams = None
    if frames > 1 and rng.random() < 0.7:
        cams = [dict(d.camera, pos=tuple(np.asarray(d.camera["pos"], dtype=np.float64) + rng.uniform(-0.5, 0.5, 3)),
                     look=tuple(np.asarray(d.camera["look"], dtype=np.float64) + rng.uniform(-0.1, 0.1, 3))) for _ in range(frames)]
    # what each frame must be: one packet-walker frame of a scene with that camera
    want = []
    for f in range(frames):
        sc = scene if cams is None else d.with_camera(**cams[f]).to_scene().build()
        if cams is None and f > 0:
            want.append(want[0])
        else:
            want.append(yahr_amd.render_image(sc, traverse=yahr_amd.TRAVERSE_PACKET, spp=spp, seed=sseed)[0])
    ctx = yahr_amd.Context(scene, 0)
    if cams is not None:
        ctx.set_cameras([[c["imW"], c["imH"], c["focal"], *c["look"], *c["up"], *c["pos"]] for c in cams])
    for trial in range(4):
        kw = dict(spp=spp, seed=sseed, frames=frames)
        form = str(rng.choice(["wave", "wave", "reference", "rows", "shard", "tuning"]))
        rows = np.arange(h)
        r0, r1 = 0, h
        if form == "reference":
            kw.update(tiles=yahr_amd.TILES_REFERENCE, n_batches=int(rng.integers(1, 12)))
        elif form == "rows":
            r0 = int(rng.integers(0, h)); r1 = int(rng.integers(r0, h + 1))
            rows = np.arange(r0, r1)
        elif form == "shard":
            ns = int(rng.integers(2, 5)); sr = int(rng.integers(1, 12)); sh = int(rng.integers(0, ns))
            kw.update(n_shards=ns, shard=sh, strip_rows=sr)
            rows = shard_global_rows(h, ns, sh, sr)
        elif form == "tuning":
            bit = int(rng.choice([yahr_amd.TUNE_XCD_BANDS, yahr_amd.TUNE_CENTRE_FIRST, yahr_amd.TUNE_REGROUP, yahr_amd.TUNE_ONE_PIXEL_PER_LANE]))
            if bit == yahr_amd.TUNE_REGROUP and d.depth > 3:
                bit = yahr_amd.TUNE_XCD_BANDS
            kw.update(tuning=bit)
            if rng.random() < 0.3:
                kw.update(tiles=yahr_amd.TILES_REFERENCE)
        kw["traverse"] = int(rng.choice([yahr_amd.TRAVERSE_PACKET, yahr_amd.TRAVERSE_LANE, yahr_amd.TRAVERSE_HYBRID, yahr_amd.TRAVERSE_AUTO]))
        env = {}
        if rng.random() < 0.6:
            env["YAHR_MAX_GRID_BLOCKS"] = str(int(rng.integers(1, 400)))
        if rng.random() < 0.5:
            env["YAHR_SPILL_BUDGET_KB"] = str(int(rng.integers(2, 64)))
        os.environ.update(env)
        try:
            out = ctx.render(r0, r1, **kw)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            if frames == 1:
                got = got[None]
            name = form + ("+cut" if env else "")
            ran[name] = ran.get(name, 0) + 1
            for f in range(frames):
                if len(rows) and not np.array_equal(bits(got[f][: len(rows)]), bits(want[f][rows])):
                    bad += 1
                    print(f"MISMATCH seed {seed} frame {f} form {form} kw {kw} env {env} size {w}x{h} depth {d.depth} cameras {cams is not None}", flush=True)
                    break
        except yahr_amd.YahrError as e:
            # the one legitimate refusal: a single frame larger than the (lowered) grid limit, or a level stack beyond the budget
            if "too large" in str(e) and env:
                ran["refused"] = ran.get("refused", 0) + 1
            else:
                bad += 1
                print(f"ERROR seed {seed} form {form} kw {kw} env {env} size {w}x{h} depth {d.depth}: {e}", flush=True)
        finally:
            for k in env:
                os.environ.pop(k, None)
    ctx.close()
    if seed % 20 == 0:
        print(f"seed {seed} ok so far (mismatches: {bad})", flush=True)
print(f"done: {count} seeds, {bad} mismatches; renders per form: {ran}")
sys.exit(1 if bad else 0)
