"""Differential fuzz of yahr_multi on ONE GPU: N entries (1..6) naming device 0 with the peer transport (the rehearsal form of
the N-device plan: strip placement, cross-stream events, per-entry counters), or one entry with either transport; random image
sizes (ragged last strips, fewer strips than entries), strip heights, frame batches with per-frame cameras, supersampling,
synchronous renders and overlapped submit / submit / wait / wait with pinned or pageable host arrays -- every frame and the
summed counters against a single-context render of the same frames, bit for bit.
usage: fuzz_multi.py [first seed] [number of seeds]   (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import yahr_amd
from random_scenes import random_scene_desc

COUNT_KEYS = ("primary_rays", "reflection_rays", "shadow_rays", "shaded_hits", "box_tests", "tri_tests", "sphere_tests")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
ran = {}


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def reference(ctx, frames, **kw):
    img = ctx.render(frames=frames, collect_stats=True, **kw).cpu().numpy()
    return (img if frames > 1 else img[None]), ctx.stats()


for seed in range(first, first + count):
    rng = np.random.default_rng(130_000 + seed)
    d = random_scene_desc(seed).with_camera(imW=int(rng.integers(9, 160)), imH=int(rng.integers(3, 130)))
    if rng.random() < 0.2:
        d = d.with_(depth=int(rng.integers(4, 8)))
    scene = d.to_scene().build()
    n = int(rng.choice([1, 1, 2, 2, 3, 4, 6]))
    strip = int(rng.choice([1, 3, 8, 8, 16, 40]))
    gather = yahr_amd.GATHER_PEER if n > 1 else int(rng.choice([yahr_amd.GATHER_AUTO, yahr_amd.GATHER_PEER, yahr_amd.GATHER_RCCL]))
    max_frames = 5
    cams = None
    if rng.random() < 0.6:
        cams = [[d.camera["imW"], d.camera["imH"], d.camera["focal"], *(np.asarray(d.camera["look"], dtype=np.float64) + rng.uniform(-0.1, 0.1, 3)),
                 *d.camera["up"], *(np.asarray(d.camera["pos"], dtype=np.float64) + rng.uniform(-0.5, 0.5, 3))] for _ in range(max_frames)]
    ctx = yahr_amd.Context(scene, 0)
    m = yahr_amd.MultiContext(scene, n, devices=[0] * n, gather=gather, strip_rows=strip)
    if cams is not None:
        ctx.set_cameras(cams); m.set_cameras(cams)
    name = f"{n} entries" + ("" if n > 1 else f" transport {m.transport}")
    if rng.random() < 0.5:
        m.render(frames=max_frames)          # buffers for the largest batch up front (else they grow on demand, between submissions only)
    try:
        for trial in range(3):
            fa, fb = int(rng.integers(1, max_frames + 1)), int(rng.integers(1, max_frames + 1))
            ka = dict(spp=int(rng.choice([1, 1, 2, 4, 16])), seed=int(rng.integers(0, 1000)))
            kb = dict(spp=int(rng.choice([1, 1, 3, 4])), seed=int(rng.integers(0, 1000)))
            wa, sa = reference(ctx, fa, **ka)
            wb, sb = reference(ctx, fb, **kb)
            if rng.random() < 0.5:                                      # synchronous
                ga, ta = m.render(frames=fa, collect_stats=True, **ka)
                gb, tb = m.render(frames=fb, collect_stats=True, **kb)
            else:                                                       # two submissions in flight
                m.submit(frames=fa, collect_stats=True, pinned=bool(rng.random() < 0.5), **ka)
                m.submit(frames=fb, collect_stats=True, pinned=bool(rng.random() < 0.5), **kb)
                ga, ta = m.wait()
                gb, tb = m.wait()
                ga, gb = np.array(ga), np.array(gb)
            ran[name] = ran.get(name, 0) + 2
            for tag, g, wimg, t, s, f in (("a", ga, wa, ta, sa, fa), ("b", gb, wb, tb, sb, fb)):
                g = g if f > 1 else g[None]
                cnt = [k for k in COUNT_KEYS if t[k] != s[k]]
                if not np.array_equal(bits(g), bits(wimg)) or cnt:
                    bad += 1
                    print(f"MISMATCH seed {seed} {name} strip {strip} size {scene.width}x{scene.height} depth {d.depth} submission {tag} frames {f} "
                          f"opts {ka if tag == 'a' else kb} cameras {cams is not None}: image {'same' if np.array_equal(bits(g), bits(wimg)) else 'DIFFERS'} counters {cnt}", flush=True)
    except yahr_amd.YahrError as e:
        if "larger frame batch needs new buffers" in str(e):      # the documented refusal: buffers cannot grow under a submission in flight
            ran["refused to grow in flight"] = ran.get("refused to grow in flight", 0) + 1
            m.close(); ctx.close()
            continue
        bad += 1
        print(f"ERROR seed {seed} {name} strip {strip} size {scene.width}x{scene.height}: {e}", flush=True)
    m.close()
    ctx.close()
    if seed % 20 == 0:
        print(f"seed {seed} ok so far (mismatches: {bad})", flush=True)
print(f"done: {count} seeds, {bad} mismatches; submissions per form: {ran}")
sys.exit(1 if bad else 0)
