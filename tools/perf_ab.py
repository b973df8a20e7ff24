"""Interleaved A/B timing of render variants in ONE process (cdna guide rule 24): N variants x M rounds,
reports min and median kernel time from HIP events.  Usage:
    python tools/perf_ab.py [--scene bunny|spheres|tris] [--rounds 15] name:traverse:tuning ..."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch

import yahr_amd
from yahr_amd import scenes

TR = {"lane": yahr_amd.TRAVERSE_LANE, "packet": yahr_amd.TRAVERSE_PACKET, "hybrid": yahr_amd.TRAVERSE_HYBRID,
      "auto": yahr_amd.TRAVERSE_AUTO, "stream": yahr_amd.TRAVERSE_STREAM, "stack": yahr_amd.TRAVERSE_STACK,
      "hybrid_stack": yahr_amd.TRAVERSE_HYBRID_STACK, "ordered": yahr_amd.TRAVERSE_ORDERED}

def _no_lights(d):
    d.lights = []
    return d


ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="bunny")
ap.add_argument("--rounds", type=int, default=15)
ap.add_argument("--spp", type=int, default=1)
ap.add_argument("variants", nargs="*", default=["lane:lane:0", "packet:packet:0", "hybrid:hybrid:0"])
a = ap.parse_args()

desc = {"bunny": lambda: scenes.pseudo_bunny(), "spheres": lambda: scenes.spheres8(),
        "bunny4": lambda: scenes.pseudo_bunny(depth=4),
        "tris": lambda: scenes.random_tris(), "tris200k": lambda: scenes.random_tris(200000, bvh_depth=16),
        "tris_d1": lambda: scenes.random_tris(depth=1),            # primary + shadow rays
        "tris_primary": lambda: _no_lights(scenes.random_tris(depth=1)),
        "tris_refl": lambda: _no_lights(scenes.random_tris(depth=2))}[a.scene]()        # primary + reflection rays, no shadow rays
scene = desc.to_scene().build()
ctx = yahr_amd.Context(scene, 0)
MAXF = 32
out = torch.zeros((MAXF, scene.height, scene.width, 3), device="cuda:0")
st = torch.cuda.current_stream()
variants = []
for v in a.variants:
    parts = v.split(":")
    name, tr, tun = parts[0], parts[1], parts[2]
    frames = int(parts[3]) if len(parts) > 3 else 1
    walker = TR[parts[4]] if len(parts) > 4 else 0          # walker of the stream form's secondary rays (diagnostics)
    assert frames <= MAXF
    variants.append((name, yahr_amd.make_opts(traverse=TR[tr], tuning=int(tun), spp=a.spp, frames=frames, stream_walker=walker)))
times = {n: [] for n, _ in variants}
ref = None
for r in range(a.rounds + 2):
    for name, o in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        ctx.render_into(out.data_ptr(), 0, scene.height, o, st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        if r >= 2:
            times[name].append(e0.elapsed_time(e1) / max(1, o.frames))
        if r == 0:
            img = out[0].clone()
            for f in range(1, max(1, o.frames)):
                if not torch.equal(out[f].view(torch.int32), img.view(torch.int32)):
                    print(f"WARNING: variant {name}: frame {f} differs from frame 0")
            if ref is None:
                ref = img
            elif not torch.equal(img.view(torch.int32), ref.view(torch.int32)):
                print(f"WARNING: variant {name} is not bit-identical to the first variant")
s = ctx.stats()
rays = None
o = yahr_amd.make_opts(collect_stats=True, spp=a.spp)
ctx.render_into(out.data_ptr(), 0, scene.height, o, st.cuda_stream)
torch.cuda.synchronize()
s = ctx.stats()
rays = s["primary_rays"] + s["reflection_rays"] + s["shadow_rays"]
print(f"scene {a.scene} {scene.width}x{scene.height} spp {a.spp}: {rays} rays/frame")
for name, _ in variants:
    t = sorted(times[name])
    print(f"  {name:14s} min {t[0]:8.3f} ms/frame  median {t[len(t) // 2]:8.3f} ms/frame   {rays / t[0] / 1e3:9.1f} Mrays/s (at min)")
