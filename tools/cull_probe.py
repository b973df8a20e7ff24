"""A/B of the fat-leaf packet culling (render_kernel.hip: leaf_culled) in ONE process: two contexts on the same scene,
one created with YAHR_NO_LEAF_CULL=1.  Images must be bit-identical and the per-ray counters equal; prints kernel times
(lone frame and 8-frame batch) and the culling counters.  usage: cull_probe.py [bunny|bunny4|spheres|tris200k|tris] [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import yahr_amd
from yahr_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
trav = {"packet": yahr_amd.TRAVERSE_PACKET, "auto": yahr_amd.TRAVERSE_AUTO, "hybrid": yahr_amd.TRAVERSE_HYBRID, "stream": yahr_amd.TRAVERSE_STREAM}[sys.argv[3] if len(sys.argv) > 3 else "packet"]
desc = {"bunny": lambda: scenes.pseudo_bunny(), "bunny4": lambda: scenes.pseudo_bunny(depth=4), "spheres": lambda: scenes.spheres8(),
        "tris200k": lambda: scenes.random_tris(200000, bvh_depth=16), "tris": lambda: scenes.random_tris(),
        "tris_d16": lambda: scenes.random_tris(bvh_depth=16)}[name]()
scene = desc.to_scene().build()
os.environ["YAHR_NO_LEAF_CULL"] = "1"
ctx_off = yahr_amd.Context(scene, 0)
del os.environ["YAHR_NO_LEAF_CULL"]
ctx_on = yahr_amd.Context(scene, 0)
F = 8
out = torch.zeros((F, scene.height, scene.width, 3), device="cuda:0")
st = torch.cuda.current_stream()

def run(ctx, frames, rounds=12, stats=False):
    o = yahr_amd.make_opts(traverse=trav, spp=spp, frames=frames, collect_stats=stats)
    ts = []
    for r in range(rounds + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); ctx.render_into(out.data_ptr(), 0, scene.height, o, st.cuda_stream); e1.record(st)
        torch.cuda.synchronize()
        if r >= 2: ts.append(e0.elapsed_time(e1) / frames)
    ts.sort()
    return ts[0], ts[len(ts) // 2], out[0].clone()

res = {}
for rnd in range(2):                       # interleaved
    for tag, ctx in (("off", ctx_off), ("on", ctx_on)):
        for frames in (1, F):
            mn, med, img = run(ctx, frames, rounds=10)
            k = (tag, frames)
            res[k] = (min(mn, res[k][0]) if k in res else mn, med, img)
for frames in (1, F):
    a, b = res[("off", frames)], res[("on", frames)]
    same = torch.equal(a[2].view(torch.int32), b[2].view(torch.int32))
    print(f"{name} spp {spp} frames/launch {frames}: cull off {a[0]:.4f} ms/frame (median {a[1]:.4f})   cull on {b[0]:.4f} (median {b[1]:.4f})   x{a[0] / b[0]:.3f}   bit-identical {same}")
sa = sb = None
for tag, ctx in (("off", ctx_off), ("on", ctx_on)):
    run(ctx, 1, rounds=1, stats=True)
    s = ctx.stats()
    print(tag, {k: s[k] for k in ("primary_rays", "reflection_rays", "shadow_rays", "shaded_hits", "box_tests", "tri_tests", "sphere_tests", "wave_node_steps", "wave_prim_steps", "wave_stage1_steps", "wave_stage2_steps", "tri_tests_culled", "wave_cull_steps", "tri_cull_violations")})
    if tag == "off": sa = s
    else: sb = s
keys = ("primary_rays", "reflection_rays", "shadow_rays", "shaded_hits", "box_tests", "tri_tests", "sphere_tests", "wave_node_steps")
print("per-ray counters equal:", all(sa[k] == sb[k] for k in keys), [(k, sa[k], sb[k]) for k in keys if sa[k] != sb[k]])
