"""1 M random triangles in a depth-16 tree (34 per leaf) under AUTO, frame batches: timing of one context (the leaf culling
follows YAHR_NO_LEAF_CULL of the environment).  usage: d16_probe.py [frames] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch, yahr_amd
from yahr_amd import scenes
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
scene = scenes.random_tris(bvh_depth=16).to_scene().build()
ctx = yahr_amd.Context(scene, 0)
out = torch.zeros((F, scene.height, scene.width, 3), device="cuda:0")
st = torch.cuda.current_stream()
for frames in (1, F):
    o = yahr_amd.make_opts(traverse=yahr_amd.TRAVERSE_AUTO, frames=frames)
    ts = []
    for r in range(R + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); ctx.render_into(out.data_ptr(), 0, scene.height, o, st.cuda_stream); e1.record(st)
        torch.cuda.synchronize()
        if r >= 2: ts.append(e0.elapsed_time(e1) / frames)
    print(f"cull {'off' if os.environ.get('YAHR_NO_LEAF_CULL') else 'on'} frames {frames}: min {min(ts):.3f} median {sorted(ts)[len(ts)//2]:.3f} ms/frame", flush=True)
