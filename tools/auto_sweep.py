"""Which traversal form wins where?  Random-triangle scenes (the generator of BASELINE config 5) of 50 k .. 4 M triangles x
BVH depth 16 / 20 x 1 / 4 / 16 / 64 samples per pixel at 1920x1080, every form timed (median of 3 launches after a
warm-up, HIP events); the table is what the AUTO policy of abi_device.cpp (auto_mode_for, the sample threshold in
yahr_ctx_render) is derived from.      usage: python tools/auto_sweep.py [out.csv]   (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch

import yahr_amd
from yahr_amd import scenes

FORMS = [("lane", yahr_amd.TRAVERSE_LANE), ("packet", yahr_amd.TRAVERSE_PACKET), ("hybrid", yahr_amd.TRAVERSE_HYBRID),
         ("stream", yahr_amd.TRAVERSE_STREAM), ("hybrid_stack", yahr_amd.TRAVERSE_HYBRID_STACK), ("auto", yahr_amd.TRAVERSE_AUTO)]
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
sizes = [int(x) for x in os.environ.get("SWEEP_N", "50000,200000,1000000,4000000").split(",")]
print("triangles,bvh_depth,nodes,largest_leaf,spp," + ",".join(n for n, _ in FORMS) + ",best", file=out, flush=True)
st = torch.cuda.current_stream()
for n in sizes:
    for depth in [int(x) for x in os.environ.get("SWEEP_DEPTHS", "16,20").split(",")]:
        d = scenes.random_tris(n, 1920, 1080, depth=2, bvh_depth=depth)
        scene = d.to_scene().build()
        a = scene.arrays()
        leaf = a["node_b"] >= 0
        ctx = yahr_amd.Context(scene, 0)
        buf = torch.zeros((1080, 1920, 3), device="cuda:0")
        for spp in [int(x) for x in os.environ.get("SWEEP_SPP", "1,4,16,64").split(",")]:
            row = []
            for name, mode in FORMS:
                o = yahr_amd.make_opts(traverse=mode, spp=spp)
                ts = []
                t_wall = time.time()
                for i in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    ctx.render_into(buf.data_ptr(), 0, 1080, o, st.cuda_stream)
                    e1.record(st)
                    torch.cuda.synchronize()
                    if i:
                        ts.append(e0.elapsed_time(e1))
                    if time.time() - t_wall > 20:            # a hopeless form on a big scene: one sample is enough
                        break
                row.append(sorted(ts)[len(ts) // 2] if ts else float("nan"))
            best = FORMS[min(range(len(FORMS) - 1), key=lambda i: row[i])][0]
            print(f"{n},{depth},{len(a['node_a'])},{int(a['node_a'][leaf].max())},{spp}," + ",".join(f"{t:.2f}" for t in row) + f",{best}",
                  file=out, flush=True)
        ctx.close()
        del buf
