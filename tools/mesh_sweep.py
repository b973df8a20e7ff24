"""AUTO on MESHES (coherent surfaces), not triangle soups: the pseudo-bunny tessellated to 70 k .. 2.2 M triangles, BVH depth
16 / 20, 1 / 4 / 16 spp at 1080p, every form, ms per frame.  usage: python tools/mesh_sweep.py [out.csv]   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import yahr_amd
from yahr_amd import scenes

FORMS = [("packet", yahr_amd.TRAVERSE_PACKET), ("hybrid", yahr_amd.TRAVERSE_HYBRID), ("stream", yahr_amd.TRAVERSE_STREAM),
         ("hybrid_stack", yahr_amd.TRAVERSE_HYBRID_STACK), ("auto", yahr_amd.TRAVERSE_AUTO)]
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
print("triangles,bvh_depth,nodes,largest_leaf,spp," + ",".join(n for n, _ in FORMS) + ",best", file=out, flush=True)
st = torch.cuda.current_stream()
for n_lon, n_lat in ((264, 133), (560, 281), (1056, 529), (1500, 751)):
    for depth in (16, 20):
        d = scenes.pseudo_bunny(1920, 1080, depth=2, n_lon=n_lon, n_lat=n_lat, bvh_depth=depth)
        scene = d.to_scene().build()
        a = scene.arrays()
        leaf = a["node_b"] >= 0
        ctx = yahr_amd.Context(scene, 0)
        buf = torch.zeros((1080, 1920, 3), device="cuda:0")
        for spp in (1, 4, 16):
            row = []
            for name, mode in FORMS:
                o = yahr_amd.make_opts(traverse=mode, spp=spp)
                ts = []
                for i in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    ctx.render_into(buf.data_ptr(), 0, 1080, o, st.cuda_stream)
                    e1.record(st)
                    torch.cuda.synchronize()
                    if i:
                        ts.append(e0.elapsed_time(e1))
                row.append(sorted(ts)[len(ts) // 2])
            best = FORMS[min(range(len(FORMS) - 1), key=lambda i: row[i])][0]
            print(f"{len(a['prim_orig'])},{depth},{len(a['node_a'])},{int(a['node_a'][leaf].max())},{spp}," + ",".join(f"{t:.2f}" for t in row) + f",{best}", file=out, flush=True)
        ctx.close()
