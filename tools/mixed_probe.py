"""AUTO on other kinds of scenes than the two sweeps hold: 200 k spheres, a tessellated mesh inside a cloud of loose triangles, a mesh
under three lights at recursion depth 4 -- every form at 1 / 4 / 16 spp, 1080p, ms per frame.  usage: python tools/mixed_probe.py (GPU box)"""
import os, sys, time
sys.path[:0] = ["/root/repo"]
import numpy as np, torch, yahr_amd
from yahr_amd import scenes
FORMS = [("packet", yahr_amd.TRAVERSE_PACKET), ("hybrid", yahr_amd.TRAVERSE_HYBRID), ("stream", yahr_amd.TRAVERSE_STREAM), ("hybrid_stack", yahr_amd.TRAVERSE_HYBRID_STACK), ("auto", yahr_amd.TRAVERSE_AUTO)]
rng = np.random.default_rng(1)
soup = scenes.random_tris(300_000, 1920, 1080, bvh_depth=20)
bun = scenes.pseudo_bunny(1920, 1080, n_lon=560, n_lat=281, bvh_depth=20)
sph = [("sphere", tuple(rng.uniform(-10, 10, 3)), 0.05, "tri") for _ in range(200_000)]
# the soup shrunk around the bunny: a mesh inside a cloud
m = soup.objects[0]
pts = (np.asarray(m[1], dtype=np.float64) * 0.3 + np.array([0.2, 1.0, 0.0])).astype(np.float32)
cloud = ("mesh", pts, None, m[3], None, "bunny")
cases = [("200k spheres d20", soup.with_(objects=sph)),
         ("mesh 314k + cloud 300k d20", bun.with_(objects=bun.objects + [cloud])),
         ("mesh 314k, 3 lights, depth 4", bun.with_(depth=4, lights=bun.lights + [((5, 6, -4), (60, 60, 60)), ((0, 12, 3), (80, 80, 80))]))]
st = torch.cuda.current_stream()
for name, d in cases:
    scene = d.to_scene().build()
    ctx = yahr_amd.Context(scene, 0)
    buf = torch.zeros((1080, 1920, 3), device="cuda:0")
    for spp in (1, 4, 16):
        row = []
        for fname, mode in FORMS:
            o = yahr_amd.make_opts(traverse=mode, spp=spp)
            ts = []
            for i in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); ctx.render_into(buf.data_ptr(), 0, 1080, o, st.cuda_stream); e1.record(st); torch.cuda.synchronize()
                if i: ts.append(e0.elapsed_time(e1))
            row.append(min(ts))
        print(f"{name:30s} spp {spp:2d}: " + "  ".join(f"{n} {t:7.2f}" for (n, _), t in zip(FORMS, row)) + f"   auto/best {row[-1] / min(row[:-1]):.3f}", flush=True)
    ctx.close()
