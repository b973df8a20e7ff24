"""The wavefront form at sample counts far beyond the benchmark's 33 M: 200 k triangles at 3840x2160 x 64 spp = 531 M paths
(queues cut into bands by the memory budget), and 7680x4320 x 16 spp; against the fused kernels bit for bit, with counters.
usage: python tools/big_stream_probe.py (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import yahr_amd
from yahr_amd import scenes

K = ("primary_rays", "reflection_rays", "shadow_rays", "shaded_hits", "box_tests", "tri_tests", "sphere_tests")
for (w, h, spp) in ((3840, 2160, 64), (7680, 4320, 16), (3840, 2160, 1)):
    d = scenes.random_tris(200_000, w, h, depth=2, bvh_depth=16)
    ctx = yahr_amd.Context(d.to_scene().build(), 0)
    res = {}
    for name, trav in (("packet", yahr_amd.TRAVERSE_PACKET), ("wavefront", yahr_amd.TRAVERSE_STREAM), ("wavefront + stack", yahr_amd.TRAVERSE_HYBRID_STACK),
                       ("adaptive", yahr_amd.TRAVERSE_HYBRID), ("auto", yahr_amd.TRAVERSE_AUTO)):
        t = time.perf_counter()
        img = ctx.render(traverse=trav, spp=spp, seed=3, collect_stats=True)
        torch.cuda.synchronize()
        st = ctx.stats()
        res[name] = (img, st)
        base_img, base_st = res["packet"]
        print(f"{w}x{h}x{spp} {name}: {st['kernel_ms']:.0f} ms counted, paths {w * h * spp / 1e6:.0f} M, rays {(st['primary_rays'] + st['reflection_rays'] + st['shadow_rays']) / 1e6:.0f} M, "
              f"== packet bit for bit {bool(torch.equal(img.view(torch.int32), base_img.view(torch.int32)))}, counters equal {all(st[k] == base_st[k] for k in K)}", flush=True)
        if name != "packet":
            del res[name]
        del img
        ms = []
        wall = []
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
        o = yahr_amd.make_opts(traverse=trav, spp=spp, seed=3)
        for rep in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            ctx.render_into(out.data_ptr(), 0, h, o, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - t) * 1e3)
            ms.append(ctx.stats()["kernel_ms"])
        del out
        print(f"{w}x{h}x{spp} {name}: {min(ms):.2f} ms uncounted (HIP events), {min(wall):.2f} ms wall", flush=True)
    ctx.close()
