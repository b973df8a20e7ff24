"""Do the launches of the wavefront form leave the chip idle in their tails?  Frames of the 1 M-triangle workload rendered
back to back on ONE stream against the same frames dealt to TWO streams (two contexts of the same scene: each has its own
queues).  usage: python tools/two_stream_probe.py [workload: tris1m | tris200k]   (GPU box)"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import yahr_amd
from yahr_amd import scenes

which = sys.argv[1] if len(sys.argv) > 1 else "tris1m"
d = scenes.random_tris(1_000_000, 1920, 1080, depth=2, bvh_depth=20) if which == "tris1m" else scenes.random_tris(200_000, 1920, 1080, depth=2, bvh_depth=16)
scene = d.to_scene().build()
ctxs = [yahr_amd.Context(scene, 0) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
bufs = [torch.zeros((1080, 1920, 3), device="cuda:0") for _ in range(2)]
o = yahr_amd.make_opts(spp=16)
N = 12


def run(n_streams):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(N):
        k = i % n_streams
        ctxs[k].render_into(bufs[k].data_ptr(), 0, 1080, o, streams[k].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / N * 1e3


for _ in range(2):
    run(1); run(2)
for n in (1, 2, 1, 2):
    print(f"{which}: {n} stream(s): {run(n):.3f} ms per frame", flush=True)
