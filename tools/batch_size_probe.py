"""ms per frame of one yahr_ctx_render call against the number of frames in it (fused kernel; kernel_ms of yahr_ctx_stats).
usage: python tools/batch_size_probe.py (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import yahr_amd
from yahr_amd import scenes

for name, d in (("spheres8", scenes.spheres8(1920, 1080)), ("bunny", scenes.pseudo_bunny(1920, 1080))):
    ctx = yahr_amd.Context(d.to_scene().build(), 0)
    for tiles, tname in ((yahr_amd.TILES_WAVE, "wave tiles"), (yahr_amd.TILES_REFERENCE, "reference tiling")):
        row = []
        for frames in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048):
            out = torch.empty((frames, 1080, 1920, 3), dtype=torch.float32, device="cuda:0")
            o = yahr_amd.make_opts(frames=frames, tiles=tiles)
            best = 1e9
            for rep in range(3):
                ctx.render_into(out.data_ptr(), 0, 1080, o, torch.cuda.current_stream().cuda_stream)
                best = min(best, ctx.stats()["kernel_ms"] / frames)
            row.append("%d: %.3f" % (frames, best))
            del out
        print(f"{name} {tname}: ms per frame by frames per call -- " + ", ".join(row), flush=True)
    ctx.close()
