"""A frame batch whose 1-D launch would pass 2^32 work-items, at its real size (no lowered limit): 2 100 frames of the
8-sphere scene at 1920x1080 in the reference's squareBatches tiling = 70.9 M wavefront-sized workgroups (2^26 = 67.1 M).
Every frame must equal a lone frame bit for bit.  52 GB of output in HBM.  usage: python tools/grid_limit_probe.py (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import yahr_amd
from yahr_amd import scenes

d = scenes.spheres8(1920, 1080)
scene = d.to_scene().build()
ctx = yahr_amd.Context(scene, 0)
for tiles, name in ((yahr_amd.TILES_REFERENCE, "reference tiling (1-D grid)"), (yahr_amd.TILES_WAVE, "wave tiles (3-D grid)")):
    one = ctx.render(tiles=tiles)
    frames = 2100
    t = time.perf_counter()
    out = ctx.render(frames=frames, tiles=tiles)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    same = 0
    for f0 in range(0, frames, 100):
        blk = out[f0:f0 + 100].view(torch.int32)
        same += int((blk == one.view(torch.int32).unsqueeze(0)).flatten(1).all(dim=1).sum())
    print(f"{name}: {frames} frames in one call, {dt:.2f} s, frames bit-identical to a lone frame: {same} of {frames}; non-black {float((one > 0).float().mean()):.3f}", flush=True)
    del out
ctx.close()
# more than 2^31 workgroups in one call (3840x2160 x 64 spp: one pixel per wavefront, 8.3 M workgroups per frame, x 260 frames)
d = scenes.spheres8(3840, 2160)
ctx = yahr_amd.Context(d.to_scene().build(), 0)
for tiles, name, frames in ((yahr_amd.TILES_WAVE, "wave tiles (3-D grid)", 260), (yahr_amd.TILES_REFERENCE, "reference tiling (1-D grid)", 20)):
    one = ctx.render(tiles=tiles, spp=64, seed=1)
    t = time.perf_counter()
    out = ctx.render(frames=frames, tiles=tiles, spp=64, seed=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    same = sum(int(torch.equal(out[f].view(torch.int32), one.view(torch.int32))) for f in range(frames))
    print(f"3840x2160 x 64 spp, {name}: {frames} frames in one call ({frames * 8.2944:.0f} M workgroups if one pixel per wavefront), {dt:.2f} s, "
          f"frames bit-identical to a lone frame: {same} of {frames}", flush=True)
    del out
ctx.close()
