"""Where does a LONE frame spend its time?  Needs the library built with EXTRA=-DYD_TILE_CLOCK: lane 0 of every wavefront
then writes (duration, start, hardware id) instead of its pixel, in ticks of the constant 100 MHz clock.
usage: tile_clock.py [bunny|spheres] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
import yahr_amd
from yahr_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scene = {"bunny": scenes.pseudo_bunny, "spheres": scenes.spheres8}[name]().to_scene().build()
ctx = yahr_amd.Context(scene, 0)
for rep in range(3):
    img = ctx.render(traverse=yahr_amd.TRAVERSE_PACKET, frames=frames)
torch.cuda.synchronize()
img = img.cpu().numpy().reshape(frames, scene.height, scene.width, 3)
t = img[:, ::8, ::8, :]                                    # lane 0 of every 8x8 tile
dur = t[..., 0].astype(np.float64) * 10e-3                 # microseconds
start = t[..., 1].astype(np.float64)
start = ((start - start.min()) % (1 << 24)) * 10e-3
end = start + dur
hw = t[..., 2].astype(np.int64)
T = end.max()
print(f"{name}: {dur.size} wavefronts, kernel span {T:.1f} us; wave duration mean {dur.mean():.1f} us, median {np.median(dur):.1f}, p90 {np.percentile(dur, 90):.1f}, p99 {np.percentile(dur, 99):.1f}, max {dur.max():.1f}")
print(f"sum of wave durations {dur.sum() / 1e3:.1f} ms = {dur.sum() / T:.0f} wave slots busy on average (of {256 * 28} at 7 waves per SIMD)")
# occupancy over time
edges = np.linspace(0, T, 21)
for a, b in zip(edges[:-1], edges[1:]):
    act = np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None).sum() / (b - a)
    started = ((start >= a) & (start < b)).sum()
    print(f"  {a:7.1f} - {b:7.1f} us: {act:7.0f} waves in flight, {started:6d} started")
# the last waves to finish
order = np.argsort(end.ravel())[::-1][:12]
ty, tx = np.unravel_index(order % (t.shape[1] * t.shape[2]), t.shape[1:3])
print("last to finish (tile row, tile col, start, duration, end):")
for o, y, x in zip(order, ty, tx):
    print(f"  row {y:3d} col {x:3d}  start {start.ravel()[o]:7.1f}  dur {dur.ravel()[o]:7.1f}  end {end.ravel()[o]:7.1f}")
rows = dur.reshape(frames, t.shape[1], t.shape[2]).mean(axis=(0, 2))
print("mean wave duration by tile row (us), every 8th row:", " ".join(f"{v:.0f}" for v in rows[::8]))

# solo speed: the heaviest tile rows rendered ALONE (240 wavefronts on 1024 SIMDs: every wave has its SIMD to itself)
full = dur.reshape(frames, t.shape[1], t.shape[2])[0]
for r in sorted(set(int(v) for v in ty[:6])):
    one = ctx.render(row0=8 * r, row1=8 * r + 8, traverse=yahr_amd.TRAVERSE_PACKET)
    one = ctx.render(row0=8 * r, row1=8 * r + 8, traverse=yahr_amd.TRAVERSE_PACKET).cpu().numpy()
    d1 = one[0, ::8, 0].astype(np.float64) * 10e-3
    k = np.argsort(full[r])[::-1][:5]
    print(f"tile row {r}: alone on their SIMDs the five heaviest waves take " + " ".join(f"{d1[c]:.0f}" for c in k) + " us; in the full frame " + " ".join(f"{full[r][c]:.0f}" for c in k))
