"""What a one-process host gets from yahr_multi (the C ABI's multi-GPU object) on the benchmark scene: frame batches through
yahr_multi_submit / yahr_multi_wait into page-locked host memory, two submissions in flight -- i.e. INCLUDING the download of
every frame over PCIe, which bench.py's device-resident figure leaves out.  usage: python tools/multi_probe.py [entries] (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import yahr_amd
from yahr_amd import scenes
sys.path.insert(0, ROOT)
import bench
bench._import_runtime()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = scenes.pseudo_bunny(1920, 1080, depth=2)
scene = d.to_scene().build()
F = 8
m = yahr_amd.MultiContext(scene, n, devices=[0] * n, gather=yahr_amd.GATHER_PEER if n > 1 else yahr_amd.GATHER_AUTO)
m.set_cameras(bench.orbit_cameras(d.camera, (0.2, 0.9, 0.0), F))
bufs = [m._out(F, True), m._out(F, True)]
o = yahr_amd.make_opts(frames=F)
import ctypes as C


def submit(i):
    yahr_amd.check(yahr_amd.lib.yahr_multi_submit(m._h, C.byref(o), bufs[i & 1].ctypes.data_as(C.POINTER(C.c_float))))


def wait():
    yahr_amd.check(yahr_amd.lib.yahr_multi_wait(m._h, None))


for rounds in (4, 40):
    t = time.perf_counter()
    submit(0)
    for i in range(1, rounds):
        submit(i)
        wait()
    wait()
    dt = time.perf_counter() - t
    print(f"{n} entr{'y' if n == 1 else 'ies'}, pipelined submit / wait, {F} frames per batch: {dt / (rounds * F) * 1e3:.3f} ms per frame "
          f"({rounds * F / dt:.0f} frames/s, {rounds * F * 24.9e-3 / dt:.1f} GB/s of pixels to the host)")
t = time.perf_counter()
for i in range(20):
    submit(i); wait()
dt = time.perf_counter() - t
print(f"  one batch at a time (yahr_multi_render): {dt / (20 * F) * 1e3:.3f} ms per frame; checksum {float(bufs[0][0].astype(np.float64).sum()):.3f}")
m.close()
