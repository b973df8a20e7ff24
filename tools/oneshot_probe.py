"""PCIe-inclusive rate of the host-buffer entry point yahr_render_rows (scene upload + render + image download)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import yahr_amd
from yahr_amd import scenes
scene = scenes.pseudo_bunny().to_scene().build()
yahr_amd.render_image(scene)          # warm the runtime
ts = []
for _ in range(5):
    t0 = time.perf_counter(); img, st = yahr_amd.render_image(scene); ts.append(time.perf_counter() - t0)
print(f"one-shot yahr_render_rows 1080p pseudo-bunny: min {min(ts) * 1e3:.2f} ms wall (kernel {st['kernel_ms']:.3f} ms, scene upload {st['upload_ms']:.3f} ms)"
      f" -> {scene.width * scene.height / min(ts) / 1e6:.1f} Mpixels/s PCIe-inclusive")
