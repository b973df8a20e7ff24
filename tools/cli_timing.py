"""End-to-end wall time of the drop-in CLI on the benchmark scene: process start, .yahrr parse, BVH build, upload,
render, download, image encode.  Usage: python tools/cli_timing.py  (GPU box)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from yahr_amd import scenes

src = "/tmp/bunny.yahrr"
open(src, "w").write(scenes.pseudo_bunny().to_text())
print("scene file %.1f MB" % (os.path.getsize(src) / 1e6))
for ext in ("png", "ppm", "pfm"):
    for i in range(3):
        t = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "bin", "yahr"), src, "/tmp/o." + ext, "-p", "hip"], capture_output=True, text=True,
                           env=dict(os.environ, YAHR_TIMING="1"))
        dt = time.perf_counter() - t
        print("%s run %d: %.3f s wall, rc %d | %s" % (ext, i, dt, r.returncode, " | ".join(r.stderr.strip().splitlines()[-2:])))
