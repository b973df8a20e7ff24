"""End-to-end wall time of the drop-in CLI on the benchmark scene: process start, .yahrr parse, BVH build, upload,
render, download, image encode.  Usage: python tools/cli_timing.py  (GPU box)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from yahr_amd import scenes

big = len(sys.argv) > 1 and sys.argv[1] == "tris1m"       # the 152 MB scene of BASELINE config 5 instead of the bunny
src = "/tmp/tris1m.yahrr" if big else "/tmp/bunny.yahrr"
open(src, "w").write((scenes.random_tris(1_000_000, 1920, 1080, depth=2, bvh_depth=20) if big else scenes.pseudo_bunny()).to_text())
print("scene file %.1f MB" % (os.path.getsize(src) / 1e6))
for ext, threads, builder in (("png", None, "auto"), ("png", None, "host"), ("ppm", None, "auto"), ("pfm", None, "auto"), ("ppm", "1", "auto")):
    for i in range(3):
        t = time.perf_counter()
        env = dict(os.environ, YAHR_TIMING="1", YAHR_BVH_BUILDER=builder)     # auto: the GPU builds the tree of a big scene
        if threads:
            env["YAHR_PARSE_THREADS"] = threads
        r = subprocess.run([os.path.join(ROOT, "bin", "yahr"), src, "/tmp/o." + ext, "-p", "hip"], capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t
        print("%s%s builder=%s run %d: %.3f s wall, rc %d | %s" % (ext, " (1 parse thread)" if threads else "", builder, i, dt, r.returncode,
                                                      " | ".join(r.stderr.strip().splitlines()[-2:])))
