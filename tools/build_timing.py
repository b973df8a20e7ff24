"""Host-side scene build timing (expand, boxes, BVH, records) for the 1 M-triangle scene, sequential vs concurrent
subtree builds.  Usage: python tools/build_timing.py   (prints the library's YAHR_BUILD_TIMING laps)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.path.insert(0, %r); import time; from yahr_amd import scenes; "
        "d = scenes.random_tris(1000000, 1920, 1080, bvh_depth=20); d.split = sys.argv[1]; s = d.to_scene(); "
        "t = time.time(); s.build(); print('build() %%.3f s' %% (time.time() - t))" % ROOT)
for split in ("Midpoint", "SurfaceAreaHeuristic"):
    for par in ("0", "1", "2", "3", "4", "5"):
        r = subprocess.run([sys.executable, "-c", code, split], capture_output=True, text=True,
                           env=dict(os.environ, YAHR_BUILD_PAR=par, YAHR_BUILD_TIMING="1"))
        laps = [l.split()[-2] for l in r.stderr.splitlines() if "host build" in l]
        print(split, "par", par, "| laps(ms): expand boxes centroids tree records =", " ".join(laps), "|", r.stdout.strip())
