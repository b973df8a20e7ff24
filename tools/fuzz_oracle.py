"""One-off: tests/test_gpu_parity.py's oracle comparison (pixels within 1e-4 rel + 1e-6, every counter exactly equal, the
six forms bit-identical) on many more seeded random scenes than the suite holds.
usage: fuzz_oracle.py [first seed] [number of seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_py

oracle_py.build_oracle()
import test_gpu_parity as T
from random_scenes import random_scene_desc

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for seed in range(first, first + count):
    d = random_scene_desc(seed)
    try:
        T.assert_parity(d)
        T.assert_parity(d, modes=[T.yahr_amd.TRAVERSE_AUTO, T.yahr_amd.TRAVERSE_STREAM, T.yahr_amd.TRAVERSE_HYBRID_STACK], spp=4, seed=seed)
    except AssertionError as e:
        bad += 1
        print(f"seed {seed}: {e}", flush=True)
    if seed % 25 == 0:
        print(f"seed {seed} done (failures: {bad})", flush=True)
print(f"done: {count} seeds, {bad} failures")
sys.exit(1 if bad else 0)
