"""Exhaustive check of the device's `**` (device_math.hpp: glibc_powf) against the host C library's powf: EVERY binary32 x
in [0, 1] (the cosine the Blinn term raises, 1 065 353 217 values) for each given exponent, bit for bit.
usage: powf_sweep.py [exponent ...]        (default: 1 2 5 10 20 50 100 200 1000 0.5 17.3)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import oracle_py
from yahr_amd._lib import lib

oracle_py.build_oracle()
L = oracle_py.lib()
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
exps = [float(v) for v in sys.argv[1:]] or [1, 2, 5, 10, 20, 50, 100, 200, 1000, 0.5, 17.3]
top = int(np.float32(1.0).view(np.uint32)) + 1
chunk = 1 << 26
total_bad = 0
for e in exps:
    t0 = time.time()
    bad = 0
    for lo in range(0, top, chunk):
        hi = min(top, lo + chunk)
        x = np.arange(lo, hi, dtype=np.uint32).view(np.float32)
        y = np.full(len(x), e, np.float32)
        ref = np.empty(len(x), np.float32); got = np.empty(len(x), np.float32)
        L.oracle_powf_array(len(x), fp(x), fp(y), fp(ref))
        assert lib.yahr_debug_powf(0, len(x), fp(x), fp(y), fp(got)) == 0
        bad += int((ref.view(np.uint32) != got.view(np.uint32)).sum())
    total_bad += bad
    print(f"exponent {e:g}: {top} values of x in [0, 1], {bad} mismatches ({time.time() - t0:.0f} s)", flush=True)
print(f"done: {len(exps)} exponents x {top} values, {total_bad} mismatches")
sys.exit(1 if total_bad else 0)
