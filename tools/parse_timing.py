"""Parse rate of the .yahrr loader on a big scene text (one TriangleMesh of n random triangles, and the same triangles as
n Triangle records), by thread count; the flattened scenes of all runs must be identical byte for byte.
usage: parse_timing.py [n triangles]"""
import os
import sys
import time

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np

import yahr_amd
from yahr_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = scenes.random_tris(n, 192, 108, depth=2, bvh_depth=20)
mesh_text = d.to_text().encode()
sc = yahr_amd.Scene.parse(mesh_text)
sc.build()
a = sc.arrays()
# the same triangles as loose Triangle records (every one carries a string: the material id)
p0, e1, e2 = a["prim_p0"], a["prim_e1"], a["prim_e2"]


def f(v):
    return "(%s)" % np.format_float_positional(np.float32(v), unique=True, trim="0") if v < 0 else np.format_float_positional(np.float32(v), unique=True, trim="0")


recs = []
for i in range(0, len(p0), max(1, len(p0) // 200_000)):          # 200 k records are enough for a rate
    q0, q1, q2 = p0[i], p0[i] + e1[i], p0[i] + e2[i]
    v = lambda x: "Vec3 %s %s %s" % (f(x[0]), f(x[1]), f(x[2]))
    recs.append("Triangle { p0 = %s, p1 = %s, p2 = %s, n0 = %s, n1 = %s, n2 = %s, materialId = \"tri\" }"
                % (v(q0), v(q1), v(q2), v(a["n0"][i]), v(a["n1"][i]), v(a["n2"][i])))
head = mesh_text[: mesh_text.index(b"objects = [")].decode()
rec_text = (head + "objects = [\n" + ",\n".join(recs) + "\n]\n}\n").encode()
for name, txt in (("one TriangleMesh", mesh_text), ("Triangle records", rec_text)):
    print(f"{name}: {len(txt) / 1e6:.1f} MB")
    ref = None
    for th in (1, 2, 4, 8, 16):
        os.environ["YAHR_PARSE_THREADS"] = str(th)
        best = 1e9
        for _ in range(3):
            t0 = time.time()
            sc = yahr_amd.Scene.parse(txt)
            best = min(best, time.time() - t0)
        sc.build()
        arr = sc.arrays()
        if ref is None:
            ref = arr
        else:
            for k in ref:
                assert np.array_equal(np.ascontiguousarray(ref[k]).view(np.uint8), np.ascontiguousarray(arr[k]).view(np.uint8)), k
        print(f"  threads {th:2d}: parse {best:.3f} s  ({len(txt) / 1e6 / best:.0f} MB/s)", flush=True)
