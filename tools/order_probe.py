"""What a near-first visiting order could save on the benchmark scenes: box / primitive tests of the secondary rays in the
reference's order (wavefront form, stack walker) against the opt-in near-first order (same walker, YAHR_TRAVERSE_ORDERED).
Camera rays are walked by the packet walker in the reference's order in both.  usage: python tools/order_probe.py (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import yahr_amd
from yahr_amd import scenes

cases = [("bunny 70k", scenes.pseudo_bunny(1920, 1080)), ("bunny 70k depth 0", scenes.pseudo_bunny(1920, 1080, depth=0)),
         ("mesh 314k d20", scenes.pseudo_bunny(1920, 1080, n_lon=560, n_lat=281, bvh_depth=20)),
         ("spheres8", scenes.spheres8(1920, 1080)),
         ("soup 200k d16", scenes.random_tris(200_000, 1920, 1080, bvh_depth=16))]
for name, d in cases:
    scene = d.to_scene().build()
    ctx = yahr_amd.Context(scene, 0)
    rows = {}
    for label, trav in (("reference order", yahr_amd.TRAVERSE_STACK), ("near first", yahr_amd.TRAVERSE_ORDERED), ("packet", yahr_amd.TRAVERSE_PACKET)):
        ctx.render(0, scene.height, traverse=trav, collect_stats=True)
        st = ctx.stats()
        rays = st["primary_rays"] + st["reflection_rays"] + st["shadow_rays"]
        rows[label] = st
        print(f"{name:18s} {label:16s}: rays {rays}  (camera {st['primary_rays']}, reflection {st['reflection_rays']}, shadow {st['shadow_rays']})  "
              f"box tests {st['box_tests']}  primitive tests {st['tri_tests'] + st['sphere_tests']}  hits {st['shaded_hits']}", flush=True)
    a, b = rows["reference order"], rows["near first"]
    print(f"{name:18s} near first / reference order: box tests {b['box_tests'] / max(1, a['box_tests']):.3f}, primitive tests "
          f"{(b['tri_tests'] + b['sphere_tests']) / max(1, a['tri_tests'] + a['sphere_tests']):.3f}", flush=True)
    ctx.close()
