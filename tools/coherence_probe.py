"""Lane utilisation of the packet walker on 64 centre rows of the scene camera's frame (counted build): the statistic the AUTO
policy gates the wavefront form on.  Random soups against tessellated meshes.  usage: python tools/coherence_probe.py (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import yahr_amd
from yahr_amd import scenes

cases = [("soup 200k d16", scenes.random_tris(200_000, 1920, 1080, bvh_depth=16)), ("soup 300k d14", scenes.random_tris(300_000, 1920, 1080, bvh_depth=14)),
         ("soup 500k d20", scenes.random_tris(500_000, 1920, 1080, bvh_depth=20)),
         ("soup 1M d16", scenes.random_tris(1_000_000, 1920, 1080, bvh_depth=16)), ("soup 1M d20", scenes.random_tris(1_000_000, 1920, 1080, bvh_depth=20)),
         ("soup 4M d20", scenes.random_tris(4_000_000, 1920, 1080, bvh_depth=20)),
         ("mesh 70k d16", scenes.pseudo_bunny(1920, 1080)), ("mesh 314k d16", scenes.pseudo_bunny(1920, 1080, n_lon=560, n_lat=281)),
         ("mesh 314k d20", scenes.pseudo_bunny(1920, 1080, n_lon=560, n_lat=281, bvh_depth=20)),
         ("mesh 1.1M d16", scenes.pseudo_bunny(1920, 1080, n_lon=1056, n_lat=529)), ("mesh 1.1M d20", scenes.pseudo_bunny(1920, 1080, n_lon=1056, n_lat=529, bvh_depth=20)),
         ("mesh 2.2M d20", scenes.pseudo_bunny(1920, 1080, n_lon=1500, n_lat=751, bvh_depth=20))]
for name, d in cases:
    scene = d.to_scene().build()
    ctx = yahr_amd.Context(scene, 0)
    for rows in ((508, 572), (0, 1080)):
        ctx.render(rows[0], rows[1], traverse=yahr_amd.TRAVERSE_PACKET, collect_stats=True)
        st = ctx.stats()
        rays = st["primary_rays"] + st["reflection_rays"] + st["shadow_rays"]
        print(f"{name:14s} rows {rows}: node-step lanes {st['box_tests'] / (64.0 * max(1, st['wave_node_steps'])):.3f}  primitive-step lanes "
              f"{(st['tri_tests'] + st['sphere_tests']) / (64.0 * max(1, st['wave_prim_steps'])):.3f}  node steps per wave-ray {st['wave_node_steps'] / max(1, rays / 64):.0f}  "
              f"box tests per ray {st['box_tests'] / max(1, rays):.0f}  hit fraction {st['shaded_hits'] / max(1, st['primary_rays'] + st['reflection_rays']):.2f}  {st['kernel_ms']:.2f} ms", flush=True)
    ctx.close()
