"""Print the counted statistics of one frame (collect_stats build).  Usage: python tools/stats_probe.py [scene] [traverse]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import yahr_amd
from yahr_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
trav = {"auto": 0, "lane": 1, "packet": 2, "hybrid": 3}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
desc = {"bunny": scenes.pseudo_bunny, "spheres": scenes.spheres8, "tris": scenes.random_tris}[name]()
img, st = yahr_amd.render_image(desc.to_scene().build(), traverse=trav, collect_stats=True)
rays = st["primary_rays"] + st["reflection_rays"] + st["shadow_rays"]
for k, v in st.items():
    print(f"{k:22s} {v}")
print(f"per wave-ray (rays/64): node steps {st.get('wave_nodes', 0) / (rays / 64):.1f}, primitive steps {st.get('wave_prims', 0) / (rays / 64):.1f}")
