"""share of the packet walker's node steps that belong to traversals whose rays share ONE direction octant (needs the
library built with EXTRA=-DYD_OCT_PROBE: wave_cull_steps then counts those node steps; YAHR_NO_LEAF_CULL=1 keeps the
real culling counter out of it).  usage: oct_probe.py [bunny|spheres|bunny4|tris] [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
os.environ["YAHR_NO_LEAF_CULL"] = "1"
import yahr_amd
from yahr_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
base = {"bunny": scenes.pseudo_bunny, "spheres": scenes.spheres8, "bunny4": lambda: scenes.pseudo_bunny(depth=4), "tris": scenes.random_tris}[name]()
for tag, d in (("full", base), ("primary only", base.with_(depth=1, lights=[])), ("primary + shadow", base.with_(depth=1)), ("primary + reflection", base.with_(lights=[]))):
    ctx = yahr_amd.Context(d.to_scene().build(), 0)
    ctx.render(traverse=yahr_amd.TRAVERSE_PACKET, spp=spp, collect_stats=True)
    s = ctx.stats()
    print(f"{name} {tag:22s} node steps {s['wave_node_steps']:10d}  in octant-uniform traversals {s['wave_cull_steps']:10d} = {s['wave_cull_steps'] / max(1, s['wave_node_steps']):.3f}")
    ctx.close()
