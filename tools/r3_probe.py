import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import yahr_amd
from yahr_amd import scenes
d = scenes.random_tris(200000, 480, 270, bvh_depth=18)
scene = d.to_scene().build()
base = None
for stats in (True, False):
    for m, name in ((yahr_amd.TRAVERSE_STREAM, "stream"), (yahr_amd.TRAVERSE_STACK, "stack"), (yahr_amd.TRAVERSE_HYBRID_STACK, "hybrid_stack"), (yahr_amd.TRAVERSE_ORDERED, "ordered")):
        for spp in (1, 4):
            img, st = yahr_amd.render_image(scene, traverse=m, collect_stats=stats, spp=spp)
            key = (spp,)
            if base is None: base = {}
            if key not in base: base[key] = img
            diff = (img.view(np.uint32) != base[key].view(np.uint32)).any(axis=2)
            print(name, "stats", stats, "spp", spp, "differing px", int(diff.sum()), "sum", float(img.astype(np.float64).sum()), st["box_tests"])
