import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import yahr_amd
from yahr_amd import scenes
W, H = int(sys.argv[1]), int(sys.argv[2])
d = scenes.random_tris(1_000_000, W, H, depth=2, bvh_depth=20)
scene = d.to_scene().build()
base = None
for m, name in ((yahr_amd.TRAVERSE_STREAM, "stream"), (yahr_amd.TRAVERSE_STACK, "stack"), (yahr_amd.TRAVERSE_STACK, "stack"), (yahr_amd.TRAVERSE_HYBRID_STACK, "hybrid_stack")):
    img, st = yahr_amd.render_image(scene, traverse=m, collect_stats=False, spp=16)
    if base is None: base = img
    diff = (img.view(np.uint32) != base.view(np.uint32)).any(axis=2)
    ys, xs = np.nonzero(diff)
    print(name, "differing px", int(diff.sum()), "sum", float(img.astype(np.float64).sum()), "rows", (ys.min(), ys.max()) if len(ys) else None, flush=True)
    if len(ys):
        for y, x in list(zip(ys, xs))[:5]:
            print("   ", y, x, img[y, x], base[y, x])
