"""Row ranges that are not multiples of the 8-row tile, against the oracle: pixels and all seven counters, four forms.
usage: python tools/partial_rows_check.py (GPU box)"""
import sys
sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, yahr_amd
from yahr_amd import scenes
from oracle_py import OracleScene
import oracle_py; oracle_py.build_oracle()
oracle_py.lib().oracle_set_shadow_any_hit(1)        # count a shadow ray's tests as the lazy `reachable` performs them (first hit ends it)
K=("primary_rays","reflection_rays","shadow_rays","shaded_hits","box_tests","tri_tests","sphere_tests")
for name,d in (("bunny", scenes.pseudo_bunny(203,117)), ("soup", scenes.random_tris(200000, 320, 180, depth=2, bvh_depth=16))):
    scene = d.to_scene().build(); ctx = yahr_amd.Context(scene,0)
    osc = OracleScene.from_desc(d).build()
    h = scene.height
    for (r0,r1) in ((h//2, h//2+2), (h//2+1, h//2+2), (8,16), (5,29), (0,h)):
        ref, ost = osc.render_rows(r0,r1,threads=16)
        for tn,trav in (("packet",yahr_amd.TRAVERSE_PACKET),("lane",yahr_amd.TRAVERSE_LANE),("stream",yahr_amd.TRAVERSE_STREAM),("auto",yahr_amd.TRAVERSE_AUTO)):
            img = ctx.render(r0,r1,traverse=trav,collect_stats=True).cpu().numpy(); st = ctx.stats()
            err=np.abs(img-ref); tol=1e-4*np.maximum(np.abs(img),np.abs(ref))+1e-6
            bad=[(k,st[k],ost[k]) for k in K if st[k]!=ost[k]]
            print(name,(r0,r1),tn,"bad px",int((err>tol).any(axis=2).sum()),"counter mismatches",bad,flush=True)
    ctx.close()
