"""Sanity of extreme image shapes on the GPU (8192x8192, 65536x8, 3x40000, 1x1): an oracle row of each, and the wavefront form
against the packet kernel bit for bit.  usage: python tools/big_probe.py   (GPU box)"""
import sys; sys.path[:0]=["/root/repo","/root/repo/tests"]
import numpy as np, yahr_amd
from yahr_amd import scenes
from oracle_py import OracleScene
import oracle_py; oracle_py.build_oracle()
for (w,h) in ((8192,8192),(65536,8),(3,40000),(1,1)):
    d = scenes.pseudo_bunny(w,h)
    scene = d.to_scene().build()
    ctx = yahr_amd.Context(scene,0)
    img = ctx.render().cpu().numpy()
    r0 = h//2 if h>2 else 0
    ref,_ = OracleScene.from_desc(d).build().render_rows(r0, min(h,r0+1), threads=16)
    err = np.abs(img[r0:r0+1]-ref); tol = 1e-4*np.maximum(np.abs(img[r0:r0+1]),np.abs(ref))+1e-6
    img2 = ctx.render(traverse=yahr_amd.TRAVERSE_STREAM).cpu().numpy() if w*h <= 1<<26 else None
    print(w,h,"bad px in oracle row:", int((err>tol).any(axis=2).sum()), "nonzero", float((img>0).mean()), "stream==packet", None if img2 is None else bool(np.array_equal(img.view(np.uint32), img2.view(np.uint32))), flush=True)
    ctx.close()
