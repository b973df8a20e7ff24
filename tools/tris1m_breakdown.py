"""Counted wave steps of the 1 M-triangle workload by ray kind (camera rays only / + shadow / + reflection), in the
wavefront form: how full the packet walker's steps are (box tests per 64 lanes of a wave node step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch

import yahr_amd
from yahr_amd import scenes

base = scenes.random_tris(1_000_000, 1920, 1080, depth=2, bvh_depth=20)
for name, d in [("depth1 no light (camera rays)", base.with_(depth=1, lights=[])),
                ("depth1 1 light (+ shadow)", base.with_(depth=1)),
                ("depth2 no light (+ reflection)", base.with_(lights=[])),
                ("depth2 1 light (full)", base)]:
    scene = d.to_scene().build()
    ctx = yahr_amd.Context(scene, 0)
    for mode, mname in ((yahr_amd.TRAVERSE_STREAM, "stream"), (yahr_amd.TRAVERSE_LANE, "lane")):
        ctx.render(spp=16, traverse=mode, collect_stats=True)
        torch.cuda.synchronize()
        st = ctx.stats()
        for _ in range(2):
            ctx.render(spp=16, traverse=mode)
        torch.cuda.synchronize()
        ms = ctx.stats()["kernel_ms"]
        print(f"{name:32s} {mname:6s} rays {st['primary_rays']}/{st['reflection_rays']}/{st['shadow_rays']} wave_nodes {st['wave_node_steps']} "
              f"wave_prims {st['wave_prim_steps']} stage1 {st['wave_stage1_steps']} stage2 {st['wave_stage2_steps']} box {st['box_tests']} "
              f"tri {st['tri_tests']} ms {ms:.2f}", flush=True)
    ctx.close()
