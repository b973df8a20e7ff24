"""Where the VALU instructions of a pseudo-bunny frame go: renders variants of the scene (full, no reflection, no
light, ...) at slightly different heights (so that rocprofv3 --pmc rows can be told apart by grid size) and prints the
counted wave steps of each.  Run under:  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d out -- python3 tools/valu_breakdown.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch

import yahr_amd
from yahr_amd import scenes

base = scenes.pseudo_bunny()
variants = [("full depth2 1 light", base, 1080),
            ("depth1 1 light (primary+shadow)", base.with_(depth=1), 1072),
            ("depth1 no light (primary only)", base.with_(depth=1, lights=[]), 1064),
            ("depth2 no light (primary+reflection)", base.with_(lights=[]), 1056)]
for name, d, h in variants:
    d = d.with_camera(imH=h)
    scene = d.to_scene().build()
    ctx = yahr_amd.Context(scene, 0)
    out = ctx.render(traverse=yahr_amd.TRAVERSE_PACKET, collect_stats=True)
    torch.cuda.synchronize()
    st = ctx.stats()
    for _ in range(3):
        ctx.render(traverse=yahr_amd.TRAVERSE_PACKET)
    torch.cuda.synchronize()
    ms = ctx.stats()["kernel_ms"]
    tiles = ((1920 + 7) // 8) * ((h + 7) // 8)
    print(f"{name:40s} h={h} grid={tiles * 64:9d} rays {st['primary_rays']}/{st['reflection_rays']}/{st['shadow_rays']} shaded {st['shaded_hits']} "
          f"wave_nodes {st['wave_node_steps']} wave_prims {st['wave_prim_steps']} stage1 {st['wave_stage1_steps']} stage2 {st['wave_stage2_steps']} box {st['box_tests']} tri {st['tri_tests']} ms {ms:.3f}", flush=True)
    ctx.close()
