"""counted statistics of single tile rows of the pseudo-bunny frame (what do the heaviest wavefronts do?).  usage: row_stats.py row..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import yahr_amd
from yahr_amd import scenes
scene = scenes.pseudo_bunny().to_scene().build()
ctx = yahr_amd.Context(scene, 0)
for r in [int(a) for a in sys.argv[1:]] or [40, 86, 100]:
    for c0, c1 in ((0, 240), (88, 128)):
        ctx.stats()
        # (a row range renders whole tile rows; columns cannot be cut, so the second figure is the whole row again -- kept for symmetry)
        ctx.render(row0=8 * r, row1=8 * r + 8, traverse=yahr_amd.TRAVERSE_PACKET, collect_stats=True)
        s = ctx.stats()
        w = 240
        print(f"tile row {r}: per wavefront: node steps {s['wave_node_steps'] / w:.0f}, primitive steps {s['wave_prim_steps'] / w:.0f} "
              f"(stage 1 {s['wave_stage1_steps'] / w:.0f}, stage 2 {s['wave_stage2_steps'] / w:.0f}), packet tests {s['wave_cull_steps'] / w:.1f}, "
              f"tri tests settled per packet {s['tri_tests_culled'] / max(1, s['tri_tests']):.2f} of {s['tri_tests'] / w / 64:.0f} per ray-lane, "
              f"rays {(s['primary_rays'] + s['reflection_rays'] + s['shadow_rays']) / w / 64:.2f} per pixel, longest wave {s['max_wave_cycles']} clk, mean {s['sum_wave_cycles'] / w:.0f}")
        break
