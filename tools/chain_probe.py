"""Critical-path probe: longest wave vs kernel time (collect_stats build records per-wave shader clocks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch, yahr_amd
from yahr_amd import scenes
scene = scenes.pseudo_bunny().to_scene().build()
ctx = yahr_amd.Context(scene, 0)
out = torch.zeros((scene.height, scene.width, 3), device="cuda:0")
st = torch.cuda.current_stream()
for name, tr in (("lane", yahr_amd.TRAVERSE_LANE), ("packet", yahr_amd.TRAVERSE_PACKET), ("hybrid", yahr_amd.TRAVERSE_HYBRID)):
    o = yahr_amd.make_opts(traverse=tr, collect_stats=True)
    for _ in range(3):
        ctx.render_into(out.data_ptr(), 0, scene.height, o, st.cuda_stream)
        torch.cuda.synchronize()
        s = ctx.stats()
    n_waves = ((scene.width + 7) // 8) * ((scene.height + 7) // 8)
    steps = s["wave_node_steps"] + s["wave_prim_steps"]
    print(f"{name:7s} kernel {s['kernel_ms']:.3f} ms | longest wave {s['max_wave_cycles']} clk, mean wave {s['sum_wave_cycles'] / n_waves:.0f} clk, "
          f"sum {s['sum_wave_cycles'] / 1e6:.1f} Mclk | wave steps {steps} ({steps / n_waves:.0f}/wave)")
