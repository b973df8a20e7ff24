"""Prints DESIGN.md §5's table of current figures from the committed bench lines (profiles/<tag>_final_bench*.json), so that
the table is copied, not retyped.  usage: figures_table.py [round tag, default r04] [previous round's tag for the (..) figures]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
prev = sys.argv[2] if len(sys.argv) > 2 else "r03"
ROWS = [("", "`bunny` (config 3, **the metric**)"), ("_driver_args", "`bunny`, the driver's arguments (`--steps 20 --warmup 5`)"),
        ("_spheres", "`spheres` (config 2)"), ("_bunny4k", "`bunny4k` (config 4: 3840×2160, 64 spp, depth 4)"),
        ("_tris1m", "`tris1m` (config 5: 1 M triangles, 16 spp)"), ("_scene_fixed", "`scene_fixed` (config 1: 256×256, plumbing)")]


def load(t, w):
    p = os.path.join(ROOT, "profiles", f"{t}_final_bench{w}.json")
    return json.load(open(p)) if os.path.exists(p) else None


def sp(x):
    return f"{x:,.0f}".replace(",", " ")


print("| workload | Mrays/s | Mpixels/s | ms per frame | lone frame ms | VALU-issue frac (of guide / of measured peak) | HBM frac | "
      "§8(d) byte-model \"frac\" | lanes busy (node / primitive steps) | CPU oracle, Mrays/s (threads) | same-run parity (pixels out / compared / bit-identical) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for w, name in ROWS:
    d = load(tag, w)
    if d is None:
        continue
    o = load(prev, w)
    r, cb, pa = d["roofline"], d.get("cpu_baseline") or {}, d.get("parity") or {}
    was = lambda k, f: f" ({f(o[k])})" if o and k in o else ""
    frac = f"{r['frac']:.3f} / {r['frac_of_measured_issue_peak']:.3f}" if r.get("frac") is not None else "–"
    hbm = f"{r['hbm']['frac']:.3f}" if r.get("hbm") else "–"
    lu = r.get("lane_utilisation") or {}
    print(f"| {name} | {sp(d['value'])}{was('value', sp)} | {sp(d['mpixels_per_s'])} | {d['ms_per_step']:.4g}{was('ms_per_step', lambda v: f'{v:.4g}')} | "
          f"{d['single_frame_ms']:.3f}{was('single_frame_ms', lambda v: f'{v:.3f}')} | {frac} | {hbm} | {r.get('survey_8d_frac', 0):.3g} | "
          f"{lu.get('node_steps', 0):.2f} / {lu.get('primitive_steps', 0):.2f} | {cb.get('value', 0):.1f} ({cb.get('cores', '?')}) | "
          f"{pa.get('pixels_outside_tolerance', '?')} / {sp(pa.get('pixels', 0))} / {sp(pa.get('bit_identical_pixels', 0))} |")
