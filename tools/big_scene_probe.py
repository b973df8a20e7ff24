"""Scenes far beyond BASELINE's largest (1 M triangles): 16 M and 48 M random triangles.  Per size: the GPU-built tree against
the host builder's bit for bit, one oracle row against the packet kernel, the wavefront form and the stack walkers against the
packet kernel bit for bit with all counters, frame times.  Nothing here is a benchmark line: it is a check that record
offsets, queue sizes, scans and the level loop of the device build hold at sizes the 288 GB of one MI355X invite.
usage: python tools/big_scene_probe.py [millions ...]   (GPU box; 48 M needs ~40 GB of host memory)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import yahr_amd
from yahr_amd import scenes
import oracle_py
from oracle_py import OracleScene

oracle_py.build_oracle()
oracle_py.lib().oracle_set_shadow_any_hit(1)        # count a shadow ray's tests as the lazy `reachable` performs them (first hit ends it)
W, H = 1920, 1080
COUNTERS = ("primary_rays", "reflection_rays", "shadow_rays", "shaded_hits", "box_tests", "tri_tests", "sphere_tests")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


for millions in [int(a) for a in sys.argv[1:]] or [16, 48]:
    n = millions * 1_000_000
    depth = 24 if millions <= 16 else 26
    t = time.perf_counter()
    d = scenes.random_tris(n, W, H, depth=2, bvh_depth=depth)
    print(f"--- {millions} M triangles, BVH {depth} Midpoint: generated in {time.perf_counter() - t:.1f} s", flush=True)
    t = time.perf_counter()
    host = d.to_scene().set_builder(yahr_amd.BUILDER_HOST).build()
    t_host = time.perf_counter() - t
    t = time.perf_counter()
    dev = d.to_scene().set_builder(yahr_amd.BUILDER_DEVICE).build()
    t_dev = time.perf_counter() - t
    used, ms = dev.builder_used()
    f = dev.flat.contents
    print(f"{f.n_prims} primitives, {f.n_nodes} nodes ({f.n_nodes * 32 / 1e9:.2f} GB of node records, {f.n_prims * 96 / 1e9:.2f} GB of primitive "
          f"records); scene + host build {t_host:.1f} s, scene + device build {t_dev:.1f} s (builder {used}, {ms:.1f} ms on the device)", flush=True)
    ha, da = host.arrays(), dev.arrays()
    diff = [k for k in ha if not np.array_equal(bits(ha[k]), bits(da[k]))]
    print("device tree == host tree bit for bit:", not diff, diff, flush=True)
    del ha, da, host
    ctx = yahr_amd.Context(dev, 0)
    base = ctx.render(traverse=yahr_amd.TRAVERSE_PACKET, collect_stats=True).cpu().numpy()
    bst = ctx.stats()
    print(f"packet kernel (counted): {bst['kernel_ms']:.1f} ms, rays {bst['primary_rays'] + bst['reflection_rays'] + bst['shadow_rays']}, "
          f"box tests {bst['box_tests']}, triangle tests {bst['tri_tests']}, non-black {float((base > 0).any(axis=2).mean()):.3f}", flush=True)
    for name, trav in (("wavefront", yahr_amd.TRAVERSE_STREAM), ("wavefront + stack walker", yahr_amd.TRAVERSE_STACK),
                       ("wavefront + packet start + stack walker", yahr_amd.TRAVERSE_HYBRID_STACK), ("lane", yahr_amd.TRAVERSE_LANE)):
        img = ctx.render(traverse=trav, collect_stats=True).cpu().numpy()
        st = ctx.stats()
        print(f"{name}: image bit-identical {bool(np.array_equal(bits(img), bits(base)))}, counters equal "
              f"{all(st[k] == bst[k] for k in COUNTERS)}, {st['kernel_ms']:.1f} ms counted", flush=True)
    for name, trav in (("AUTO", yahr_amd.TRAVERSE_AUTO), ("packet", yahr_amd.TRAVERSE_PACKET), ("wavefront", yahr_amd.TRAVERSE_STREAM)):
        ctx.render(traverse=trav)
        ctx.render(traverse=trav)
        print(f"{name}: {ctx.stats()['kernel_ms']:.1f} ms per frame (uncounted)", flush=True)
    t = time.perf_counter()
    osc = OracleScene.from_desc(d).build()
    r0 = H // 2
    ref, ost = osc.render_rows(r0, r0 + 2, threads=16)
    err = np.abs(base[r0:r0 + 2] - ref)
    tol = 1e-4 * np.maximum(np.abs(base[r0:r0 + 2]), np.abs(ref)) + 1e-6
    print(f"oracle rows {r0}..{r0 + 1} ({time.perf_counter() - t:.1f} s incl. its build): pixels outside tolerance {int((err > tol).any(axis=2).sum())}, "
          f"bit-identical {float((bits(base[r0:r0 + 2]) == bits(ref)).all(axis=2).mean()):.4f}", flush=True)
    img2 = ctx.render(r0, r0 + 2, traverse=yahr_amd.TRAVERSE_PACKET, collect_stats=True).cpu().numpy()
    st2 = ctx.stats()
    print("counters of those rows equal the oracle's:", all(st2[k] == ost[k] for k in COUNTERS), [(k, st2[k], ost[k]) for k in COUNTERS if st2[k] != ost[k]],
          "; rows rendered alone == rows of the full frame:", bool(np.array_equal(bits(img2), bits(base[r0:r0 + 2]))), flush=True)
    ctx.close()
    del osc, dev, d
