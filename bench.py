#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json: Mrays/s (+ Mpixels/s) on the bunny scene at
1920x1080, on N GPUs of one node, beside the CPU baseline.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one render of the whole frame: every rank renders its row strips (scene resident in its HBM
before the timed region), a gather over RCCL brings the strips to rank 0, rank 0 scatters them into the
image.  Total work is fixed as N grows ("scaling": "strong").  Frames are rendered in batches of F per kernel
launch (--frames-per-launch, default 8; tiles of the F frames interleaved) with ONE gather per launch: a single
1080p frame is bounded by the serial chain of its heaviest tile (~0.9 ms), not by throughput, so the tail of one
frame is overlapped with the other frames of the batch.  K steps = K/F launches; every frame is fully rendered,
gathered and assembled inside the timed region.  Workload: BASELINE.json configs[2] -- the
metric's own configuration -- with the documented stand-in mesh: the reference ships no bunny mesh
(SURVEY finding 3), so the scene is the procedural 69,696-triangle "pseudo-bunny" of yahr_amd/scenes.py on the
+-1000 plane, 1 light, 1 spp, WhittedIntegrator depth 2, BVH 16 Midpoint.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field's definition).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT]

import torch
import torch.distributed as dist

import yahr_amd
from yahr_amd import scenes
from yahr_amd.dist import ShardPipeline, ShardPlan

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s

WORKLOADS = {
    "bunny": lambda: (scenes.pseudo_bunny(1920, 1080, depth=2),
                      "pseudo-bunny (procedural stand-in, 69,696 tris + 2-tri plane), 1920x1080, 1 spp, "
                      "Whitted depth 2, BVH 16 Midpoint, 1 point light"),
    "spheres": lambda: (scenes.spheres8(1920, 1080, depth=1),
                        "8 spheres + plane + 2 point lights, 1920x1080, 1 spp, Whitted depth 1, BVH 16 Midpoint"),
    "bunny_small": lambda: (scenes.pseudo_bunny(480, 270, depth=2), "pseudo-bunny 480x270 (smoke size)"),
    # BASELINE.json configs[3] and [4]; spp > 1 and "stratified" are build extensions (sampler)
    "bunny4k": lambda: (scenes.pseudo_bunny(3840, 2160, depth=4),
                        "pseudo-bunny 3840x2160, 64 spp stratified, Whitted depth 4 (primary + 3 reflection segments)"),
    "tris1m": lambda: (scenes.random_tris(1_000_000, 1920, 1080, depth=2, bvh_depth=20),
                       "1M random triangles (splitmix64 seed 0x5EED), 1920x1080, 16 spp, Whitted depth 2, BVH 20 Midpoint"),
}
WORKLOAD_SPP = {"bunny4k": 64, "tris1m": 16}


def algorithmic_bytes(st, n_lights, n_pixels):
    """32 B per box test, 72 B per triangle test, 16 B per sphere test, (32 + 24 L) B per shaded
    hit, 12 B per output pixel."""
    return (32 * st["box_tests"] + 72 * st["tri_tests"] + 16 * st["sphere_tests"]
            + (32 + 24 * n_lights) * st["shaded_hits"] + 12 * n_pixels)


def cpu_baseline(desc, cores, spp=1, min_cpu_seconds=10.0):
    """The oracle (CPU restatement of the reference, kind "port": the Haskell reference cannot be built here)
    timed on the host cores over the same workload: whole frames, repeated until ~min_cpu_seconds of CPU work
    (spp == 1), or a bounded sample of evenly spaced 8-row strips of one frame (supersampled workloads)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py                      # the checker, timed beside the product; never part of the product path
    osc = oracle_py.OracleScene.from_desc(desc).build()
    reps, wall, rays, st = 0, 0.0, 0, None
    if spp > 1:
        w, h = osc.size()
        strips = list(range(0, h - 7, 8))
        order = [strips[(i * 37) % len(strips)] for i in range(len(strips))]      # spread over the image
        prim, done = 0, 0
        for r0 in order:
            t0 = time.perf_counter()
            _, st = osc.render_rows(r0, r0 + 8, spp=spp, seed=0, threads=cores)
            wall += time.perf_counter() - t0
            rays += st["primary_rays"] + st["reflection_rays"] + st["shadow_rays"]
            prim += st["primary_rays"]
            done += 1
            if wall * cores >= min_cpu_seconds:
                break
        return dict(value=round(rays / wall / 1e6, 3), unit="Mrays/s", cores=cores, kind="port",
                    sample=f"{done} strips of 8 rows (of {len(strips)}) spread over one frame at {spp} spp, render only, "
                           f"{wall:.2f} s wall on {cores} threads (one thread per image row)",
                    mpixels_per_s=round(prim / spp / wall / 1e6, 4)), None
    while wall * cores < min_cpu_seconds and reps < 50:
        t0 = time.perf_counter()
        _, st = osc.render(threads=cores)
        wall += time.perf_counter() - t0
        reps += 1
        rays += st["primary_rays"] + st["reflection_rays"] + st["shadow_rays"]
    return dict(value=round(rays / wall / 1e6, 3), unit="Mrays/s", cores=cores, kind="port",
                sample=f"{reps} full frame(s) of the same workload, render only (scene build excluded), "
                       f"{wall:.2f} s wall on {cores} threads over the reference's squareBatches tiles",
                mpixels_per_s=round(reps * st["primary_rays"] / wall / 1e6, 3)), st


def load_traffic(workload):
    """HBM bytes per launch from rocprofv3 PMC passes (profiles/, collected separately as the microarch guide
    prescribes; FETCH_SIZE already doubled for gfx950).  None when no profile matches this workload."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        j = json.load(open(p))
        if j.get("workload") == workload:
            return j.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def load_valu_profile(workload, frames_per_launch):
    """(VALU wave-instructions per launch, measured VALU issue peak in G instr/s) from the committed PMC profile,
    valid for the profiled configuration only (8 frames per launch, 1 GPU); else None."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        j = json.load(open(p))
        if j.get("workload") == workload and frames_per_launch == 8 and "valu_insts_per_launch" in j:
            return j["valu_insts_per_launch"], j["valu_issue_peak_ginst_s"]
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="bunny", choices=sorted(WORKLOADS))
    ap.add_argument("--traverse", default="auto", choices=["auto", "lane", "packet", "hybrid"])
    ap.add_argument("--strip-rows", type=int, default=8)
    ap.add_argument("--frames-per-launch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="finish gather + assembly of a launch before the next render starts (diagnostics)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "YAHR_BENCH_FORCE_DEVICE" in os.environ:       # rehearsal of N ranks on one GPU (diagnostics only)
        local_rank = int(os.environ["YAHR_BENCH_FORCE_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU: yahr_amd has no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    desc, workload_name = WORKLOADS[args.workload]()
    t0 = time.perf_counter()
    scene = desc.to_scene().build()
    build_s = time.perf_counter() - t0
    ctx = yahr_amd.Context(scene, local_rank)
    w, h = scene.width, scene.height
    trav = {"auto": yahr_amd.TRAVERSE_AUTO, "lane": yahr_amd.TRAVERSE_LANE, "packet": yahr_amd.TRAVERSE_PACKET,
            "hybrid": yahr_amd.TRAVERSE_HYBRID}[args.traverse]
    # frames per launch: the largest divisor of K not above the request, so that EXACTLY K frames are timed
    # (the request scales with the rank count: every rank then gets as many tiles per launch as in the 1-GPU case,
    # so the per-launch tail stays amortised and the gathers get larger and fewer
    # ... but never fewer than ~8 launches when K allows it: the gather + assembly of the LAST launch cannot hide
    # behind a render, so it must stay a small part of the timed region)
    F = max(1, min(args.frames_per_launch * world, max(args.frames_per_launch, args.steps // 8), args.steps))
    while args.steps % F:
        F -= 1
    n_launches = args.steps // F
    plan = ShardPlan(h, w, world, args.strip_rows, device=dev, frames=F)
    # double-buffered: gather + assembly of launch i overlap the render of launch i+1 (yahr_amd/dist.py)
    pipe = ShardPipeline(plan, rank)
    bufs = pipe.slots[0]
    stream = torch.cuda.current_stream()
    spp = WORKLOAD_SPP.get(args.workload, 1)
    o_run = yahr_amd.make_opts(traverse=trav, n_shards=world, shard=rank, strip_rows=args.strip_rows, frames=F,
                               frame_rows=plan.max_rows, spp=spp)
    o_stat = yahr_amd.make_opts(traverse=trav, n_shards=world, shard=rank, strip_rows=args.strip_rows, collect_stats=True,
                                spp=spp)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_launches)]
    step_no = [-1]

    def render_fn(shard, n_shards, strip_rows, out):
        # launched on torch's current stream, so torch.cuda.Event brackets exactly this kernel
        i = step_no[0]
        if i >= 0:
            ev[i][0].record(stream)
        ctx.render_into(out.data_ptr(), 0, h, o_run, stream.cuda_stream)
        if i >= 0:
            ev[i][1].record(stream)

    # ---- one counted single-frame launch per rank (deterministic counters; not timed)
    ctx.render_into(bufs[0].data_ptr(), 0, h, o_stat, stream.cuda_stream)
    torch.cuda.synchronize()
    st = ctx.stats()
    my = torch.tensor([st["primary_rays"], st["reflection_rays"], st["shadow_rays"]], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(my)
    rays_per_step = int(my.sum().item())
    primary_per_step = int(my[0].item())

    # ---- warmup (>= W frames), then EXACTLY K timed steps (= K/F launches) bracketed by barrier + synchronize
    # (at least three launches, whatever W: the first launches after start-up run at ramping clocks)
    for _ in range(max(3, (args.warmup + F - 1) // F)):
        pipe.submit(render_fn)
    pipe.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = None
    for i in range(n_launches):
        step_no[0] = i
        pipe.submit(render_fn)
        if args.no_overlap:
            pipe.drain()
    img = pipe.drain()                      # every launch rendered, gathered and assembled before the clock stops
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_no[0] = -1
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kernel_ms_avg = sum(kern_ms) / len(kern_ms)

    if rank == 0:
        assert img is not None and tuple(img.shape) == (F, h, w, 3)
        checksum = float(img[F - 1].double().sum().item())
        n_px = w * h
        # roofline of the dominant (only) kernel on this rank: algorithmic bytes of ITS launch (F frames of this
        # rank's strips) / its average duration
        alg = F * algorithmic_bytes(st, scene.n_lights, yahr_amd.shard_rows(h, world, rank, args.strip_rows) * w)
        achieved = alg / (kernel_ms_avg * 1e-3) / 1e9
        out = {
            "metric": "Mrays/s (primary + reflection + shadow rays cast), bunny scene 1920x1080" if args.workload == "bunny"
                      else f"Mrays/s, {args.workload}",
            "value": round(rays_per_step * args.steps / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "mpixels_per_s": round(primary_per_step / spp * args.steps / elapsed / 1e6, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic (procedural pseudo-bunny: no bunny mesh ships with the reference)"
                     if args.workload.startswith("bunny") else "synthetic (generated scene, yahr_amd/scenes.py)"),
            "config": {"workload": workload_name, "rays_per_frame": rays_per_step, "spp": spp, "traverse": args.traverse,
                       "sharding": f"{world} rank(s), strips of {args.strip_rows} rows round-robin, one gather to rank 0 per "
                                   "launch, gather+assembly of launch i overlapped with the render of launch i+1",
                       "frames_per_launch": F, "launches": n_launches,
                       "host_scene_build_s": round(build_s, 3)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic(args.workload),
                         "kernel": "k_render", "kernel_ms_avg": round(kernel_ms_avg, 4), "frames_per_launch": F,
                         "algorithmic_bytes_per_launch": alg,
                         "note": "algorithmic bytes = per-ray tests x record sizes (SURVEY 8d), counted by the kernel's "
                                 "own counters (== oracle, first-hit shadow rays); node/primitive records are served "
                                 "from SGPR broadcasts and L2, so HBM traffic is far below this figure and frac can "
                                 "exceed what a DRAM stream could reach -- the kernel is VALU/latency bound"},
            "image_checksum": checksum,
        }
        vp = load_valu_profile(args.workload, F) if world == 1 else None
        if vp is not None:
            # what actually bounds the kernel: vector-instruction issue
            g = vp[0] / (kernel_ms_avg * 1e-3) / 1e9
            out["roofline"]["valu_issue"] = {"achieved": round(g, 1), "peak": vp[1], "unit": "G wave-instr/s",
                                             "frac": round(g / vp[1], 4),
                                             "source": "SQ_INSTS_VALU of profiles/ / live launch time; peak measured by tools/ubench/valu_rate.hip"}
        if world == 1 and not args.no_cpu_baseline:
            cores = min(16, os.cpu_count() or 1)
            cb, ost = cpu_baseline(desc, cores, spp)
            out["cpu_baseline"] = cb
            if ost is not None:
                out["roofline"]["algorithmic_bytes_reference_order"] = algorithmic_bytes(ost, scene.n_lights, n_px)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
