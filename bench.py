#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json: Mrays/s (+ Mpixels/s) on the bunny scene at
1920x1080, on N GPUs of one node, beside the CPU baseline.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one render of the whole frame: every rank renders its row strips (scene resident in its HBM
before the timed region), a gather over RCCL brings the strips to rank 0, rank 0 scatters them into the
image.  Total work is fixed as N grows ("scaling": "strong").

Frames are rendered F = 8 per kernel launch (x the rank count; tiles of the F frames interleaved) with ONE
gather per launch: a single 1080p frame is bounded by the serial chain of its heaviest tile, not by
throughput, so the tail of one frame overlaps the other frames of the batch.  The F frames of a launch have F
DISTINCT cameras (a small orbit around the model), so no frame shares rays with another.  A block of K steps
is floor(K/F) launches of F frames plus one launch of K mod F frames: exactly K frames, whatever K.  Because a
block of K = 20 frames lasts only ~9 ms, the block is REPEATED R times inside the timed region (R chosen so
that the region lasts >= ~1 s, printed as "repeats"); ms_per_step = timed region / (R * K).  Every frame is
fully rendered, gathered and assembled inside the timed region.

Workload: BASELINE.json configs[2] -- the metric's own configuration -- with the documented stand-in mesh: the
reference ships no bunny mesh (SURVEY finding 3), so the scene is the procedural 69,696-triangle
"pseudo-bunny" of yahr_amd/scenes.py on the +-1000 plane, 1 light, 1 spp, WhittedIntegrator depth 2, BVH 16
Midpoint.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field's definition).
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT]

# torch, numpy and yahr_amd are imported in main() only: with --gpus N > 1 and no WORLD_SIZE in the environment this
# process is a LAUNCHER -- it starts N ranks as a child `python -m torch.distributed.run` and must not touch the GPU
# (nor pay for the imports) itself.
np = torch = dist = yahr_amd = scenes = ShardPipeline = ShardPlan = None

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
# vector-instruction issue ceiling of the chip from the microarch guide: 256 CUs x 4 SIMD-32, a wave64 VALU
# instruction occupies its SIMD for 2 cycles, 2.4 GHz  ->  1024 * 2.4e9 / 2 = 1228.8 G wave-instructions/s
VALU_PEAK_GINST_S = 1228.8

# workload -> (scene description, text, point the camera orbit turns around)
def _import_runtime():
    global np, torch, dist, yahr_amd, scenes, ShardPipeline, ShardPlan
    import numpy as np_
    import torch as torch_
    import torch.distributed as dist_
    import yahr_amd as yahr_amd_
    from yahr_amd import scenes as scenes_
    from yahr_amd.dist import ShardPipeline as SP, ShardPlan as SPl
    np, torch, dist, yahr_amd, scenes, ShardPipeline, ShardPlan = np_, torch_, dist_, yahr_amd_, scenes_, SP, SPl


WORKLOADS = {
    "bunny": lambda: (scenes.pseudo_bunny(1920, 1080, depth=2),
                      "pseudo-bunny (procedural stand-in, 69,696 tris + 2-tri plane), 1920x1080, 1 spp, "
                      "Whitted depth 2, BVH 16 Midpoint, 1 point light", (0.2, 0.9, 0.0)),
    "spheres": lambda: (scenes.spheres8(1920, 1080, depth=1),
                        "8 spheres + plane + 2 point lights, 1920x1080, 1 spp, Whitted depth 1, BVH 16 Midpoint",
                        (0.0, -1.2, 15.0)),
    "bunny_small": lambda: (scenes.pseudo_bunny(480, 270, depth=2), "pseudo-bunny 480x270 (smoke size)", (0.2, 0.9, 0.0)),
    # BASELINE.json configs[0]: the bundled scene (repaired copy, scenes/scene_fixed.yahrr) at 256x256, 1 spp -- the
    # reference's own CPU-runnable case; the CPU baseline of this line runs SINGLE-THREADED, as the config says
    "scene_fixed": lambda: (scenes.scene_fixed(256, 256), "scene.yahrr (repaired: 7 spheres + 2-triangle plane, 1 point light), "
                            "256x256, 1 spp, Whitted depth 3, BVH 16 Midpoint", (0.0, -1.2, 10.0)),
    # BASELINE.json configs[3] and [4]; spp > 1 and "stratified" are build extensions (sampler)
    "bunny4k": lambda: (scenes.pseudo_bunny(3840, 2160, depth=4),
                        "pseudo-bunny 3840x2160, 64 spp stratified, Whitted depth 4 (primary + 3 reflection segments)",
                        (0.2, 0.9, 0.0)),
    "tris1m": lambda: (scenes.random_tris(1_000_000, 1920, 1080, depth=2, bvh_depth=20),
                       "1M random triangles (splitmix64 seed 0x5EED), 1920x1080, 16 spp, Whitted depth 2, BVH 20 Midpoint",
                       (0.0, 0.0, 0.0)),
}
WORKLOAD_SPP = {"bunny4k": 64, "tris1m": 16}
ORBIT_STEP_DEG = 1.5           # angle between neighbouring cameras of a frame batch


def orbit_cameras(cam, target, n):
    """n cameras on a horizontal circle around `target` through the scene camera's position, ORBIT_STEP_DEG
    apart and centred on the scene camera, each looking at the target like the scene camera does (same offset
    between its look direction and the direction to the target).  Rows of yahr_ctx_set_cameras."""
    pos = np.array(cam["pos"], dtype=np.float64)
    look = np.array(cam["look"], dtype=np.float64)
    tgt = np.array(target, dtype=np.float64)
    rows = []
    for i in range(n):
        a = math.radians((i - (n - 1) / 2.0) * ORBIT_STEP_DEG)
        c, s = math.cos(a), math.sin(a)
        rot = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
        p = tgt + rot @ (pos - tgt)
        rows.append([cam["imW"], cam["imH"], cam["focal"], *(rot @ look), *cam["up"], *p])
    return np.array(rows, dtype=np.float32)


def algorithmic_bytes(st, n_lights, n_pixels):
    """32 B per box test, 72 B per triangle test, 16 B per sphere test, (32 + 24 L) B per shaded
    hit, 12 B per output pixel."""
    return (32 * st["box_tests"] + 72 * st["tri_tests"] + 16 * st["sphere_tests"]
            + (32 + 24 * n_lights) * st["shaded_hits"] + 12 * n_pixels)


def parity_of(gpu, cpu):
    """the parity check of the same run -- per channel |g - c| <= 1e-4 * max(|g|, |c|) + 1e-6"""
    g, c = gpu.astype(np.float64), cpu.astype(np.float64)
    err = np.abs(g - c)
    tol = 1e-4 * np.maximum(np.abs(g), np.abs(c)) + 1e-6
    mag = np.maximum(np.abs(g), np.abs(c))
    big = mag >= 1e-3                                  # relative error where a relative error means something
    return dict(pixels=int(g.shape[0] * g.shape[1]), pixels_outside_tolerance=int((err > tol).any(axis=2).sum()),
                max_rel_err=float((err[big] / mag[big]).max()) if big.any() else 0.0, max_abs_err=float(err.max()),
                bit_identical_pixels=int((gpu.view(np.uint32) == cpu.view(np.uint32)).all(axis=2).sum()),
                tolerance="|g - c| <= 1e-4 max(|g|, |c|) + 1e-6 per channel; max_rel_err over channels of magnitude >= 1e-3")


def cpu_baseline(desc, cores, spp=1, min_cpu_seconds=10.0, gpu_rows=None):
    """The oracle (CPU restatement of the reference, kind "port": the Haskell reference cannot be built here)
    timed on the host cores over the same workload: whole frames (scene camera), repeated until ~min_cpu_seconds
    of CPU work (spp == 1), or a bounded sample of evenly spaced 8-row strips of one frame (supersampled).
    gpu_rows(r0, r1) -> the HIP path's rows of the same frame (scene camera): what the oracle has rendered anyway is
    compared with them, the parity check of the same run.  Returns (baseline, oracle stats or None, parity or None)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py                      # the checker, timed beside the product; never part of the product path
    osc = oracle_py.OracleScene.from_desc(desc).build()
    reps, wall, rays, st = 0, 0.0, 0, None
    if spp > 1:
        w, h = osc.size()
        strips = list(range(0, h - 7, 8))
        order = [strips[(i * 37) % len(strips)] for i in range(len(strips))]      # spread over the image
        prim, done = 0, 0
        cpu_strips, gpu_strips = [], []
        for r0 in order:
            t0 = time.perf_counter()
            rows, st = osc.render_rows(r0, r0 + 8, spp=spp, seed=0, threads=cores)
            wall += time.perf_counter() - t0
            if gpu_rows is not None:
                cpu_strips.append(rows); gpu_strips.append(gpu_rows(r0, r0 + 8))
            rays += st["primary_rays"] + st["reflection_rays"] + st["shadow_rays"]
            prim += st["primary_rays"]
            done += 1
            if wall * cores >= min_cpu_seconds:
                break
        return dict(value=round(rays / wall / 1e6, 3), unit="Mrays/s", cores=cores, kind="port",
                    sample=f"{done} strips of 8 rows (of {len(strips)}) spread over one frame at {spp} spp, render only, "
                           f"{wall:.2f} s wall on {cores} threads (one thread per image row)",
                    mpixels_per_s=round(prim / spp / wall / 1e6, 4)), None, \
            (parity_of(np.concatenate(gpu_strips), np.concatenate(cpu_strips)) if gpu_strips else None)
    img = None
    while wall * cores < min_cpu_seconds and reps < 50:
        t0 = time.perf_counter()
        img, st = osc.render(threads=cores)
        wall += time.perf_counter() - t0
        reps += 1
        rays += st["primary_rays"] + st["reflection_rays"] + st["shadow_rays"]
    return dict(value=round(rays / wall / 1e6, 3), unit="Mrays/s", cores=cores, kind="port",
                sample=f"{reps} full frame(s) of the same workload (scene camera), render only (scene build excluded), "
                       f"{wall:.2f} s wall on {cores} threads over the reference's squareBatches tiles",
                mpixels_per_s=round(reps * st["primary_rays"] / wall / 1e6, 3)), st, \
        (parity_of(gpu_rows(0, img.shape[0]), img) if gpu_rows is not None and img is not None else None)


def usable_cores():
    """Every core this process may use: the scheduler affinity mask, capped by the cgroup's CPU quota when there is one (a
    container that sees all 256 cores of its host but may run 16 cores' worth of time is a 16-core machine: 256 threads on
    it measured 10.6 Mrays/s where 16 threads measure 24)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    src = "sched_getaffinity"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(math.ceil(int(quota) / int(period))))
            if q < n:
                n, src = q, "cgroup cpu.max (%s / %s) below the affinity mask" % (quota, period)
    except (OSError, ValueError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0 and int(math.ceil(q / per)) < n:
                n, src = int(math.ceil(q / per)), "cgroup cfs quota below the affinity mask"
        except (OSError, ValueError):
            pass
    return n, src


def load_profile(workload):
    """Per-frame counter figures of the committed rocprofv3 PMC profile of this workload (profiles/traffic_<workload>.json;
    traffic_latest.json for the headline workload; written by tools/pmc_to_traffic.py from separate --pmc passes), or None."""
    for name in ("traffic_%s.json" % workload, "traffic_latest.json"):
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", name)))
            if j.get("workload") == workload and "valu_insts_per_frame" in j:
                return j
        except (OSError, ValueError):
            pass
    return None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a torch.distributed environment: start the N ranks ourselves, as a CHILD
    process (never an exec, and before anything here touches the GPU), relay rank 0's JSON line and leave with the
    child's status.  Same command shape as the driver's own multi-rank launch."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["YAHR_BENCH_SPAWNED"] = "1"
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:                       # stderr passes straight through
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks finished without printing a result line\n")
        rc = 1
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="bunny", choices=sorted(WORKLOADS))
    ap.add_argument("--traverse", default="auto", choices=["auto", "lane", "packet", "hybrid", "stream", "stack", "hybrid_stack", "ordered"])
    ap.add_argument("--strip-rows", type=int, default=8)
    ap.add_argument("--frames-per-launch", type=int, default=20, help="per rank; a launch holds this x ranks frames")
    ap.add_argument("--repeats", type=int, default=0, help="repeats of the K-step block inside the timed region; 0 = "
                                                           "as many as make the region last --min-seconds")
    ap.add_argument("--min-seconds", type=float, default=3.0)
    ap.add_argument("--same-camera", action="store_true", help="every frame of a batch uses the scene camera (diagnostics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="finish gather + assembly of a launch before the next render starts (diagnostics)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])                 # does not return
    _import_runtime()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "YAHR_BENCH_FORCE_DEVICE" in os.environ:       # rehearsal of N ranks on one GPU (diagnostics only)
        local_rank = int(os.environ["YAHR_BENCH_FORCE_DEVICE"])
    if world != args.gpus:
        args.gpus = world                                     # the launcher's world size is what runs
    assert torch.cuda.is_available(), "bench.py needs a GPU: yahr_amd has no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  YAHR_BENCH_BACKEND=gloo (with YAHR_BENCH_FORCE_DEVICE=0) rehearses the N-rank control
        # flow on a one-GPU box, where RCCL refuses two ranks on one device; its numbers mean nothing.
        backend = os.environ.get("YAHR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    desc, workload_name, target = WORKLOADS[args.workload]()
    t0 = time.perf_counter()
    scene = desc.to_scene().build()
    build_s = time.perf_counter() - t0
    ctx = yahr_amd.Context(scene, local_rank)
    w, h = scene.width, scene.height
    K = args.steps
    trav = {"auto": yahr_amd.TRAVERSE_AUTO, "lane": yahr_amd.TRAVERSE_LANE, "packet": yahr_amd.TRAVERSE_PACKET,
            "hybrid": yahr_amd.TRAVERSE_HYBRID, "stream": yahr_amd.TRAVERSE_STREAM, "stack": yahr_amd.TRAVERSE_STACK,
            "hybrid_stack": yahr_amd.TRAVERSE_HYBRID_STACK, "ordered": yahr_amd.TRAVERSE_ORDERED}[args.traverse]
    # frames per launch: 20 per rank (every rank then gets as many tiles per launch as in the 1-GPU case, so the
    # per-launch tail stays amortised and the gathers get larger and fewer), never more than K.  (8 until the second half
    # of round 4: with the shorter kernel the fixed ~0.15 ms per launch weighs more -- 8 / 16 / 20 / 25 / 40 distinct
    # cameras per launch: 0.2742 / 0.2707 / 0.2664 / 0.2655 / 0.2657 ms per frame, profiles/r04_frames_per_launch.txt)
    F = max(1, min(args.frames_per_launch * world, K))
    n_full, f_tail = K // F, K % F
    if not args.same_camera:
        ctx.set_cameras(orbit_cameras(desc.camera, target, F))
    plan = ShardPlan(h, w, world, args.strip_rows, device=dev, frames=F)
    # double-buffered: gather + assembly of launch i overlap the render of launch i+1 (yahr_amd/dist.py)
    pipe = ShardPipeline(plan, rank)
    bufs = pipe.slots[0]
    stream = torch.cuda.current_stream()
    spp = WORKLOAD_SPP.get(args.workload, 1)
    shard = dict(traverse=trav, n_shards=world, shard=rank, strip_rows=args.strip_rows, spp=spp, frame_rows=plan.max_rows)
    o_full = yahr_amd.make_opts(frames=F, **shard)
    o_tail = yahr_amd.make_opts(frames=f_tail, **shard) if f_tail else None
    timed_ev = []                            # HIP event pairs around the F-frame launches of the timed region

    def render_with(opts, timed):
        def render_fn(shard_, n_shards_, strip_rows_, out):
            # launched on torch's current stream, so the torch.cuda.Event pair brackets exactly this kernel
            if timed:
                e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                e[0].record(stream)
            ctx.render_into(out.data_ptr(), 0, h, opts, stream.cuda_stream)
            if timed:
                e[1].record(stream)
                timed_ev.append(e)
        return render_fn

    def block(timed):
        """one block = exactly K frames"""
        for _ in range(n_full):
            pipe.submit(render_with(o_full, timed))
            if args.no_overlap:
                pipe.drain()
        if o_tail is not None:
            pipe.submit(render_with(o_tail, False))
            if args.no_overlap:
                pipe.drain()

    # ---- counted launches per rank (deterministic counters; not timed): the F-frame launch, the tail launch
    def counted(frames):
        o = yahr_amd.make_opts(frames=frames, collect_stats=True, **shard)
        ctx.render_into(bufs[0].data_ptr(), 0, h, o, stream.cuda_stream)
        torch.cuda.synchronize()
        return ctx.stats()
    st_full = counted(F)
    st_tail = counted(f_tail) if f_tail else None

    def ray_counts(st):
        t = torch.tensor([st["primary_rays"], st["reflection_rays"], st["shadow_rays"], st["box_tests"],
                          st["wave_node_steps"], st["tri_tests"] + st["sphere_tests"] - st["tri_tests_culled"], st["wave_prim_steps"],
                          st["tri_tests_culled"], st["wave_cull_steps"], st["tri_cull_violations"]],
                         dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(t)
        return int(t[:3].sum().item()), int(t[0].item()), [int(v) for v in t[3:].tolist()]
    rays_full, prim_full, steps_full = ray_counts(st_full)
    rays_tail, prim_tail, _ = ray_counts(st_tail) if st_tail else (0, 0, None)
    rays_per_block = n_full * rays_full + rays_tail
    prim_per_block = n_full * prim_full + prim_tail

    # ---- warmup: >= W frames and at least three launches (the first launches after start-up run at ramping
    # clocks), then one calibration block to choose R
    for _ in range(max(3, (args.warmup + F - 1) // F)):
        pipe.submit(render_with(o_full, False))
    pipe.drain()
    torch.cuda.synchronize()
    R = args.repeats
    if R <= 0:
        t0 = time.perf_counter()
        block(False)
        pipe.drain()
        torch.cuda.synchronize()
        tb = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)      # every rank must agree on R
        R = max(1, min(100000, int(math.ceil(args.min_seconds / max(float(tb.item()), 1e-6)))))

    # ---- timed region: R blocks of EXACTLY K steps, bracketed by barrier + synchronize on both sides
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.reset_timing()
    pipe.timing = True                      # (events only: per launch one behind the render, one behind gather + assembly)
    for _ in range(R):
        block(True)
    img = pipe.drain()                      # every launch rendered, gathered and assembled before the clock stops
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    # (taken now: the lone-frame launches below reuse the first slot's buffer, which `img` may be a view of)
    checksum = float(img[0].double().sum().item()) if rank == 0 else 0.0
    kern_ms = [a.elapsed_time(b) for a, b in timed_ev]
    kernel_ms_avg = sum(kern_ms) / len(kern_ms) if kern_ms else float("nan")
    pipe.timing = False
    # the exchange step of every timed launch on this rank: its duration and the part of it the next render did not hide
    gt = pipe.gather_times()
    gather_ms_avg = sum(g for g, _ in gt) / len(gt) if gt else 0.0
    gather_exposed_ms_avg = sum(e for _, e in gt) / len(gt) if gt else 0.0
    # what every rank saw: its device, the world size of its process group, its own launch time
    mine = torch.tensor([float(local_rank), float(dist.get_world_size() if world > 1 else 1), kernel_ms_avg,
                         float(yahr_amd.shard_rows(h, world, rank, args.strip_rows)), gather_ms_avg, gather_exposed_ms_avg],
                        dtype=torch.float64, device=dev)
    table = torch.zeros((world, mine.numel()), dtype=torch.float64, device=dev)
    table[rank] = mine
    if world > 1:
        dist.all_reduce(table)                 # (a sum of one-hot rows: all_reduce is what every backend offers)
    per_rank = [[float(x) for x in row] for row in table.tolist()]

    # ---- a lone frame per launch (the scene camera): bound by its longest wave, reported beside the batch figure
    o_one = yahr_amd.make_opts(frames=1, **shard)
    one_ev = []
    for i in range(24):
        e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        e[0].record(stream)
        ctx.render_into(bufs[0].data_ptr(), 0, h, o_one, stream.cuda_stream)
        e[1].record(stream)
        if i >= 4:
            one_ev.append(e)
    torch.cuda.synchronize()
    single_ms = sorted(a.elapsed_time(b) for a, b in one_ev)[len(one_ev) // 2]

    if rank == 0:
        assert img is not None and tuple(img.shape) == (F, h, w, 3)
        n_steps = R * K
        my_px = yahr_amd.shard_rows(h, world, rank, args.strip_rows) * w
        alg = algorithmic_bytes(st_full, scene.n_lights, F * my_px)      # of rank 0's F-frame launch
        prof = load_profile(args.workload)
        from yahr_amd.srchash import kernel_source_hash
        src_hash = kernel_source_hash()
        matches = bool(prof is not None and prof.get("kernel_source_sha256") == src_hash)
        # a launch ends when its slowest rank does; instruction counts add up over the ranks exactly, because a strip of
        # 8 rows is one row of 8x8 wave tiles: the N ranks together run the very wavefronts of the one-rank launch
        kernel_ms_job = max(r[2] for r in per_rank)
        roof = {"kernel": (prof or {}).get("kernels", "k_render"), "kernel_ms_avg": round(kernel_ms_job, 4),
                "frames_per_launch": F, "launches_timed": len(kern_ms),
                "kernel_source_sha256": src_hash[:16], "profile_matches_source": matches,
                "per_rank": [{"rank": i, "device": int(r[0]), "world_size_seen": int(r[1]), "rows": int(r[3]),
                              "kernel_ms_avg": round(r[2], 4), "gather_ms_avg": round(r[4], 4),
                              "gather_exposed_ms_avg": round(r[5], 4)} for i, r in enumerate(per_rank)]}
        node_steps, node_lanes, prim_lanes, prim_steps = steps_full[1], steps_full[0], steps_full[2], steps_full[3]
        # lanes that were logically at work per wavefront step (counted launch of the same batch): an issue fraction
        # says the VALU is busy, this says how much of what it issues is useful
        roof["lane_utilisation"] = {
            "node_steps": round(node_lanes / (64.0 * node_steps), 4) if node_steps else None,
            "primitive_steps": round(prim_lanes / (64.0 * prim_steps), 4) if prim_steps else None,
            "source": "box tests / (64 x wavefront node steps), EXECUTED primitive tests / (64 x wavefront primitive steps) of "
                      "the counted launch; wavefront steps are counted by the packet walker only"}
        # fat multiLeafs: triangle tests whose outcome was proven per packet instead of being executed (part of tri_tests)
        roof["leaf_culling"] = {"tri_tests_settled_per_packet": steps_full[4], "packet_tests": steps_full[5],
                                "share_of_primitive_tests": round(steps_full[4] / max(1, steps_full[4] + steps_full[2]), 4),
                                "violations_found_by_the_counting_build": steps_full[6]}
        if prof is not None and kern_ms and matches:
            # what the counters say bounds the kernel: vector-instruction issue.  Instruction
            # counts are deterministic per frame, so the profile's per-frame figures (mean over the cameras of the
            # orbit) scale to any number of frames per launch and add up over row shards.
            valu = int(prof["valu_insts_per_frame"] * F)
            g = valu / (kernel_ms_job * 1e-3) / 1e9
            traffic = int(prof["hbm_bytes_per_frame"] * F)
            peak = VALU_PEAK_GINST_S * world
            roof.update({
                "bound": "valu-issue", "achieved": round(g, 1), "peak": peak, "unit": "G wave-instr/s",
                "frac": round(g / peak, 4),
                "peak_source": "MI355X_MICROARCH.md: 1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction, x ranks",
                "frac_of_measured_issue_peak": round(g / (prof["valu_issue_peak_measured_ginst_s"] * world), 4),
                "measured_issue_peak": prof["valu_issue_peak_measured_ginst_s"] * world,
                "valu_insts_per_launch": valu,
                "traffic": traffic,
                "hbm": {"achieved": round(traffic / (kernel_ms_job * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                        "frac": round(traffic / (kernel_ms_job * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
                        "source": "rocprofv3 FETCH_SIZE (x2, gfx950) + WRITE_SIZE per frame x frames per launch"},
                "profile": prof.get("source", "profiles/traffic_latest.json"),
            })
        else:
            roof.update({"bound": "valu-issue", "achieved": None, "peak": VALU_PEAK_GINST_S * world, "unit": "G wave-instr/s",
                         "frac": None, "traffic": None,
                         "note": ("the committed PMC profile was taken from other kernel sources (kernel_source_sha256 differs): "
                                  "re-run tools/profile_bench.sh + tools/pmc_to_traffic.py" if prof is not None and not matches
                                  else "no committed PMC profile for this workload (profiles/traffic_<workload>.json)")})
        # SURVEY 8(d)'s algorithmic-byte model, kept as a FIELD: the records are served by scalar broadcasts and L2,
        # so this is not HBM traffic and is not a fraction of anything
        roof["algorithmic_bytes_per_launch"] = alg
        roof["algorithmic_gbs"] = round(alg / (kernel_ms_avg * 1e-3) / 1e9, 1) if kern_ms else None
        # ... and as SURVEY 8(d) literally defines the roofline: algorithmic bytes / kernel time / (8 TB/s x GPUs).  A value
        # above 1 does NOT mean skipped work (the counters equal the oracle's): the model charges every LANE 32 B per box
        # test, the packet walker fetches a record once per WAVEFRONT by scalar load, so the model does not bound it
        roof["survey_8d_frac"] = round(alg / (kernel_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kern_ms else None
        roof["survey_8d_note"] = ("algorithmic bytes of rank 0's launch / its kernel time / 8 TB/s; > 1 is possible because the "
                                  "per-lane byte model does not bound a per-wavefront (scalar) fetch")
        out = {
            "metric": "Mrays/s (primary + reflection + shadow rays cast), bunny scene 1920x1080" if args.workload == "bunny"
                      else f"Mrays/s, {args.workload}",
            "value": round(rays_per_block * R / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "mpixels_per_s": round(prim_per_block / spp * R / elapsed / 1e6, 2),
            "n_gpus": world, "world_size": int(per_rank[0][1]), "steps": K, "warmup": args.warmup, "repeats": R, "frames_timed": n_steps,
            "timed_region_s": round(elapsed, 4),
            "ms_per_step": round(elapsed / n_steps * 1e3, 4),
            "single_frame_ms": round(single_ms, 4),
            # the exchange step per launch on rank 0 (gather over RCCL + assembly on a side stream): from the end of the
            # launch's render to the assembled images, and the part of that which the NEXT launch's render did not hide
            # (one rank: there is no exchange step, both are 0; every rank's own figures are in roofline.per_rank)
            "gather_ms_avg": round(per_rank[0][4], 4), "gather_exposed_ms_avg": round(per_rank[0][5], 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic (procedural pseudo-bunny: no bunny mesh ships with the reference)"
                     if args.workload.startswith("bunny") else "synthetic (generated scene, yahr_amd/scenes.py)"),
            "config": {"workload": workload_name, "rays_per_frame": round(rays_per_block / K, 1), "spp": spp,
                       "traverse": args.traverse,
                       "cameras": ("the scene camera for every frame" if args.same_camera else
                                   f"{F} distinct cameras per launch, an orbit of {ORBIT_STEP_DEG} degree steps around the model"),
                       "sharding": f"{world} rank(s), strips of {args.strip_rows} rows round-robin, one gather to rank 0 per "
                                   "launch, gather+assembly of launch i overlapped with the render of launch i+1",
                       "frames_per_launch": F, "launches_per_block": n_full + (1 if f_tail else 0),
                       "tail_launch_frames": f_tail,
                       "host_scene_build_s": round(build_s, 3)},
            "roofline": roof,
            "image_checksum": checksum,
        }
        if world == 1 and not args.no_cpu_baseline:
            cores, cores_source = usable_cores()
            def gpu_rows(r0, r1):               # the HIP path's rows of the scene camera's frame, same sampler and seed
                t = torch.zeros((r1 - r0, w, 3), dtype=torch.float32, device=dev)
                ctx.render_into(t.data_ptr(), r0, r1, yahr_amd.make_opts(traverse=trav, spp=spp, seed=0), stream.cuda_stream)
                torch.cuda.synchronize()
                return t.cpu().numpy()
            if args.workload == "scene_fixed":
                cores = 1                       # configs[0]: "Haskell CPU path single-threaded"
            cb, ost, par = cpu_baseline(desc, cores, spp, gpu_rows=gpu_rows)
            cb["cores_source"] = cores_source if args.workload != "scene_fixed" else "BASELINE configs[0]: single-threaded"
            out["cpu_baseline"] = cb
            out["parity"] = par
            if ost is not None:
                roof["algorithmic_bytes_reference_order_per_frame"] = algorithmic_bytes(ost, scene.n_lights, w * h)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
