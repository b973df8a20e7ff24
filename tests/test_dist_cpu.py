"""World-size-2 (and 3) rehearsal of the multi-GPU path on the CPU with the gloo backend: the sharding plan,
the single gather and the assembly in yahr_amd/dist.py.  The per-rank "render" is the oracle restricted to the
rank's rows (tests may use the oracle; on a GPU box the same code path runs with the HIP context, see
tests/test_gpu_parity.py::test_sharded_render_single_process and bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yahr_amd import scenes
from yahr_amd.dist import ShardPlan, render_sharded


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, strip, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from oracle_py import OracleScene
        from yahr_amd import shard_global_rows
        desc = scenes.scene_fixed(w, h)
        osc = OracleScene.from_desc(desc).build()
        plan = ShardPlan(h, w, world, strip, frames=2)

        def render_fn(shard, n_shards, strip_rows, out):
            rows = shard_global_rows(h, n_shards, shard, strip_rows)
            for i, gv in enumerate(rows):                      # oracle stand-in for the HIP launch
                r, _ = osc.render_rows(int(gv), int(gv) + 1, threads=1)
                for f in range(out.shape[0]):                  # every frame of the batch (same camera here)
                    out[f, i] = torch.from_numpy(r[0])
        img = render_sharded(render_fn, plan, rank)
        if rank == 0:
            assert img.shape[0] == plan.frames and all(torch.equal(img[0], img[f]) for f in range(plan.frames))
            np.save(out_path, img[0].numpy())
        else:
            assert img is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,strip", [(2, 40, 30, 8), (3, 24, 19, 4), (2, 16, 5, 8)])
def test_sharded_render_gloo(tmp_path, world, w, h, strip):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, strip, out), nprocs=world, join=True)
    from oracle_py import OracleScene
    ref, _ = OracleScene.from_desc(scenes.scene_fixed(w, h)).build().render(threads=2)
    got = np.load(out)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_plan_single_rank_is_identity():
    plan = ShardPlan(11, 4, 1, 8)
    x = torch.arange(11 * 4 * 3, dtype=torch.float32).reshape(11, 4, 3)
    assert torch.equal(plan.assemble([x]), x)
    img = render_sharded(lambda s, n, st, out: out.copy_(x.unsqueeze(0)), plan, 0)
    assert torch.equal(img[0], x)


def _pipe_worker(rank, world, port, w, h, strip, frames, n_launches, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yahr_amd import shard_global_rows
        from yahr_amd.dist import ShardPipeline
        plan = ShardPlan(h, w, world, strip, frames=frames)
        pipe = ShardPipeline(plan, rank)
        seen = {}
        pipe.on_assembled = lambda i, imgs: seen.__setitem__(i, imgs.clone())
        launch = [0]

        def render_fn(shard, n_shards, strip_rows, out):
            # synthetic "render": pixel value encodes (launch, frame, global row, column) -- any mix-up of slots,
            # ranks, frames or rows shows up in the assembled image
            rows = torch.from_numpy(shard_global_rows(h, n_shards, shard, strip_rows)).float()
            out.zero_()
            for f in range(frames):
                val = launch[0] * 1000.0 + f * 100.0 + rows[:, None] + torch.arange(w).float()[None, :] / 64.0
                out[f, : len(rows)] = val[:, :, None].expand(-1, -1, 3)
        for i in range(n_launches):
            launch[0] = i
            pipe.submit(render_fn)
        last = pipe.drain()
        if rank == 0:
            assert sorted(seen) == list(range(n_launches)), sorted(seen)
            rows = torch.arange(h).float()
            for i in range(n_launches):
                for f in range(frames):
                    want = i * 1000.0 + f * 100.0 + rows[:, None] + torch.arange(w).float()[None, :] / 64.0
                    assert torch.equal(seen[i][f, :, :, 0], want), (i, f)
            assert torch.equal(last, seen[n_launches - 1])
            open(out_path, "w").write("ok")
        else:
            assert last is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,frames,n_launches", [(2, 2, 5), (3, 1, 4), (2, 3, 1)])
def test_overlapped_pipeline_gloo(tmp_path, world, frames, n_launches):
    out = str(tmp_path / "ok.txt")
    mp.spawn(_pipe_worker, args=(world, _free_port(), 33, 29, 4, frames, n_launches, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_pipeline_single_rank():
    from yahr_amd.dist import ShardPipeline
    plan = ShardPlan(11, 4, 1, 8, frames=2)
    pipe = ShardPipeline(plan, 0)
    for i in range(3):
        pipe.submit(lambda s, n, st, out, i=i: out.fill_(float(i)))
    img = pipe.drain()
    assert img.shape == (2, 11, 4, 3) and float(img.min()) == 2.0 == float(img.max())


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` (the driver's command shape) must launch N ranks itself: a child
    `python -m torch.distributed.run`, started before anything touches the GPU, rank 0's JSON line relayed and the
    child's exit status passed on."""
    import importlib
    import io
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    assert bench.torch is None and bench.yahr_amd is None          # nothing heavy imported by the launcher
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = io.StringIO('noise\n{"metric": "x", "n_gpus": 4}\n')

        def wait(self):
            return seen.get("rc", 0)

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    out = io.StringIO()
    monkeypatch.setattr(sys, "stdout", out)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert out.getvalue().strip() == '{"metric": "x", "n_gpus": 4}'
    seen["rc"] = 3                                                  # a failing child fails the launcher
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3
