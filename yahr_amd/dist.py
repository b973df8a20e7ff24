"""Row-sharded rendering across the GPUs of one node, one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" for the CPU rehearsal in tests).

The reference parallelises over independent pixel tiles inside one process (main.hs:75-96).  Here the image
is cut into strips of `strip_rows` rows dealt round-robin to the ranks (strip s -> rank s % world), every
rank renders its strips with the scene replicated in its own HBM, and there is exactly ONE collective per
launch: a gather of the float RGB strips (of every frame of the launch's frame batch) to rank 0, which
scatters them to their image rows (samplesToImage, main.hs:98-107) with a single index_copy.  No exchange
step exists inside a frame.

`ShardPipeline` double-buffers that: the gather (RCCL's own stream) and the assembly (a side stream on
rank 0) of launch i run while launch i+1 renders, so a steady stream of launches costs
max(render, gather + assemble) instead of their sum.
"""
import torch
import torch.distributed as dist

from . import shard_global_rows, shard_rows


class ShardPlan:
    """Which rows each rank owns, and where they land in the final image(s)."""

    def __init__(self, height, width, world, strip_rows=8, device="cpu", frames=1):
        self.height, self.width, self.world, self.strip_rows = height, width, world, strip_rows
        self.frames = frames
        self.device = torch.device(device)
        self.rows = [shard_rows(height, world, r, strip_rows) for r in range(world)]
        self.max_rows = max(self.rows) if self.rows else 0
        assert sum(self.rows) == height
        # destination row of every (rank, local row) in gather order; padding rows go to a scratch row
        dest = torch.full((world, self.max_rows), height, dtype=torch.int64)
        for r in range(world):
            if self.rows[r]:
                dest[r, : self.rows[r]] = torch.from_numpy(shard_global_rows(height, world, r, strip_rows))
        self.dest = dest.reshape(-1).to(self.device)

    def buffers(self, rank):
        """(local strips [frames, max_rows, w, 3]; on rank 0 also the gather buffer [world, frames, max_rows, w, 3]
        and the image buffer [frames, height + 1, w, 3] whose last row is scratch for padding rows)"""
        f = self.frames
        local = torch.zeros((f, self.max_rows, self.width, 3), dtype=torch.float32, device=self.device)
        if rank != 0:
            return local, None, None
        gathered = torch.zeros((self.world, f, self.max_rows, self.width, 3), dtype=torch.float32, device=self.device)
        image = torch.zeros((f, self.height + 1, self.width, 3), dtype=torch.float32, device=self.device)
        return local, gathered, image

    def assemble(self, parts, image=None):
        """parts: [world, frames, max_rows, width, 3] (or [world, max_rows, width, 3], or a list of per-rank
        [>= rows[r], width, 3] tensors for one frame).  Returns [frames, height, width, 3] (or [height, width, 3])."""
        single = False
        if isinstance(parts, (list, tuple)):
            buf = torch.zeros((self.world, 1, self.max_rows, self.width, 3), dtype=parts[0].dtype, device=parts[0].device)
            for r, p in enumerate(parts):
                buf[r, 0, : self.rows[r]] = p[: self.rows[r]]
            parts, single = buf, True
        elif parts.dim() == 4:
            parts, single = parts.unsqueeze(1), True
        f = parts.shape[1]
        if image is None:
            image = torch.empty((f, self.height + 1, self.width, 3), dtype=parts.dtype, device=parts.device)
        src = parts.permute(1, 0, 2, 3, 4).reshape(f, self.world * self.max_rows, self.width, 3)
        image.index_copy_(1, self.dest.to(parts.device), src)
        out = image[:, : self.height]
        return out[0] if single else out


def render_sharded(render_fn, plan, rank, group=None, bufs=None, force_collective=False):
    """Unpipelined form: render_fn(shard, n_shards, strip_rows, out) renders this rank's strips of every frame of
    the batch into `out` ([frames, max_rows, width, 3]; rows beyond plan.rows[rank] are padding).  Returns the
    assembled images [frames, height, width, 3] on rank 0, None elsewhere.  One gather per call."""
    local, gathered, image = bufs if bufs is not None else plan.buffers(rank)
    render_fn(rank, plan.world, plan.strip_rows, local)
    if plan.world == 1 and not force_collective:
        return local[:, : plan.height]
    dist.gather(local, gather_list=[gathered[r] for r in range(plan.world)] if rank == 0 else None, dst=0, group=group)
    if rank != 0:
        return None
    return plan.assemble(gathered, image)


class ShardPipeline:
    """Double-buffered launches: submit() renders launch i and starts its gather asynchronously; the gather and
    (on rank 0) the assembly of launch i complete while launch i+1 renders.  drain() finishes everything and
    returns the images of the LAST launch on rank 0.  Every launch is fully rendered, gathered and assembled."""

    def __init__(self, plan, rank, group=None, force_collective=False):
        # force_collective: run the gather + side-stream assembly even with a single rank (a one-rank RCCL group
        # is the only way to exercise that path on a one-GPU box; tests/test_gpu_parity.py)
        self.plan, self.rank, self.group = plan, rank, group
        self.collective = plan.world > 1 or force_collective
        self.slots = [plan.buffers(rank), plan.buffers(rank)]
        self.cuda = plan.device.type == "cuda"
        self.side = torch.cuda.Stream(device=plan.device) if (self.cuda and rank == 0 and self.collective) else None
        self.work = [None, None]             # per slot: handle of the gather in flight
        self.done = [None, None]             # per slot (rank 0, cuda): event recorded after the slot's assembly
        self.launch = [-1, -1]               # per slot: index of the launch occupying it
        self.n = 0
        self.last = None
        self.on_assembled = None             # optional hook(launch_index, images) on rank 0 (tests)
        # timing (cuda only; bench.py): per launch an event behind the render and one behind the launch's gather (+ assembly
        # on rank 0) -- gather_times() turns them into "how long did the exchange step take" and "how much of it was NOT
        # hidden behind the next launch's render"
        self.timing = False
        self.ev_render, self.ev_gathered = {}, {}

    def _complete(self, slot):
        """finish the launch occupying `slot`: its gather, and on rank 0 the assembly (on the side stream when
        there is one, so that the render stream is not held up)"""
        work = self.work[slot]
        if work is None:
            return
        self.work[slot] = None
        local, gathered, image = self.slots[slot]
        if self.side is not None:
            with torch.cuda.stream(self.side):
                work.wait()                                    # the side stream waits for the gather
                self.last = self.plan.assemble(gathered, image)
                if self.on_assembled:
                    self.on_assembled(self.launch[slot], self.last)
                ev = torch.cuda.Event(enable_timing=self.timing)
                ev.record(self.side)
            self.done[slot] = ev
            if self.timing:
                self.ev_gathered[self.launch[slot]] = ev
        else:
            work.wait()                                        # (cuda: the render stream waits; cpu: blocks)
            if self.rank == 0:
                self.last = self.plan.assemble(gathered, image)
                if self.on_assembled:
                    self.on_assembled(self.launch[slot], self.last)
            if self.timing and self.cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(self.plan.device))
                self.ev_gathered[self.launch[slot]] = ev

    def submit(self, render_fn):
        slot = self.n & 1
        self._complete(slot)                                   # normally already done at the previous submit
        if self.done[slot] is not None:                        # rank 0: slot buffers are free once assembled
            torch.cuda.current_stream(self.plan.device).wait_event(self.done[slot])
            self.done[slot] = None
        local, gathered, image = self.slots[slot]
        self.launch[slot] = self.n
        render_fn(self.rank, self.plan.world, self.plan.strip_rows, local)
        if self.timing and self.cuda and self.collective:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.plan.device))
            self.ev_render[self.n] = ev
        if not self.collective:
            self.last = local[:, : self.plan.height]
            if self.on_assembled:
                self.on_assembled(self.n, self.last)
        else:
            self.work[slot] = dist.gather(
                local, gather_list=[gathered[r] for r in range(self.plan.world)] if self.rank == 0 else None,
                dst=0, group=self.group, async_op=True)
            if self.side is not None:
                self._complete(slot ^ 1)                       # assemble the PREVIOUS launch while this one renders
        self.n += 1

    def drain(self):
        first = self.n & 1                                     # older launch first
        self._complete(first)
        self._complete(first ^ 1)
        if self.side is not None:
            torch.cuda.current_stream(self.plan.device).wait_stream(self.side)
        self.done = [None, None]
        return self.last if self.rank == 0 else None

    def gather_times(self):
        """After drain() + a device synchronise, with `timing` set before the launches: per launch (gather_ms, exposed_ms).
        gather_ms = from the end of the launch's render to the end of its exchange step on this rank (rank 0: gather +
        assembly on the side stream; other ranks: their send); exposed_ms = the part of it that was not over when the NEXT
        launch's render ended -- what the exchange step added to the stream of launches (for the last launch: all of it).
        Without a collective (one rank) there is no exchange step: an empty list."""
        out = []
        for i in sorted(self.ev_gathered):
            if i not in self.ev_render:
                continue
            g = self.ev_render[i].elapsed_time(self.ev_gathered[i])
            nxt = self.ev_render.get(i + 1)
            e = max(0.0, nxt.elapsed_time(self.ev_gathered[i])) if nxt is not None else g
            out.append((g, e))
        return out

    def reset_timing(self):
        self.ev_render, self.ev_gathered = {}, {}
