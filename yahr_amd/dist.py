"""Row-sharded rendering across the GPUs of one node, one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" for the CPU rehearsal in tests).

The reference parallelises over independent pixel tiles inside one process (main.hs:75-96).  Here the image
is cut into strips of `strip_rows` rows dealt round-robin to the ranks (strip s -> rank s % world), every
rank renders its strips with the scene replicated in its own HBM, and there is exactly ONE collective per
frame: a gather of the float RGB strips to rank 0, which scatters them to their image rows
(samplesToImage, main.hs:98-107).  No exchange step exists inside the frame.
"""
import torch
import torch.distributed as dist

from . import shard_global_rows, shard_rows


class ShardPlan:
    """Which rows each rank owns, and where they land in the final image."""

    def __init__(self, height, width, world, strip_rows=8, device="cpu"):
        self.height, self.width, self.world, self.strip_rows = height, width, world, strip_rows
        self.rows = [shard_rows(height, world, r, strip_rows) for r in range(world)]
        self.max_rows = max(self.rows) if self.rows else 0
        assert sum(self.rows) == height
        # destination row of every (rank, local row) in gather order; padding rows are dropped
        idx = [torch.from_numpy(shard_global_rows(height, world, r, strip_rows)) for r in range(world)]
        self.dest = [i.to(device) for i in idx]

    def assemble(self, parts):
        """parts[r]: [max_rows or rows[r], width, 3] tensor of rank r -> full image [height, width, 3]"""
        img = torch.empty((self.height, self.width, 3), dtype=parts[0].dtype, device=parts[0].device)
        for r, p in enumerate(parts):
            if self.rows[r]:
                img.index_copy_(0, self.dest[r], p[: self.rows[r]])
        return img


def render_sharded(render_fn, plan, rank, group=None, out_buffers=None):
    """render_fn(shard, n_shards, strip_rows, out) renders this rank's strips into `out`
    ([max_rows, width, 3], rows beyond plan.rows[rank] are padding).  Returns the assembled image on rank 0,
    None elsewhere.  One gather per call."""
    if out_buffers is None:
        dev = plan.dest[0].device
        local = torch.zeros((plan.max_rows, plan.width, 3), dtype=torch.float32, device=dev)
        parts = [torch.empty_like(local) for _ in range(plan.world)] if rank == 0 else None
    else:
        local, parts = out_buffers
    render_fn(rank, plan.world, plan.strip_rows, local)
    if plan.world == 1:
        return plan.assemble([local])
    dist.gather(local, gather_list=parts if rank == 0 else None, dst=0, group=group)
    if rank != 0:
        return None
    return plan.assemble(parts)
