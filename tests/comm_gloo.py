"""CPU stand-in for sharding.RcclComm in the world_size-2 tests: the same interface (rank, world,
barrier, allreduce_max, gather) over torch.distributed's gloo backend.  Test infrastructure only --
the product's transport is RCCL through the C ABI (rescan_line_sted_amd/sharding.py)."""
import numpy as np


class GlooComm:
    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def allreduce_max(self, x):
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, local, counts, root=0):
        import torch
        local = np.ascontiguousarray(local, dtype=np.float64)
        shape = [tuple(local.shape[1:])] if self.rank == root else [None]
        self.dist.broadcast_object_list(shape, src=root)
        item = tuple(shape[0])
        nmax = int(max(counts)) if len(counts) else 0
        pad = torch.zeros((nmax,) + item, dtype=torch.float64)
        if local.shape[0]:
            pad[:local.shape[0]] = torch.from_numpy(local.reshape((local.shape[0],) + item))
        bufs = [torch.empty_like(pad) for _ in range(self.world)] if self.rank == root else None
        self.dist.gather(pad, bufs, dst=root)
        if self.rank != root:
            return None
        return torch.cat([bufs[r][:counts[r]] for r in range(self.world)], dim=0).numpy()
