"""Multi-GPU decomposition of the hot path: trajectories are independent given the model
(maximum_likelihood.py:383-385, bayesian_sampling.py:288-290), so they are partitioned over
ranks (one process per GPU) and the only exchange is ONE all-reduce of the packed sufficient
statistics per EM iteration / Gibbs sweep -- the distributed form of the Python sums at
maximum_likelihood.py:271-282.  On GPUs the all-reduce runs over RCCL/xGMI on the device
buffer the E-step wrote (torch.distributed backend "nccl"); the message is ~1 KB-35 KB, i.e.
latency-bound, so it is a single un-bucketed call.
"""
import numpy as np


def lpt_partition(lengths, world_size):
    """Greedy longest-processing-time assignment of trajectories to ranks.
    Returns a list (one entry per rank) of sorted trajectory indices; deterministic, so every
    rank computes the same partition without communication."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = sorted(range(len(lengths)), key=lambda k: (-int(lengths[k]), k))
    load = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for k in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        parts[r].append(k)
        load[r] += int(lengths[k])
    return [sorted(p) for p in parts]


class Comm(object):
    """Thin view of torch.distributed (or a single process)."""

    def __init__(self, group=None):
        self.group = group
        self.dist = None
        self.rank, self.world = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist
                self.rank = dist.get_rank(group)
                self.world = dist.get_world_size(group)
        except ImportError:
            pass

    @property
    def active(self):
        return self.world > 1

    def allreduce_sum_(self, tensor):
        """In-place sum over ranks of a torch tensor (device tensor -> RCCL, CPU -> gloo)."""
        if self.active:
            self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM, group=self.group)
        return tensor

    def allreduce_sum_numpy(self, arr):
        """Sum a host array over ranks (used for int64 Gibbs counts and CPU tests)."""
        if not self.active:
            return arr
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        backend = self.dist.get_backend(self.group)
        if backend == 'nccl':
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def gather_objects(self, obj):
        if not self.active:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.group)
        return out
