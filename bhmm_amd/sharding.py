"""Multi-GPU decomposition of the hot path: trajectories are independent given the model
(maximum_likelihood.py:383-385, bayesian_sampling.py:288-290), so they are partitioned over
ranks (one process per GPU) and the only exchange is ONE all-reduce of the packed sufficient
statistics per EM iteration / Gibbs sweep -- the distributed form of the Python sums at
maximum_likelihood.py:271-282.  On GPUs the all-reduce runs over RCCL/xGMI on the device
buffer the E-step wrote (torch.distributed backend "nccl"); the message is ~1 KB-35 KB, i.e.
latency-bound, so it is a single un-bucketed call.
"""
import os

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
    """Thin view of torch.distributed (or a single process), bound to the GPU of ONE engine.

    All device traffic of the collectives happens on `device` (the engine's HIP ordinal), never
    on torch's "current" device: a caller that did not `torch.cuda.set_device` would otherwise
    land every rank on GPU 0 (RCCL: "duplicate GPU")."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.dist = None
        self.backend = None
        self.rank, self.world = 0, 1
        self.device = device
        self._bufs = {}
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist
                self.rank = dist.get_rank(group)
                self.world = dist.get_world_size(group)
                self.backend = dist.get_backend(group)
        except ImportError:
            pass

    @property
    def active(self):
        """Several ranks -- or one rank with BHMM_AMD_FORCE_COMM=1, which sends a single process
        through the same collectives (how the RCCL code path is exercised on a one-GPU box)."""
        return self.world > 1 or (self.dist is not None and
                                  os.environ.get("BHMM_AMD_FORCE_COMM") == "1")

    def bind_device(self, device):
        self.device = None if device is None else int(device)
        self._bufs = {}

    def _torch_device(self):
        import torch
        if self.device is None:
            raise RuntimeError("Comm is not bound to an engine device")
        return torch.device("cuda", self.device)

    # -- device-resident statistics ------------------------------------------------------
    def stats_buffer(self, n):
        """fp64 device tensor of n entries on the engine's GPU (cached): the E-step / Gibbs step
        writes its packed statistics into it, `allreduce_stats` sums it over ranks."""
        import torch
        t = self._bufs.get(n)
        if t is None:
            # empty, not zeros: a fill kernel on torch's stream would be unordered against the
            # engine's own stream, which overwrites the whole vector (ranks without a shard call
            # zero_() themselves); the synchronize orders the allocation against first use
            dev = self._torch_device()
            t = torch.empty(int(n), dtype=torch.float64, device=dev)
            torch.cuda.current_stream(dev).synchronize()
            self._bufs[n] = t
        return t

    def allreduce_stats(self, t):
        """Sum the device tensor `t` over ranks and return the result as a host array: with RCCL
        one in-place all-reduce on the device buffer and one device-to-host copy; with gloo
        (tests: several ranks on one GPU) one copy and a host all-reduce."""
        if not self.active:
            return t.cpu().numpy()
        if self.backend == 'nccl':
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            return t.cpu().numpy()
        h = t.cpu()
        self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
        return h.numpy()

    # -- host arrays ---------------------------------------------------------------------
    def _on_wire(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.backend == 'nccl':
            t = t.to(self._torch_device())
        return t

    def allreduce_sum_numpy(self, arr):
        """Sum a host array over ranks (CPU test doubles; small host-side quantities)."""
        if not self.active:
            return arr
        t = self._on_wire(arr)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def broadcast_numpy(self, arr, src=0):
        """Every rank returns rank `src`'s array (same shape/dtype on all ranks)."""
        if not self.active:
            return arr
        t = self._on_wire(arr)
        gsrc = src if self.group is None else self.dist.get_global_rank(self.group, src)
        self.dist.broadcast(t, src=gsrc, group=self.group)
        return t.cpu().numpy()

    def gather_objects(self, obj):
        if not self.active:
            return [obj]
        out = [None] * self.world
        if self.backend == 'nccl':
            import torch
            with torch.cuda.device(self._torch_device()):
                self.dist.all_gather_object(out, obj, group=self.group)
        else:
            self.dist.all_gather_object(out, obj, group=self.group)
        return out
