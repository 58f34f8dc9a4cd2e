"""Collective hook for the sharded path: the library asks the host program for an in-place
sum all-reduce of a buffer (include/scanrs_amd.h, scanrs_allreduce_fn); here it is served by
torch.distributed — backend "nccl" (= RCCL over xGMI) on device buffers, "gloo" on host buffers
or, for single-GPU multi-process tests, on device buffers staged through the host."""
from __future__ import annotations

import ctypes

import numpy as np


class _DevArr:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def make_allreduce(dist, device=None, stage_through_host: bool = False, group=None):
    """Returns callback(ptr, count, dtype) -> 0. dtype 0 = f64, 1 = u64 (summed as i64).

    device=None: `ptr` is host memory (CPU tests). Otherwise `ptr` is device memory on `device`;
    with stage_through_host the reduction itself runs on a host copy (gloo backend)."""
    import torch

    def host_cb(ptr, count, dtype):
        ct = ctypes.c_double if dtype == 0 else ctypes.c_int64
        arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), shape=(count,))
        t = torch.from_numpy(arr)
        dist.all_reduce(t, group=group)
        return 0

    def dev_cb(ptr, count, dtype):
        t = torch.as_tensor(_DevArr(ptr, count, "<f8" if dtype == 0 else "<i8"), device=device)
        if stage_through_host:
            h = t.cpu()
            dist.all_reduce(h, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=group)
        torch.cuda.synchronize(device)
        return 0

    return host_cb if device is None else dev_cb


def shard_bounds(n: int, world: int):
    """Equal-count contiguous partition [lo, hi) per rank (cells are i.i.d. in the synthetic bench)."""
    per = (n + world - 1) // world
    return [(min(n, r * per), min(n, (r + 1) * per)) for r in range(world)]
