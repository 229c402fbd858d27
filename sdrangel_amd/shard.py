"""Multi-GPU harness: one process per GPU, one independent device stream (or several) per process.

SURVEY.md §8(e): the path shards over DEVICE STREAMS -- stream s (and all its channels) lives on
GPU s mod N; there is no exchange step, hence no collective on the data path.  torch.distributed
is used only to rendezvous, barrier and max-reduce the timing (RCCL on GPUs, gloo in CPU tests).
"""
from __future__ import annotations

import os
import time
from typing import Callable


def world_from_env():
    """(rank, local_rank, world_size) as torch.distributed.run exports them; single process otherwise."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def streams_of_rank(n_streams: int, rank: int, world: int) -> list[int]:
    """stream s -> rank s mod world (FileSource device set s -> GPU s mod 8 in BASELINE config 5)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return [s for s in range(n_streams) if s % world == rank]


def init_process_group(backend: str, rank: int, world: int, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def timed_region(step: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None], dist=None, device=None) -> float:
    """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + sync on both sides;
    returns the MAX elapsed seconds over all ranks."""
    import torch
    for _ in range(warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=device if device is not None else "cpu")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.barrier()
    return float(el.item())


def aggregate_rate(units_per_rank_per_step: int, world: int, steps: int, elapsed_max: float) -> float:
    """whole-job throughput (units / s): weak scaling, every rank does the same amount of work"""
    return world * steps * units_per_rank_per_step / elapsed_max
