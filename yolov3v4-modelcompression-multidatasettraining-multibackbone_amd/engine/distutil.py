"""Multi-process timing harness shared by bench.py: one process per GPU, no data-path collective.

Inference over independent frames shards with no exchange ("replicas only"); the only cross-rank traffic is
the barrier that brackets the timed region and the MAX-reduction of the per-rank wall time.  Works on any
torch.distributed backend (``nccl`` = RCCL on the GPU node, ``gloo`` in the CPU tests).
"""
import time

import torch


def sync(device):
    if device is not None and device.type == 'cuda':
        torch.cuda.synchronize(device)


def barrier(dist):
    if dist is not None and dist.is_initialized():
        dist.barrier()


def timed_region(step, steps, dist=None, device=None):
    """Run ``step()`` ``steps`` times between barriers; returns the slowest rank's wall seconds on every rank."""
    sync(device)
    barrier(dist)
    sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync(device)
    barrier(dist)
    sync(device)
    elapsed = time.perf_counter() - t0
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None and device.type == 'cuda' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def aggregate_rate(units_per_rank_step, steps, elapsed, world):
    """Whole-job throughput: units all ranks processed / slowest rank's time."""
    return world * units_per_rank_step * steps / elapsed
