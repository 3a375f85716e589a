"""Multi-GPU execution of the hot path: independent pairs sharded over ranks.

Every pair is independent end to end (per-cloud neighbour search and
InstanceNorm, segment-bounded attention, per-pair pose), so N GPUs run N
replicas of the path on disjoint slices of the pair stream -- the semantics of
the reference's DistributedSampler (src/data_loaders/__init__.py:76) -- with
NO data-path collective.  The only communication is around the timed region:
a barrier on each side and a MAX reduction of the elapsed time.

Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os
import time
from typing import Callable, List, Optional, Tuple


def rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def pair_seeds(rank: int, pairs_per_rank: int, stride: int = 1000) -> List[int]:
    """Seeds of the synthetic pairs a rank owns (disjoint across ranks)."""
    if pairs_per_rank > stride:
        raise ValueError("pairs_per_rank must not exceed the seed stride")
    return [stride * rank + i for i in range(pairs_per_rank)]


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous slice of a global pair stream owned by `rank` (sizes differ
    by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def timed_steps(step: Callable[[], object], steps: int, dist=None, sync: Optional[Callable[[], None]] = None,
                device=None, return_own: bool = False):
    """Run `step` exactly `steps` times between two (barrier + device sync)
    brackets; returns the elapsed wall time, MAX over ranks (and, with
    return_own, this rank's own time up to its device sync as a second value)."""
    import torch

    def fence():
        if sync is not None:
            sync()
        if dist is not None:
            dist.barrier()

    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if sync is not None:
        sync()
    own = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return (elapsed, own) if return_own else elapsed


def gather_ms(ms: float, dist=None, device=None) -> List[float]:
    """Every rank's value, in rank order (all_gather; [ms] without a process group): lets rank 0
    print per-rank step times beside the MAX it reports."""
    import torch
    if dist is None:
        return [round(float(ms), 3)]
    world = dist.get_world_size()
    t = torch.tensor([ms], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [round(float(o.item()), 3) for o in out]


def throughput(pairs_per_rank_per_step: int, steps: int, world: int, elapsed: float) -> float:
    """Whole-job pairs/s: all ranks' pairs over the slowest rank's time."""
    return pairs_per_rank_per_step * steps * world / elapsed
