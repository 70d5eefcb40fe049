"""One-process-per-GPU vertex enumeration: rank-range sharding + the incumbent reduce.

SURVEY.md §8(e): the combination-rank space [0, C(n,m)) is cut into `world` contiguous
equal ranges; each process enumerates its range on its own GPU (no data-path collective);
the only exchange is the incumbent optimum: all-reduce(max) of the best score (8 bytes over
RCCL/xGMI, latency-bound), then all-reduce(min) of the smallest rank within 1e-9 of it
(the shard-independent tie rule of SURVEY.md §8 row E1), then a sum of the counts.

The collectives are passed in as callables so that the same driver runs over
torch.distributed (nccl == RCCL on ROCm, or gloo in the CPU tests) or, for world == 1,
with no communication at all.
"""
import numpy as np

TIE_TOL = 1e-9  # mirrors Solver::EPS, /root/reference/src/SimplexSolover.h:13
U64_MAX = (1 << 64) - 1


def shard_bounds(total, rank, world):
    """Contiguous equal cut of [0, total): shard `rank` of `world`."""
    return (total * rank) // world, (total * (rank + 1)) // world


class LocalComm:
    """world == 1: reductions are identities."""
    rank, world = 0, 1

    def max_f64(self, v):
        return v

    def min_u64(self, v):
        return v

    def sum_i64(self, a):
        return a

    def barrier(self):
        pass


class TorchComm:
    """torch.distributed collectives on `device` ('cuda:<i>' with the nccl/RCCL backend, or
    'cpu' with gloo)."""

    def __init__(self, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def max_f64(self, v):
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_u64(self, v):
        # ranks are < 2^63 here (C(64,32) < 2^61), so int64 carries them; U64_MAX -> int64 max
        x = min(v, (1 << 63) - 1)
        t = self.torch.tensor([x], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        r = int(t.item())
        return U64_MAX if r == (1 << 63) - 1 else r

    def sum_i64(self, a):
        t = self.torch.tensor(list(a), dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    def barrier(self):
        self.dist.barrier()


def enum_solve_sharded(comm, total, maximize, range_fn, first_fn):
    """Runs pass 1 / pass 2 on this process's shard and reduces over `comm`.

    range_fn(begin, end) -> (zbest (±inf if no feasible subset), counts[3])
    first_fn(begin, end, zstar, tol) -> smallest qualifying rank or U64_MAX
    Returns dict(feasible, zstar, rank, counts) — identical on every process and for every
    `world` (the tie rule does not depend on how the range was cut).
    """
    lo, hi = shard_bounds(total, comm.rank, comm.world)
    z, counts = range_fn(lo, hi)
    score = z if maximize else -z
    if np.isnan(score):
        score = -np.inf
    gscore = comm.max_f64(score)
    gcounts = comm.sum_i64(counts)
    if gscore == -np.inf:
        return dict(feasible=False, zstar=None, rank=None, counts=gcounts)
    zstar = gscore if maximize else -gscore
    local_first = first_fn(lo, hi, zstar, TIE_TOL) if score >= gscore - TIE_TOL else U64_MAX
    grank = comm.min_u64(local_first)
    return dict(feasible=True, zstar=zstar, rank=grank, counts=gcounts)
