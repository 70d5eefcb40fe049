"""One-process-per-GPU vertex enumeration: rank-range sharding + the incumbent reduce.

SURVEY.md §8(e): the combination-rank space [0, C(n,m)) is cut into `world` contiguous
ranges; each process enumerates its range on its own GPU (no data-path collective); the only
exchange is the incumbent optimum: one all-gather of (best score, smallest rank within 1e-9 of
it, counts) over RCCL/xGMI, latency-bound, and — only for near-ties across shards — an
all-reduce(min) of recomputed ranks (the shard-independent tie rule of SURVEY.md §8 row E1).

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


def batched_shard_bounds(batch, rank, world):
    """BASELINE configs[4] "1 -> 8 GPUs": participant `rank` solves the LPs [lo, hi) of a batch of
    independent LPs — the convention of lp_batched_shard_bounds in include/simplexmethod_amd.h
    (tests/test_dist_cpu.py checks the two agree): contiguous, disjoint, covering, sizes differing by
    at most one.  There is no collective in the data path: replicas of the code; the caller concatenates
    the outputs in rank order."""
    if batch < 0 or world < 1 or not 0 <= rank < world:
        raise ValueError("batched_shard_bounds: bad (batch, rank, world)")
    return (batch * rank) // world, (batch * (rank + 1)) // world


def batched_solve_sharded(comm, batch, solve_fn):
    """One-process-per-GPU batched simplex (reference shape: a loop of Solver::solve(),
    /root/reference/src/main.cpp:111-113, over independent problems).  `solve_fn(lo, hi)` solves the LPs
    [lo, hi) on this process's device and returns (status[hi-lo], iters[hi-lo]) as int arrays.  No
    exchange is needed for the answers; the summary below (worst status, total pivots, LPs solved) is
    the only thing reduced, after the timed region of whoever calls this."""
    lo, hi = batched_shard_bounds(batch, comm.rank, comm.world)
    status, iters = solve_fn(lo, hi)
    status, iters = np.asarray(status, dtype=np.int64), np.asarray(iters, dtype=np.int64)
    if status.shape != (hi - lo,) or iters.shape != (hi - lo,):
        raise ValueError("solve_fn must return one status and one pivot count per LP of its shard")
    bad = int((status != 0).sum())
    tot = comm.sum_i64(np.array([hi - lo, bad, int(iters.sum())], dtype=np.int64))
    return dict(bounds=(lo, hi), solved=int(tot[0]), not_optimal=int(tot[1]), pivots=int(tot[2]))


def _binom(n, k):
    from math import comb
    return comb(n, k) if 0 <= k <= n else 0


def _unrank(n, m, rank):
    """k-th sorted m-subset of {0..n-1} in lexicographic order (same order as the kernels)."""
    s, a = [], 0
    for t in range(m):
        j = a
        while True:
            cnt = _binom(n - 1 - j, m - 1 - t)
            if rank < cnt:
                break
            rank -= cnt
            j += 1
        s.append(j)
        a = j + 1
    return s


def _lexrank(universe, subset):
    r, a, k = 0, 0, len(subset)
    for t, v in enumerate(subset):
        for j in range(a, v):
            r += _binom(universe - 1 - j, k - 1 - t)
        a = v + 1
    return r


RECORD_COST = 160  # one depth m-7 tree node costs about as much as 160 subsets (scripts/scan_record_cost.py, MI355X, on the round-3 kernels)


def balanced_shard_bounds(n, m, rank, world, record_cost=RECORD_COST):
    """Cut [0, C(n,m)) into `world` contiguous ranges of equal estimated COST rather than equal
    size.  The shared-prefix enumeration pays per subset and per depth m-7 tree node (its record:
    one pivot in the last breadth-first level, two HBM round trips, one thin-kernel pass), and the
    nodes are not spread evenly along the rank axis (late prefixes have few subsets each), so
    equal-size shards differ by up to 2x in run time.  cost(x) = x + record_cost * (number of
    depth m-7 prefixes before the x-th subset), exact combinatorics on the host.  The solver's
    answer does not depend on where the cuts are (tie rule of SURVEY.md 8 row E1)."""
    total = _binom(n, m)
    d0 = m - 7
    # (the cost model is that of the tuned kernels' box; the general kernel's shapes and the direct
    # kernel get equal-size cuts)
    prefix_shape = 7 <= m <= 16 and 2 <= n - m <= 16
    if world <= 1 or not prefix_shape or total < (1 << 20):
        return shard_bounds(total, rank, world)
    universe = n - m + d0

    def cost(x):
        if x >= total:
            return total + record_cost * _binom(universe, d0)
        return x + record_cost * _lexrank(universe, _unrank(n, m, x)[:d0])

    full = cost(total)

    def cut(k):
        if k <= 0:
            return 0
        if k >= world:
            return total
        target = full * k // world
        lo, hi = 0, total
        while lo < hi:
            mid = (lo + hi) // 2
            if cost(mid) < target:
                lo = mid + 1
            else:
                hi = mid
        return lo

    return cut(rank), cut(rank + 1)


class LocalComm:
    """world == 1: reductions are identities."""
    rank, world = 0, 1

    def max_f64(self, v):
        return v

    def min_u64(self, v):
        return v

    def sum_i64(self, a):
        return a

    def gather_i64(self, a):
        return [list(a)]

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

    def gather_i64(self, a):
        """all-gather of a short int64 record: one collective, one host sync."""
        t = self.torch.tensor(list(a), dtype=self.torch.int64, device=self.device)
        out = self.torch.empty(self.world * t.numel(), dtype=self.torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(out, t)
        return out.view(self.world, t.numel()).tolist()

    def barrier(self):
        self.dist.barrier()


def _f64_bits(v):
    return int(np.array([v], dtype=np.float64).view(np.int64)[0])


def _bits_f64(b):
    return float(np.array([b], dtype=np.int64).view(np.float64)[0])


def enum_solve_sharded(comm, total, maximize, range_fn, first_fn, bounds=None):
    """Runs pass 1 / pass 2 on this process's shard and reduces over `comm`.

    range_fn(begin, end) -> (zbest (±inf if no feasible subset), counts[3])
    first_fn(begin, end, zstar, tol) -> smallest qualifying rank or U64_MAX
    bounds: this process's (begin, end); default = equal-size cut (shard_bounds), or pass
            balanced_shard_bounds(n, m, rank, world) for the cost-balanced cut
    Returns dict(feasible, zstar, rank, counts) — identical on every process and for every
    `world` (the tie rule does not depend on how the range was cut).

    ONE collective in the common case: every process applies the tie rule against its OWN best
    score and all-gathers (score, that rank, counts).  The global optimum is the largest score; a
    process whose own best IS the optimum has already applied the rule against the right value,
    so the answer is the smallest of those ranks.  Only if another process's best lies within the
    tolerance below the optimum without being equal to it (two distinct vertices within 1e-9 of
    each other in different shards) are its candidates recomputed against the optimum and reduced
    with a second collective — every process sees the same gathered records, so all take the
    same branch.
    """
    lo, hi = bounds if bounds is not None else shard_bounds(total, comm.rank, comm.world)
    z, counts = range_fn(lo, hi)
    score = z if maximize else -z
    if np.isnan(score):
        score = -np.inf
    local_first = U64_MAX
    if score != -np.inf:
        local_first = first_fn(lo, hi, z, TIE_TOL)
    rec = [_f64_bits(score), min(local_first, (1 << 63) - 1)] + [int(c) for c in counts]
    allrec = comm.gather_i64(rec)
    scores = [_bits_f64(r[0]) for r in allrec]
    gcounts = [sum(r[2 + k] for r in allrec) for k in range(3)]
    gscore = max(scores)
    if gscore == -np.inf:
        return dict(feasible=False, zstar=None, rank=None, counts=gcounts)
    zstar = gscore if maximize else -gscore
    exact = [r[1] for r, s in zip(allrec, scores) if s == gscore]
    near = any(gscore - TIE_TOL <= s < gscore for s in scores)
    if not near:
        grank = min(exact)
    else:  # rare: redo the tie rule against the global optimum where it can matter
        redo = first_fn(lo, hi, zstar, TIE_TOL) if score >= gscore - TIE_TOL else U64_MAX
        grank = comm.min_u64(redo)
    grank = U64_MAX if grank >= (1 << 63) - 1 else grank
    return dict(feasible=True, zstar=zstar, rank=grank, counts=gcounts)
