"""world_size-2 (and 3) gloo runs of the sharded-enumeration driver on the CPU: the per-shard
work is done by the oracle here, so what is under test is the sharding + reduction logic that
bench.py and the multi-GPU path use (simplexmethod_amd/dist.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, m, n, maximize, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import pyoracle as o
    from simplexmethod_amd import dist as lpdist
    from tests import lpcases
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, b, c, _ = lpcases.random_lp(seed, m, n)
    total = o.binom(n, m)

    def range_fn(lo, hi):
        st, z, counts = o.enum_range(A, b, c, maximize, lo, hi)
        return z, counts

    def first_fn(lo, hi, zstar, tol):
        return o.enum_first_within(A, b, c, maximize, lo, hi, zstar, tol)

    res = lpdist.enum_solve_sharded(lpdist.TorchComm("cpu"), total, maximize, range_fn, first_fn)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("maximize", [True, False])
def test_sharded_enumeration_gloo(world, maximize):
    import torch.multiprocessing as mp
    from oracle import pyoracle as o
    from tests import lpcases
    seed, m, n = 17, 5, 12
    A, b, c, _ = lpcases.random_lp(seed, m, n)
    ref = o.enum_solve(A, b, c, maximize, n)
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, m, n, maximize, out))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert len(out) == world
    for r in range(world):
        res = out[r]
        assert res["feasible"] and res["rank"] == ref["rank"] and res["counts"] == ref["counts"]
        assert res["zstar"] == ref["obj"] or abs(res["zstar"] - ref["obj"]) <= 1e-9


def test_shard_bounds_cover_exactly():
    from simplexmethod_amd import dist as lpdist
    for total in (0, 1, 10, 601080390):
        for world in (1, 2, 3, 8):
            cuts = [lpdist.shard_bounds(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_local_comm_single_process():
    from oracle import pyoracle as o
    from simplexmethod_amd import dist as lpdist
    from tests import lpcases
    A, b, c, _, no = lpcases.input_symmetric_lp()
    res = lpdist.enum_solve_sharded(
        lpdist.LocalComm(), 10, True,
        lambda lo, hi: o.enum_range(A, b, c, True, lo, hi)[1:],
        lambda lo, hi, z, tol: o.enum_first_within(A, b, c, True, lo, hi, z, tol))
    assert res == dict(feasible=True, zstar=35.0, rank=2, counts=[7, 3, 0])


def test_balanced_shard_bounds_cover_and_match_oracle_ranking():
    from oracle import pyoracle as o
    from simplexmethod_amd import dist as lpdist
    # the host-side combinatorics agree with the oracle's
    for n, m, k in [(9, 4, 77), (20, 10, 123456), (32, 16, 592101131)]:
        assert lpdist._unrank(n, m, k) == o.unrank(n, m, k).tolist()
    for n, m in [(32, 16), (28, 14), (24, 8)]:
        total = o.binom(n, m)
        for world in (2, 3, 8):
            cuts = [lpdist.balanced_shard_bounds(n, m, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
            assert all(hi > lo for lo, hi in cuts)
    # early shards (few, large subtrees) are longer than late ones in subsets, never the reverse
    cuts = [lpdist.balanced_shard_bounds(32, 16, r, 8) for r in range(8)]
    sizes = [hi - lo for lo, hi in cuts]
    assert sizes[0] > sizes[-1]
    # small trees keep the equal-size cut
    assert lpdist.balanced_shard_bounds(12, 5, 1, 2) == lpdist.shard_bounds(o.binom(12, 5), 1, 2)


class _ThreadComm:
    """Two (or more) simulated processes in threads: enough to drive both branches of
    enum_solve_sharded's exchange without a process group."""

    def __init__(self, rank, world, shared):
        self.rank, self.world, self.s = rank, world, shared

    def _exchange(self, key, value, combine):
        import threading
        with self.s["lock"]:
            self.s.setdefault(key, {})[self.rank] = value
        self.s["barrier"].wait()
        vals = [self.s[key][r] for r in range(self.world)]
        self.s["barrier"].wait()
        return combine(vals)

    def gather_i64(self, a):
        self.s["collectives"][self.rank] += 1
        return self._exchange("g", list(a), lambda v: v)

    def min_u64(self, v):
        self.s["collectives"][self.rank] += 1
        return self._exchange("m", v, min)


def _run_threads(entries, total, world, maximize=True):
    """entries: {rank: score}.  Returns (results per process, collectives per process)."""
    import threading
    from simplexmethod_amd import dist as lpdist

    def range_fn(lo, hi):
        inside = [s for r, s in entries.items() if lo <= r < hi]
        best = (max(inside) if maximize else min(inside)) if inside else (-np.inf if maximize else np.inf)
        return best, [len(inside), (hi - lo) - len(inside), 0]

    def first_fn(lo, hi, z, tol):
        ok = [r for r, s in entries.items() if lo <= r < hi and ((s >= z - tol) if maximize else (s <= z + tol))]
        return min(ok) if ok else lpdist.U64_MAX

    shared = dict(lock=threading.Lock(), barrier=threading.Barrier(world), collectives=[0] * world)
    out = [None] * world

    def work(r):
        out[r] = lpdist.enum_solve_sharded(_ThreadComm(r, world, shared), total, maximize, range_fn, first_fn)

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    return out, shared["collectives"]


def test_exchange_single_collective_and_near_tie_branch():
    # unique optimum: one collective, the winner's own tie rule is already the answer
    out, coll = _run_threads({3: 9.0, 5: 9.5, 12: 10.0, 14: 10.0 - 2e-10}, 16, 2)
    assert all(o == dict(feasible=True, zstar=10.0, rank=12, counts=[4, 12, 0]) for o in out)
    assert coll == [1, 1]
    # a different vertex within 1e-9 of the optimum sits in the OTHER shard at a smaller rank:
    # the tie rule must pick it, which needs the second round
    out, coll = _run_threads({3: 10.0 - 5e-10, 5: 9.0, 12: 10.0}, 16, 2)
    assert all(o["rank"] == 3 and o["zstar"] == 10.0 for o in out)
    assert coll == [2, 2]
    # minimisation mirrors it; three shards; an empty shard
    out, coll = _run_threads({1: 4.0, 9: 4.0 + 3e-10, 10: 7.0}, 18, 3, maximize=False)
    assert all(o["rank"] == 1 and o["zstar"] == 4.0 and o["counts"] == [3, 15, 0] for o in out)
    # nothing feasible anywhere
    out, coll = _run_threads({}, 8, 2)
    assert all(o["feasible"] is False and o["counts"] == [0, 8, 0] for o in out)


# ---- BASELINE configs[4] across GPUs: LP-index shards, replicas, no exchange -----------------------------

def test_batched_shard_bounds_cover_and_match_the_c_abi():
    """Every LP of a batch belongs to exactly one participant, and the Python driver cuts where
    lp_batched_shard_bounds (what a C++ host binds) cuts."""
    import ctypes as C
    from simplexmethod_amd import capi, dist as lpdist
    lib = capi.load()
    lo, hi = C.c_int(0), C.c_int(0)
    for batch in (0, 1, 7, 4096, 4099):
        for world in (1, 2, 3, 8, 16):
            seen = np.zeros(batch, dtype=np.int64)
            sizes = []
            for r in range(world):
                a, b = lpdist.batched_shard_bounds(batch, r, world)
                assert lib.lp_batched_shard_bounds(batch, r, world, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == (a, b)
                seen[a:b] += 1
                sizes.append(b - a)
            assert (seen == 1).all() and max(sizes) - min(sizes) <= 1
    assert lib.lp_batched_shard_bounds(8, 2, 2, C.byref(lo), C.byref(hi)) == capi.BAD_ARG
    assert lib.lp_batched_shard_bounds(8, 0, 0, C.byref(lo), C.byref(hi)) == capi.BAD_ARG
    with pytest.raises(ValueError):
        lpdist.batched_shard_bounds(8, 2, 2)


def _batched_worker(rank, world, port, batch, m, n, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import pyoracle as o
    from simplexmethod_amd import dist as lpdist
    from tests import lpcases
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    objs = {}

    def solve_fn(lo, hi):   # the oracle stands in for the device here: what is under test is the split
        st, it = [], []
        for k in range(lo, hi):
            A, b, c, basis = lpcases.random_lp(k, m, n)
            r = o.simplex_tableau(A, b, c, basis, True, n - m)
            st.append(r["status"])
            it.append(r["iters"])
            objs[k] = r["obj"]
        return st, it

    res = lpdist.batched_solve_sharded(lpdist.TorchComm("cpu"), batch, solve_fn)
    res["objs"] = objs
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_batched_simplex_sharded_gloo(world):
    """world-size-2/3 gloo run of the batched driver: every LP solved exactly once, by the rank the
    convention names, with the answers of an unsharded run."""
    import torch.multiprocessing as mp
    from oracle import pyoracle as o
    from tests import lpcases
    batch, m, n = 11, 6, 14
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_batched_worker, args=(r, world, port, batch, m, n, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert len(out) == world
    ref = {}
    for k in range(batch):
        A, b, c, basis = lpcases.random_lp(k, m, n)
        ref[k] = o.simplex_tableau(A, b, c, basis, True, n - m)
    owner = {}
    for r in range(world):
        res = out[r]
        assert res["solved"] == batch and res["not_optimal"] == 0
        assert res["pivots"] == sum(v["iters"] for v in ref.values())
        lo, hi = res["bounds"]
        assert sorted(res["objs"]) == list(range(lo, hi))
        for k, z in res["objs"].items():
            assert k not in owner
            owner[k] = r
            assert z == ref[k]["obj"]
    assert sorted(owner) == list(range(batch))
