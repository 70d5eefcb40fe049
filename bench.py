#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X.

BASELINE.json metric: "basis subsets solved/sec (enum) + tableau-update GB/s vs HBM peak".

  value     = vertex-enumeration throughput, basis subsets solved per second, whole job:
              C(32,16) = 601,080,390 subsets (BASELINE.json configs[3], m=16 n=32, seed 0),
              the rank space cut into N contiguous shards (one process per GPU); a step is
              one full enumeration = pass 1 on the shard, all-reduce(max) of the incumbent
              over RCCL, pass 2 (tie rule) and all-reduce(min) of the winning rank.
              The same C(32,16) job at every N => "scaling": "strong".
  roofline  = the tableau rank-1 update of the simplex pivot (BASELINE.json configs[1]:
              m=512, n=1024, seed 0) — algorithmic bytes 16*m*(n+1) per pivot (SURVEY.md
              §8(d)) against the 8 TB/s HBM3E peak, measured with HIP events on the stream
              the kernels run on, on rank 0.
  cpu_baseline = the oracle's restatement of the reference CPU path ("port"), 1 thread, on a
              bounded sample, rank 0, N=1 only.

Usage: python bench.py --gpus N --steps K --warmup W      (N>1: under torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6   # MI355X fp64 vector peak (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--enum-m", type=int, default=16)
    ap.add_argument("--enum-n", type=int, default=32)
    ap.add_argument("--enum-algo", type=int, default=0, help="0 auto, 1 direct, 2 prefix")
    ap.add_argument("--simplex-algo", type=int, default=0, help="0 auto, 1 launch, 2 persistent")
    ap.add_argument("--pivot-m", type=int, default=512)
    ap.add_argument("--pivot-n", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pivot", action="store_true")
    ap.add_argument("--no-batched", action="store_true")
    ap.add_argument("--batch", type=int, default=4096)
    return ap.parse_args()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/): bench.py cannot
    run rocprofv3 on itself; the collection recipe is in profiles/README.md."""
    try:
        data = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_update_traffic.json")))
        return float(data[kernel]["hbm_bytes_per_launch"])
    except Exception:
        return None


def pivot_leg(ctx, args):
    """Simplex on the m=512 x n=1024 random LP (BASELINE configs[1]).

    roofline = the dominant kernel of the solve, the rank-J tableau update of the look-ahead
    path: ALGORITHMIC bytes per launch = (pivots it applies) x 16*m*(n+1)  [SURVEY.md 8(d):
    one pivot reads and writes every tableau element once], divided by the launch duration
    measured with HIP events on the solver's stream.  Also reported: the whole-solve rate
    (selector + update + launch gaps) and the classic one-launch-per-pivot rank-1 update."""
    from simplexmethod_amd import capi
    m, n = args.pivot_m, args.pivot_n
    A, b, c, basis = capi.gen_lp(0, m, n)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    bytes_per_pivot = 16.0 * m * (n + 1)
    rc, st = p.run(algo=args.simplex_algo)          # warm-up solve
    best = None
    for _ in range(5):
        p.reset()
        rc, st = p.run(algo=args.simplex_algo)
        cur = dict(solve_ms=st.solve_ms, pivots=st.pivots, launches=st.launches)
        if best is None or cur["solve_ms"] < best["solve_ms"]:
            best = cur
    # one more solve with every tableau-update launch bracketed by HIP events (slower: the
    # events add ~1-2 us per launch, which is why solve_ms comes from the runs above)
    p.profile(True)
    best["update_ms"], best["update_launches"] = 0.0, 0
    for _ in range(3):
        p.reset()
        rc, st = p.run(algo=args.simplex_algo)
        if st.update_launches and (best["update_launches"] == 0 or st.update_ms < best["update_ms"]):
            best["update_ms"], best["update_launches"] = st.update_ms, st.update_launches
    p.profile(False)
    p.reset()
    upd1_ms = min(p.bench_update(0, 0, 200) for _ in range(3))   # ms per rank-1 update launch
    try:   # rank-J update alone: 200 back-to-back launches between two events
        p.reset()
        updj = min((p.bench_update_rankj(200) for _ in range(3)), key=lambda t: t[0])
    except capi.LPError:
        updj = None
    p.free()
    pivots = max(best["pivots"], 1)
    out = {
        "workload": f"simplex m={m} n={n} seed=0 (BASELINE configs[1])",
        "status": int(rc), "pivots": int(best["pivots"]), "launches": int(best["launches"]),
        "solve_ms": round(best["solve_ms"], 3),
        "us_per_pivot_whole_solve": round(1e3 * best["solve_ms"] / pivots, 3),
        "whole_solve_equiv_GBs": round(bytes_per_pivot * pivots / (best["solve_ms"] * 1e-3) / 1e9, 1),
        "rank1_update_us_per_launch": round(1e3 * upd1_ms, 3),
        "rank1_update_GBs": round(bytes_per_pivot / (upd1_ms * 1e-3) / 1e9, 1),
        "rank1_update_traffic_bytes": pmc_traffic("k_simplex_update") if (m, n) == (512, 1024) else None,
    }
    if best["update_launches"] > 0 and best["update_ms"] > 0:
        out["update_launch_us_inside_solve_event_bracketed"] = round(
            1e3 * best["update_ms"] / best["update_launches"], 3)
    if updj is not None:
        per_launch_ms, pivots_per_launch = updj
        achieved = bytes_per_pivot * pivots_per_launch / (per_launch_ms * 1e-3) / 1e9
        roofline = {
            "kernel": "k_look_update (rank-J Gauss-Jordan update: J staged pivots applied in one "
                      "pass over the tableau)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic("k_look_update") if (m, n) == (512, 1024) else None,
            "launches": 200,
            "avg_launch_us": round(1e3 * per_launch_ms, 3),
            "pivots_per_launch": int(pivots_per_launch),
            "algorithmic_bytes_per_launch": round(bytes_per_pivot * pivots_per_launch, 1),
            "hbm_bytes_per_launch_by_construction": 2.0 * 8.0 * (m + 1) * (8 * ((n + 1 + 7) // 8)),
            "note": "achieved = algorithmic bytes (16*m*(n+1) per pivot, SURVEY 8(d)) x J pivots "
                    "per launch / (HIP-event time of 200 back-to-back launches / 200, launch "
                    "boundary included); each launch moves the tableau once (read + write), "
                    "i.e. 1/J of the algorithmic bytes - see hbm_bytes_per_launch_by_construction",
        }
    else:
        achieved = bytes_per_pivot / (upd1_ms * 1e-3) / 1e9
        roofline = {
            "kernel": "k_simplex_update (rank-1 Gauss-Jordan update, one launch per pivot)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic("k_simplex_update") if (m, n) == (512, 1024) else None,
            "algorithmic_bytes_per_launch": bytes_per_pivot,
        }
    return out, roofline


def batched_leg(ctx, args):
    """BASELINE configs[4]: 4096 random LPs of m=128, n=256 (seeds 0..4095), one LP per workgroup."""
    from simplexmethod_amd import capi
    batch, m, n = args.batch, 128, 256
    A = np.empty((batch, m, n)); b = np.empty((batch, m)); c = np.empty((batch, n))
    basis = np.empty((batch, m), dtype=np.int32)
    for k in range(batch):
        A[k], b[k], c[k], basis[k] = capi.gen_lp(k, m, n)
    p = ctx.batched_problem(A, b, c, basis, True, n - m)
    p.run()
    ms = min(p.run() for _ in range(3))
    d = p.download()
    p.free()
    piv = int(d["iters"].sum())
    return {
        "workload": f"{batch} LPs m={m} n={n} seeds 0..{batch - 1} (BASELINE configs[4])",
        "ms": round(ms, 3), "lps_per_s": round(batch / ms * 1e3, 1), "pivots": piv,
        "all_optimal": bool((d["status"] == 0).all()),
        "equiv_tableau_GBs": round(16.0 * m * (n + 1) * piv / (ms * 1e-3) / 1e9, 1),
    }


def two_phase_leg(ctx, args):
    """a Symmetrical-style MIN problem (no starting basis) through
    lp_simplex_two_phase; host-buffer entry point, so the time includes both uploads."""
    m = k = 256
    rng = np.random.default_rng(0)
    A = np.hstack([rng.uniform(0.0, 1.0, size=(m, k)), -np.eye(m)])
    b = rng.uniform(1.0, 2.0, size=m)
    c = np.concatenate([rng.uniform(0.1, 1.0, size=k), np.zeros(m)])
    ctx.two_phase(A, b, c, maximize=False, n_orig=k)  # warm-up
    best, r = 1e9, None
    for _ in range(3):
        t0 = time.perf_counter()
        r = ctx.two_phase(A, b, c, maximize=False, n_orig=k)
        best = min(best, time.perf_counter() - t0)
    return {
        "workload": f"min c.x, A0 x >= b, A0 {m}x{k} U(0,1) seed 0: canonical [A0|-I] {m}x{m + k}, no starting basis",
        "status": int(r["status"]), "pivots_phase1_driveout_phase2": r["iters"],
        "crash_pivots_phase2": m, "ms_host_inclusive": round(best * 1e3, 3), "objective": r["obj"],
    }


def cpu_baseline_leg(args):
    """Oracle (restatement of the reference CPU path) on this host, 1 thread, bounded sample."""
    from oracle import pyoracle as o
    from simplexmethod_amd import capi
    m, n = args.enum_m, args.enum_n
    A, b, c, _ = capi.gen_lp(0, m, n)
    sample = 12_000_000
    total = o.binom(n, m)
    sample = min(sample, total)
    t0 = time.perf_counter()
    o.enum_range(A, b, c, True, 0, sample)
    t_enum = time.perf_counter() - t0
    pm, pn = args.pivot_m, args.pivot_n
    A2, b2, c2, basis2 = capi.gen_lp(0, pm, pn)
    piv = 40
    t0 = time.perf_counter()
    r = o.simplex_reference(A2, b2, c2, basis2, True, pn - pm, max_iter=piv, dense_eta_product=True)
    t_piv = time.perf_counter() - t0
    return {
        "value": round(sample / t_enum, 1), "unit": "subsets/s", "cores": 1, "kind": "port",
        "sample": f"oracle orc_enum_range on the first {sample} ranks of C({n},{m}) seed 0 "
                  f"({t_enum:.1f} s); reference-shaped simplex (full-pivot LU inverse + dense "
                  f"F*Binv per pivot, SimplexSolover.h:117-133,198-206,446) {r['iters']} pivots "
                  f"of m={pm} n={pn} in {t_piv:.1f} s",
        "simplex_pivots_per_s": round(r["iters"] / t_piv, 3),
        "simplex_equiv_GBs": round(16.0 * pm * (pn + 1) * r["iters"] / t_piv / 1e9, 4),
        "host_cores_available": os.cpu_count(),
        "note": "Eigen unavailable - restated baseline",
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(one process per GPU)")
        args.gpus = world

    import torch
    from simplexmethod_amd import capi
    from simplexmethod_amd import dist as lpdist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP hot path has no CPU fallback")
    # LP_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the
    # collectives run over gloo on the CPU and ranks share devices round-robin.  The driver's
    # multi-GPU runs use the default: nccl (= RCCL over xGMI), one device per rank.
    backend = os.environ.get("LP_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(device_index)
    reduce_device = "cpu" if backend == "gloo" else f"cuda:{device_index}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{device_index}"))
        comm = lpdist.TorchComm(reduce_device)
    else:
        comm = lpdist.LocalComm()

    ctx = capi.Context(device_index)
    m, n = args.enum_m, args.enum_n
    A, b, c, _ = capi.gen_lp(0, m, n)
    ep = ctx.enum_problem(A, b, c, True)
    total = ep.total
    kernel_ms = []

    def range_fn(lo, hi):
        rc, z, counts, st = ep.range(lo, hi, args.enum_algo)
        kernel_ms.append(st.kernel_ms)
        return z, counts

    def first_fn(lo, hi, zstar, tol):
        return ep.first_within(lo, hi, zstar, tol)

    # cost-balanced cut of the rank space (identical answer for any cut; see dist.py)
    my_bounds = lpdist.balanced_shard_bounds(n, m, rank, world)

    def step():
        return lpdist.enum_solve_sharded(comm, total, True, range_fn, first_fn, bounds=my_bounds)

    for _ in range(args.warmup):
        res = step()
    comm.barrier()
    torch.cuda.synchronize()
    kernel_ms.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    comm.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = total * args.steps / elapsed
    flops_per_subset = (2.0 / 3.0) * m ** 3 + 2.0 * m ** 2
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    shard = my_bounds[1] - my_bounds[0]

    winner = ep.vertex(res["rank"], n - m) if res["feasible"] else None
    line = None
    if rank == 0:
        line = {
            "metric": "basis subsets solved/sec (enum) + tableau-update GB/s vs HBM peak",
            "value": round(value, 1), "unit": "subsets/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"vertex enumeration C({n},{m})={total} subsets, dense random LP "
                            f"seed 0 (BASELINE configs[3]); rank space sharded over {world} GPU(s)",
                "enum_algo": args.enum_algo, "subsets_per_gpu": shard,
                "parallelism": f"cost-balanced rank-range shards x{world}, one all-gather of the "
                               "incumbent record (score, rank, counts) per step",
            },
            "enum": {
                "optimum": None if winner is None else winner["obj"],
                "rank": res["rank"], "counts": res["counts"],
                "kernel_ms_pass1_rank0": round(k_ms, 4),
                "algorithmic_flops_per_subset": flops_per_subset,
                "algorithmic_TFLOPs": round(value * flops_per_subset / 1e12, 3),
                "frac_of_fp64_vector_peak": round(value * flops_per_subset / 1e12 /
                                                  (FP64_VECTOR_PEAK_TF * world), 4),
            },
        }
    if rank == 0 and not args.no_pivot:
        pivot, roofline = pivot_leg(ctx, args)
        line["pivot"] = pivot
        line["roofline"] = roofline
    if rank == 0 and not args.no_batched:
        line["batched"] = batched_leg(ctx, args)
        line["two_phase"] = two_phase_leg(ctx, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_leg(args)
    ep.free()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
