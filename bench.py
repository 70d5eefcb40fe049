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
    ap.add_argument("--no-large-shape", action="store_true",
                    help="skip the 2048x4096 leg (profiles: keeps k_simplex_update's per-launch average to the 512x1024 workload)")
    ap.add_argument("--no-batched", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the two rocprofv3 --pmc passes for `traffic` (the committed profile is used if its hash matches)")
    ap.add_argument("--batch", type=int, default=4096)
    return ap.parse_args()


def kernel_source_hash():
    """sha256 over the HIP sources: a committed PMC file counts only for the kernels it was taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "simplexmethod_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


PMC_PROFILE = "r04_pmc_traffic.json"


def pmc_profile():
    """profiles/r04_pmc_traffic.json (written by scripts/pmc_to_json.py from separate rocprofv3 --pmc
    passes: FETCH_SIZE and WRITE_SIZE of scripts/pmc_traffic.py, the VALU and fp64 instruction counters of
    scripts/pmc_enum.py; plus the rocprofv3 --kernel-trace average of k_simplex_update): HBM bytes per
    launch of the tableau kernels, executed fp64 operations per enumerated subset.  The tableau kernels' traffic is
    measured again by every default run (live_pmc_traffic: child processes under rocprofv3 --pmc); the other
    counters are reported only when the file was taken on exactly these kernel sources, otherwise they are null."""
    try:
        data = json.load(open(os.path.join(ROOT, "profiles", PMC_PROFILE)))
    except Exception:
        return {}
    return data if data.get("kernel_source_hash") == kernel_source_hash() else {}


def traffic_of(prof, kernel):
    """HBM bytes per launch: measured by this run's own PMC passes (live_pmc_traffic) if they ran, else the
    committed profile's figure for these kernel sources, else None."""
    if kernel in _LIVE_PMC.get("kernels", {}):
        return float(_LIVE_PMC["kernels"][kernel]["hbm_bytes_per_launch"])
    try:
        return float(prof["kernels"][kernel]["hbm_bytes_per_launch"])
    except Exception:
        return None


_LIVE_PMC = {}
_PMC_KERNELS = ("k_simplex_resident", "k_simplex_update", "k_look_update", "k_simplex_overlap")


def live_pmc_traffic(timeout_s=150):
    """HBM traffic of the tableau kernels measured IN THIS RUN (the line's `traffic` used to come
    from a committed file only): two child processes, `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
    passes, no trace domains: MI355X_MICROARCH.md's HBM recipe) on scripts/pmc_traffic.py, before this process
    touches the GPU; per launch (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes — rocprofv3 reports KB and FETCH_SIZE
    reads half of a wide read on gfx950, the guide's correction.  Skipped (the committed, hash-gated profile is
    used instead, and the line says so) when rocprofv3 is missing, when this process itself runs under a profiler,
    or when a pass fails or times out."""
    import collections, csv, glob, shutil, subprocess, tempfile, time
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTX")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return "skipped: this process runs under a profiler"
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return "skipped: rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="lp_pmc_")
    t0 = time.perf_counter()
    try:
        per = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            r = subprocess.run([tool, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "scripts", "pmc_traffic.py")], capture_output=True, text=True,
                               timeout=timeout_s, cwd=out)
            files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
            if r.returncode != 0 or not files:
                return "skipped: rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            acc = collections.defaultdict(lambda: [0.0, 0])
            for row in csv.DictReader(open(files[-1])):
                if row["Counter_Name"] != counter:
                    continue
                for k in _PMC_KERNELS:
                    if k in row["Kernel_Name"]:
                        acc[k][0] += float(row["Counter_Value"])
                        acc[k][1] += 1
            per[counter] = {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}
        kernels = {}
        for k in _PMC_KERNELS:
            if k in per["FETCH_SIZE"] and k in per["WRITE_SIZE"]:
                f, w = per["FETCH_SIZE"][k], per["WRITE_SIZE"][k]
                kernels[k] = {"FETCH_SIZE_KB": round(f[0], 2), "WRITE_SIZE_KB": round(w[0], 2), "launches_sampled": [f[1], w[1]],
                              "hbm_bytes_per_launch": round((2.0 * f[0] + w[0]) * 1024.0, 1)}
        if not kernels:
            return "skipped: no kernel of the workload in the counter files"
        _LIVE_PMC["kernels"] = kernels
        return "measured in this run: rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE, on scripts/pmc_traffic.py (%.0f s)" % (time.perf_counter() - t0)
    except subprocess.TimeoutExpired:
        return "skipped: a rocprofv3 --pmc pass exceeded %d s" % timeout_s
    except Exception as ex:   # never let the profiler cost the bench its line
        return "skipped: %s" % (str(ex)[:120],)
    finally:
        shutil.rmtree(out, ignore_errors=True)


def pivot_leg(ctx, args):
    """Simplex on the m=512 x n=1024 random LP (BASELINE configs[1]).

    Every figure is ALGORITHMIC bytes (one pivot reads and writes every tableau
    element once, 16*m*(n+1) B) divided by a HIP-event time on the solver's stream, so a fraction
    of the 8 TB/s HBM peak can exceed 1 only if the kernel does not move those bytes - which is
    stated next to it (traffic, bytes moved by construction)."""
    from simplexmethod_amd import capi
    m, n = args.pivot_m, args.pivot_n
    A, b, c, basis = capi.gen_lp(0, m, n)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    bytes_per_pivot = 16.0 * m * (n + 1)
    prof = pmc_profile() if (m, n) == (512, 1024) else {}

    def best_of(algo, reps=7):
        """MEAN over the repetitions (what the rooflines are computed from), the minimum beside it."""
        runs = []
        for _ in range(reps):
            p.reset()
            rc, st = p.run(algo=algo)
            runs.append(dict(rc=int(rc), solve_ms=st.solve_ms, pivots=st.pivots, launches=st.launches,
                             kernel_ms=st.update_ms, kernel_launches=st.update_launches,
                             algo_used=int(st.algo_used), fell_back=int(st.fell_back)))
        out = dict(runs[0])
        out["solve_ms"] = sum(r["solve_ms"] for r in runs) / len(runs)
        out["kernel_ms"] = sum(r["kernel_ms"] for r in runs) / len(runs)
        out["solve_ms_min"] = min(r["solve_ms"] for r in runs)
        out["kernel_ms_min"] = min(r["kernel_ms"] for r in runs)
        out["fell_back"] = max(r["fell_back"] for r in runs)
        out["reps"] = len(runs)
        return out

    p.run(algo=args.simplex_algo)                   # warm-up solve
    auto = best_of(args.simplex_algo)
    pivots = max(auto["pivots"], 1)
    resident = auto["algo_used"] == capi.SIMPLEX_RESIDENT and not auto["fell_back"]   # the chip-resident path ran
    try:
        look = best_of(capi.SIMPLEX_LOOKAHEAD, 3)
    except capi.LPError:
        look = None
    # where a pivot of the chip-resident kernel goes: interval timings of diagnostic builds (two clock reads per
    # pivot each: scripts/resident_marks.py), committed with the hash of the kernel sources they were taken on
    phases = None
    if resident and (m, n) == (512, 1024):
        try:
            mk = json.load(open(os.path.join(ROOT, "profiles", "r04_resident_marks.json")))
            if mk.get("kernel_source_hash") == kernel_source_hash():
                phases = {"source": "profiles/r04_resident_marks.json (scripts/resident_marks.py)",
                          "cycles_per_pivot_mean_over_workgroups": mk["intervals"],
                          "critical_path_cycles_per_pivot": mk.get("critical_path_cycles_per_pivot")}
        except Exception:
            phases = None
    p.reset()
    upd1_ms = min(p.bench_update(0, 0, 200) for _ in range(3))   # ms per rank-1 update launch
    try:   # rank-J update alone: 200 back-to-back launches between two events
        p.reset()
        updj = min((p.bench_update_rankj(200) for _ in range(3)), key=lambda t: t[0])
    except capi.LPError:
        updj = None
    p.free()
    # one-shot entry point: host buffers in, host buffers out (upload, solve, download, free)
    ctx.simplex_solve(A, b, c, basis, True, n - m)
    t_one = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.simplex_solve(A, b, c, basis, True, n - m)
        t_one = min(t_one, time.perf_counter() - t0)

    out = {
        "workload": f"simplex m={m} n={n} seed=0 (BASELINE configs[1])",
        "algorithm": "chip-resident tableau (one launch per solve)" if resident else "launch-based",
        "algo_used": {capi.SIMPLEX_LAUNCH: "launch", capi.SIMPLEX_LOOKAHEAD: "lookahead",
                      capi.SIMPLEX_RESIDENT: "resident"}.get(auto["algo_used"], str(auto["algo_used"])),
        "fell_back": bool(auto["fell_back"]),
        "status": auto["rc"], "pivots": int(auto["pivots"]), "launches": int(auto["launches"]),
        "solve_ms": round(auto["solve_ms"], 3), "solve_ms_min": round(auto["solve_ms_min"], 3),
        "timing": f"mean of {auto['reps']} solves (HIP events on the solver's stream); *_min = the fastest of them",
        "us_per_pivot_whole_solve": round(1e3 * auto["solve_ms"] / pivots, 3),
        "us_per_pivot_whole_solve_min": round(1e3 * auto["solve_ms_min"] / pivots, 3),
        "one_shot_host_buffers_ms": round(1e3 * t_one, 3),
        "budget_us_per_pivot_at_70pct_of_8TBs": round(bytes_per_pivot / (0.7 * HBM_PEAK_GBS * 1e9) * 1e6, 3),
    }
    out["resident_kernel_intervals"] = phases   # null unless the committed marks belong to these kernel sources
    if look is not None:
        out["lookahead_path_solve_ms"] = round(look["solve_ms"], 3)
        out["lookahead_path_us_per_pivot"] = round(1e3 * look["solve_ms"] / max(look["pivots"], 1), 3)

    whole = bytes_per_pivot * pivots / (auto["solve_ms"] * 1e-3) / 1e9
    roofline_whole = {
        "what": "whole pivot (pricing + ratio test + update + hand-offs): algorithmic bytes / (solve time / pivots)",
        "bound": "handoff-latency" if resident else "hbm", "roofline_it_is_priced_against": "hbm", "achieved": round(whole, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(whole / HBM_PEAK_GBS, 4), "us_per_pivot": out["us_per_pivot_whole_solve"],
    }
    r1 = bytes_per_pivot / (upd1_ms * 1e-3) / 1e9
    roofline_rank1 = {
        "kernel": "k_simplex_update (rank-1 Gauss-Jordan update, one launch per pivot: the tableau streams "
                  "through HBM/L2 once per pivot; the path for tableaus that do not fit on chip)",
        "bound": "hbm", "achieved": round(r1, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(r1 / HBM_PEAK_GBS, 4), "traffic": traffic_of(prof, "k_simplex_update"),
        "launches": 200, "avg_launch_us": round(1e3 * upd1_ms, 3),
        "algorithmic_bytes_per_launch": bytes_per_pivot,
        "timing": "HIP events around 200 back-to-back launches (this run)",
    }
    # the same kernel's average under rocprofv3 --kernel-trace (committed with the PMC file, same sources):
    # the profiler's per-launch duration excludes the back-to-back overlap of the launch ramps
    try:
        rp_us = float(prof["kernels"]["k_simplex_update"]["rocprof_avg_launch_us"])
        roofline_rank1["rocprof_avg_launch_us"] = rp_us
        roofline_rank1["rocprof_frac"] = round(bytes_per_pivot / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    except Exception:
        roofline_rank1["rocprof_avg_launch_us"] = None
        roofline_rank1["rocprof_frac"] = None
    rankj = None
    if updj is not None:
        per_launch_ms, J = updj
        moved = 2.0 * 8.0 * (m + 1) * (8 * ((n + 1 + 7) // 8))     # one read + one write of the padded tableau
        rankj = {
            "kernel": "k_look_update (rank-J update of the look-ahead path)", "J": int(J),
            "avg_launch_us": round(1e3 * per_launch_ms, 3),
            "bytes_moved_per_launch_by_construction": moved,
            "bytes_moved_GBs": round(moved / (per_launch_ms * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak_by_bytes_moved": round(moved / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": traffic_of(prof, "k_look_update"),
            "algorithmic_equiv_GBs_no_frac": round(bytes_per_pivot * J / (per_launch_ms * 1e-3) / 1e9, 1),
        }
    if resident and auto["kernel_ms"] > 0:
        k = bytes_per_pivot * pivots / (auto["kernel_ms"] * 1e-3) / 1e9
        roofline = {
            "kernel": "k_simplex_resident (every pivot of the solve in ONE launch: the tableau stays in the "
                      "registers of ceil(n/32) co-resident workgroups; per pivot one all-to-all hand-off through L2: a 16-byte "
                      "pricing record, eight 32-byte ratio-test slice records and an 8 KB candidate column per workgroup)",
            "bound": "handoff-latency", "roofline_it_is_priced_against": "hbm",
            "achieved": round(k, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(k / HBM_PEAK_GBS, 4), "traffic": traffic_of(prof, "k_simplex_resident"),
            "launches": auto["reps"], "avg_launch_us": round(1e3 * auto["kernel_ms"], 3),
            "min_launch_us": round(1e3 * auto["kernel_ms_min"], 3),
            "pivots_per_launch": int(pivots),
            "algorithmic_bytes_per_launch": bytes_per_pivot * pivots,
            "hbm_bytes_per_launch_by_construction": 2.0 * 8.0 * (m + 1) * (n + 1),
            "note": "achieved/frac = SURVEY 8(d)'s algorithmic rate, 16*m*(n+1) B x pivots of the launch / HIP-event "
                    "time of the launch, against the HBM peak the north star prices the pivot on; the kernel "
                    "reads the tableau from HBM once and writes it once per SOLVE, so the HBM traffic is "
                    "1/pivots of the algorithmic bytes and what binds it is the latency of the per-pivot "
                    "hand-off (bound), not bandwidth; the HBM-streaming kernels are in roofline_rank1_update / rankj_update",
        }
    else:
        roofline = dict(roofline_rank1)
    return out, roofline, roofline_whole, roofline_rank1, rankj


def large_shape_leg(ctx, args):
    """A tableau beyond the chip-resident shapes (m > 960): 2048 x 4096, 67 MB, 300 pivots.  AUTO takes the
    one-launch-per-pivot path (simplex_overlap.hip): the out-of-place rank-1 update of pivot k streams the
    tableau while one more workgroup of the same launch selects pivot k+1.  Here the pivot really is the
    HBM-bound rank-1 update SURVEY 8(d) prices: algorithmic bytes 16*m*(n+1) per pivot / time."""
    from simplexmethod_amd import capi
    m, n, piv = 2048, 4096, 300
    A, b, c, basis = capi.gen_lp(0, m, n)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    bytes_per_pivot = 16.0 * m * (n + 1)
    names = {capi.SIMPLEX_LAUNCH: "launch", capi.SIMPLEX_LOOKAHEAD: "lookahead", capi.SIMPLEX_RESIDENT: "resident",
             capi.SIMPLEX_OVERLAP: "overlap"}

    def run(algo):   # (mean of 3 solves behind a warm-up that takes the path's one-time allocations: the second tableau)
        p.reset()
        p.run(algo=algo, max_iter=piv)
        acc, last = 0.0, None
        for _ in range(3):
            p.reset()
            rc, st = p.run(algo=algo, max_iter=piv)
            acc += st.solve_ms
            last = st
        return (acc / 3.0, last.pivots, int(last.algo_used), int(last.launches))
    auto = run(capi.SIMPLEX_AUTO)
    two = run(capi.SIMPLEX_LAUNCH)
    p.reset()
    upd_ms = min(p.bench_update(m // 3, n // 5, 50) for _ in range(3))
    p.free()
    us = 1e3 * auto[0] / max(auto[1], 1)
    rate = bytes_per_pivot / us / 1e3
    traffic = traffic_of(pmc_profile(), "k_simplex_overlap")   # HBM bytes per launch (= per pivot) from the PMC passes, same sources only
    upd = bytes_per_pivot / (upd_ms * 1e3) / 1e3
    return {
        "workload": f"simplex m={m} n={n} seed 0, first {piv} pivots (tableau {8e-6 * (m + 1) * (n + 1):.0f} MB)",
        "algo_used": names.get(auto[2], str(auto[2])), "pivots": int(auto[1]), "launches": auto[3],
        "us_per_pivot": round(us, 3),
        "roofline": {"what": "whole pivot (selection overlapped with the update), algorithmic bytes / time",
                     "bound": "hbm", "achieved": round(rate, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(rate / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_over_algorithmic_bytes": None if traffic is None else round(traffic / bytes_per_pivot, 3),
                     "algorithmic_bytes_per_launch": bytes_per_pivot},
        "two_launches_per_pivot_us": round(1e3 * two[0] / max(two[1], 1), 3),
        "rank1_update_kernel_alone": {"kernel": "k_simplex_update (in place)", "avg_launch_us": round(1e3 * upd_ms, 3),
                                      "achieved": round(upd, 1), "unit": "GB/s", "frac": round(upd / HBM_PEAK_GBS, 4),
                                      "timing": "HIP events around 50 back-to-back launches"},
    }


def batched_leg(ctx, args, rank, world, reduce_device):
    """BASELINE configs[4]: 4096 random LPs of m=128, n=256 (seeds 0..4095), one LP per workgroup,
    "1 -> 8 GPUs": the LPs are independent, so rank r uploads and solves the LPs
    [batch*r/N, batch*(r+1)/N) (lp_batched_shard_bounds) and nobody exchanges anything: replicas of the
    code, no collective in the data path.  The time is the MAX over the ranks of each rank's best
    device time (HIP events around its launch), the pivots are summed over the ranks.

    Rooflines ("LDS bandwidth if the tableau is chip-resident"): the kernel keeps each
    LP's condensed tableau (m+1) x (n-m+1) in registers and moves only what crosses threads through LDS.
      fp64: executed fused multiply-adds of the rank-1 update, 2*(m+1)*(n-m+1) flop per pivot, against
            the fp64 vector peak (the ratio test's divisions, the eta column and pricing are not counted).
      lds : the bytes one pivot must move through LDS by construction of the register form - every
            updating thread reads its share of the eta column (m+1 doubles per column of threads) and one
            pivot-row entry, the owners store the entering column, the pivot row and the eta column -
            against 128 B/clk/CU at the 2.4 GHz peak clock."""
    from simplexmethod_amd import capi
    import ctypes as C
    batch, m, n = args.batch, 128, 256
    lo, hi = C.c_int(0), C.c_int(0)
    capi.load().lp_batched_shard_bounds(batch, rank, world, C.byref(lo), C.byref(hi))
    lo, hi = lo.value, hi.value
    mine = hi - lo
    ms, piv, all_opt = 0.0, 0, True
    if mine > 0:
        A = np.empty((mine, m, n)); b = np.empty((mine, m)); c = np.empty((mine, n))
        basis = np.empty((mine, m), dtype=np.int32)
        for k in range(mine):
            A[k], b[k], c[k], basis[k] = capi.gen_lp(lo + k, m, n)
        p = ctx.batched_problem(A, b, c, basis, True, n - m)
        p.run()
        ms = min(p.run() for _ in range(3))
        d = p.download()
        p.free()
        piv = int(d["iters"].sum())
        all_opt = bool((d["status"] == 0).all())
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([ms], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        q = torch.tensor([piv, 0 if all_opt else 1, mine], dtype=torch.int64, device=reduce_device)
        dist.all_reduce(q, op=dist.ReduceOp.SUM)
        piv, all_opt, solved = int(q[0].item()), int(q[1].item()) == 0, int(q[2].item())
    else:
        solved = mine
    if rank != 0:
        return None
    cols = n - m + 1                                   # condensed tableau: non-basic columns + right-hand side
    flops = 2.0 * (m + 1) * cols * piv
    tf = flops / (ms * 1e-3) / 1e12
    # LDS bytes per pivot and LP by construction (batched_simplex.hip, register form): thread (column j,
    # row group g) reads the eta entries of its rows (8 B each, broadcast pairs) and its pivot-row entry;
    # the entering column, the eta column and the pivot row are each stored once and the scanning wave
    # reads the entering column, xB and the reduced-cost row once
    lds_bytes = 8.0 * ((m + 1) * cols + cols) + 8.0 * (3 * (m + 1) + cols) + 8.0 * (2 * (m + 1) + cols)
    lds_peak_cu = 128.0 * 2.4e9                        # B/s per CU (MI355X_MICROARCH.md: 128 B/clk/CU)
    cus = 256 * world
    lds_floor_us = lds_bytes / lds_peak_cu * 1e6       # per pivot of one LP on its CU
    us_per_pivot_cu = ms * 1e3 * cus / max(piv, 1)
    return {
        "workload": f"{batch} LPs m={m} n={n} seeds 0..{batch - 1} (BASELINE configs[4])",
        "n_gpus": world, "lps_solved": solved,
        "parallelism": f"LP-index shards x{world} (lp_batched_shard_bounds): replicas, no exchange",
        "ms": round(ms, 3), "lps_per_s": round(batch / ms * 1e3, 1), "pivots": piv,
        "all_optimal": all_opt,
        "roofline": {"bound": "fp64-vector", "achieved": round(tf, 3), "peak": FP64_VECTOR_PEAK_TF * world,
                     "unit": "TFLOP/s", "frac": round(tf / (FP64_VECTOR_PEAK_TF * world), 4),
                     "what": "executed fp64 flops of the rank-1 updates, 2*(m+1)*(n-m+1) per pivot"},
        "roofline_lds": {"bound": "lds", "lds_bytes_per_pivot_by_construction": lds_bytes,
                         "floor_us_per_pivot_per_cu": round(lds_floor_us, 3),
                         "achieved_us_per_pivot_per_cu": round(us_per_pivot_cu, 3),
                         "frac": round(lds_floor_us / us_per_pivot_cu, 4)},
    }


def enum_inputs_leg(ctx, args):
    """The enumeration's throughput depends on the data (wave-level early exit of infeasible
    subsets, number of feasible bases): the headline seed next to three more seeds and to the worst
    case for the feasible list, a fully degenerate LP (b = 0: every non-singular basis is feasible;
    the first pass stops once its list is far over capacity and the range is redone in the dense form —
    every subset's score by rank, no list).  One step = pass 1 + tie rule; the degenerate case is
    reported for its first call (stopped listing pass + 4.8 GB allocation + dense pass) and for a
    repeat (dense pass only)."""
    from simplexmethod_amd import capi
    m, n = args.enum_m, args.enum_n
    out = []
    for label, seed, degenerate in [("seed 1", 1, False), ("seed 2", 2, False), ("seed 3", 3, False),
                                    ("seed 0 with b = 0 (fully degenerate)", 0, True)]:
        A, b, c, _ = capi.gen_lp(seed, m, n)
        if degenerate:
            b = np.zeros_like(b)
        p = ctx.enum_problem(A, b, c, True)
        best, res, first = 1e9, None, None
        for _ in range(2 if degenerate else 3):
            t0 = time.perf_counter()
            rc, z, counts, st = p.range(0, p.total, args.enum_algo)
            k = p.first_within(0, p.total, z) if rc == 0 else None
            dt = time.perf_counter() - t0
            first = dt if first is None else first
            best = min(best, dt)
            res = (rc, z, counts, k)
        p.free()
        row = {"input": label, "ms_per_step": round(1e3 * best, 3),
               "subsets_per_s": round(p.total / best, 1), "status": int(res[0]),
               "optimum": res[1], "rank": res[3], "counts": res[2]}
        if degenerate:
            row["first_call_ms"] = round(1e3 * first, 3)
            row["first_call_subsets_per_s"] = round(p.total / first, 1)
        out.append(row)
    return out


def enum_wide_leg(ctx):
    """Shapes outside the tuned kernels' 16 x 16 box: 32-row records (m > 16) and more than 16 selectable
    columns run the general leaf kernel over the same shared prefixes; the direct kernel (one m x m solve
    per subset) timed beside it on the smaller two.  One step = pass 1 + tie rule, seed 5."""
    from simplexmethod_amd import capi
    out = []
    for m, n, with_direct in [(18, 30, True), (12, 32, True), (16, 34, False)]:
        A, b, c, _ = capi.gen_lp(5, m, n)
        p = ctx.enum_problem(A, b, c, True)
        row = {"shape": f"C({n},{m})", "subsets": p.total}
        answers = {}
        for name, algo in (("prefix", capi.ENUM_PREFIX), ("direct", capi.ENUM_DIRECT)):
            if name == "direct" and not with_direct:
                continue
            best = 1e9
            for _ in range(2 if name == "prefix" else 1):
                t0 = time.perf_counter()
                rc, z, counts, st = p.range(0, p.total, algo)
                k = p.first_within(0, p.total, z) if rc == 0 else None
                best = min(best, time.perf_counter() - t0)
            answers[name] = (rc, z, counts, k)
            row[f"{name}_ms"] = round(1e3 * best, 3)
            row[f"{name}_subsets_per_s"] = round(p.total / best, 1)
        if len(answers) == 2:
            row["same_answers"] = answers["prefix"] == answers["direct"]
        row["optimum"], row["rank"] = answers["prefix"][1], answers["prefix"][3]
        p.free()
        out.append(row)
    return out


def enum_config2_leg(ctx):
    """BASELINE configs[2]: vertex enumeration n = 28, m = 14 (C(28,14) = 40.1 M subsets) on one GPU, seed 0:
    pass 1 + tie rule, device-resident problem, best of 5; an eighth-of-C(32,16)-sized problem, so the fixed
    costs of a pass weigh as they do on an 8-way shard."""
    from simplexmethod_amd import capi
    m, n = 14, 28
    A, b, c, _ = capi.gen_lp(0, m, n)
    p = ctx.enum_problem(A, b, c, True)
    p.range(0, p.total)
    best, kern = 1e9, 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        rc, z, counts, st = p.range(0, p.total)
        k = p.first_within(0, p.total, z) if rc == 0 else None
        best = min(best, time.perf_counter() - t0)
        kern = min(kern, st.kernel_ms)
    out = {"workload": f"vertex enumeration C({n},{m}) = {p.total} subsets, seed 0 (BASELINE configs[2])",
           "ms": round(1e3 * best, 3), "kernel_ms": round(kern, 3), "subsets_per_s": round(p.total / best, 1),
           "status": int(rc), "optimum": z, "rank": k, "counts": list(counts)}
    p.free()
    return out


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
        "phase2": "continues on the phase-I tableau (costs re-priced on the device, artificial columns barred)",
        "ms_host_inclusive": round(best * 1e3, 3), "objective": r["obj"],
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
    t0 = time.perf_counter()
    rt = o.simplex_tableau(A2, b2, c2, basis2, True, pn - pm)     # the SAME algorithm the GPU runs, 1 core
    t_tab = time.perf_counter() - t0
    return {
        "value": round(sample / t_enum, 1), "unit": "subsets/s", "cores": 1, "kind": "port",
        "sample": f"oracle orc_enum_range on the first {sample} ranks of C({n},{m}) seed 0 "
                  f"({t_enum:.1f} s); reference-shaped simplex (full-pivot LU inverse + dense "
                  f"F*Binv per pivot, SimplexSolover.h:117-133,198-206,446) {r['iters']} pivots "
                  f"of m={pm} n={pn} in {t_piv:.1f} s",
        "simplex_pivots_per_s": round(r["iters"] / t_piv, 3),
        "simplex_equiv_GBs": round(16.0 * pm * (pn + 1) * r["iters"] / t_piv / 1e9, 4),
        "simplex_same_algorithm_tableau_pivots_per_s": round(rt["iters"] / t_tab, 1),
        "simplex_same_algorithm_tableau_solve_ms": round(1e3 * t_tab, 2),
        "simplex_note": "two CPU baselines: the reference-SHAPED path (O(m^3) re-inversion per pivot, what the "
                        "reference executes) and the tableau form the GPU executes (like-for-like algorithm)",
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

    # `traffic` of the tableau kernels: measured by two child processes before this one touches the GPU (N = 1 only)
    pmc_source = None
    if rank == 0 and world == 1 and not args.no_pivot and not args.no_live_pmc and (args.pivot_m, args.pivot_n) == (512, 1024):
        pmc_source = live_pmc_traffic()

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
    my_bounds = lpdist.balanced_shard_bounds(n, m, rank, world)   # (what lp_enum_shard_bounds gives the C path)

    # The exchange of the incumbent: the C ABI's own RCCL communicator (lp_comm_create_rccl +
    # lp_enum_solve_sharded: ONE ncclAllGather of a 48-byte record per step) — the entry point a C++
    # host uses.  torch.distributed only carries the 128-byte ncclUniqueId to the other ranks and
    # the barriers / max-over-ranks of the timing.  If the communicator cannot be created on every
    # rank, all ranks fall back together to the torch.distributed form of the same protocol
    # (simplexmethod_amd/dist.py) and the line says so.
    c_comm, exchange = None, "single participant (no collective)"
    if world > 1:
        ok = 0
        if backend != "gloo":
            try:
                idt = torch.zeros(128, dtype=torch.uint8, device=reduce_device)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(capi.Comm.unique_id()), dtype=torch.uint8))
                torch.distributed.broadcast(idt, src=0)
                c_comm = capi.Comm.rccl(ctx, rank, world, bytes(idt.cpu().numpy().tobytes()))
                ok = 1
            except Exception as e:   # noqa: BLE001 - any failure means "use the other exchange"
                print(f"[bench rank {rank}] C-ABI RCCL communicator unavailable: {e}", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int64, device=reduce_device)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if int(flag.item()) == 1:
            exchange = "C ABI: lp_enum_solve_sharded over lp_comm_create_rccl (one ncclAllGather of 48 B per step)"
        else:
            if c_comm is not None:
                c_comm.destroy()
                c_comm = None
            exchange = "torch.distributed (dist.py): one all_gather of the same record per step"

    def range_fn(lo, hi):
        rc, z, counts, st = ep.range(lo, hi, args.enum_algo)
        return z, counts

    def first_fn(lo, hi, zstar, tol):
        return ep.first_within(lo, hi, zstar, tol)

    use_c = world == 1 or c_comm is not None

    def step():
        if use_c:
            # the whole drop-in solve(): pass 1, exchange, tie rule AND the winning vertex x (every step)
            r = ep.solve_sharded(c_comm, n - m, want_vertex=True)
            return dict(feasible=r["status"] == 0, rank=r["rank"], counts=r["counts"], zstar=r["obj"], x=r["x"])
        return lpdist.enum_solve_sharded(comm, total, True, range_fn, first_fn, bounds=my_bounds)

    for _ in range(args.warmup):
        res = step()
    comm.barrier()
    torch.cuda.synchronize()
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
    # kernel time of pass 1 on this rank's shard alone (outside the timed region)
    k_ms = float(ep.range(my_bounds[0], my_bounds[1], args.enum_algo)[3].kernel_ms) if my_bounds[1] > my_bounds[0] else 0.0
    shard = my_bounds[1] - my_bounds[0]

    winner = ep.vertex(res["rank"], n - m) if res["feasible"] else None
    # BASELINE configs[4] on every rank (its LP-index shard; replicas, no exchange), while the ranks are
    # still in step: the legs below run on rank 0 only
    bl = None if args.no_batched else batched_leg(ctx, args, rank, world, reduce_device)
    prof_all = pmc_profile()
    try:
        fp64_per_subset = float(prof_all["enum_fp64"]["flops_per_subset"]) if (m, n) == (16, 32) else None
    except Exception:
        fp64_per_subset = None
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
                "exchange": exchange,
            },
            "enum": {
                "optimum": None if winner is None else winner["obj"],
                "timed_step": ("pass 1 + exchange + tie rule + the winning vertex x (lp_enum_solve_sharded with x_out: what the "
                               "drop-in EnumerationSolver::solve() returns)" if use_c else
                               "pass 1 + exchange + tie rule (torch.distributed form; the vertex is evaluated once, outside)"),
                "rank": res["rank"], "counts": res["counts"],
                "kernel_ms_pass1_rank0": round(k_ms, 4),
                "algorithmic_flops_per_subset": flops_per_subset,
                "algorithmic_equiv_TFLOPs_no_frac": round(value * flops_per_subset / 1e12, 3),
                # what the shared-prefix kernels EXECUTE per subset (not the 3,243 flop an independent solve
                # would need): fp64 instruction counters of one C(32,16) pass (scripts/pmc_enum.py under
                # rocprofv3 --pmc, scripts/pmc_to_json.py), reported only for the kernel sources they were taken on
                "executed_fp64_flops_per_subset": fp64_per_subset,
                "executed_TFLOPs": None if fp64_per_subset is None else round(value * fp64_per_subset / 1e12, 3),
                "executed_frac_of_fp64_vector_peak": None if fp64_per_subset is None else
                    round(value * fp64_per_subset / 1e12 / (FP64_VECTOR_PEAK_TF * world), 4),
                "valu_issue_busy": (prof_all.get("enum_valu_issue_busy") if (m, n) == (16, 32) else None),
                "mfma": "not used: per-subset row pivoting is data-dependent (DESIGN.md 4.6); no MFMA utilisation to report",
            },
        }
    if rank == 0 and not args.no_pivot:
        pivot, roofline, roofline_whole, roofline_rank1, rankj = pivot_leg(ctx, args)
        line["pivot"] = pivot
        roofline["traffic_source"] = (pmc_source if _LIVE_PMC else
                                      ("committed profile profiles/%s (taken on these kernel sources)" % PMC_PROFILE if prof_all else "none")
                                      + ("" if pmc_source is None else "; live passes " + pmc_source))
        line["roofline"] = roofline
        line["roofline_whole_pivot"] = roofline_whole
        line["roofline_rank1_update"] = roofline_rank1
        if rankj is not None:
            line["rankj_update"] = rankj
        if world == 1 and (args.pivot_m, args.pivot_n) == (512, 1024) and not args.no_large_shape:
            line["pivot_beyond_resident"] = large_shape_leg(ctx, args)
    if rank == 0 and world == 1 and not args.no_batched:
        line["enum"]["other_inputs"] = enum_inputs_leg(ctx, args)
        line["enum"]["wide_shapes"] = enum_wide_leg(ctx)
        line["enum"]["config2_c28_14"] = enum_config2_leg(ctx)
        line["enum"]["worst_case_subsets_per_s"] = min(r.get("first_call_subsets_per_s", r["subsets_per_s"])
                                                       for r in line["enum"]["other_inputs"])
    if rank == 0 and not args.no_batched:
        line["batched"] = bl
        line["two_phase"] = two_phase_leg(ctx, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_leg(args)
    ep.free()
    if c_comm is not None:
        c_comm.destroy()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
