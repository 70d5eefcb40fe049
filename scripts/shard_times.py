"""How evenly does the C(32,16) rank space split?  Times pass 1 + pass 2 of every shard for
N = 1, 2, 4, 8 on ONE GPU (what each rank of an N-GPU run would do on its own GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np
import bench   # kernel_source_hash
from simplexmethod_amd import capi, dist as lpdist

m, n = 16, 32
ctx = capi.Context(0)
A, b, c, _ = capi.gen_lp(0, m, n)
p = ctx.enum_problem(A, b, c, True)
total = p.total
p.range(0, total)          # warm-up (allocations)
report = {"what": "C(32,16) seed 0, every shard of an N-way cost-balanced cut timed ALONE on one MI355X "
                  "(pass 1 + tie rule, best of 3; what each rank of an N-GPU run does on its own GPU, before "
                  "the one all-gather of 48 B per rank and without process skew) - a projection, not a measured "
                  "multi-GPU run", "kernel_source_hash": bench.kernel_source_hash(), "cuts": {}}
for parts in (1, 2, 4, 8):
    times = []
    rows = []
    for r in range(parts):
        lo, hi = lpdist.balanced_shard_bounds(n, m, r, parts)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            rc, z, counts, st = p.range(lo, hi)
            t1 = time.perf_counter()
            k = p.first_within(lo, hi, z) if rc == 0 else None
            t2 = time.perf_counter()
            if t2 - t0 < best:
                best = t2 - t0
                detail = (st.kernel_ms, (t1 - t0) * 1e3, (t2 - t1) * 1e3)
        times.append(best * 1e3)
        rows.append({"shard": r, "subsets": hi - lo, "wall_ms": round(best * 1e3, 3), "pass1_kernel_ms": round(detail[0], 3),
                     "pass1_wall_ms": round(detail[1], 3), "tie_rule_wall_ms": round(detail[2], 3)})
        if parts == 8:
            print("   shard", r, "pass1 kernels %.2f ms, pass1 wall %.2f ms, pass2 wall %.2f ms" % detail)
    print(f"N={parts}: shard ms {[round(t, 2) for t in times]}  max {max(times):.2f}  ideal {times and sum(times)/parts:.2f}"
          f"  speedup vs N=1 by max: {single / max(times):.2f}x" if parts > 1 else f"N=1: {times[0]:.2f} ms")
    if parts == 1:
        single = times[0]
    report["cuts"][str(parts)] = {"shards": rows, "max_wall_ms": round(max(times), 3),
                                  "projected_speedup_vs_1_by_slowest_shard": round(single / max(times), 3)}
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "shard_times.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(report, open(out, "w"), indent=1)
print("wrote", out)
