"""Workload for the HBM-traffic PMC passes at m=512 x n=1024 (BASELINE configs[1]): 5 chip-resident
solves, 50 launches of the rank-1 update, 50 of the rank-J update.  Run under
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 scripts/pmc_traffic.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 scripts/pmc_traffic.py
(separate passes, no trace domains: gpurun refuses --pmc combined with them), then
  python3 scripts/pmc_to_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi

ctx = capi.Context(0)
m, n = 512, 1024
A, b, c, basis = capi.gen_lp(0, m, n)
p = ctx.simplex_problem(A, b, c, basis, True, n - m)
for _ in range(5):
    p.reset()
    rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
    print("resident: rc", rc, "pivots", st.pivots, "launches", st.launches, "kernel ms", st.update_ms)
p.reset()
print("rank-1 ms/launch", p.bench_update(0, 0, 50))
p.reset()
print("rank-J (ms/launch, J)", p.bench_update_rankj(50))
p.free()
# the one-launch-per-pivot path on a tableau that streams (2048 x 4096, 67 MB): k_simplex_overlap, one launch per pivot
m, n = 2048, 4096
A, b, c, basis = capi.gen_lp(0, m, n)
p = ctx.simplex_problem(A, b, c, basis, True, n - m)
rc, st = p.run(algo=capi.SIMPLEX_OVERLAP, max_iter=40)
print("overlap 2048x4096: rc", rc, "pivots", st.pivots, "launches", st.launches, "solve ms", st.solve_ms)
p.free()
