"""Workload for the PMC passes: 50 launches of the rank-1 update and 50 of the rank-J update on the
m=512 x n=1024 tableau (BASELINE configs[1]).  Run under
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python scripts/pmc_update.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir> -- python scripts/pmc_update.py
(separate passes, no trace domains: gpurun refuses --pmc combined with them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi

ctx = capi.Context(0)
m, n = 512, 1024
A, b, c, basis = capi.gen_lp(0, m, n)
p = ctx.simplex_problem(A, b, c, basis, True, n - m)
print("rank-1 ms/launch", p.bench_update(0, 0, 50))
p.reset()
print("rank-J (ms/launch, J)", p.bench_update_rankj(50))
p.free()
