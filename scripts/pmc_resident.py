"""Workload for PMC passes on the chip-resident simplex kernel: 5 solves of 512 x 1024 (345 pivots each).
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d <dir> -- python3 scripts/pmc_resident.py
then python3 scripts/pmc_summary.py <dir> k_simplex_resident"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
ctx = capi.Context(0)
A, b, c, basis = capi.gen_lp(0, 512, 1024)
p = ctx.simplex_problem(A, b, c, basis, True, 512)
for _ in range(5):
    p.reset()
    rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
    print("resident: rc", rc, "pivots", st.pivots, "kernel ms", st.update_ms)
p.free()
