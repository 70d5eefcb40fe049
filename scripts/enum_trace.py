"""Kernel-trace driver: the full C(32,16) range once, then shard 7 of 8 once (after a warm-up).
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/etrace -- python3 scripts/enum_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi, dist as lpdist

m, n = 16, 32
ctx = capi.Context(0)
A, b, c, _ = capi.gen_lp(0, m, n)
p = ctx.enum_problem(A, b, c, True)
p.range(0, p.total)
p.range(0, p.total)
part = int(sys.argv[1]) if len(sys.argv) > 1 else 7
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lo, hi = lpdist.balanced_shard_bounds(n, m, part, parts) if parts == 8 else lpdist.shard_bounds(p.total, part, parts)
p.range(lo, hi)
p.range(lo, hi)
