"""One C(32,16) enumeration pass under rocprofv3 --pmc (library chosen by LP_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
ctx = capi.Context(0)
A, b, c, _ = capi.gen_lp(0, 16, 32)
p = ctx.enum_problem(A, b, c, True)
p.range(0, p.total)
p.range(0, p.total)
