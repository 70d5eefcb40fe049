"""Writes tests/golden/*.json.

The reference cannot be built or run here (Eigen 3.4.0 is absent and fetched from the network by
/root/reference/CMakeLists.txt:12-17) and holds no recorded solver outputs, so these vectors are
RESTATEMENT-DERIVED: produced by oracle/lp_oracle.c (pinned against the reference's own fixtures
in tests/test_oracle.py) and cross-checked here against scipy.optimize.linprog (HiGHS) for the
optimum.  Inputs are regenerated from seeds by capi.gen_lp; only expected outputs are stored.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as o          # noqa: E402
from simplexmethod_amd import capi        # noqa: E402
from scipy.optimize import linprog        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

simplex = []
for seed, m, n in [(0, 4, 9), (1, 8, 16), (2, 16, 32), (3, 32, 64), (4, 64, 128), (5, 128, 256),
                   (6, 24, 100), (0, 512, 1024)]:
    A, b, c, basis = capi.gen_lp(seed, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14)
    assert r["status"] == 0
    ref = linprog(-c[:n - m], A_ub=A[:, :n - m], b_ub=b, bounds=(0, None), method="highs")
    assert ref.status == 0 and abs(r["obj"] + ref.fun) <= 1e-9 * abs(ref.fun), (seed, m, n)
    # the reference-SHAPED restatement (Binv recomputed by full-pivot LU every iteration,
    # /root/reference/src/SimplexSolover.h:429-447): ~20 s at 512 x 1024, run once here
    rr = o.simplex_reference(A, b, c, basis, True, n - m, trace_cap=1 << 14)
    assert rr["status"] == 0 and rr["trace"] == r["trace"] and np.array_equal(rr["basis"], r["basis"])
    assert np.allclose(rr["x"], r["x"], rtol=1e-10, atol=1e-12)
    assert abs(rr["obj"] - r["obj"]) <= 1e-10 * abs(r["obj"])
    case = dict(seed=seed, m=m, n=n, iters=r["iters"], basis=r["basis"].tolist(),
                obj=r["obj"], trace_head=r["trace"][:16],
                x_nonzero={str(j): v for j, v in enumerate(r["x"].tolist()) if v != 0.0})
    # outputs of the reference-shaped form: what the GPU is held to within the north star's
    # 1e-10 (trace and basis exactly)
    case["reference_shaped"] = dict(
        iters=rr["iters"], basis=rr["basis"].tolist(), obj=rr["obj"],
        trace=[list(t) for t in rr["trace"]],
        x_nonzero={str(j): v for j, v in enumerate(rr["x"].tolist()) if v != 0.0})
    simplex.append(case)
json.dump(simplex, open(os.path.join(HERE, "simplex_cases.json"), "w"), indent=1)

enum = []
for seed, m, n in [(0, 2, 5), (1, 4, 9), (2, 6, 13), (3, 8, 16), (4, 10, 20)]:
    A, b, c, basis = capi.gen_lp(seed, m, n)
    e = o.enum_solve(A, b, c, True, n - m)
    s = o.simplex_tableau(A, b, c, basis, True, n - m)
    assert abs(e["obj"] - s["obj"]) <= 1e-10 * abs(s["obj"])
    enum.append(dict(seed=seed, m=m, n=n, rank=e["rank"], basis=e["basis"].tolist(), obj=e["obj"],
                     counts=e["counts"], x=e["x"].tolist()))
# the input_symmetric.txt table of SURVEY.md §4, rank by rank
A = np.array([[1, 2, 3, 1, 0], [4, 5, 6, 0, 1.0]]); b = [10, 20.0]; c = [7, 8, 3, 0, 0.0]
table = []
for k in range(10):
    st, xB, z = o.enum_subset(A, b, c, o.unrank(5, 2, k))
    table.append(dict(rank=k, subset=o.unrank(5, 2, k).tolist(), verdict=st, xB=xB.tolist(), z=z))
json.dump(dict(random=enum, input_symmetric=table), open(os.path.join(HERE, "enum_cases.json"), "w"), indent=1)

# two-phase simplex: min problems without a starting basis, optimum cross-checked
# against scipy's HiGHS; inputs are regenerated from seeds by tests/lpcases.py
from tests import lpcases                 # noqa: E402
two = []
for kind, seed, args in [("min", 0, (5, 4, 0, 0, 0)), ("min", 1, (6, 5, 1, 2, 0)), ("min", 3, (16, 12, 3, 4, 2)),
                         ("min", 5, (64, 64, 4, 8, 3)), ("deg", 7, ()), ("deg", 34, ()), ("deg", 40, ())]:
    if kind == "min":
        A, b, c, no = lpcases.min_lp(seed, args[0], args[1], equalities=args[2], negative_rows=args[3],
                                     zero_rhs=args[4])
    else:
        A, b, c, no = lpcases.degenerate_eq_lp(seed)
    r = o.two_phase(A, b, c, maximize=False, n_orig=no)
    assert r["status"] == 0
    ref = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    assert ref.status == 0 and abs(r["obj"] - ref.fun) <= 1e-9 * max(1.0, abs(ref.fun)), (kind, seed)
    two.append(dict(kind=kind, seed=seed, args=list(args), iters=r["iters"], basis=r["basis"].tolist(),
                    obj=r["obj"], x=r["x"].tolist()))
json.dump(two, open(os.path.join(HERE, "two_phase_cases.json"), "w"), indent=1)
print("wrote", len(simplex), "simplex cases,", len(enum), "enumeration cases and", len(two), "two-phase cases")
