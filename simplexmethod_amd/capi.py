"""ctypes binding of include/simplexmethod_amd.h (libsimplexmethod_hip.so).

There is no CPU fallback here or in the library: `load()` raises if the shared
object is missing, and `Context()` raises if no gfx950 device is usable.
Matrices are passed as (m, n) numpy arrays and converted to the ABI's column-major
layout (Eigen's default, /root/reference/src/ProblemTypes/Canonical.cpp:10).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

OPTIMAL, UNBOUNDED, ITER_LIMIT, SINGULAR, INFEASIBLE, BAD_ARG = range(6)
SUBSET_FEASIBLE, SUBSET_INFEASIBLE, SUBSET_SINGULAR = range(3)
SIMPLEX_AUTO, SIMPLEX_LAUNCH, SIMPLEX_LOOKAHEAD, SIMPLEX_RESIDENT, SIMPLEX_OVERLAP = 0, 1, 2, 3, 4
ENUM_AUTO, ENUM_DIRECT, ENUM_PREFIX = 0, 1, 2
U64_MAX = (1 << 64) - 1
EPS = 1e-9        # Solver::EPS, SimplexSolover.h:13
MAX_ITER = 10000  # SimplexSolover.h:426

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_u64p = C.POINTER(C.c_uint64)
_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


class SimplexStats(C.Structure):
    _fields_ = [("status", C.c_int), ("pivots", C.c_int), ("launches", C.c_int),
                ("solve_ms", C.c_float), ("update_ms", C.c_float), ("update_launches", C.c_int),
                ("bytes_per_pivot", C.c_double), ("algo_used", C.c_int), ("fell_back", C.c_int)]


class EnumStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_float), ("subsets", C.c_uint64), ("launches", C.c_int)]


# name -> (restype, argtypes); also the list tests check against the header's declarations
SIGNATURES = {
    "lp_abi_version": (C.c_int, []),
    "lp_device_count": (C.c_int, []),
    "lp_context_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "lp_context_destroy": (None, [_vp]),
    "lp_last_error": (C.c_char_p, [_vp]),
    "lp_status_string": (C.c_char_p, [C.c_int]),
    "lp_context_sync": (C.c_int, [_vp]),
    "lp_simplex_solve": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _dp, _dp, _ip, C.c_int, C.c_int,
                                   C.c_double, C.c_int, _dp, _ip, _dp, _ip]),
    "lp_simplex_upload": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _dp, _dp, _ip, C.c_int, C.c_int,
                                    C.POINTER(_vp)]),
    "lp_simplex_reset": (C.c_int, [_vp]),
    "lp_simplex_run": (C.c_int, [_vp, C.c_double, C.c_int, C.c_int, C.POINTER(SimplexStats)]),
    "lp_simplex_profile": (C.c_int, [_vp, C.c_int]),
    "lp_simplex_download": (C.c_int, [_vp, _dp, _ip, _dp, _ip, _ip, C.c_int, _dp]),
    "lp_simplex_free": (None, [_vp]),
    "lp_simplex_two_phase": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int,
                                       C.c_double, C.c_int, _dp, _ip, _dp, _ip]),
    "lp_simplex_row": (C.c_int, [_vp, C.c_int, _dp]),
    "lp_simplex_force_pivot": (C.c_int, [_vp, C.c_int, C.c_int]),
    "lp_bench_rank1_update": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _fp]),
    "lp_bench_rankj_update": (C.c_int, [_vp, C.c_int, _fp, _ip]),
    "lp_debug_simplex_stamps": (C.c_int, [_vp, C.c_int, _u64p]),
    "lp_simplex_solve_batched": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, _ip,
                                           C.c_int, C.c_int, C.c_double, C.c_int, _dp, _ip, _dp,
                                           _ip, _ip]),
    "lp_batched_upload": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, _ip, C.c_int,
                                    C.c_int, C.POINTER(_vp)]),
    "lp_batched_run": (C.c_int, [_vp, C.c_double, C.c_int, _fp]),
    "lp_batched_download": (C.c_int, [_vp, _dp, _ip, _dp, _ip, _ip]),
    "lp_batched_free": (None, [_vp]),
    "lp_batched_shard_bounds": (C.c_int, [C.c_int, C.c_int, C.c_int, _ip, _ip]),
    "lp_binom": (C.c_uint64, [C.c_int, C.c_int]),
    "lp_enum_shard_bounds": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _u64p, _u64p]),
    "lp_enum_solve": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, _ip,
                                _u64p, _dp, _u64p]),
    "lp_enum_upload": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.POINTER(_vp)]),
    "lp_enum_range": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_int, _dp, _u64p,
                                C.POINTER(EnumStats)]),
    "lp_enum_first_within": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_double, C.c_double, _u64p]),
    "lp_enum_vertex": (C.c_int, [_vp, C.c_uint64, C.c_int, _dp, _ip, _dp, _ip]),
    "lp_enum_free": (None, [_vp]),
    "lp_enum_exact_division": (C.c_int, [_vp]),
    "lp_debug_reciprocal": (C.c_int, [_vp, _dp, C.c_int, _dp, _dp]),
    "lp_debug_division": (C.c_int, [_vp, _dp, _dp, C.c_int, _dp, _dp]),
    "lp_comm_unique_id": (C.c_int, [_vp]),
    "lp_comm_create_rccl": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "lp_comm_create_local": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "lp_comm_rank": (C.c_int, [_vp]),
    "lp_comm_world": (C.c_int, [_vp]),
    "lp_comm_destroy": (None, [_vp]),
    "lp_enum_solve_sharded": (C.c_int, [_vp, _vp, C.c_int, _dp, _ip, _u64p, _dp, _u64p]),
    "lp_enum_shard_abstain": (C.c_int, [_vp, C.c_int]),
}

# phases of the chip-resident kernel's diagnostic instantiation (lp_debug_simplex_stamps after a
# chip-resident run: cycle sums per workgroup, in this order): slots 0-5 the communication wave
# (polls, decides, finishes the ratio test, publishes), slots 6-15 row wave 0
RESIDENT_STAMP_NAMES = [
    "comm: poll_records (the hop)", "comm: decide_and_decision_block",
    "comm: waits while the rows read the decision and stage the pivot row",
    "comm: waits for the rows' pricing + ratio slices, then ratio stage 2 + record",
    "comm: loop", "comm: commit",
    "rows: wait for the decision (publish -> hop -> decide)",
    "rows: read_decision_request_column_pivot_row_to_lds_two_quotients", "rows: pivot_row_barrier",
    "rows: reduced_costs_and_next_pricing", "rows: column_wait_eta_entry",
    "rows: candidate_ratio_slice", "rows: ratio_barrier",
    "rows: rank1_update_and_eta_column_publication", "rows: loop", "rows: commit"]

_lib = None


def lib_path():
    # LP_LIB_PATH: A/B experiments with an alternative build of the same library
    return os.environ.get("LP_LIB_PATH") or _build.HIP_LIB


def load():
    """Loads the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        if path == _build.HIP_LIB:   # a library older than its sources measures and tests something else
            srcs = [os.path.join(_build.CSRC, f) for f in os.listdir(_build.CSRC) if f.endswith((".hip", ".hpp"))]
            stale = [os.path.basename(f) for f in srcs if os.path.getmtime(f) > os.path.getmtime(path) + 1.0]
            if stale:
                import sys
                print(f"simplexmethod_amd: {os.path.basename(path)} is OLDER than {', '.join(sorted(stale))} - rebuild "
                      "(python -c 'import __graft_entry__ as g; g.build()')", file=sys.stderr)
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError = ABI symbol missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class LPError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"status {code}: {msg}")
        self.code = code


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def colmajor(A):
    A = np.asarray(A, dtype=np.float64)
    return np.ascontiguousarray(A.T).reshape(-1)


# ---- synthetic LPs; bit-identical to oracle/lp_oracle.c:orc_gen_lp ----
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _u01(key, count, start=0):
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = _mix64(np.uint64(key) + k * np.uint64(0x9E3779B97F4A7C15))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def gen_lp(seed, m, n):
    """Dense random canonical LP [A_orig | I]: A_orig ~ U(0,1), b ~ U(1,2)*(n-m)/2, c ~ U(0,1)
    on the original columns, slack basis, maximise.  Returns (A (m,n), b, c, basis)."""
    no = n - m
    with np.errstate(over="ignore"):
        key = int(_mix64(np.uint64(seed) + np.uint64(0x5851F42D4C957F2D)))
    u = _u01(key, no * m + m + no)
    A = np.zeros((m, n))
    A[:, :no] = u[:no * m].reshape(no, m).T
    A[:, no:] = np.eye(m)
    b = (1.0 + u[no * m:no * m + m]) * (no * 0.5)
    c = np.zeros(n)
    c[:no] = u[no * m + m:]
    basis = np.arange(no, n, dtype=np.int32)
    return A, b, c, basis


class Context:
    """One HIP device + stream (lp_context)."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = _vp()
        rc = self.lib.lp_context_create(device, stream, C.byref(h))
        if rc != 0:
            raise LPError(rc, "lp_context_create failed: " +
                          (self.lib.lp_last_error(None) or b"").decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.lp_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def error(self):
        return (self.lib.lp_last_error(self.h) or b"").decode()

    def check(self, rc):
        if rc < 0 or rc == BAD_ARG:
            raise LPError(rc, self.error() or self.lib.lp_status_string(rc).decode())
        return rc

    # ---- simplex -------------------------------------------------------------------------
    def debug_reciprocal(self, x):
        """(fast, plain): the leaf kernels' fast reciprocal of x and 1.0 / x, both from the device."""
        x = _f64(x).reshape(-1)
        fast, plain = np.empty_like(x), np.empty_like(x)
        self.check(self.lib.lp_debug_reciprocal(self.h, _d(x), len(x), _d(fast), _d(plain)))
        return fast, plain

    def debug_division(self, num, den):
        """(fast, plain): the chip-resident simplex's quotient sequence and num / den, both from the device."""
        num, den = _f64(num).reshape(-1), _f64(den).reshape(-1)
        fast, plain = np.empty_like(num), np.empty_like(num)
        self.check(self.lib.lp_debug_division(self.h, _d(num), _d(den), len(num), _d(fast), _d(plain)))
        return fast, plain

    def simplex_solve(self, A, b, c, basis, maximize=True, n_orig=None, eps=EPS,
                      max_iter=MAX_ITER):
        A = np.asarray(A, dtype=np.float64)
        m, n = A.shape
        n_orig = n if n_orig is None else n_orig
        Af, b, c = colmajor(A), _f64(b), _f64(c)
        basis = np.ascontiguousarray(basis, dtype=np.int32)
        x = np.zeros(max(n_orig, 1))
        bo = np.zeros(m, dtype=np.int32)
        obj = C.c_double(float("nan"))
        it = C.c_int(0)
        rc = self.check(self.lib.lp_simplex_solve(self.h, _d(Af), m, n, _d(b), _d(c), _i(basis),
                                                  int(maximize), n_orig, eps, max_iter, _d(x),
                                                  _i(bo), C.byref(obj), C.byref(it)))
        return dict(status=rc, x=x[:n_orig], basis=bo, obj=obj.value, iters=it.value)

    def two_phase(self, A, b, c, maximize=False, n_orig=None, eps=EPS, max_iter=MAX_ITER):
        """lp_simplex_two_phase: no starting basis needed."""
        A = np.asarray(A, dtype=np.float64)
        m, n = A.shape
        n_orig = n if n_orig is None else n_orig
        Af, b, c = colmajor(A), _f64(b), _f64(c)
        x = np.zeros(n_orig)
        bo = np.full(m, -1, dtype=np.int32)
        obj = C.c_double(float("nan"))
        it = np.zeros(3, dtype=np.int32)
        rc = self.check(self.lib.lp_simplex_two_phase(self.h, _d(Af), m, n, _d(b), _d(c),
                                                      int(maximize), n_orig, eps, max_iter, _d(x),
                                                      _i(bo), C.byref(obj), _i(it)))
        return dict(status=rc, x=x, basis=bo, obj=obj.value, iters=it.tolist())

    def simplex_problem(self, A, b, c, basis, maximize=True, n_orig=None):
        return SimplexProblem(self, A, b, c, basis, maximize, n_orig)

    def simplex_solve_batched(self, A, b, c, basis, maximize=True, n_orig=None, eps=EPS,
                              max_iter=MAX_ITER):
        """A: (batch, m, n); b: (batch, m); c: (batch, n); basis: (batch, m)."""
        A = np.asarray(A, dtype=np.float64)
        batch, m, n = A.shape
        n_orig = n if n_orig is None else n_orig
        Af = np.ascontiguousarray(np.transpose(A, (0, 2, 1))).reshape(-1)
        b, c = _f64(b).reshape(-1), _f64(c).reshape(-1)
        basis = np.ascontiguousarray(basis, dtype=np.int32).reshape(-1)
        x = np.zeros((batch, n_orig))
        bo = np.zeros((batch, m), dtype=np.int32)
        obj = np.full(batch, np.nan)
        it = np.zeros(batch, dtype=np.int32)
        st = np.zeros(batch, dtype=np.int32)
        self.check(self.lib.lp_simplex_solve_batched(self.h, batch, _d(Af), m, n, _d(b), _d(c),
                                                     _i(basis), int(maximize), n_orig, eps,
                                                     max_iter, _d(x), _i(bo), _d(obj), _i(it),
                                                     _i(st)))
        return dict(status=st, x=x, basis=bo, obj=obj, iters=it)

    def batched_problem(self, A, b, c, basis, maximize=True, n_orig=None):
        return BatchedProblem(self, A, b, c, basis, maximize, n_orig)

    # ---- enumeration ---------------------------------------------------------------------
    def enum_solve(self, A, b, c, maximize=True, n_orig=None):
        A = np.asarray(A, dtype=np.float64)
        m, n = A.shape
        n_orig = n if n_orig is None else n_orig
        Af, b, c = colmajor(A), _f64(b), _f64(c)
        x = np.zeros(n_orig)
        bo = np.zeros(m, dtype=np.int32)
        rank = C.c_uint64(0)
        obj = C.c_double(float("nan"))
        counts = (C.c_uint64 * 3)()
        rc = self.check(self.lib.lp_enum_solve(self.h, _d(Af), m, n, _d(b), _d(c), int(maximize),
                                               n_orig, _d(x), _i(bo), C.byref(rank), C.byref(obj),
                                               counts))
        return dict(status=rc, x=x, basis=bo, rank=int(rank.value), obj=obj.value,
                    counts=[int(v) for v in counts])

    def enum_problem(self, A, b, c, maximize=True):
        return EnumProblem(self, A, b, c, maximize)


class SimplexProblem:
    """Device-resident tableau (lp_simplex_problem)."""

    def __init__(self, ctx, A, b, c, basis, maximize=True, n_orig=None):
        A = np.asarray(A, dtype=np.float64)
        self.ctx, self.m, self.n = ctx, A.shape[0], A.shape[1]
        self.n_orig = self.n if n_orig is None else n_orig
        Af, b, c = colmajor(A), _f64(b), _f64(c)
        basis = np.ascontiguousarray(basis, dtype=np.int32)
        h = _vp()
        ctx.check(ctx.lib.lp_simplex_upload(ctx.h, _d(Af), self.m, self.n, _d(b), _d(c), _i(basis),
                                            int(maximize), self.n_orig, C.byref(h)))
        self.h = h

    def reset(self):
        self.ctx.check(self.ctx.lib.lp_simplex_reset(self.h))

    def profile(self, on=True):
        self.ctx.check(self.ctx.lib.lp_simplex_profile(self.h, int(on)))

    def run(self, eps=EPS, max_iter=MAX_ITER, algo=SIMPLEX_AUTO):
        st = SimplexStats()
        rc = self.ctx.check(self.ctx.lib.lp_simplex_run(self.h, eps, max_iter, algo, C.byref(st)))
        return rc, st

    def download(self, trace_cap=0, want_tableau=False):
        x = np.zeros(self.n_orig)
        bo = np.zeros(self.m, dtype=np.int32)
        obj = C.c_double(float("nan"))
        te = np.full(max(trace_cap, 1), -1, dtype=np.int32)
        tl = np.full(max(trace_cap, 1), -1, dtype=np.int32)
        tab = np.zeros((self.m + 1, self.n + 1)) if want_tableau else None
        self.ctx.check(self.ctx.lib.lp_simplex_download(self.h, _d(x), _i(bo), C.byref(obj), _i(te),
                                                        _i(tl), trace_cap, _d(tab)))
        return dict(x=x, basis=bo, obj=obj.value, trace_enter=te[:trace_cap],
                    trace_leave=tl[:trace_cap], tableau=tab)

    def row(self, row):
        """Row `row` of the current tableau (m = reduced costs): n + 1 doubles."""
        out = np.zeros(self.n + 1)
        self.ctx.check(self.ctx.lib.lp_simplex_row(self.h, int(row), _d(out)))
        return out

    def force_pivot(self, row, col):
        return self.ctx.check(self.ctx.lib.lp_simplex_force_pivot(self.h, int(row), int(col)))

    def bench_update(self, row, col, iters):
        ms = C.c_float(0.0)
        self.ctx.check(self.ctx.lib.lp_bench_rank1_update(self.h, row, col, iters, C.byref(ms)))
        return ms.value

    def bench_update_rankj(self, iters):
        ms = C.c_float(0.0)
        piv = C.c_int(0)
        self.ctx.check(self.ctx.lib.lp_bench_rankj_update(self.h, iters, C.byref(ms), C.byref(piv)))
        return ms.value, piv.value

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.lp_simplex_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BatchedProblem:
    def __init__(self, ctx, A, b, c, basis, maximize=True, n_orig=None):
        A = np.asarray(A, dtype=np.float64)
        self.ctx = ctx
        self.batch, self.m, self.n = A.shape
        self.n_orig = self.n if n_orig is None else n_orig
        Af = np.ascontiguousarray(np.transpose(A, (0, 2, 1))).reshape(-1)
        b, c = _f64(b).reshape(-1), _f64(c).reshape(-1)
        basis = np.ascontiguousarray(basis, dtype=np.int32).reshape(-1)
        h = _vp()
        ctx.check(ctx.lib.lp_batched_upload(ctx.h, self.batch, _d(Af), self.m, self.n, _d(b), _d(c),
                                            _i(basis), int(maximize), self.n_orig, C.byref(h)))
        self.h = h

    def run(self, eps=EPS, max_iter=MAX_ITER):
        ms = C.c_float(0.0)
        self.ctx.check(self.ctx.lib.lp_batched_run(self.h, eps, max_iter, C.byref(ms)))
        return ms.value

    def download(self):
        x = np.zeros((self.batch, self.n_orig))
        bo = np.zeros((self.batch, self.m), dtype=np.int32)
        obj = np.full(self.batch, np.nan)
        it = np.zeros(self.batch, dtype=np.int32)
        st = np.zeros(self.batch, dtype=np.int32)
        self.ctx.check(self.ctx.lib.lp_batched_download(self.h, _d(x), _i(bo), _d(obj), _i(it),
                                                        _i(st)))
        return dict(status=st, x=x, basis=bo, obj=obj, iters=it)

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.lp_batched_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class EnumProblem:
    """Device-resident enumeration problem (lp_enum_problem)."""

    def __init__(self, ctx, A, b, c, maximize=True):
        A = np.asarray(A, dtype=np.float64)
        self.ctx, self.m, self.n = ctx, A.shape[0], A.shape[1]
        self.maximize = bool(maximize)
        Af, b, c = colmajor(A), _f64(b), _f64(c)
        h = _vp()
        ctx.check(ctx.lib.lp_enum_upload(ctx.h, _d(Af), self.m, self.n, _d(b), _d(c),
                                         int(maximize), C.byref(h)))
        self.h = h
        self.total = int(ctx.lib.lp_binom(self.n, self.m))

    def range(self, begin, end, algo=ENUM_AUTO):
        z = C.c_double(0.0)
        counts = (C.c_uint64 * 3)()
        st = EnumStats()
        rc = self.ctx.check(self.ctx.lib.lp_enum_range(self.h, begin, end, algo, C.byref(z), counts,
                                                       C.byref(st)))
        return rc, z.value, [int(v) for v in counts], st

    def first_within(self, begin, end, zstar, tol=1e-9):
        r = C.c_uint64(0)
        self.ctx.check(self.ctx.lib.lp_enum_first_within(self.h, begin, end, zstar, tol,
                                                         C.byref(r)))
        return int(r.value)

    def solve_sharded(self, comm=None, n_orig=None, want_vertex=True):
        """lp_enum_solve_sharded: this participant's shard of the rank space + the one exchange.
        comm: a Comm (or None = single participant).  Blocks until every participant has called.
        want_vertex=False: rank, objective and counts only (no vertex kernel)."""
        n_orig = self.n if n_orig is None else n_orig
        x = np.zeros(n_orig)
        bo = np.zeros(self.m, dtype=np.int32)
        rank = C.c_uint64(0)
        obj = C.c_double(float("nan"))
        counts = (C.c_uint64 * 3)()
        rc = self.ctx.check(self.ctx.lib.lp_enum_solve_sharded(
            comm.h if comm is not None else None, self.h, n_orig, _d(x) if want_vertex else None,
            _i(bo) if want_vertex else None, C.byref(rank), C.byref(obj), counts))
        return dict(status=rc, x=x, basis=bo, rank=int(rank.value), obj=obj.value,
                    counts=[int(v) for v in counts])

    @property
    def exact_division(self):
        """True once the leaf kernels divide plainly (include/simplexmethod_amd.h: lp_enum_exact_division)."""
        return bool(self.ctx.lib.lp_enum_exact_division(self.h))

    def vertex(self, rank, n_orig=None):
        n_orig = self.n if n_orig is None else n_orig
        x = np.zeros(n_orig)
        bo = np.zeros(self.m, dtype=np.int32)
        obj = C.c_double(float("nan"))
        verdict = C.c_int(-1)
        self.ctx.check(self.ctx.lib.lp_enum_vertex(self.h, rank, n_orig, _d(x), _i(bo),
                                                   C.byref(obj), C.byref(verdict)))
        return dict(x=x, basis=bo, obj=obj.value, verdict=verdict.value)

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.lp_enum_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Comm:
    """lp_comm: the exchange of the sharded enumeration (one per participant)."""

    def __init__(self, handle, lib):
        self.h, self.lib = handle, lib

    @staticmethod
    def unique_id():
        """128 bytes for Comm.rccl (rank 0 creates them and hands them to the other participants)."""
        buf = C.create_string_buffer(128)
        if load().lp_comm_unique_id(buf) != OPTIMAL:
            raise LPError(BAD_ARG, "RCCL is not available (lp_comm_unique_id)")
        return buf.raw

    @staticmethod
    def rccl(ctx, rank, world, unique_id):
        """Collective: every participant calls it with the same id (ncclCommInitRank)."""
        h = _vp()
        ctx.check(ctx.lib.lp_comm_create_rccl(ctx.h, rank, world, C.c_char_p(unique_id), C.byref(h)))
        return Comm(h, ctx.lib)

    @staticmethod
    def local(world):
        """`world` communicators for host threads of this process (exchange through host memory)."""
        lib = load()
        arr = (_vp * world)()
        if lib.lp_comm_create_local(world, arr) != OPTIMAL:
            raise LPError(BAD_ARG, "lp_comm_create_local failed")
        return [Comm(_vp(arr[r]), lib) for r in range(world)]

    def rank(self):
        return self.lib.lp_comm_rank(self.h)

    def world(self):
        return self.lib.lp_comm_world(self.h)

    def destroy(self):
        if self.h:
            self.lib.lp_comm_destroy(self.h)
            self.h = None
