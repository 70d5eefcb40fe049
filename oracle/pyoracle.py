"""ctypes binding of oracle/_build/liblp_oracle.so — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never from simplexmethod_amd (the product path).  See lp_oracle.h for what each
function restates (reference file:line) and its parity-pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblp_oracle.so")

OPTIMAL, UNBOUNDED, ITER_LIMIT, SINGULAR, INFEASIBLE, BAD_ARG = range(6)
SUBSET_FEASIBLE, SUBSET_INFEASIBLE, SUBSET_SINGULAR = range(3)
U64_MAX = (1 << 64) - 1


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("lp_oracle.c", "lp_oracle.h", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_simplex_reference.restype = C.c_int
        L.orc_simplex_reference.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, _ip, C.c_int, C.c_int,
                                            C.c_double, C.c_int, C.c_int, _dp, _ip, _dp, _ip, _ip,
                                            _ip, C.c_int]
        L.orc_simplex_tableau.restype = C.c_int
        L.orc_simplex_tableau.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, _ip, C.c_int, C.c_int,
                                          C.c_double, C.c_int, _dp, _ip, _dp, _ip, _ip, _ip, C.c_int,
                                          _dp]
        L.orc_two_phase.restype = C.c_int
        L.orc_two_phase.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double,
                                    C.c_int, _dp, _ip, _dp, _ip]
        L.orc_chain_select.restype = C.c_int
        L.orc_chain_select.argtypes = [_dp, C.c_char_p, C.c_int, C.c_int, C.c_double, _dp]
        L.orc_basic_solution.restype = C.c_int
        L.orc_basic_solution.argtypes = [_dp, C.c_int, C.c_int, _dp, _ip, _dp]
        L.orc_is_feasible_basis.restype = C.c_int
        L.orc_is_feasible_basis.argtypes = [_dp, C.c_int, C.c_int, _dp, _ip]
        L.orc_evaluate.restype = C.c_double
        L.orc_evaluate.argtypes = [_dp, _dp, C.c_int]
        L.orc_binom.restype = C.c_uint64
        L.orc_binom.argtypes = [C.c_int, C.c_int]
        L.orc_unrank.restype = None
        L.orc_unrank.argtypes = [C.c_int, C.c_int, C.c_uint64, _ip]
        L.orc_rank.restype = C.c_uint64
        L.orc_rank.argtypes = [C.c_int, C.c_int, _ip]
        L.orc_next_subset.restype = C.c_int
        L.orc_next_subset.argtypes = [C.c_int, C.c_int, _ip]
        L.orc_enum_subset.restype = C.c_int
        L.orc_enum_subset.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, _ip, _dp, _dp]
        L.orc_enum_range.restype = C.c_int
        L.orc_enum_range.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_uint64,
                                     C.c_uint64, _dp, _u64p]
        L.orc_enum_first_within.restype = C.c_uint64
        L.orc_enum_first_within.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_uint64,
                                            C.c_uint64, C.c_double, C.c_double]
        L.orc_enum_solve.restype = C.c_int
        L.orc_enum_solve.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, _ip,
                                     _u64p, _dp, _u64p]
        L.orc_gen_lp.restype = None
        L.orc_gen_lp.argtypes = [C.c_uint64, C.c_int, C.c_int, _dp, _dp, _dp, _ip]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _colmajor(A):
    """(m, n) array -> flat column-major float64 buffer."""
    A = np.asarray(A, dtype=np.float64)
    return np.ascontiguousarray(A.T).reshape(-1)


def gen_lp(seed, m, n):
    """Synthetic canonical LP [A_orig | I]; returns (A (m,n), b, c, basis)."""
    A = np.empty(m * n)
    b = np.empty(m)
    c = np.empty(n)
    basis = np.empty(m, dtype=np.int32)
    lib().orc_gen_lp(seed, m, n, _d(A), _d(b), _d(c), _i(basis))
    return A.reshape(n, m).T.copy(), b, c, basis


def _simplex(fn_name, A, b, c, basis, maximize, n_orig, eps, max_iter, trace_cap, extra):
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    Af, b, c = _colmajor(A), _f64(b), _f64(c)
    basis = np.ascontiguousarray(basis, dtype=np.int32)
    x = np.zeros(max(n_orig, 1))
    basis_out = np.zeros(m, dtype=np.int32)
    obj = C.c_double(float("nan"))
    iters = C.c_int(0)
    te = np.full(max(trace_cap, 1), -1, dtype=np.int32)
    tl = np.full(max(trace_cap, 1), -1, dtype=np.int32)
    L = lib()
    if fn_name == "reference":
        st = L.orc_simplex_reference(_d(Af), m, n, _d(b), _d(c), _i(basis), int(maximize), n_orig,
                                     eps, max_iter, int(extra), _d(x), _i(basis_out), C.byref(obj),
                                     C.byref(iters), _i(te), _i(tl), trace_cap)
        tab = None
    else:
        tab = np.zeros((m + 1, n + 1)) if extra else None
        st = L.orc_simplex_tableau(_d(Af), m, n, _d(b), _d(c), _i(basis), int(maximize), n_orig,
                                   eps, max_iter, _d(x), _i(basis_out), C.byref(obj),
                                   C.byref(iters), _i(te), _i(tl), trace_cap,
                                   _d(tab) if tab is not None else None)
    k = min(iters.value, trace_cap)
    return dict(status=st, x=x[:n_orig], basis=basis_out, obj=obj.value, iters=iters.value,
                trace=list(zip(te[:k].tolist(), tl[:k].tolist())), tableau=tab)


def simplex_reference(A, b, c, basis, maximize=True, n_orig=None, eps=1e-9, max_iter=10000,
                      trace_cap=0, dense_eta_product=False):
    n_orig = A.shape[1] if n_orig is None else n_orig
    return _simplex("reference", A, b, c, basis, maximize, n_orig, eps, max_iter, trace_cap,
                    dense_eta_product)


def simplex_tableau(A, b, c, basis, maximize=True, n_orig=None, eps=1e-9, max_iter=10000,
                    trace_cap=0, want_tableau=False):
    n_orig = A.shape[1] if n_orig is None else n_orig
    return _simplex("tableau", A, b, c, basis, maximize, n_orig, eps, max_iter, trace_cap,
                    want_tableau)


def two_phase(A, b, c, maximize=False, n_orig=None, eps=1e-9, max_iter=10000):
    """orc_two_phase: no starting basis needed."""
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    n_orig = n if n_orig is None else n_orig
    Af, b, c = _colmajor(A), _f64(b), _f64(c)
    x = np.zeros(n_orig)
    basis_out = np.full(m, -1, dtype=np.int32)
    obj = C.c_double(float("nan"))
    iters = np.zeros(3, dtype=np.int32)
    st = lib().orc_two_phase(_d(Af), m, n, _d(b), _d(c), int(maximize), n_orig, eps, max_iter,
                             _d(x), _i(basis_out), C.byref(obj), _i(iters))
    return dict(status=st, x=x, basis=basis_out, obj=obj.value, iters=iters.tolist())


def chain_select(v, mask=None, want_max=True, eps=1e-9):
    v = _f64(v)
    best = C.c_double(0.0)
    mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).tobytes()
    j = lib().orc_chain_select(_d(v), mk, len(v), int(want_max), eps, C.byref(best))
    return j, best.value


def basic_solution(A, b, basis):
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    x = np.zeros(n)
    basis = np.ascontiguousarray(basis, dtype=np.int32)
    st = lib().orc_basic_solution(_d(_colmajor(A)), m, n, _d(_f64(b)), _i(basis), _d(x))
    return st, x


def is_feasible_basis(A, b, basis):
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    basis = np.ascontiguousarray(basis, dtype=np.int32)
    return bool(lib().orc_is_feasible_basis(_d(_colmajor(A)), m, n, _d(_f64(b)), _i(basis)))


def evaluate(c, x):
    c, x = _f64(c), _f64(x)
    return lib().orc_evaluate(_d(c), _d(x), len(c))


def binom(n, k):
    return lib().orc_binom(n, k)


def unrank(n, m, rank):
    s = np.zeros(m, dtype=np.int32)
    lib().orc_unrank(n, m, rank, _i(s))
    return s


def rank_of(n, subset):
    s = np.ascontiguousarray(subset, dtype=np.int32)
    return lib().orc_rank(n, len(s), _i(s))


def enum_subset(A, b, c, subset):
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    s = np.ascontiguousarray(subset, dtype=np.int32)
    xB = np.zeros(m)
    z = C.c_double(float("nan"))
    st = lib().orc_enum_subset(_d(_colmajor(A)), m, n, _d(_f64(b)), _d(_f64(c)), _i(s), _d(xB),
                               C.byref(z))
    return st, xB, z.value


def enum_range(A, b, c, maximize, begin, end):
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    z = C.c_double(0.0)
    counts = (C.c_uint64 * 3)()
    st = lib().orc_enum_range(_d(_colmajor(A)), m, n, _d(_f64(b)), _d(_f64(c)), int(maximize),
                              begin, end, C.byref(z), counts)
    return st, z.value, [int(v) for v in counts]


def enum_first_within(A, b, c, maximize, begin, end, zstar, tol=1e-9):
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    return int(lib().orc_enum_first_within(_d(_colmajor(A)), m, n, _d(_f64(b)), _d(_f64(c)),
                                           int(maximize), begin, end, zstar, tol))


def enum_solve(A, b, c, maximize=True, n_orig=None):
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    n_orig = n if n_orig is None else n_orig
    x = np.zeros(n_orig)
    basis = np.zeros(m, dtype=np.int32)
    rank = C.c_uint64(0)
    obj = C.c_double(float("nan"))
    counts = (C.c_uint64 * 3)()
    st = lib().orc_enum_solve(_d(_colmajor(A)), m, n, _d(_f64(b)), _d(_f64(c)), int(maximize),
                              n_orig, _d(x), _i(basis), C.byref(rank), C.byref(obj), counts)
    return dict(status=st, x=x, basis=basis, rank=int(rank.value), obj=obj.value,
                counts=[int(v) for v in counts])
