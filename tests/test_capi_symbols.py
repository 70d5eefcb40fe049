"""CPU-only: the C-ABI library loads and exports every symbol include/simplexmethod_amd.h
declares (no compute calls — there is no GPU here), and refuses to work without a device."""
import ctypes
import os
import re

import pytest

from simplexmethod_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "simplexmethod_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?\s+\*?(lp_[a-z0-9_]+)\s*\(", text,
                       flags=re.M)
    return sorted(set(names))


def test_header_symbols_exported():
    build.build_hip()
    lib = ctypes.CDLL(build.HIP_LIB)
    declared = _declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    # and the binding covers exactly the header
    assert sorted(capi.SIGNATURES) == declared


def test_abi_version_and_strings():
    lib = capi.load()
    assert lib.lp_abi_version() == 4
    assert lib.lp_status_string(0) == b"optimal"
    assert lib.lp_status_string(3) == b"singular basis matrix"
    assert lib.lp_binom(32, 16) == 601080390 and lib.lp_binom(28, 14) == 40116600


def test_no_cpu_fallback():
    """Without a usable GPU the product path must fail loudly, not compute on the CPU."""
    lib = capi.load()
    if lib.lp_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(capi.LPError):
        capi.Context(0)


def test_product_path_does_not_touch_the_oracle():
    """simplexmethod_amd/ and include/ never import, link or mention oracle/."""
    bad = []
    for base in ("simplexmethod_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "_build" in dirpath or "__pycache__" in dirpath:
                continue
            for f in files:
                if not f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    continue
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"(import\s+oracle|from\s+oracle|pyoracle|lp_oracle\.h|liblp_oracle|#include\s+\"[^\"]*oracle)", src):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_generator_matches_oracle_generator():
    import numpy as np
    from oracle import pyoracle as o
    for seed, m, n in [(0, 3, 8), (5, 16, 32), (4095, 7, 9)]:
        a = capi.gen_lp(seed, m, n)
        b = o.gen_lp(seed, m, n)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_shard_bounds_match_the_python_driver():
    """lp_enum_shard_bounds (used by the C++ EnumerationSolver's multi-GPU mode) cuts the rank space
    exactly where simplexmethod_amd.dist.balanced_shard_bounds (used by bench.py) does."""
    import ctypes as C
    from simplexmethod_amd import capi, dist as lpdist
    lib = capi.load()
    for n, m in [(32, 16), (28, 14), (24, 8), (20, 10), (12, 5), (40, 9)]:
        for world in (1, 2, 3, 8):
            for r in range(world):
                lo, hi = C.c_uint64(0), C.c_uint64(0)
                assert lib.lp_enum_shard_bounds(n, m, r, world, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == lpdist.balanced_shard_bounds(n, m, r, world), (n, m, r, world)
    lo, hi = C.c_uint64(0), C.c_uint64(0)
    assert lib.lp_enum_shard_bounds(32, 16, 8, 8, C.byref(lo), C.byref(hi)) == capi.BAD_ARG
    assert lib.lp_enum_shard_bounds(10, 12, 0, 2, C.byref(lo), C.byref(hi)) == capi.BAD_ARG
