"""Runs the C++ tests of the host classes (tests/cpp/*.cpp): the CPU-only one mirrors the
reference's gtest cases for Canonical / Symmetrical / SymmetricalParser; the GPU one drives the
drop-in Solver / EnumerationSolver classes on the device."""
import os
import subprocess

import pytest

from simplexmethod_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exe(name):
    """The prebuilt test executable (built by __graft_entry__.build()); built here only if missing —
    on the GPU box the snapshot's file times say nothing about staleness."""
    path = os.path.join(build.TESTS_OUT, name)
    if not os.path.exists(path):
        exes = {os.path.basename(e): e for e in build.build_cpp_tests()}
        assert name in exes, f"{name} was not built"
        path = exes[name]
    return path


def test_host_classes_cpu():
    r = subprocess.run([_exe("test_host")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failed" in r.stdout


@pytest.mark.gpu
def test_solver_classes_gpu():
    r = subprocess.run([_exe("test_solvers_gpu")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failed" in r.stdout
