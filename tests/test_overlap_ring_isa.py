"""The persistent update of k_simplex_overlap<true> keeps eight slots of loads in flight behind inline asm and waits
for them with a hand-counted s_waitcnt (simplex_overlap.hip).  The compiler does not know that a slot's destination
registers are in flight between the request and the wait: this test compiles the file to gfx950 assembly (no GPU
needed) and checks on the ISA that nothing touches them there — a spill, a copy or a reuse would read or clobber
stale data without any parity test necessarily noticing on every run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ring_registers_are_untouched_between_request_and_wait():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_ring_asm.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "slot requests checked: 24 violations: 0" in r.stdout
