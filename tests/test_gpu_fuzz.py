"""random interleavings of the public operations on one agent (profiles/diag/op_fuzz.py): episodes added with evictions, fused steps
at changing batch sizes, data-parallel steps, the reference's op-by-op calls, inference, debug fetches -- nothing may fault, raise or
go non-finite, and the replay bookkeeping follows the oracle memory throughout."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_operation_fuzz(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "diag", "op_fuzz.py"), str(seed)], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    done = [l for l in out.splitlines() if l.startswith("FUZZ seed %d " % seed)]
    assert r.returncode == 0 and len(done) == 1 and " ok " in done[0], out[-1500:]


@pytest.mark.parametrize("seed", [1, 2])
def test_operation_fuzz_naf(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "diag", "op_fuzz_naf.py"), str(seed)], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    done = [l for l in out.splitlines() if l.startswith("FUZZNAF seed %d " % seed)]
    assert r.returncode == 0 and len(done) == 1 and " ok " in done[0], out[-1500:]


def test_random_geometries_against_the_f64_oracle():
    """30 random render sizes (8..72 in both directions, odd ones included), camera / repeat counts and batch sizes through the fused
    step (graph replay or eager, at random) against the float64 oracle at the usual bar (profiles/diag/random_geometry_parity.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "diag", "random_geometry_parity.py"), "7", "30"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    out = r.stdout.decode()
    assert r.returncode == 0 and "GEODONE bad 0" in out, "\n".join(l for l in out.splitlines() if l.startswith("GEO"))[-3000:]
