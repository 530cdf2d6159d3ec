"""The RELEASE build of the library (libcup3d_hip.so: no debug-option map, no in-process communicator, no A/B kernel variants) in
subprocesses -- the rest of the suite loads libcup3d_hip_testing.so (tests/conftest.py).  Same sources, same kernels: the smoke
check (advect-diffuse bit-exact against the oracle, projection to solver round-off) and a 64^3 solve must behave the same, and the
test-support entry points must refuse to work instead of silently doing nothing."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(code, flavour):
    env = dict(os.environ, CUP3D_HIP_FLAVOUR=flavour)
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


def test_smoke_on_the_release_build():
    out = run("import __graft_entry__ as g; g.smoke(); import cup3d_amd.capi as c; print('LIB', c.LIB_PATH)", "release")
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert b"smoke ok" in out.stdout and b"libcup3d_hip.so" in out.stdout and b"testing" not in out.stdout.split(b"LIB")[-1]


SOLVE = r"""
import numpy as np, cup3d_amd as cu
cu.device_init(0)
sim = cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=4, levelStart=3, extent=1.0, BC_x="wall", BC_y="wall", BC_z="wall")
rhs = np.random.default_rng(3).uniform(-1, 1, (sim.nblocks, 8, 8, 8))
sim.upload("lhs", rhs)
r = cu.makePoissonSolver(sim).solve()
print("RESULT", r.iterations, r.restarts, repr(float(r.norm)), sim.checksum("pres"))
rc = cu.lib().cup3d_debug_set_option(b"no_fuse", 1)
print("DEBUG_RC", rc, cu.lib().cup3d_debug_virtual_comm(2))
"""


def test_release_and_testing_builds_compute_the_same_bits():
    res = {}
    for flavour in ("release", "testing"):
        out = run(SOLVE, flavour)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        lines = out.stdout.decode().splitlines()
        res[flavour] = ([l for l in lines if l.startswith("RESULT")][0], [l for l in lines if l.startswith("DEBUG_RC")][0])
    assert res["release"][0] == res["testing"][0]            # iterations, restarts, final norm and the pressure's checksum
    assert res["release"][1] == "DEBUG_RC -5 -5"             # CUP3D_ESTATE: test support is not in this build
    assert res["testing"][1] == "DEBUG_RC 0 0"
