"""BASELINE configs[3] and configs[4] END TO END on real MPI ranks: the unmodified reference translation unit (fish geometry,
CreateObstacles, UpdateObstacles, Penalization with its force all-reduce 13913-13938, adaptMesh with the LoadBalancer 4729-5021,
ComputeForces) with the two hot-path operators swapped for the HIP ones by the C++ shim (cup3d_amd/host/cup3d_hip_operators.h) --
every rank a process of its own, the shim's multi-rank branch live: gather the leaves, take the rank's view of the mesh
(cup3d_grid_rank_view), ghost-block and face-flux exchanges, all-reduced BiCGSTAB scalars, the mirror rebuilt after every adaptation
and migration the reference performs.

The builder's and the driver's test boxes have ONE GPU and RCCL refuses two ranks on one device, so the bytes travel through the
library's host-memory test transport (cup3d_debug_host_transport, include/cup3d_hip_testing.h) carried by the reference's own MPI --
everything but ncclSend / ncclRecv / ncclAllReduce themselves is the production path.  Compared with the SAME harness running the
reference's CPU operators on the same number of ranks: per-rank block lists (level, Z, ownership) identical half way and at the end (30 steps for configs[3], 8 for configs[4]: every one of the first ten steps adapts the mesh),
chi / velocity / pressure to solver round-off (Poisson tolerance 1e-9 / 1e-8 on both sides).  "Solver round-off" is MEASURED, not
assumed: on these three- and four-level meshes the reference's BiCGSTAB stagnates above its tolerance, and the reference differs from
ITSELF when nothing but the order of its reductions changes -- 1 or 3 OpenMP threads on one rank: 7e-4 of the velocity and 1e-2 of
the pressure within 8 steps (scripts/diag_fish_mpi.py; at levelMax 3 both are 1e-7).  Here the reference on ONE rank is the second
opinion (with more than one thread per MPI rank this build of the reference crashes): the device on N ranks must stay within 5x the
spread between the reference on N ranks and the reference on one (or 1e-6 / 1e-4 relative where that spread is smaller).

  configs[3]: single StefanFish, chi-penalisation, 3 levels (levelMax 4, levels 1-3, ~320 blocks), 2 ranks
  configs[4]: two-fish school (the factory of the reference's run.sh), 4 levels (levelMax 5, levels 1-4, ~820 blocks = a 256^3-
              effective mesh; 1024^3-effective is levelMax 7 and minutes of CPU reference per step), 8 ranks
"""
import os
import signal
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
REF_MPI = O.REF_TOOL_MPI
REF_HIP_MPI = os.path.join(O.ORACLE_DIR, "_ref", "ref_tool_hip_mpi_testing")
# (this file sorts first among the GPU tests on purpose: N ranks + the test process share ONE GPU here, and the more queues the test
#  process has already opened the slower the ranks' many tiny synchronisations get -- the 8-rank case took 21 s, 46 s, 285 s and, twice,
#  did not end within 7 and 20 minutes on boxes of the same pool: nine processes time-slicing one device, ~10 blocking host
#  synchronisations per BiCGSTAB iteration and rank in this transport.  Each rank is held to two hardware queues; a launch that is slow but
#  still moving runs on until the whole test's budget is used up, and only then does the 8-rank case skip -- a HUNG launch always fails)
ENV = dict(os.environ, OMP_NUM_THREADS="1", LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib:" + os.environ.get("LD_LIBRARY_PATH", ""),
           HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="2")
for k in ("OMP_PROC_BIND", "GOMP_CPU_AFFINITY", "OMP_PLACES"):
    ENV.pop(k, None)
# CUP3D_CONFIGS4_LEVELMAX=7 (builder runs, profiles/r04): the 8-rank case on the 1024^3-effective mesh configs[4] names (10 060 blocks)
# instead of the 256^3-effective one the suite runs by default (820 blocks); the limits scale with it
BIG = int(os.environ.get("CUP3D_CONFIGS4_LEVELMAX", "5"))
RUN_LIMIT = 240 if BIG <= 5 else 1500   # seconds a launch of the harness is EXPECTED to take at most (the longest takes 10-40 s when the GPU switches between the ranks quickly)
STALL_LIMIT = int(os.environ.get("CUP3D_STALL_LIMIT", "0")) or (150 if BIG <= 5 else 600)  # ... and seconds without a single line of output before the launch counts as HUNG (a failure, never a skip)
# a launch that is past RUN_LIMIT but still producing output is SLOW, not wrong: it keeps running until the whole test's budget (below the
# pytest timeout of the test) is used up -- round 4 turned that case into a skip at RUN_LIMIT
TEST_BUDGET = 1650 if BIG <= 5 else 3300


def _all_cpus():   # the ranks must not inherit a narrowed affinity mask from whatever ran in this process before
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except OSError:
        pass
    os.setsid()


class HarnessTimeout(Exception):
    pass
COMMON = ["-bMeanConstraint", "2", "-bpdx", "2", "-bpdy", "2", "-bpdz", "2", "-CFL", "0.4", "-Ctol", "0.1", "-extentx", "1", "-levelStart", "1", "-nu", "0.001",
          "-poissonSolver", "iterative", "-Rtol", "5", "-tdump", "0", "-tend", "0", "-factory", "", "-poissonTol", "1e-9", "-poissonTolRel", "1e-8"]
ONE_FISH = "StefanFish L=0.4 T=1.0 xpos=0.5 ypos=0.5 zpos=0.5 heightProfile=danio widthProfile=stefan bFixFrameOfRef=1"
TWO_FISH = ("StefanFish L=0.4 T=1.0 xpos=0.25 ypos=0.5 zpos=0.5 planarAngle=180 heightProfile=danio widthProfile=stefan bFixFrameOfRef=1\n"
            "StefanFish L=0.4 T=1.0 xpos=0.7 ypos=0.5 zpos=0.5 heightProfile=danio widthProfile=stefan")
def script(nsteps):   # two check points of the block lists; adaptMesh runs at every step below 10 and every 20th (main.cpp:15314)
    return [f"op steps {nsteps // 2}", "tables t12.bin", f"op steps {nsteps - nsteps // 2}", "tables t30.bin", "dump vel v.bin", "dump pres p.bin", "dump chi c.bin"]

_launcher = None


def launcher():
    """mpiexec with whatever this box needs to start local ranks (the container hostname may not resolve)."""
    global _launcher
    if _launcher is None:
        for extra in ([], ["-hosts", "127.0.0.1"], ["-launcher", "fork", "-hosts", "127.0.0.1"], ["-iface", "lo"]):
            try:
                if subprocess.run([O.MPIEXEC] + extra + ["-n", "2", "/bin/true"], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60).returncode == 0:
                    _launcher = [O.MPIEXEC] + extra
                    break
            except Exception:
                pass
        else:
            _launcher = []
    return _launcher


def rank_words(wd, nranks):
    """every rank's last word (oracle/ref_harness.cpp writes alive.r<rank> every 5 s: exchange / all-reduce / iteration counters and since
    when they stand still) -- quoted when a launch fails, hangs or is cut off, so that the log says WHERE each rank stood"""
    out = []
    for r in range(nranks):
        try:
            out.append(open(os.path.join(wd, f"alive.r{r}")).read().strip())
        except OSError:
            out.append(f"rank {r}: no word (ended within 5 s, or not a drop-in launch)")
    return " || ".join(out)


def run(tool, nranks, pre, args, wd, extra_env=None, nsteps=30, deadline=None):
    os.makedirs(wd)
    with open(os.path.join(wd, "script.txt"), "w") as f:
        f.write("\n".join(pre + script(nsteps)) + "\n")
    proc = subprocess.Popen((launcher() + ["-n", str(nranks)] if nranks > 1 else []) + [tool, "script.txt", "--"] + args, cwd=wd, env=dict(ENV, **(extra_env or {})),
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, preexec_fn=_all_cpus)
    # SLOW and HUNG are told apart: the reference prints "main.cpp: step: N" from every calcMaxTimestep (rank 0) and the harness a sign of
    # life while the library's exchange counters move, so a run that is merely slow keeps producing lines; one that produced nothing for
    # STALL_LIMIT seconds is hung and FAILS (never a skip).  A slow one runs on until the test's budget ends.
    import threading
    import time
    out_chunks, err_chunks, last = [], [], [time.time()]

    def pump(stream, sink):
        for line in iter(stream.readline, b""):
            sink.append(line)
            last[0] = time.time()

    pumps = [threading.Thread(target=pump, args=(proc.stdout, out_chunks), daemon=True), threading.Thread(target=pump, args=(proc.stderr, err_chunks), daemon=True)]
    for t in pumps:
        t.start()
    t0 = time.time()
    warned = False
    while proc.poll() is None:
        time.sleep(0.5)
        now = time.time()
        stalled, late = now - last[0] > STALL_LIMIT, now > (deadline if deadline is not None else t0 + RUN_LIMIT)
        if now - t0 > RUN_LIMIT and not warned:
            warned = True
            print(f"{os.path.basename(tool)} on {nranks} ranks: past {RUN_LIMIT} s and still making progress (slow box) -- running on; {rank_words(wd, nranks)}", flush=True)
        if stalled or late:
            words = rank_words(wd, nranks)
            os.killpg(proc.pid, signal.SIGKILL)   # mpiexec and every rank (own session, see _all_cpus)
            proc.wait()
            tail = b"".join(out_chunks[-5:]).decode()[-400:] + " | stderr: " + b"".join(err_chunks[-8:]).decode()[-800:] + " | ranks: " + words
            if stalled:
                raise AssertionError(f"{os.path.basename(tool)} on {nranks} ranks produced no output for {STALL_LIMIT} s: HUNG, not slow.  Last output: {tail}")
            raise HarnessTimeout(f"{os.path.basename(tool)} on {nranks} ranks was still making progress after {now - t0:.0f} s when the test's budget ended (slow box).  Last output: {tail}")
    for t in pumps:
        t.join(timeout=10)
    so, se = b"".join(out_chunks), b"".join(err_chunks)
    assert proc.returncode == 0, (so.decode()[-1500:], se.decode()[-3000:], rank_words(wd, nranks))
    res = []
    for r in range(nranks):
        suf = f".r{r}" if nranks > 1 else ""
        t12, t30 = O.read_tables(os.path.join(wd, "t12.bin" + suf))[0], O.read_tables(os.path.join(wd, "t30.bin" + suf))[0]
        nb = len(t30)
        res.append((t12, t30, O.read_blocks(os.path.join(wd, "v.bin" + suf), nb, 3), O.read_blocks(os.path.join(wd, "p.bin" + suf), nb, 1),
                    O.read_blocks(os.path.join(wd, "c.bin" + suf), nb, 1)))
    return res


# The stand-in for librccl (tests/fake_rccl): in round 4 these cases were opt-in after one whole-suite run in which this case and the
# 8-rank case after it failed without a traceback.  Round 5: the stand-in orders its copies with stream memory operations instead of
# blocking host functions (no host thread can sit in front of another stream's release any more), every rank leaves its last exchange
# counters behind (rank_words), the suite's GPU sessions log tracebacks line-buffered -- and the cases run in the DEFAULT suite
# (three whole-suite logs: profiles/r05/).
FAKE_RCCL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "fake_rccl", "librccl_fake.so")


@pytest.mark.timeout(1800 if BIG <= 5 else 3600)
@pytest.mark.parametrize("name,nranks,level_max,fish,min_levels,nsteps,transport", [
    ("configs3_one_fish_3_levels_2_ranks", 2, 4, ONE_FISH, 3, 30, "host"),
    # the same through the PRODUCTION branch of the shim and of comm.hip -- unique id by MPI_Bcast, cup3d_comm_init, grouped ncclSend /
    # ncclRecv of sub-boxes, face fluxes and migrating blocks, ncclAllReduce, the status agreement -- with librccl replaced at its own API by
    # tests/fake_rccl (RCCL refuses two ranks on one device): everything but RCCL's internals executes
    ("configs3_one_fish_3_levels_2_ranks", 2, 4, ONE_FISH, 3, 30, "rccl_stand_in"),
    ("configs4_two_fish_4_levels_8_ranks", 8, BIG, TWO_FISH, 4, 8, "host")])
def test_fish_with_amr_over_mpi_ranks_through_the_shim(tmp_path, name, nranks, level_max, fish, min_levels, nsteps, transport):
    if not (os.path.exists(REF_MPI) and os.path.exists(REF_HIP_MPI) and os.path.exists(O.MPIEXEC)):
        pytest.skip("needs oracle/_ref/ref_tool_mpi, ref_tool_hip_mpi_testing (built where /root/reference exists) and an mpiexec")
    if not launcher():
        pytest.skip("mpiexec cannot start local ranks on this box")
    args = COMMON + ["-levelMax", str(level_max), "-factory-content", fish]
    if transport == "rccl_stand_in" and not os.path.exists(FAKE_RCCL):
        pytest.skip("tests/fake_rccl/librccl_fake.so is not built")
    hip_env = {"CUP3D_HIP_HOST_TRANSPORT": "1"} if transport == "host" else {"CUP3D_RCCL_LIBRARY": FAKE_RCCL}
    import time
    deadline = time.time() + TEST_BUDGET   # of the three launches together; each is expected to take RUN_LIMIT at most and may take what is left
    try:
        hip = run(REF_HIP_MPI, nranks, ["hip on"], args, str(tmp_path / "hip"), hip_env, nsteps=nsteps, deadline=deadline)
        cpu = run(REF_MPI, nranks, [], args, str(tmp_path / "cpu"), nsteps=nsteps, deadline=deadline)
        one = run(O.REF_TOOL, 1, [], args, str(tmp_path / "one"), {"OMP_NUM_THREADS": "4"}, nsteps=nsteps, deadline=deadline)   # the reference against itself: one rank, four threads
    except HarnessTimeout as e:
        if nranks <= 2:
            raise
        # eight processes time-slicing one GPU with the test process, still making progress when the whole test's budget (27 minutes) was
        # used up: a property of the box, not of the code under test -- the only case left in which this test does not give a verdict
        pytest.skip(str(e))
    levels, nblocks, vmax, pmax, wet = set(), 0, 0.0, 0.0, 0
    for c in cpu:
        levels |= set(c[1][:, 0].tolist())
        nblocks += len(c[1])
        vmax, pmax, wet = max(vmax, np.abs(c[2]).max()), max(pmax, np.abs(c[3]).max()), wet + int((c[4] > 0).sum())
    assert len(levels) >= min_levels and nblocks > 100 * nranks // 2      # the mesh the config names, spread over the ranks
    assert wet > 100 and vmax > (1e-3 if level_max <= 5 else 1e-4)        # there IS a fish, and it moves the fluid (finer mesh, smaller dt: less after 8 steps)
    # the one-rank run holds the same leaves (the mesh does not depend on the number of ranks): match blocks by (level, Z)
    where = {(int(l), int(z)): i for i, (l, z) in enumerate(one[0][1][:, :2])}
    noise_v = noise_p = 0.0
    for c in cpu:
        idx = [where.get((int(l), int(z)), -1) for l, z in c[1][:, :2]]
        if min(idx) < 0:            # the two reference runs adapted differently: no second opinion on the fields
            noise_v = noise_p = 0.0
            break
        noise_v = max(noise_v, np.abs(c[2] - one[0][2][idx]).max())
        noise_p = max(noise_p, np.abs(c[3] - one[0][3][idx]).max())
    tol_v, tol_p = 5 * max(noise_v, 1e-6 * vmax), 5 * max(noise_p, 1e-4 * pmax)
    dv = dp = 0.0
    for r, (c, h) in enumerate(zip(cpu, hip)):
        assert np.array_equal(c[0], h[0]) and np.array_equal(c[1], h[1]), f"rank {r}: block lists differ"   # incl. who owns what
        assert np.abs(c[4] - h[4]).max() <= 1e-6, f"rank {r}: chi"
        dv, dp = max(dv, np.abs(c[2] - h[2]).max()), max(dp, np.abs(c[3] - h[3]).max())
    print(f"{name} [{transport}]: {nblocks} blocks on levels {sorted(levels)}, {nranks} ranks: block lists identical; max|dv| = {dv:.2e} (reference vs itself "
          f"{noise_v:.2e}, |v| = {vmax:.2e}), max|dp| = {dp:.2e} (reference vs itself {noise_p:.2e}, |p| = {pmax:.2e})")
    assert dv <= tol_v, (dv, tol_v)
    assert dp <= tol_p, (dp, tol_p)
