"""bench.py over real processes and real RCCL (one process per GPU), as the driver launches it.

On a box with >= 2 devices: `python bench.py --gpus 2` (no launcher environment: bench.py re-executes itself under
torch.distributed.run) must print ONE JSON line whose partition-independent checksum equals the CPU oracle's constant -- the halo
exchange (ncclSend / ncclRecv of packed face slabs, comm.hip; reference: SynchronizerMPI_AMR::sync, main.cpp:2356-2405) carried the
right bits -- and whose BiCGSTAB iteration counts stay within the erratic-case band of the one-process run of the same command (the all-reduces,
main.cpp:14486 / 14546, only reorder the sums).  On a one-GPU box those tests are skipped and the refusal path is checked instead."""
import json
import os
import subprocess
import sys

import pytest

import cup3d_amd as cu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


POISSON_KEYS = ("lhs_exact_field", "precond_exact_field", "fused_iteration")
ALL_KEYS = ("exact_field", "taylor_green") + POISSON_KEYS


def same_bits_at_every_n(ck1, ckn):
    """every bitwise signal of config.checksum: the N-rank value equals the one-process value (and the recorded constant where one exists)"""
    assert ckn["ok"] is not False and ck1["ok"] is not False, (ck1, ckn)
    for k in ALL_KEYS:
        assert ckn[k]["value"] == ck1[k]["value"], (k, ck1[k], ckn[k])
        assert ckn[k]["ok"] is not False, (k, ckn[k])
    assert ckn["fused_iteration"]["vectors"] == ck1["fused_iteration"]["vectors"]


FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "librccl_fake.so")


def run_bench(*args, timeout=900, extra_env=None, alt_early=False):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    # --full-line: these tests read the full record (checksum values, per-step arrays); the default stdout line is the compact summary
    # --no-alt-early: the optional early-all-reduce region of N > 1 runs has its own test below
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--detail-out", "", *([] if alt_early else ["--no-alt-early"]), *args], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_one_process_bench_carries_a_matching_checksum():
    out = run_bench("--size", "128", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-alt", "--no-pcie")
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = out.stdout.decode().strip().splitlines()
    assert len(lines) == 1
    r = json.loads(lines[0])
    ck = r["config"]["checksum"]
    assert ck["ok"] is True and ck["exact_field"]["ok"] is True and ck["exact_field"]["value"] == ck["exact_field"]["expected"]
    assert ck["taylor_green"]["value"] == ck["taylor_green"]["expected"]   # holds as long as numpy's sin / cos equal the build container's
    # the Poisson path: A p against the CPU oracle's constant; M^-1 p and one fused iteration against the recorded one-GPU values
    for k in POISSON_KEYS:
        assert ck[k]["ok"] is True and ck[k]["value"] == ck[k]["expected"], (k, ck[k])
    assert len(ck["fused_iteration"]["vectors"]) == 18
    assert r["config"]["communication"]["rccl_ranks"] == 1 and r.get("valid", True) is True
    assert len(r["config"]["bicgstab_iters_by_step"]) == 2
    assert r["ms_per_bicgstab_iteration"] > 0


def test_iteration_count_over_the_bench_window_against_the_references_recorded_one():
    """The solver's iteration count per step is erratic (it swings by 20-30 % between two summation orders of the SAME algorithm, the
    multi-threaded reference's own included), so a per-step comparison says little (tests/test_gpu_parity.py keeps a gross factor-2 bound per step
    and asserts windows).  Over a WINDOW of steps the swings average out: the compiled reference's own time loop was recorded for 25 steps from step 21
    at 256^3 (profiles/r03/reference_window_256.json: 71 ... 195 per step, mean 120.6) -- the device, running the same 25 steps of the
    bench's workload, must land within 15 % of that mean (round 4: 111; at 512^3, the driver's window: 170.9 against 183.1, in every
    bench line as config.ref_iters_per_step)."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r03", "reference_window_256.json")))
    ref = [st["iters"] for st in rec["steps"]]
    out = run_bench("--size", "256", "--steps", str(len(ref)), "--warmup", "0", "--no-cpu", "--no-alt", "--no-pcie", "--no-checksum", timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    r = json.loads([l for l in out.stdout.decode().strip().splitlines() if l.startswith("{")][-1])
    dev = r["config"]["bicgstab_iters_by_step"]
    assert len(dev) == len(ref) == 25 and r["config"]["ref_iters_per_step"]["by_step"] == ref
    mean_dev, mean_ref = sum(dev) / len(dev), sum(ref) / len(ref)
    print(f"256^3, steps 21..45: device {mean_dev:.1f} iterations per step ({min(dev)}..{max(dev)}), compiled reference {mean_ref:.1f} ({min(ref)}..{max(ref)})")
    assert abs(mean_dev - mean_ref) <= 0.15 * mean_ref, (mean_dev, mean_ref, dev, ref)


def test_more_gpus_than_devices_is_refused_clearly():
    n = cu.capi.device_count()
    out = run_bench("--gpus", str(n + 1), "--size", "64", "--steps", "1", "--warmup", "0", "--no-cpu")
    assert out.returncode == 2 and f"needs {n + 1} devices".encode() in out.stderr and out.stdout.strip() == b""


@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_over_rccl_matches_the_one_process_run(n):
    if cu.capi.device_count() < n:
        pytest.skip(f"{n} devices needed, {cu.capi.device_count()} visible")
    args = ("--size", "128", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-alt", "--no-pcie")
    one = run_bench(*args)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    many = run_bench("--gpus", str(n), *args)
    assert many.returncode == 0, many.stderr.decode()[-3000:]
    lines = many.stdout.decode().strip().splitlines()
    assert len(lines) == 1, lines
    r1, rn = json.loads(one.stdout.decode().strip()), json.loads(lines[0])
    assert rn["n_gpus"] == n and rn["config"]["communication"]["rccl_ranks"] == n
    # the bits: vector halo (advect-diffuse), scalar halo read in place by the fused kernels, block CG -- equal at every N
    same_bits_at_every_n(r1["config"]["checksum"], rn["config"]["checksum"])
    assert rn["config"]["checksum"]["ok"] is True
    # the solver over ranks: the all-reduces only reorder the dot-product sums, both runs project to the same stopping tolerance
    u1, un = r1["config"]["umax_by_step"], rn["config"]["umax_by_step"]
    assert max(abs(a - b) / a for a, b in zip(u1, un)) <= 2e-3, (u1, un)
    com = rn["config"]["communication"]
    assert com["halo_exchanges_per_iteration"] > 0
    for k in ("halo_ms_per_iteration", "allreduce_ms_per_iteration", "exposed_ms_per_iteration"):
        assert com[k] >= 0, (k, com)
    assert com["halo_ms_per_iteration"] > 0 and com["allreduce_ms_per_iteration"] > 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n", [2, 3])
def test_bench_multi_process_path_over_the_host_transport(n):
    """`bench.py --gpus N --transport host`: bench.py's own multi-process path -- it re-executes itself under torch.distributed.run, every
    rank builds its Hilbert range of the grid, the checksum pass and the timed steps run with halo exchanges and all-reduced BiCGSTAB
    scalars -- on ONE GPU: the library's exchanges are staged through host memory and carried by gloo (cup3d_debug_host_transport)
    where production uses RCCL.  The partition-independent checksums must equal the CPU oracle's constants on 2 and on 3 ranks (an
    odd split of the Hilbert curve), and the iteration counts stay in the band of the one-process run."""
    args = ("--size", "128", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-alt", "--no-pcie")
    one = run_bench(*args)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    many = run_bench("--gpus", str(n), "--transport", "host", *args, timeout=800)
    assert many.returncode == 0, many.stderr.decode()[-3000:]
    lines = [l for l in many.stdout.decode().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, many.stdout.decode()[-2000:]
    r1, rn = json.loads(one.stdout.decode().strip()), json.loads(lines[0])
    assert rn["n_gpus"] == n and rn["config"]["communication"]["rccl_ranks"] == n and "host-memory" in rn["config"]["transport"]
    ck1, ckn = r1["config"]["checksum"], rn["config"]["checksum"]
    assert ckn["ok"] is True and ckn["exact_field"]["value"] == ck1["exact_field"]["value"] == ckn["exact_field"]["expected"]
    # every bitwise signal -- incl. A p over the scalar halo, the block CG and ONE fused iteration's 18 vectors -- equal on 1, 2, 3 ranks
    same_bits_at_every_n(ck1, ckn)
    # the solver over ranks: both runs project to the same stopping tolerance, so max|u| along the run agrees to ~1e-4; the iteration
    # COUNT of this all-wall workload is not a usable signal at 128^3 (131 / 105 on one rank, 93 / 53 on two: the count at which the
    # residual first dips below the tolerance moves by a factor of two with the order of the sums)
    u1, un = r1["config"]["umax_by_step"], rn["config"]["umax_by_step"]
    assert len(u1) == len(un) == 3 and max(abs(a - b) / a for a, b in zip(u1, un)) <= 2e-3, (u1, un)
    i1, im = r1["config"]["bicgstab_iters_by_step"], rn["config"]["bicgstab_iters_by_step"]
    print(f"{n} ranks over the host transport: iterations {im} (one process: {i1}); max|u| {un} vs {u1}")
    assert rn["config"]["communication"]["halo_exchanges_per_iteration"] >= 2 and rn["config"]["communication"]["allreduces_per_iteration"] >= 2
    com = rn["config"]["communication"]
    for k in ("halo_ms_per_iteration", "allreduce_ms_per_iteration", "exposed_ms_per_iteration"):   # the keys a SCALE run is read by
        assert k in com and com[k] >= 0, (k, com)


def parse_error_line(out):
    lines = [l for l in out.stdout.decode().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.stdout.decode()[-1500:], out.stderr.decode()[-1500:])
    return json.loads(lines[0])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("how, stage", [("hang", "timed"), ("exit", "warmup"), ("raise", "checksum"), ("hang", "comm_init")])
def test_a_rank_that_hangs_or_dies_yields_one_diagnostic_line(how, stage):
    """First contact with N > 1 must not be able to fail silently: when ONE rank hangs (a stuck collective), leaves (a crash) or raises
    in any stage, the run ends within the stall limit with ONE JSON line {"valid": false, "error", "stage", "rank_progress"} and a
    non-zero exit code -- never a result line, never the launcher's 1800 s."""
    import time
    t0 = time.time()
    out = run_bench("--gpus", "2", "--transport", "host", "--size", "64", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-alt", "--no-pcie",
                    "--stall-timeout", "25", "--fail-at", f"{stage}:1:{how}", timeout=400)
    took = time.time() - t0
    assert out.returncode != 0
    r = parse_error_line(out)
    assert r["valid"] is False and r["value"] is None and r["n_gpus"] == 2 and r["error"], r
    assert len(r["rank_progress"]) == 2
    stages = [p["stage"] for p in r["rank_progress"]]
    assert stage in stages, r      # the line says where the run was
    assert took < 390, took        # (25 s stall limit + start-up -- importing torch in two fresh processes took from 3 s to minutes on the pool's boxes; a silent hang would sit here for the launcher's half hour)


@pytest.mark.timeout(600)
def test_bench_on_a_multilevel_mesh_over_rank_views():
    """`bench.py --amr --gpus 2 --transport host`: every rank builds the multi-level mesh, takes its contiguous run of the block order as a
    rank view (ghost blocks in sub-boxes, face fluxes, all-reduced scalars) and steps it.  Sub-boxes against whole ghost blocks
    (`whole_ghost_blocks=1`): the same BiCGSTAB iteration counts and the same max|u| to the last bit -- the exchange form changes which
    cells travel, not what is read -- with several times fewer bytes per iteration; the line says how the ghost blocks travelled."""
    args = ("--amr", "--amr-base", "3", "--amr-levels", "3", "--gpus", "2", "--transport", "host", "--steps", "2", "--warmup", "1")
    runs = {}
    for name, extra in (("subbox", ()), ("whole", ("--debug-option", "whole_ghost_blocks=1"))):
        out = run_bench(*args, *extra, timeout=500)
        assert out.returncode == 0, out.stderr.decode()[-3000:]
        lines = [l for l in out.stdout.decode().strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout.decode()[-2000:]
        runs[name] = json.loads(lines[0])
    a, b = runs["subbox"]["config"], runs["whole"]["config"]
    assert runs["subbox"]["n_gpus"] == 2 and a["blocks"] == b["blocks"] > 1000 and len(a["blocks_per_level"]) == 3
    assert a["bicgstab_iters_by_step"] == b["bicgstab_iters_by_step"] and a["umax_by_step"] == b["umax_by_step"]
    ca, cb = a["communication"], b["communication"]
    assert "sub-boxes" in ca["ghost_blocks_travel_as"] and "whole" in cb["ghost_blocks_travel_as"]
    assert ca["ghost_and_flux_exchanges_per_iteration"] == cb["ghost_and_flux_exchanges_per_iteration"] > 2
    ratio = cb["MB_sent_per_iteration (rank 0)"] / ca["MB_sent_per_iteration (rank 0)"]
    print(f"bench --amr on 2 rank views: {a['blocks']} blocks, {ca['MB_sent_per_iteration (rank 0)']} MB per iteration in sub-boxes, x{ratio:.2f} as whole blocks")
    assert ratio > 2.5


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n,mode", [(2, "values"), (3, "values"), (2, "hostfunc")])
def test_bench_rccl_code_path_with_the_stand_in_library(n, mode):
    """`bench.py --gpus N` in its PRODUCTION configuration -- `--transport rccl`, the RELEASE library, comm.hip's RCCL branch: dlopen,
    ncclGetUniqueId on rank 0, the id over gloo, ncclCommInitRank, grouped ncclSend / ncclRecv of the packed face slabs on the
    communication stream, ncclAllReduce + the recurrence step behind it, the status agreement's flag, cup3d_comm_finalize -- on ONE GPU:
    librccl is replaced AT ITS OWN API by tests/fake_rccl (CUP3D_RCCL_LIBRARY; stream-ordered copies through shared-memory mailboxes),
    because RCCL refuses two ranks on one device.  Everything but RCCL's internals executes.  All five bitwise signals equal the
    one-process run's and the recorded constants; max|u| along the run agrees to the projections' stopping tolerance.
    mode: how the stand-in orders the copies of two processes -- "values": hipStreamWriteValue64 / hipStreamWaitValue64 on sequence words in
    the pinned shared segment (device-side waits, as RCCL's are); "hostfunc": round 4's blocking host functions, kept as a variant.
    (Round 4 ran these cases opt-in; they are part of the default suite again, tests/test_gpu_00_dropin_mpi.py says why.)"""
    if not os.path.exists(FAKE_RCCL):
        pytest.skip("tests/fake_rccl/librccl_fake.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    args = ("--size", "128", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-alt", "--no-pcie")
    one = run_bench(*args)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    many = run_bench("--gpus", str(n), *args, timeout=800, extra_env={"CUP3D_RCCL_LIBRARY": FAKE_RCCL, "CUP3D_BENCH_SHARE_DEVICE": "1",
                                                                             "CUP3D_HIP_FLAVOUR": "release", "FAKE_RCCL_MODE": mode, "FAKE_RCCL_VERBOSE": "1"})   # (the suite's own processes load the testing build)
    assert many.returncode == 0, (many.stdout.decode()[-1500:], many.stderr.decode()[-3000:])
    lines = [l for l in many.stdout.decode().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, many.stdout.decode()[-2000:]
    r1, rn = json.loads(one.stdout.decode().strip()), json.loads(lines[0])
    assert rn["n_gpus"] == n and rn["config"]["transport"] == "rccl" and rn["config"]["library"] == "libcup3d_hip.so"   # the release build
    assert rn["config"]["checksum"]["ok"] is True
    same_bits_at_every_n(r1["config"]["checksum"], rn["config"]["checksum"])
    u1, un = r1["config"]["umax_by_step"], rn["config"]["umax_by_step"]
    assert len(u1) == len(un) == 3 and max(abs(a - b) / a for a, b in zip(u1, un)) <= 2e-3, (u1, un)
    com = rn["config"]["communication"]
    assert com["rccl_ranks"] == n and com["halo_exchanges_per_iteration"] >= 2 and com["allreduces_per_iteration"] >= 2
    assert com["halo_ms_per_iteration"] > 0 and com["allreduce_ms_per_iteration"] > 0 and com["exposed_ms_per_iteration"] >= 0
    assert f"mode {mode}".encode() in many.stderr, many.stderr.decode()[-1500:]   # the stand-in says which ordering it runs
    print(f"{n} ranks through comm.hip's RCCL branch (stand-in library, {mode}): iterations {rn['config']['bicgstab_iters_by_step']} "
          f"(one process: {r1['config']['bicgstab_iters_by_step']}); per iteration: halo {com['halo_ms_per_iteration']} ms, all-reduce "
          f"{com['allreduce_ms_per_iteration']} ms, exposed {com['exposed_ms_per_iteration']} ms")



@pytest.mark.timeout(900)
def test_early_allreduce_is_deterministic_and_solves_the_same_problem():
    """The all-reduce of a fused loop's dot products can start when the LAST BLOCK LEAVES ITS VECTOR PHASE instead of when the kernel ends
    (poisson.hip, `early`: the loop kernels total their per-block values themselves -- k_loop1_cg_tot / k_loop2_cg_tot -- and raise a flag
    that k_wait_totals on the communication stream waits for; the mean-constraint total follows in an all-reduce of its own behind a flag
    that only the corner block's wavefront waits for).  What must hold:
      * forcing the scalars of ONE process through a one-rank communicator (the stand-in library, 30 us injected per all-reduce) in the
        DEFAULT order changes nothing at all: the same totalling launch, the same recurrence functions -- iteration counts and max|u|
        identical to the plain one-process run (with its every-50th iterations in the same launch-by-launch form), bit for bit;
      * the early order adds the dot products in another tree (the in-kernel one), so its iteration counts differ like those of any two
        summation orders of this erratic solver (66 / 84 / 82 against 131 / 105 / 113 seen) -- but it is DETERMINISTIC (two runs: identical
        counts and max|u|), every solve converges, the five bitwise signals hold, and max|u| along the run agrees with the default order to
        the projections' stopping tolerance; the same on two processes sharing the device."""
    if not os.path.exists(FAKE_RCCL):
        pytest.skip("tests/fake_rccl/librccl_fake.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    args = ("--size", "128", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-alt", "--no-pcie")
    fake = {"CUP3D_RCCL_LIBRARY": FAKE_RCCL, "FAKE_RCCL_ALLREDUCE_US": "30"}

    def go(*extra, env=None, timeout=500):
        out = run_bench(*args, *extra, timeout=timeout, extra_env=env)
        assert out.returncode == 0, (out.stdout.decode()[-1500:], out.stderr.decode()[-3000:])
        lines = [l for l in out.stdout.decode().strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout.decode()[-2000:]
        return json.loads(lines[0])

    def same_problem(c, c1):
        assert c["checksum"]["ok"] is True
        assert all(3 < i < 1000 for i in c["bicgstab_iters_by_step"]), c["bicgstab_iters_by_step"]
        assert max(abs(a - b) / a for a, b in zip(c["umax_by_step"], c1["umax_by_step"])) <= 2e-3, (c["umax_by_step"], c1["umax_by_step"])

    # (any debug option selects the testing build: the same flavour in all runs.  no_fuse_refresh: the every-50th iterations in the launch-by-
    #  launch form the runs through a communicator use -- their fused form, one rank only, adds its dot products per block first)
    one = go("--debug-option", "no_fuse_refresh=1")["config"]
    forced = lambda early: go("--debug-option", "force_allreduce=1", "--debug-option", f"early_allreduce={early}", env=dict(fake, CUP3D_FORCE_COMM="1"))["config"]
    d = forced(0)
    assert d["bicgstab_iters_by_step"] == one["bicgstab_iters_by_step"] and d["umax_by_step"] == one["umax_by_step"], (d["bicgstab_iters_by_step"], one["bicgstab_iters_by_step"])
    assert d["communication"]["allreduces_per_iteration"] >= 2
    e1, e2 = forced(1), forced(1)
    assert e1["bicgstab_iters_by_step"] == e2["bicgstab_iters_by_step"] and e1["umax_by_step"] == e2["umax_by_step"], "the in-kernel totals are not deterministic"
    same_problem(e1, one)
    assert e1["communication"]["allreduces_per_iteration"] >= 4   # the mean-constraint total travels on its own
    two = {}
    for early in (0, 1):
        two[early] = go("--gpus", "2", env=dict(fake, CUP3D_BENCH_SHARE_DEVICE="1", CUP3D_HIP_FLAVOUR="release", CUP3D_EARLY_ALLREDUCE=str(early)), timeout=800)["config"]
    same_problem(two[0], one)
    same_problem(two[1], one)
    same_bits_at_every_n(one["checksum"], two[1]["checksum"])
    assert two[1]["communication"]["allreduces_per_iteration"] > two[0]["communication"]["allreduces_per_iteration"]
    print("early all-reduce: one process", e1["bicgstab_iters_by_step"], "(default order", one["bicgstab_iters_by_step"], "), two processes", two[1]["bicgstab_iters_by_step"],
          "(default order", two[0]["bicgstab_iters_by_step"], "); exposed scalar wait per iteration, one process:", d["communication"]["exposed_scalar_wait_ms_per_iteration"], "->",
          e1["communication"]["exposed_scalar_wait_ms_per_iteration"], "ms (30 us injected per all-reduce)")


@pytest.mark.timeout(900)
def test_bench_times_the_early_order_as_an_optional_region_that_cannot_lose_the_result():
    """`bench.py --gpus N` over RCCL: `value` comes from the DEFAULT order of the all-reduces (the one every multi-rank test has run); the
    early order (CUP3D_EARLY_ALLREDUCE, poisson.hip) is timed afterwards in the same process as `alt_early_allreduce` -- the reference
    hides its two MPI_Iallreduce by default (main.cpp:14486-14490, 14546-14550).  The region is optional: (a) it runs and reports (two
    ranks, the stand-in library); (b) a rank that hangs inside it does not cost the run its result -- the watchdog prints the MAIN line
    with alt_early_allreduce = {"error": ...} and the exit code is 0."""
    if not os.path.exists(FAKE_RCCL):
        pytest.skip("tests/fake_rccl/librccl_fake.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    args = ("--gpus", "2", "--size", "128", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-alt", "--no-pcie")
    env = {"CUP3D_RCCL_LIBRARY": FAKE_RCCL, "CUP3D_BENCH_SHARE_DEVICE": "1", "CUP3D_HIP_FLAVOUR": "release"}
    ok = run_bench(*args, timeout=800, extra_env=env, alt_early=True)
    assert ok.returncode == 0, (ok.stdout.decode()[-1500:], ok.stderr.decode()[-3000:])
    lines = [l for l in ok.stdout.decode().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    e = r["alt_early_allreduce"]
    assert "error" not in e and e["value"] > 0 and e["bicgstab_iters_per_step"] > 3 and e["allreduce_ms_per_iteration"] > 0
    assert r["config"]["communication"]["early_allreduce"] is False and r["config"]["checksum"]["ok"] is True      # `value` is the default order's
    print(f"2 ranks (stand-in): default order {r['ms_per_bicgstab_iteration']} ms per iteration, early order {e['ms_per_bicgstab_iteration']}")
    hang = run_bench(*args, "--fail-at", "alt_early:1:hang", "--stall-timeout", "25", timeout=800, extra_env=env, alt_early=True)
    assert hang.returncode == 0, (hang.stdout.decode()[-1500:], hang.stderr.decode()[-3000:])
    lines = [l for l in hang.stdout.decode().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, hang.stdout.decode()[-2000:]
    r2 = json.loads(lines[0])
    assert r2["value"] > 0 and r2.get("valid", True) is True and r2["config"]["checksum"]["ok"] is True
    assert "error" in r2["alt_early_allreduce"] and "alt_early" in r2["alt_early_allreduce"]["error"]
