"""The one JSON line of bench.py (the driver's contract): every required key, the roofline of the dominant kernel computed from the
algorithmic bytes and the measured launch time, and the committed bench records under profiles/ still obey it.  No GPU: report() is fed
a recorded kernel profile."""
import argparse
import io
import json
import os
import subprocess
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline"]


class _Sim:
    nblocks = 262144


def _args(**kw):
    a = argparse.Namespace(size=512, steps=5, warmup=2, stencil_only=False, implicit_diffusion=False, nu=0.01, block_solver=0, no_cpu=True, cpu_threads=32,
                           cpu_size=256, cpu_steps=1, full_line=True, detail_out=None)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_help_runs_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, timeout=120)
    assert b"--gpus" in out.stdout and b"--steps" in out.stdout and b"--warmup" in out.stdout


def test_report_emits_the_contract():
    prof = {"bicgstab_loop2_cg": (762, 762 * 3.9), "bicgstab_loop1_cg": (762, 762 * 3.8), "poisson_lhs": (1667, 1667 * 0.53), "project_pointwise": (20, 14.0)}
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.report(_args(), _Sim(), prof, 7.4, [156, 156, 156, 156, 156], 1)
    lines = buf.getvalue().strip().splitlines()
    assert len(lines) == 1                                  # ONE JSON line
    r = json.loads(lines[0])
    for k in REQUIRED:
        assert k in r, k
    assert r["metric"].startswith("Mcell-updates/s") and r["unit"] == "Mcell-updates/s" and r["higher_is_better"] is True
    assert r["n_gpus"] == 1 and r["steps"] == 5 and r["warmup"] == 2 and r["dtype"] == "f64" and r["data"] == "synthetic" and r["vs_baseline"] is None
    assert abs(r["value"] - 512 ** 3 * 5 / 7.4 / 1e6) < 0.01 and abs(r["ms_per_step"] - 1480.0) < 1e-6
    assert "workload" in r["config"] and "model" not in r["config"]
    roof = r["roofline"]
    assert roof["kernel"] == "bicgstab_loop2_cg" and roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    want = 136 * 512 ** 3 / 3.9e-3 / 1e9                    # algorithmic bytes per launch / measured launch time
    assert abs(roof["achieved"] - want) < 0.1 and abs(roof["frac"] - want / 8000.0) < 1e-4
    assert roof["traffic"] is None or roof["traffic"] > 0.9 * 136 * 512 ** 3


def test_multigrid_kernels_get_a_roofline_per_level():
    """alt_multigrid.kernels (round 5; north_star names these kernels): profile entries "mg_smooth@L6" ... -> one record per (kernel, level) with
    the level's block count, the algorithmic bytes per cell of DESIGN.md section 4 and achieved / frac from the measured launch time; levels of
    <= 4096 blocks say that they are launch-latency bound; the finest level carries the recorded PMC traffic."""
    fine = 512 ** 3 // 512
    prof = {"mg_smooth@L6": (450, 450 * 0.70), "mg_smooth_from_zero@L6": (150, 150 * 0.40), "mg_residual_restrict@L6": (150, 150 * 0.50), "mg_prolong_add@L6": (150, 150 * 0.35),
            "mg_smooth@L5": (450, 450 * 0.095), "mg_smooth@L3": (450, 450 * 0.008), "mg_smooth_from_zero@L0": (150, 150 * 0.044), "poisson_lhs": (100, 50.0), "mg_gather": (10, 1.0)}
    ks = bench.multigrid_kernels(prof, fine)
    assert len(ks) == 7 and all("@" not in k["kernel"] for k in ks)          # only the per-level entries, split into (kernel, level)
    by = {(k["kernel"], k["level"]): k for k in ks}
    top = by["mg_smooth", 6]
    assert top["blocks"] == fine and top["algorithmic_bytes_per_cell"] == 24.0 and top["bound"] == "hbm" and top["peak"] == 8000.0
    want = 24.0 * 512 ** 3 / 0.70e-3 / 1e9
    assert abs(top["achieved"] - want) < 0.1 and abs(top["frac"] - want / 8000.0) < 1e-4 and "note" not in top
    assert top["traffic"] is None or top["traffic"] > 0.9 * 24 * 512 ** 3        # profiles/traffic.json: PMC bytes per launch on the finest level
    assert by["mg_smooth_from_zero", 6]["algorithmic_bytes_per_cell"] == 16.0 and by["mg_residual_restrict", 6]["algorithmic_bytes_per_cell"] == 17.0
    assert by["mg_smooth", 5]["blocks"] == fine // 8 and "traffic" not in by["mg_smooth", 5]
    assert by["mg_smooth", 3]["blocks"] == fine // 512 and "launch-latency" in by["mg_smooth", 3]["note"]
    assert by["mg_smooth_from_zero", 0]["blocks"] == 1
    assert [k["total_ms"] for k in ks] == sorted((k["total_ms"] for k in ks), reverse=True)   # largest share first


def test_default_stdout_line_is_compact_and_carries_roofline_and_cpu_baseline(tmp_path):
    """round 5's ~20 KB line was dropped by the driver's parser (BENCH_r05.parsed = null).  The stdout line is now a
    summary of at most bench.COMPACT_LIMIT (6 KB) bytes with the contract's keys, `roofline` and `cpu_baseline`; the full record goes
    to --detail-out.  Fed with round 5's full record (the largest line this bench ever printed) and with a synthetic report()."""
    f = os.path.join(ROOT, "profiles", "r05", "bench_512_fullstep_unfused_refresh.json")
    full = json.loads(open(f).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 15000
    line = json.dumps(bench.compact_line(full, "bench_detail.json"))
    assert len(line) < bench.COMPACT_LIMIT < 8192, len(line)
    r = json.loads(line)
    for k in REQUIRED + ["cpu_baseline", "ms_per_bicgstab_iteration"]:
        assert k in r, k
    assert r["value"] == full["value"] and r["ms_per_step"] == full["ms_per_step"] and r["config"]["workload"] == full["config"]["workload"]
    assert r["roofline"]["kernel"] == full["roofline"]["kernel"] and r["roofline"]["frac"] == full["roofline"]["frac"] and r["roofline"]["traffic"]
    assert r["roofline"]["avg_ms"] > 0 and r["cpu_baseline"]["kind"] == "reference" and r["cpu_baseline"]["cores"] and r["cpu_baseline"]["value"] > 0
    assert r["cpu_baseline"]["recorded_512"]["value"] > 0 and len(r["cpu_baseline"]["sample"]) <= 200
    assert r["config"]["checksum"]["ok"] is True and r["config"]["bicgstab_iters_per_step"] == full["config"]["bicgstab_iters_per_step"]
    assert 0.4 < r["stencil_only"]["frac_264"] < 1 and r["stencil_only"]["frac_288"] > r["stencil_only"]["frac_264"]
    assert r["alt"]["value"] > r["value"] and 0 < r["alt_multigrid"]["roofline"]["frac"] < 1
    assert "bicgstab_iters_by_step" not in r["config"] and "umax_by_step" not in r["config"] and all("note" not in k for k in r["kernels"])
    # report() itself: default = compact on stdout, full record in the detail file
    prof = {"bicgstab_loop2_cg": (762, 762 * 3.9), "bicgstab_loop1_cg": (762, 762 * 3.8), "poisson_lhs": (1667, 1667 * 0.53)}
    detail = str(tmp_path / "detail.json")
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.report(_args(full_line=False, detail_out=detail, umax_by_step=[1.0] * 6), _Sim(), prof, 7.4, [156] * 5, 1)
    lines = buf.getvalue().strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < bench.COMPACT_LIMIT
    r, d = json.loads(lines[0]), json.load(open(detail))
    assert r["roofline"]["kernel"] == "bicgstab_loop2_cg" and r["value"] == d["value"] and d["config"]["bicgstab_iters_by_step"] == [156] * 5
    assert "bicgstab_iters_by_step" not in r["config"] and r["detail"]
    # strict JSON whatever the numbers: a NaN in the record becomes null in the line
    bad = dict(full, value=float("nan"), roofline=dict(full["roofline"], frac=float("inf")))
    line = json.dumps(bench.compact_line(bad), allow_nan=False)
    assert json.loads(line)["value"] is None and json.loads(line)["roofline"]["frac"] is None


def test_iteration_count_at_512_over_the_drivers_window_against_the_recorded_reference():
    """SURVEY 8c (+-10 % on the iteration count) at the HEADLINE size, as a window mean: the device's counts over the driver's window
    (steps 26..45 of --warmup 5 --steps 20; committed with the round's own run of the driver's command) against the compiled reference's
    recorded counts for the same steps (profiles/r03/reference_window_512.json: 10-27 minutes per step, recorded once).  The same
    assertion at 256^3 runs live on the GPU (tests/test_gpu_rccl.py)."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r03", "reference_window_512.json")))
    by_step = {st["step"]: st["iters"] for st in rec["steps"]}
    seen = 0
    for f in ("r06/bench_512_driver_command_detail.json", "r06/bench_512_driver_command_detail_before_the_block_cg_change.json", "r05/bench_512_fullstep_unfused_refresh.json"):
        path = os.path.join(ROOT, "profiles", f)
        if not os.path.exists(path):
            continue
        r = json.loads(open(path).read().strip().splitlines()[-1])
        assert r["steps"] == 20 and r["warmup"] == 5 and r["config"]["cells"] == 512 ** 3
        dev = r["config"]["bicgstab_iters_by_step"]
        ref = [by_step[n] for n in range(26, 46)]
        assert len(dev) == 20 and r["config"]["ref_iters_per_step"]["by_step"] == ref
        mean_dev, mean_ref = sum(dev) / 20, sum(ref) / 20
        assert abs(mean_dev - mean_ref) <= 0.10 * mean_ref, (f, mean_dev, mean_ref)
        seen += 1
    assert seen >= 2


def test_this_rounds_bench_records_carry_the_completed_line():
    """metric (A) (`stencil_only`) beside metric (B) (`value`), per-level multigrid kernels, the recorded 512^3 CPU figure."""
    f = os.path.join(ROOT, "profiles", "r05", "bench_512_fullstep_unfused_refresh.json")
    r = json.loads(open(f).read().strip().splitlines()[-1])
    for k in REQUIRED:
        assert k in r, k
    so = r["stencil_only"]
    assert so["unit"] == "Mcell-updates/s" and so["value"] > 11100 and 0.40 < so["frac_at_264_B_per_cell (stage 1 reads no tmpV: 72 + 96 + 96)"] < 1   # north_star: >= 0.40 of the roof
    mg = r["alt_multigrid"]
    assert mg["roofline"]["bound"] == "hbm" and 0 < mg["roofline"]["frac"] < 1 and len(mg["kernels"]) >= 12
    assert {k["kernel"] for k in mg["kernels"]} >= {"mg_smooth", "mg_smooth_from_zero", "mg_residual_restrict", "mg_prolong_add"}
    assert r["cpu_baseline"]["recorded_512"]["size"] == 512 and r["cpu_baseline"]["kind"] == "reference"
    assert r["roofline"]["traffic"] and r["config"]["checksum"]["ok"] is True
    assert abs(r["ms_per_bicgstab_iteration"] - r["ms_per_step"] / r["config"]["bicgstab_iters_per_step"]) < 1e-2


def test_committed_bench_records_obey_the_contract():
    d = os.path.join(ROOT, "profiles", "r02")
    seen = 0
    for f in sorted(os.listdir(d)):
        if not (f.startswith("bench_") and f.endswith(".json")):
            continue
        r = json.loads(open(os.path.join(d, f)).read().strip().splitlines()[-1])
        for k in REQUIRED:
            assert k in r, (f, k)
        roof = r["roofline"]
        assert roof is None or (abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0 < roof["frac"] < 1), f
        assert r["value"] > 0 and r["ms_per_step"] > 0
        seen += 1
    assert seen >= 3


def test_gpus_n_without_a_launcher_relaunches_or_says_what_it_needs():
    """`python bench.py --gpus 2` as the driver types it (no WORLD_SIZE): with fewer than 2 devices it must exit 2 with a clear message,
    not fall over inside torch.distributed or run on one device (here: zero devices)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 2, out.stderr.decode()[-400:]
    assert b"needs 2 devices" in out.stderr and out.stdout.strip() == b""


def test_report_carries_the_checksum_and_marks_a_mismatch_invalid():
    prof = {"bicgstab_loop2_cg": (10, 39.0)}
    ok = {"ok": True, "exact_field": {"value": 1, "expected": 1, "ok": True}, "taylor_green": {"value": 2, "expected": 2, "ok": True}}
    bad = {"ok": False, "exact_field": {"value": 1, "expected": 3, "ok": False}, "taylor_green": {"value": 2, "expected": 2, "ok": True}}
    for ck, valid in ((ok, True), (bad, False)):
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.report(_args(checksum=ck, comm={"rccl_ranks": 2}), _Sim(), prof, 7.4, [156, 150], 2)
        r = json.loads(buf.getvalue().strip())
        assert r["config"]["checksum"] == ck and r["config"]["communication"]["rccl_ranks"] == 2 and r["config"]["bicgstab_iters_by_step"] == [156, 150]
        assert r.get("valid", True) is valid and r["n_gpus"] == 2


def test_a_size_without_recorded_constants_is_unchecked_not_invalid(capsys):
    """`no golden constant for this --size` is not a mismatch -- ok = None, no `valid: false`, exit code 0."""
    prof = {"bicgstab_loop2_cg": (10, 39.0)}
    ck = {"ok": None, "unchecked": ["exact_field"], "exact_field": {"value": 1, "expected": None, "ok": None}}
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.report(_args(checksum=ck, comm={"rccl_ranks": 1}), _Sim(), prof, 7.4, [156, 150], 1)
    r = json.loads(buf.getvalue().strip())
    assert "valid" not in r and r["config"]["checksum"]["ok"] is None
    assert r["ms_per_bicgstab_iteration"] == round(7.4e3 / 306, 4)


def test_multi_rank_run_that_cannot_start_prints_one_error_line():
    """`python bench.py --gpus 2 --transport host` on a box WITHOUT a GPU: every rank fails at device initialisation.  The run must
    end quickly with ONE JSON line {"valid": false, "error", "stage", "rank_progress"} on stdout and a non-zero exit code (the
    watchdog / failure path of N > 1 runs, exercised where no device exists)."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "host", "--size", "64", "--steps", "1", "--warmup", "0",
                          "--no-cpu"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode != 0
    lines = [l for l in out.stdout.decode().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.stdout.decode()[-1000:], out.stderr.decode()[-1000:])
    r = json.loads(lines[0])
    assert r["valid"] is False and r["value"] is None and r["n_gpus"] == 2 and r["stage"] and len(r["rank_progress"]) == 2
    assert "rror" in r["error"] or "device" in r["error"].lower(), r["error"]


def test_solver_stress_input_is_the_standard_generator_and_partition_independent():
    """bench.py --input random (SURVEY 8d's second input): std::mt19937_64 restated in numpy gives the C++ standard's check value (the
    10000th output of the default-seeded engine), and the field -- one engine per block, seeded 12345 + Hilbert index -- is the same
    bits whether one rank builds it or three."""
    import numpy as np
    import cup3d_amd as cu
    x = bench._mt19937_64(np.array([5489, 5489 + 1], dtype=np.uint64), 10000)
    assert int(x[0, 9999]) == 9981545732273789042 and int(x[1, 9999]) != int(x[0, 9999])
    ext, bc = 2 * np.pi, ("wall",) * 3
    whole = cu.Grid((1, 1, 1), 3, 2, ext, bc)
    v = bench.random_velocity_blocks(whole)
    assert v.shape == (64, 8, 8, 8, 3) and -1.0 <= v.min() < -0.99 and 0.99 < v.max() < 1.0 and abs(v.mean()) < 0.01
    by_z = {int(z): v[i] for i, z in enumerate(whole.tables[:, 1])}
    for r in range(3):
        part = cu.Grid((1, 1, 1), 3, 2, ext, bc, rank=r, nranks=3)
        vr = bench.random_velocity_blocks(part)
        for i, z in enumerate(part.tables[:, 1]):
            assert np.array_equal(vr[i], by_z[int(z)])


def test_checksum_fixture_is_what_the_oracle_produces():
    """tests/golden/advdiff_checksums.json (the constants bench.py compares the device with at every N) regenerated for the small sizes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import cup3d_amd as cu
    import oracle_lib as O
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "advdiff_checksums.json")))
    assert {"64", "128", "256", "512"} <= set(gold)
    ext = 2 * np.pi
    for size in (64, 128):
        level = (size // 8).bit_length() - 1
        g = O.OracleGrid((1, 1, 1), level + 1, level, ext, ("wall",) * 3)
        G = cu.Grid((1, 1, 1), level + 1, level, ext, ("wall",) * 3)
        assert gold[str(size)]["dt"] == bench.checksum_dt(size)
        for key, vel in (("exact_field", bench.exact_test_field_blocks(G, size)), ("taylor_green", bench.taylor_green_blocks(G, [ext] * 3, 1.0))):
            g.advect_diffuse(vel, np.zeros_like(vel), bench.checksum_dt(size), 0.01, (0.0, 0.0, 0.0))
            assert int(vel.view(np.uint64).sum(dtype=np.uint64)) == gold[str(size)][key], (size, key)
        pres = np.ascontiguousarray(bench.exact_test_field_blocks(G, size)[..., 0])        # the Poisson path's oracle-side constant: A p
        assert int(g.lhs(pres, 0).view(np.uint64).sum(dtype=np.uint64)) == gold[str(size)]["lhs_exact_field"], size
        assert {"precond_exact_field", "fused_iteration"} <= set(gold[str(size)])             # the device-side ones: recorded on one GPU


def _checksum_worker(rank, world, port, size, q):
    import numpy as np
    import torch
    import torch.distributed as dist
    import cup3d_amd as cu
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        level = (size // 8).bit_length() - 1
        G = cu.Grid((1, 1, 1), level + 1, level, 2 * np.pi, ("wall",) * 3, rank, world)   # this rank's Hilbert range (main.cpp:2970-2986)
        f = bench.exact_test_field_blocks(G, size)
        mine = int(f.view(np.uint64).sum(dtype=np.uint64))
        # bench.py's own gather (gloo, CPU tensors -- what every N > 1 run of the bench uses): one value, and a list of values at once
        a = argparse.Namespace(tdev="cpu")
        one = bench.gather_sum(mine, a, dist, world)
        per_comp = bench.gather_sum([int(np.ascontiguousarray(f[..., c]).view(np.uint64).sum(dtype=np.uint64)) for c in range(3)], a, dist, world)
        q.put((rank, one, int(G.nblocks), per_comp))
    finally:
        dist.destroy_process_group()


def test_checksum_is_independent_of_the_partition_over_gloo():
    """The per-rank wrapping sums of the exact test field, gathered and added as bench.py does, equal the one-rank sum on 2 and 3 ranks
    (the operator in between is bit-exact per block: tests/test_gpu_multirank.py)."""
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    import cup3d_amd as cu
    size = 64
    G = cu.Grid((1, 1, 1), 4, 3, 2 * np.pi, ("wall",) * 3)
    whole = bench.exact_test_field_blocks(G, size)
    want = int(whole.view(np.uint64).sum(dtype=np.uint64))
    want_comp = [int(np.ascontiguousarray(whole[..., c]).view(np.uint64).sum(dtype=np.uint64)) for c in range(3)]
    assert bench.gather_sum(want, argparse.Namespace(tdev="cpu"), None, 1) == want
    for world in (2, 3):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_checksum_worker, args=(r, world, port, size, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(60)
        assert sum(g[2] for g in got) == G.nblocks
        assert all(g[1] == want for g in got), (world, got, want)
        assert all(g[3] == want_comp for g in got), (world, got, want_comp)


def test_a_failure_inside_an_optional_region_prints_the_main_result_and_exits_zero(tmp_path):
    """bench.py --gpus N times the early all-reduce order AFTER the main result exists (`alt_early_allreduce`); while that region runs,
    Progress.fallback holds a function that prints the main line.  A failure there -- the watchdog's stall limit, an exception -- must
    print THAT line (with the error noted in it) and leave with exit code 0, where a failure anywhere else prints the error line and
    leaves non-zero.  No GPU: Progress.fail() itself, in a subprocess (it ends the process)."""
    code = r"""
import argparse, json, sys, os
sys.path.insert(0, %r)
import bench
a = argparse.Namespace(stall_timeout=5.0, timeout=100.0, watchdog=False, steps=3, warmup=1)
os.environ["CUP3D_BENCH_PROGRESS_DIR"] = %r
p = bench.Progress(0, 1, a)
p.set("alt_early", "optional region")
if sys.argv[1] == "optional":
    p.fallback = lambda error, rank_progress: print(json.dumps({"metric": "m", "value": 95.2, "alt_early_allreduce": {"error": error, "ranks": len(rank_progress)}}))  # (Progress.fail flushes stdout before it leaves)
p.fail("no progress for 5 s in stage 'alt_early'")
""" % (ROOT, str(tmp_path))
    opt = subprocess.run([sys.executable, "-c", code, "optional"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert opt.returncode == 0, opt.stderr.decode()[-500:]
    r = json.loads(opt.stdout.decode().strip().splitlines()[-1])
    assert r["value"] == 95.2 and "alt_early" in r["alt_early_allreduce"]["error"] and r["alt_early_allreduce"]["ranks"] == 1
    assert b"the main result stands" in opt.stderr
    hard = subprocess.run([sys.executable, "-c", code, "mandatory"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert hard.returncode != 0
    r = json.loads(hard.stdout.decode().strip().splitlines()[-1])
    assert r["valid"] is False and r["value"] is None and r["stage"] == "alt_early"
