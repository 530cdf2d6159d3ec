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
                           cpu_size=256, cpu_steps=1)
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
