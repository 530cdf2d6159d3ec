#!/usr/bin/env python3
"""BASELINE configs[4] at the size it names -- two-fish school, 1024^3-effective mesh (bpd 2, levelMax 7: levels 2..6) -- run ONCE with
the reference's CPU operators and once with the drop-in, and the Amdahl split of a step recorded:

    python scripts/configs4_measure.py [--level-max 7] [--steps 5] [--threads 32] [--out profiles/r04/configs4_1024_effective.json]

Both runs are the UNMODIFIED reference translation unit (oracle/_ref/ref_tool, ref_tool_hip: fish geometry, CreateObstacles,
UpdateObstacles, Penalization, ComputeForces, adaptMesh with its LoadBalancer all stay the reference's CPU code -- SURVEY section 2
#24-26 are out of scope); in the second one the two hot-path operators are swapped by the C++ shim in its device-led mode
(`hip resident3`, cup3d_amd/host/cup3d_hip_operators.h).  The harness wraps every entry of sim.pipeline in a wall-clock timer
(`timeops`).  Script: 10 initial steps (each one adapts the mesh: main.cpp:15314) untimed, then --steps timed steps; block lists at
the end must be identical.  Output: per-operator seconds of both runs, hot_path_fraction_cpu / _hip, end-to-end speed-up of a step,
PCIe bytes per step.  One rank: the mesh is ~10 000 blocks (5 M cells); it is the MEASUREMENT of what a CUP3D user gets from the
drop-in on the fish case, not a scaling run (tests/test_gpu_00_dropin_mpi.py covers 8 ranks at levelMax 5)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

TWO_FISH = ("StefanFish L=0.4 T=1.0 xpos=0.25 ypos=0.5 zpos=0.5 planarAngle=180 heightProfile=danio widthProfile=stefan bFixFrameOfRef=1\n"
            "StefanFish L=0.4 T=1.0 xpos=0.7 ypos=0.5 zpos=0.5 heightProfile=danio widthProfile=stefan")
HOT = ("AdvectionDiffusion", "PressureProjection")   # the two operators of the hot path; everything else stays on the CPU


def run(tool, pre, args, threads, warm, steps, timeout):
    wd = tempfile.mkdtemp(prefix="cup3d_c4_")
    script = pre + [f"op steps {warm}", "timeops", f"op steps {steps}", "tables t.bin", "dump vel v.bin", "dump pres p.bin"]
    # `timeops` after the warm-up steps: the first `op steps` prints nothing per operator, the second one the split of the timed steps
    open(os.path.join(wd, "script.txt"), "w").write("\n".join(script) + "\n")
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    t0 = time.time()
    out = subprocess.run([tool, "script.txt", "--"] + args, cwd=wd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if out.returncode != 0:
        sys.exit(f"{tool} failed: {out.stderr.decode()[-2000:]}")
    ops, steps_rec = {}, []
    for line in out.stdout.decode().splitlines():
        p = line.split()
        if line.startswith("REF optime"):
            kv = dict(x.split("=") for x in p[2:])
            ops[kv["name"]] = {"calls": int(kv["calls"]), "seconds": float(kv["seconds"])}
        elif line.startswith("REF steps"):
            kv = dict(x.split("=") for x in p[2:])
            steps_rec.append({"seconds": float(kv["seconds"]), "iters": int(float(kv["iters"]))})
    t = O.read_tables(os.path.join(wd, "t.bin"))[0]
    nb = len(t)
    return {"ops": ops, "steps": steps_rec, "wall": time.time() - t0, "tables": t, "vel": O.read_blocks(os.path.join(wd, "v.bin"), nb, 3),
            "pres": O.read_blocks(os.path.join(wd, "p.bin"), nb, 1)}


def split(r, nsteps):
    ops = {k: v["seconds"] for k, v in r["ops"].items() if k not in ("blocks", "pcie_MB_up", "pcie_MB_down", "bicgstab_iterations")}
    total = sum(ops.values())
    hot = sum(v for k, v in ops.items() if any(h in k for h in HOT))   # (the drop-in's classes are cup3d_hip::<name>HIP)
    return {"seconds_per_step": round(total / nsteps, 5), "hot_path_seconds_per_step": round(hot / nsteps, 5), "hot_path_fraction": round(hot / total, 4),
            "per_operator_seconds_per_step": {k: round(v / nsteps, 5) for k, v in sorted(ops.items(), key=lambda kv: -kv[1])}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level-max", type=int, default=7)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warm", type=int, default=10, help="untimed steps first (every one of the first ten adapts the mesh)")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--cpu-only", action="store_true", help="the reference run only (no GPU needed)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04", "configs4_1024_effective.json"))
    a = ap.parse_args()
    args = ["-bMeanConstraint", "2", "-bpdx", "2", "-bpdy", "2", "-bpdz", "2", "-CFL", "0.4", "-Ctol", "0.1", "-extentx", "1", "-levelStart", "1", "-nu", "0.001",
            "-poissonSolver", "iterative", "-Rtol", "5", "-tdump", "0", "-tend", "0", "-factory", "", "-levelMax", str(a.level_max), "-factory-content", TWO_FISH]
    cpu = run(O.REF_TOOL, [], args, a.threads, a.warm, a.steps, 7200)
    t = cpu["tables"]
    lv, cnt = np.unique(t[:, 0], return_counts=True)
    rec = {"config": "BASELINE configs[4]: two-fish school (the factory of the reference's run.sh), bpd 2, levelMax %d = %d^3-effective, one rank"
                     % (a.level_max, 16 << (a.level_max - 1)),
           "reference_args": " ".join(args[:-1]) + " '<two StefanFish>'", "blocks": int(len(t)), "cells": int(len(t)) * 512,
           "blocks_per_level": {int(l): int(c) for l, c in zip(lv, cnt)}, "finest_uniform_equivalent_cells": (16 << (a.level_max - 1)) ** 3,
           "warmup_steps": a.warm, "timed_steps": a.steps, "host_threads": a.threads, "host_cores_available": os.cpu_count(),
           "cpu": split(cpu, a.steps)}
    # (the harness prints the number of 7-double reductions = BiCGSTAB iterations of each `op`: the last record is the timed steps')
    rec["cpu"]["bicgstab_iters_per_step"] = round(cpu["steps"][-1]["iters"] / a.steps, 1) if cpu["steps"] else None
    rec["hot_path_fraction_cpu"] = rec["cpu"]["hot_path_fraction"]
    if not a.cpu_only:
        tool = os.path.join(O.ORACLE_DIR, "_ref", "ref_tool_hip")
        hip = run(tool, ["hip resident3"], args, a.threads, a.warm, a.steps, 7200)
        rec["hip"] = split(hip, a.steps)
        rec["hot_path_fraction_hip"] = rec["hip"]["hot_path_fraction"]
        rec["end_to_end_speedup"] = round(rec["cpu"]["seconds_per_step"] / rec["hip"]["seconds_per_step"], 3)
        rec["hot_path_speedup"] = round(rec["cpu"]["hot_path_seconds_per_step"] / max(1e-9, rec["hip"]["hot_path_seconds_per_step"]), 2)
        rec["amdahl_limit_of_this_split"] = round(1.0 / (1.0 - rec["hot_path_fraction_cpu"]), 2)
        rec["pcie_MB_per_step"] = {"up": round(hip["ops"].get("pcie_MB_up", {"calls": 0})["calls"] / a.steps, 2),
                                   "down": round(hip["ops"].get("pcie_MB_down", {"calls": 0})["calls"] / a.steps, 2)}
        rec["hip"]["bicgstab_iters_per_step"] = round(hip["ops"].get("bicgstab_iterations", {"calls": 0})["calls"] / a.steps, 1)
        same = np.array_equal(cpu["tables"], hip["tables"])
        rec["block_lists_identical"] = bool(same)
        if same:
            vmax, pmax = float(np.abs(cpu["vel"]).max()), float(np.abs(cpu["pres"]).max())
            rec["max_abs_dvel"] = float(np.abs(cpu["vel"] - hip["vel"]).max())
            rec["max_abs_dpres"] = float(np.abs(cpu["pres"] - hip["pres"]).max())
            rec["max_abs_vel"], rec["max_abs_pres"] = vmax, pmax
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
