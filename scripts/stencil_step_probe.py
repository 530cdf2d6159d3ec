#!/usr/bin/env python3
"""Where does a step of BASELINE configs[1] (256^3 periodic Taylor-Green, advect-diffuse only) spend its wall clock?
bench.py --stencil-only --size 256 measured 1.75 ms per step with 1.12 ms of kernels (profiles/r06): this probe times, over N
repetitions each, (a) AdvectionDiffusion(dt) alone with a fixed dt (nothing returns to the host), (b) findMaxU alone (kernel +
8-byte copy + host wait), (c) the step as bench.py runs it (calcMaxTimestep + AdvectionDiffusion), with and without the per-kernel
hipEvents of the profiler.  One JSON line per case.

    python scripts/stencil_step_probe.py [--size 256] [--reps 200]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import cup3d_amd as cu  # noqa: E402
from cup3d_amd.capi import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    cu.device_init(0)
    nb1 = a.size // 8
    level = (nb1 & -nb1).bit_length() - 1
    bpd = nb1 >> level
    ext = 2 * np.pi
    sim = cu.SimulationData(bpdx=bpd, bpdy=bpd, bpdz=bpd, levelMax=level + 1, levelStart=level, extent=ext, nu=0.01, CFL=0.3, BC_x="periodic", BC_y="periodic",
                            BC_z="periodic", uMax_forced=1.0, rampup=0)
    sim.upload("vel", bench.taylor_green_blocks(sim.grid, [ext] * 3, 1.0))
    sim.step = 21
    S = cu.Simulation(sim)
    adv = S.pipeline[0]
    dt = S.calcMaxTimestep()
    sync = lib().cup3d_device_synchronize

    def timed(name, fn, profile):
        lib().cup3d_profile_enable(1 if profile else 0)
        lib().cup3d_profile_reset()
        for _ in range(10):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        sync()
        sec = (time.perf_counter() - t0) / a.reps
        cells = float(a.size) ** 3
        rec = {"case": name, "size": a.size, "per_kernel_events": bool(profile), "ms_per_call": round(sec * 1e3, 4), "Mcell_per_s": round(cells / sec / 1e6, 1),
               "frac_of_8TBs_at_264_B_per_cell": round(264.0 * cells / sec / 8e12, 4)}
        if profile:
            p = bench.read_profile()
            rec["kernel_ms_per_call"] = round(sum(ms for _, ms in p.values()) / a.reps, 4)
        print(json.dumps(rec))
        lib().cup3d_profile_enable(0)

    def step():
        d = S.calcMaxTimestep()
        adv(d)
        sim.step += 1

    for prof in (True, False):
        timed("AdvectionDiffusion(dt) alone, fixed dt", lambda: adv(dt), prof)
        timed("findMaxU alone", lambda: cu.findMaxU(sim), prof)
        timed("calcMaxTimestep + AdvectionDiffusion (bench.py --stencil-only)", step, prof)


if __name__ == "__main__":
    main()
