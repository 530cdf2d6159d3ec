#!/usr/bin/env python3
"""Does the face-slab halo exchange of the BiCGSTAB loop kernels hide behind the INNER blocks' pass?  (the reference's
order: main.cpp:5598-5618 -- inner blocks while the messages travel, halo blocks after; SynchronizerMPI_AMR::sync 2356-2405.)

One process, two THREAD ranks of the testing build (cup3d_debug_virtual_comm: the exchanges are device copies on each rank's
communication stream, handed to / from the compute stream by events exactly as the RCCL calls are), a 256 x 256 x 512 grid so that each
rank holds the 256^3 share it holds of the headline on 8 GPUs.  `halo_delay_us` (comm.hip, testing build) holds the communication stream
for the given time behind every slab transfer; the solver runs a fixed number of iterations; the profiler's entries say what the
exchange took on its stream (comm_halo) and how long the compute stream waited for it in halo_finish (comm_exposed_halo_wait).
Both ranks share ONE device, so the absolute times are those of two ranks on half a GPU each -- the ratio exposed / injected is the result.

    python scripts/halo_overlap_probe.py [--level 5] [--delays 0,50,100,200,400,800] [--iters 40]   -> one JSON line per delay"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CUP3D_HIP_FLAVOUR", "testing")
import bench  # noqa: E402
import cup3d_amd as cu  # noqa: E402
from cup3d_amd.capi import check, lib  # noqa: E402


def run_ranks(fn, n):
    errs = [None] * n

    def work(r):
        try:
            fn(r)
        except BaseException as e:  # noqa: BLE001
            errs[r] = e
    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for e in errs:
        if e is not None:
            raise e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=5, help="blocks per side of the x / y directions = 2^level (5: 256 cells); z has twice as many")
    ap.add_argument("--delays", default="0,50,100,200,400,800")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--ranks", type=int, default=2)
    a = ap.parse_args()
    cu.device_init(0)
    n = a.ranks
    kw = dict(bpdx=1, bpdy=1, bpdz=2, levelMax=a.level + 1, levelStart=a.level, extent=2 * np.pi, BC_x="wall", BC_y="wall", BC_z="wall")
    check(lib().cup3d_debug_virtual_comm(n))
    try:
        sims = [cu.SimulationData(rank=r, nranks=n, **kw) for r in range(n)]
        rhs = []
        for s in sims:
            rng = np.random.default_rng(11 + s.grid.rank)
            rhs.append(rng.uniform(-1, 1, (s.nblocks, 8, 8, 8)))
        info = {"blocks_per_rank": [int(s.nblocks) for s in sims], "inner_blocks_per_rank": [int(lib().cup3d_grid_ninner(s.grid.handle)) for s in sims]}
        for delay in [int(x) for x in a.delays.split(",")]:
            check(lib().cup3d_debug_set_option(b"halo_delay_us", delay))
            res = {}
            wall = {}

            def rank(r, timed):
                s = sims[r]
                s.upload("lhs", rhs[r])
                s.fill("pres", 0.0)
                p = s.poisson_params()
                p.max_iter = a.iters + 1          # iteration 0 is the host-driven one (k % 50 == 0); the rest are fused
                out = cu.capi.PoissonResult()
                lib().cup3d_device_synchronize()
                t0 = time.perf_counter()
                check(lib().cup3d_poisson_solve(s.handle, C.byref(p), C.byref(out)))
                lib().cup3d_device_synchronize()
                wall[r] = time.perf_counter() - t0
                res[r] = out.iterations

            run_ranks(lambda r: rank(r, False), n)       # warm-up (allocations, plans)
            lib().cup3d_profile_enable(1)
            lib().cup3d_profile_reset()
            run_ranks(lambda r: rank(r, True), n)
            prof = bench.read_profile()
            lib().cup3d_profile_enable(0)
            its = res[0]
            per = lambda name: round(prof.get(name, (0, 0.0))[1] / max(1, its) / n, 5)    # ms per iteration and rank
            exch = prof.get("comm_halo", (0, 0.0))[0] / max(1, its) / n
            rec = {"probe": "halo_overlap", "ranks": n, "cells_per_rank": int(sims[0].nblocks) * 512, **info, "injected_delay_us_per_exchange": delay,
                   "bicgstab_iterations": its, "exchanges_per_iteration": round(exch, 2),
                   "wall_ms_per_iteration": round(max(wall.values()) * 1e3 / max(1, its), 4),
                   "comm_halo_ms_per_iteration (pack + transfer + injected delay, on the communication stream)": per("comm_halo"),
                   "exposed_halo_wait_ms_per_iteration (compute stream blocked in halo_finish)": per("comm_exposed_halo_wait"),
                   "injected_ms_per_iteration": round(delay * 1e-3 * exch, 5),
                   "loop1_ms_per_launch": round(prof.get("bicgstab_loop1_cg", (1, 0.0))[1] / max(1, prof.get("bicgstab_loop1_cg", (1, 0.0))[0]), 4),
                   "loop2_ms_per_launch": round(prof.get("bicgstab_loop2_cg", (1, 0.0))[1] / max(1, prof.get("bicgstab_loop2_cg", (1, 0.0))[0]), 4),
                   "loop_launches_per_iteration_and_rank": round((prof.get("bicgstab_loop1_cg", (0, 0))[0] + prof.get("bicgstab_loop2_cg", (0, 0))[0]) / max(1, its) / n, 2)}
            inj = rec["injected_ms_per_iteration"]
            rec["exposed_over_injected"] = round(rec["exposed_halo_wait_ms_per_iteration (compute stream blocked in halo_finish)"] / inj, 3) if inj else None
            print(json.dumps(rec))
            sys.stdout.flush()
        check(lib().cup3d_debug_set_option(b"halo_delay_us", 0))
        del sims
    finally:
        lib().cup3d_device_synchronize()
        lib().cup3d_debug_virtual_comm(0)


if __name__ == "__main__":
    main()
