#!/usr/bin/env python3
"""Minimal reproducer of round 5's exit-time `double free or corruption`: which combination of
{the library's RCCL communicator, torch's ROCm runtime} in one process aborts at exit?

    python scripts/exit_repro.py <order>      order: comma-separated list of steps, executed left to right
        rccl     cup3d_comm_unique_id + cup3d_comm_init(0, 1) + one all-reduce (findMaxU) + cup3d_comm_finalize
        torch    import torch; torch.cuda.synchronize(); torch.cuda.mem_get_info()
        sim      one small simulation life (create, upload, one step, destroy) with profiling on (the event pool gets used)
        rcclkeep like rccl but WITHOUT cup3d_comm_finalize (the communicator is alive at exit)

The exit code of the process is the evidence (134 = glibc abort); the librccl / libamdhip64 copies mapped at the end go to stderr."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CUP3D_HIP_FLAVOUR", "testing")
import numpy as np  # noqa: E402

import cup3d_amd as cu  # noqa: E402


def step_rccl(keep=False):
    raw = (C.c_ubyte * 128)()
    cu.capi.check(cu.lib().cup3d_debug_set_option(b"force_allreduce", 1))
    cu.capi.check(cu.lib().cup3d_comm_unique_id(raw))
    cu.capi.check(cu.lib().cup3d_comm_init(0, 1, raw))
    sim = cu.SimulationData(bpdx=2, bpdy=2, bpdz=2, levelMax=1, extent=1.0, BC_x="periodic", BC_y="periodic", BC_z="periodic")
    v = np.random.default_rng(2).uniform(-1, 1, (sim.nblocks, 8, 8, 8, 3))
    sim.upload("vel", v)
    assert cu.findMaxU(sim) == np.abs(v).max()
    del sim
    if not keep:
        cu.lib().cup3d_comm_finalize()
    cu.capi.check(cu.lib().cup3d_debug_set_option(b"force_allreduce", 0))


def step_torch():
    import torch
    torch.cuda.synchronize()
    torch.cuda.mem_get_info()


def step_sim():
    cu.lib().cup3d_profile_enable(1)
    sim = cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=4, levelStart=3, extent=2 * np.pi, nu=0.01, BC_x="wall", BC_y="periodic", BC_z="freespace")
    sim.upload("vel", np.random.default_rng(7).uniform(-1, 1, (sim.grid.nblocks, 8, 8, 8, 3)))
    S = cu.Simulation(sim)
    S.sim.step = 5
    S.advance(0.01)
    cu.lib().cup3d_profile_enable(0)


def main():
    cu.device_init(0)
    for st in sys.argv[1].split(","):
        {"rccl": step_rccl, "rcclkeep": lambda: step_rccl(True), "torch": step_torch, "sim": step_sim}[st]()
        sys.stderr.write(f"exit_repro: step {st} done\n")
    maps = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if any(k in ln for k in ("rccl", "amdhip64", "hsa-runtime"))})
    sys.stderr.write("exit_repro: mapped: " + ", ".join(maps) + "\n")


if __name__ == "__main__":
    main()
