#!/usr/bin/env python3
"""Kernel A/B and ablation timing on one GPU, without torch (fast start).

    python scripts/kernel_probe.py adv  --size 512 [--variants 0,1,2,3,10]   advect-diffuse variants
    python scripts/kernel_probe.py pre  --size 256                          block-CG variants on solver-like input
    python scripts/kernel_probe.py one  --size 512 --kernel adv|lhs|pre|loop  a few launches of one kernel (rocprof --pmc target)
Times come from the library's own hipEvent profile (cup3d_profile_*)."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CUP3D_HIP_FLAVOUR", "testing")  # the A/B switches (cup3d_debug_set_option) exist in libcup3d_hip_testing.so only
import cup3d_amd as cu  # noqa: E402
from bench import taylor_green_blocks  # noqa: E402
from cup3d_amd.capi import ProfileEntry, check, lib  # noqa: E402


def profile():
    ents = (ProfileEntry * 64)()
    n = C.c_int(0)
    lib().cup3d_profile_read(ents, 64, C.byref(n))
    return {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n.value)}


def make(size, bc="periodic"):
    level = int(round(np.log2(size // 8)))
    sim = cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=level + 1, levelStart=level, extent=2 * np.pi, nu=0.01, CFL=0.3,
                            BC_x=bc, BC_y=bc, BC_z=bc, uMax_forced=1.0, rampup=0)
    sim.upload("vel", taylor_green_blocks(sim.grid, [2 * np.pi] * 3, 1.0))
    return sim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["adv", "pre", "one", "loops", "pcie", "cgvar", "lhs", "fusedwaves"])
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--variants", default="0,1,2,3")
    ap.add_argument("--kernel", default="adv")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--no-fuse", action="store_true")
    a = ap.parse_args()
    cu.device_init(0)
    if a.no_fuse:
        check(lib().cup3d_debug_set_option(b"no_fuse", 1))
    lib().cup3d_profile_enable(1)
    cells = float(a.size) ** 3
    if a.what == "adv":
        sim = make(a.size)
        adv = cu.AdvectionDiffusion(sim)
        dt = 0.3 * sim.grid.h
        for v in [int(x) for x in a.variants.split(",")]:
            check(lib().cup3d_debug_set_option(b"advdiff_variant", v))
            adv(dt)
            lib().cup3d_device_synchronize()
            lib().cup3d_profile_reset()
            for _ in range(a.reps):
                adv(dt)
            n, ms = profile()["advdiff_stage"]
            avg = ms / n
            print(json.dumps({"probe": "advdiff", "size": a.size, "variant": v, "avg_ms": round(avg, 4),
                              "GBps_algorithmic": round(96 * cells / avg / 1e6, 1), "frac_8TBs": round(96 * cells / avg / 1e6 / 8000, 4)}))
        check(lib().cup3d_debug_set_option(b"advdiff_variant", 0))
    elif a.what == "lhs":
        # KernelLHSPoisson alone (no mean constraint); variant 1 (TUNING builds only, wrong results): no x-face ghost gathers
        sim = make(a.size, "wall")
        sim.upload("pres", np.random.default_rng(1).uniform(-1, 1, (sim.nblocks, 8, 8, 8)))
        for v in [int(x) for x in a.variants.split(",")]:
            if lib().cup3d_debug_set_option(b"lhs_variant", v) != 0:
                continue
            check(lib().cup3d_compute_lhs(sim.handle, 0))
            lib().cup3d_device_synchronize()
            lib().cup3d_profile_reset()
            for _ in range(a.reps * 5):
                check(lib().cup3d_compute_lhs(sim.handle, 0))
            n, ms = profile()["poisson_lhs"]
            avg = ms / n
            print(json.dumps({"probe": "poisson_lhs", "size": a.size, "variant": v, "avg_ms": round(avg, 4), "GBps_algorithmic": round(16 * cells / avg / 1e6, 1),
                              "frac_8TBs": round(16 * cells / avg / 1e6 / 8000, 4)}))
        lib().cup3d_debug_set_option(b"lhs_variant", 0)
    elif a.what == "fusedwaves":
        # occupancy A/B of the second fused solver kernel (debug option "loop2_four_waves": variant 1 = 122 registers, 4 waves / SIMD;
        # 0 = production, 96 registers, 5 waves): one pressure projection per variant, same input
        sim = make(a.size, "wall")
        sim.step = 21
        dt = 0.3 * sim.grid.h
        vel0 = sim.download("vel")
        for v in [int(x) for x in a.variants.split(",")]:
            check(lib().cup3d_debug_set_option(b"loop2_four_waves", v))
            sim.upload("vel", vel0); sim.fill("pres", 0.0)
            lib().cup3d_device_synchronize()
            lib().cup3d_profile_reset()
            r = cu.PressureProjection(sim)(dt)
            p = profile()
            print(json.dumps({"probe": "loop2_four_waves", "size": a.size, "variant": v, "iterations": r.iterations,
                              "loop1_cg_ms": round(p["bicgstab_loop1_cg"][1] / p["bicgstab_loop1_cg"][0], 4),
                              "loop2_cg_ms": round(p["bicgstab_loop2_cg"][1] / p["bicgstab_loop2_cg"][0], 4)}))
        check(lib().cup3d_debug_set_option(b"loop2_four_waves", 0))
    elif a.what == "pcie":
        # host <-> device rate of the boundary's block transfers (reference layout AoS on the host, SoA slab on the device)
        import time
        sim = make(a.size)
        vel = sim.download("vel")
        out = np.empty_like(vel)          # reused destination: the reference's blocks are resident pages, not fresh allocations
        fid = cu.operators.FIELDS["vel"]
        ptr_in = (C.c_void_p * sim.nblocks)(*[vel[i].ctypes.data for i in range(sim.nblocks)])   # one pointer per block, like Info::block
        ptr_out = (C.c_void_p * sim.nblocks)(*[out[i].ctypes.data for i in range(sim.nblocks)])
        cases = (("upload_vel_block_pointers", lambda: check(lib().cup3d_sim_upload_blocks(sim.handle, fid, ptr_in))),
                 ("download_vel_block_pointers", lambda: check(lib().cup3d_sim_download_blocks(sim.handle, fid, ptr_out))),
                 ("upload_vel_contiguous", lambda: check(lib().cup3d_sim_upload(sim.handle, fid, vel))),
                 ("download_vel_contiguous", lambda: check(lib().cup3d_sim_download(sim.handle, fid, out))))
        for name, fn in cases:
            fn()
            lib().cup3d_device_synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                fn()
            lib().cup3d_device_synchronize()
            sec = (time.perf_counter() - t0) / a.reps
            print(json.dumps({"probe": name, "size": a.size, "GB": round(vel.nbytes / 1e9, 3), "seconds": round(sec, 4), "GBps": round(vel.nbytes / sec / 1e9, 2)}))
        assert np.array_equal(out, vel)
    elif a.what == "loops":
        # grid size of the fused BiCGSTAB vector loops (cup3d_debug_set_option "vec_groups")
        sim = make(a.size, "wall")
        sim.step = 21
        dt = 0.3 * sim.grid.h
        vel0 = sim.download("vel")
        for vg in [int(x) for x in a.variants.split(",")]:
            nt = 1 if vg < 0 else 0   # negative grid size: nontemporal variant of the two fused loops
            check(lib().cup3d_debug_set_option(b"loops_no_nt", 0 if nt else 1))
            vg = abs(vg)
            check(lib().cup3d_debug_set_option(b"vec_groups", vg))
            sim.upload("vel", vel0); sim.fill("pres", 0.0)
            lib().cup3d_device_synchronize()
            lib().cup3d_profile_reset()
            r = cu.PressureProjection(sim)(dt)
            p = profile()
            print(json.dumps({"probe": "bicgstab_loops", "size": a.size, "vec_groups": vg, "nt": nt, "iterations": r.iterations,
                              "loop1_ms": round(p["bicgstab_loop1"][1] / p["bicgstab_loop1"][0], 4),
                              "loop2_ms": round(p["bicgstab_loop2"][1] / p["bicgstab_loop2"][0], 4)}))
        check(lib().cup3d_debug_set_option(b"vec_groups", 0))
        check(lib().cup3d_debug_set_option(b"loops_no_nt", 0))
    elif a.what == "cgvar":
        # the eight evaluations of the production block CG (bit 0: wave sums on the matrix pipe, 1: single-width LDS reads,
        # 2: reciprocal divisions) on solver-like input, against the reference-association kernel (block_solver 2)
        sim = make(a.size, "wall")
        check(lib().cup3d_pressure_rhs(sim.handle, 0.3 * sim.grid.h))
        rhs = sim.download("lhs")
        sim.upload("pres", rhs)
        check(lib().cup3d_preconditioner(sim.handle, 2))
        ref = sim.download("pres")
        todo = [int(x) for x in a.variants.split(",")] if a.variants != "0,1,2,3" else list(range(16)) + ["pair"]
        for bits in todo:
            solver = 4 if bits == "pair" else 0
            if solver == 0:
                check(lib().cup3d_debug_set_option(b"cg_variant", 8 + bits))
            sim.upload("pres", rhs)
            check(lib().cup3d_preconditioner(sim.handle, solver))
            err = float(np.abs(sim.download("pres") - ref).max() / np.abs(ref).max())
            lib().cup3d_device_synchronize()
            lib().cup3d_profile_reset()
            for _ in range(a.reps):
                sim.upload("pres", rhs)
                check(lib().cup3d_preconditioner(sim.handle, solver))
            n, ms = profile()["poisson_block_cg"]
            rec = {"probe": "block_cg_variant", "size": a.size, "avg_ms": round(ms / n, 4), "max_rel_diff_vs_reference_association": err}
            rec.update({"kernel": "two blocks per wavefront"} if bits == "pair" else
                       {"bits": bits, "mfma_sums": bits & 1, "single_width_lds": (bits >> 1) & 1, "reciprocal_div": (bits >> 2) & 1, "fma3_p_update": (bits >> 3) & 1})
            print(json.dumps(rec))
        check(lib().cup3d_debug_set_option(b"cg_variant", 0))
    elif a.what == "pre":
        sim = make(a.size, "wall")
        # solver-like inputs: the pressure RHS of the initial field, then A M^-1 of it, then noise
        check(lib().cup3d_pressure_rhs(sim.handle, 0.3 * sim.grid.h))
        rhs = sim.download("lhs")
        rng = np.random.default_rng(0)
        inputs = {"rhs": rhs, "noise": rng.uniform(-1, 1, rhs.shape)}
        for name, x in inputs.items():
            for solver in (2, 0, 3, 1):
                fma = int(solver != 2)
                sim.upload("pres", x)
                check(lib().cup3d_preconditioner(sim.handle, solver))
                ref = sim.download("pres") if solver == 2 else ref  # reference point: uncontracted block CG
                err = float(np.abs(sim.download("pres") - ref).max() / np.abs(ref).max())
                lib().cup3d_device_synchronize()
                lib().cup3d_profile_reset()
                for _ in range(a.reps):
                    sim.upload("pres", x)
                    check(lib().cup3d_preconditioner(sim.handle, solver))
                key = "poisson_block_fdm" if solver == 1 else "poisson_block_cg"
                n, ms = profile()[key]
                print(json.dumps({"probe": key, "size": a.size, "input": name, "block_solver": solver, "fma": fma, "avg_ms": round(ms / n, 4),
                                  "max_rel_diff_vs_cg": err}))
    else:
        bc = "periodic" if a.kernel == "adv" else "wall"
        sim = make(a.size, bc)
        dt = 0.3 * sim.grid.h
        if a.kernel == "adv":
            for _ in range(a.reps):
                cu.AdvectionDiffusion(sim)(dt)
        else:
            sim.step = 21
            cu.PressureProjection(sim)(dt)
        lib().cup3d_device_synchronize()
        print(json.dumps({k: {"launches": v[0], "avg_ms": round(v[1] / max(1, v[0]), 4)} for k, v in profile().items()}))


if __name__ == "__main__":
    main()
