#!/usr/bin/env python3
"""bench.py — whole-step throughput of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 512]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher environment re-executes itself under torch.distributed.run (one process per GPU,
127.0.0.1 rendezvous) and still prints ONE JSON line; with fewer than N visible devices it says so and exits 2.

One "step" = one reference time step of the hot path on a synthetic uniform grid:
calcMaxTimestep (findMaxU) + AdvectionDiffusion (fused RK3) + ExternalForcing +
PressureProjection (pressure RHS, pipelined BiCGSTAB with block-CG preconditioner,
mean removal, gradient update) — main.cpp:15254-15326 with the obstacle-free pipeline of
15229-15246.  Workload (BASELINE.json configs[2] as far as the reference can express it,
SURVEY §0 F3): Taylor-Green initial condition on an all-`wall` box, 2*pi extent, nu=0.01,
CFL 0.3, -rampup 0, default Poisson tolerances, steps numbered from 21 (2nd-order pressure
path, no adaptMesh step).  Fields are resident in HBM before the timed region.

Prints ONE JSON line (rank 0): metric Mcell-updates/s = cells * steps / seconds over all
ranks (strong scaling: the grid is fixed, blocks are sharded by Hilbert ranges), plus
`roofline` (dominant kernel by device time, algorithmic bytes / measured kernel time vs
8 TB/s) and `cpu_baseline` (the compiled reference on this host's cores, bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# FP64 vector peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop (FMA) x 2.4 GHz = 78.6 TFLOP/s (= half the guide's 157.3 TF FP32 vector
# figure; AMD's public MI355X FP64-vector number).  A wave64 FP64 instruction occupies its SIMD for 4 cycles, FMA or not.
FP64_PEAK_TFLOPS = 78.6
# block CG, algorithmic FP64 work per cell per CG iteration (kernelPoissonGetZInner + the updates, main.cpp:14662-14699):
# stencil 1 FMA + 5 adds, p.Ap 1 FMA, x 1 FMA, r 1 FMA, r.r 1 FMA, p 1 FMA  =  6 FMA + 5 add = 17 flop in 11 instructions
BLOCK_CG_FLOPS_PER_CELL_ITER = 17.0
# algorithmic HBM bytes per cell per launch, SURVEY.md §8(d)
ALGO_BYTES = {
    "advdiff_stage": 96.0,     # RK stages 2, 3: vel 24 in + tmpV 24 in + vel' 24 out + tmpV 24 out
    "advdiff_stage1": 72.0,    # RK stage 1 reads no tmpV (it is zero there)
    "bicgstab_loop1_cg": 152.0,  # loop 1 fused with the block CG on z: 11 reads + 7 writes + zhat out (z never re-read)
    "bicgstab_loop2_cg": 136.0,  # loop 2 fused with the block CG on w: 12 reads + 4 writes + what out
    "bicgstab_loop1": 144.0,   # 11 reads + 7 writes
    "bicgstab_loop2": 128.0,   # 12 reads + 4 writes
    "poisson_lhs": 16.0,       # p in, Ap out
    "poisson_block_cg": 16.0,  # r in, z out (FP64-VALU bound, shown against HBM for reference)
    "poisson_block_fdm": 16.0,  # r in, z out (direct block solve)
    # --implicit-diffusion (AdvectionDiffusionImplicit, main.cpp:10030-10119)
    "advect_implicit": 72.0,   # vel 24 in + vel' 24 out + tmpV 24 out
    "diffusion_rhs": 48.0,     # vel 24 in + tmpV 24 out
    "diffusion_lhs": 16.0,
    "diffusion_block_cg": 16.0,
}


def taylor_green_blocks(grid, ext, umax):
    """KernelIC_taylorGreen (main.cpp:12516-12539) for this rank's blocks, block order."""
    h = grid.h
    idx = grid.index.astype(np.float64)
    cell = np.arange(8) + 0.5
    px = idx[:, 0:1] * 8 * h + h * cell[None, :]
    py = idx[:, 1:2] * 8 * h + h * cell[None, :]
    pz = idx[:, 2:3] * 8 * h + h * cell[None, :]
    a, b, c = 2 * np.pi / ext[0], 2 * np.pi / ext[1], 2 * np.pi / ext[2]
    A, B = umax, -umax * ext[1] / ext[0]
    vel = np.zeros((grid.nblocks, 8, 8, 8, 3))
    sz = np.sin(c * pz)[:, :, None, None]
    vel[..., 0] = (A * np.cos(a * px))[:, None, None, :] * np.sin(b * py)[:, None, :, None] * sz
    vel[..., 1] = (B * np.sin(a * px))[:, None, None, :] * np.cos(b * py)[:, None, :, None] * sz
    return vel


def exact_test_field_blocks(grid, size):
    """Velocity field of the partition-independent checksum (config.checksum): polynomials of the cell-centre coordinates evaluated
    with correctly rounded +, -, *, / only -- no libm, so the input bits are the same on every machine and the constant in
    tests/golden/advdiff_checksums.json (the CPU oracle's result, tests/golden/make_checksums.py) is portable.  Both signs of every
    component occur (both upwind branches of KernelAdvectDiffuse, main.cpp:9474-9483); zero at no wall in particular."""
    n = float(size)
    idx = grid.index.astype(np.float64)
    cell = np.arange(8) + 0.5
    x = ((idx[:, 0:1] * 8 + cell[None, :]) / n)[:, None, None, :]
    y = ((idx[:, 1:2] * 8 + cell[None, :]) / n)[:, None, :, None]
    z = ((idx[:, 2:3] * 8 + cell[None, :]) / n)[:, :, None, None]
    vel = np.empty((grid.nblocks, 8, 8, 8, 3))
    vel[..., 0] = ((4.0 * x) * (1.0 - x)) * (y - 0.5) * (0.25 + z * z)
    vel[..., 1] = (0.5 - x) * ((4.0 * y) * (1.0 - y)) * (z + 0.125)
    vel[..., 2] = (x * y - 0.25) * ((4.0 * z) * (1.0 - z))
    return vel


def checksum_dt(size):
    return 0.3 * (2 * np.pi / size)


def advdiff_checksums(sim, a, dist, world):
    """config.checksum: one AdvectionDiffusion::operator() (fixed dt = 0.3 h, nu = 0.01) applied to (a) the exact test field, (b) the
    Taylor-Green field of the run; the wrapping 64-bit sums of the bit patterns of `vel` afterwards, added over the ranks mod 2^64.
    The stencil path is bit-exact with the CPU oracle under any sharding of the blocks, so both values must equal the oracle's
    constants at every N (the reference's own multi-rank run equals its one-rank run the same way: tests/test_oracle_vs_ref.py)."""
    import torch
    import cup3d_amd as cu
    ext = 2 * np.pi
    golden = os.path.join(ROOT, "tests", "golden", "advdiff_checksums.json")
    expected = json.load(open(golden)).get(str(a.size), {}) if os.path.exists(golden) else {}
    out = {"what": "wrapping uint64 sum of the bit patterns of vel after ONE AdvectionDiffusion (dt = 0.3 h, nu = 0.01) on this workload's grid, "
                   "summed over ranks mod 2^64; `expected` = the CPU oracle's value (tests/golden/advdiff_checksums.json)", "dt": checksum_dt(a.size)}
    adv = cu.AdvectionDiffusion(sim)
    nu0, sim.nu = sim.nu, 0.01
    for key, field in (("exact_field", exact_test_field_blocks(sim.grid, a.size)), ("taylor_green", taylor_green_blocks(sim.grid, [ext] * 3, 1.0))):
        sim.upload("vel", field)
        adv(checksum_dt(a.size))
        mine = sim.checksum("vel")
        if dist is not None:
            parts = [torch.zeros(1, dtype=torch.int64, device=a.tdev) for _ in range(world)]
            dist.all_gather(parts, torch.tensor([mine - (1 << 64) if mine >= (1 << 63) else mine], dtype=torch.int64, device=a.tdev))
            mine = sum(int(p.item()) for p in parts) % (1 << 64)
        exp = expected.get(key)
        out[key] = {"value": mine, "expected": exp, "ok": (mine == exp) if exp is not None else None}
    sim.nu = nu0
    out["ok"] = all(out[k]["ok"] is not False for k in ("exact_field", "taylor_green")) and out["exact_field"]["ok"] is not None
    return out


def install_host_transport(dist, rank, world):
    """--transport host: the library's exchanges (face slabs, scalar all-reduces) staged through host memory and carried by
    torch.distributed's gloo backend -- cup3d_debug_host_transport of libcup3d_hip_testing.so (include/cup3d_hip_testing.h), the same
    stand-in for RCCL the MPI-rank tests of the C++ shim use.  Everything else is the production multi-process path of this file."""
    import torch
    from cup3d_amd.capi import check, lib
    EXCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long))
    ARED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)

    class Transport(C.Structure):
        _fields_ = [("ctx", C.c_void_p), ("exchange", EXCH), ("allreduce", ARED)]

    def exchange(ctx, sb, so, sn, rb, ro, rn):
        try:
            reqs, recvs = [], []
            for p in range(world):
                if rn[p]:
                    t = torch.empty(rn[p], dtype=torch.uint8)
                    recvs.append((t, ro[p]))
                    reqs.append(dist.irecv(t, src=p))
            for p in range(world):
                if sn[p]:
                    src = np.ctypeslib.as_array((C.c_ubyte * sn[p]).from_address(sb + so[p]))
                    reqs.append(dist.isend(torch.from_numpy(src.copy()), dst=p))
            for r in reqs:
                r.wait()
            for t, off in recvs:
                C.memmove(rb + off, t.numpy().ctypes.data, t.numel())
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            sys.stderr.write(f"bench: host transport exchange failed on rank {rank}: {e}\n")
            return 1

    def allreduce(ctx, buf, n, is_max):
        try:
            t = torch.tensor([buf[i] for i in range(n)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if is_max else dist.ReduceOp.SUM)
            for i in range(n):
                buf[i] = float(t[i])
            return 0
        except Exception as e:
            sys.stderr.write(f"bench: host transport all-reduce failed on rank {rank}: {e}\n")
            return 1

    tr = Transport(None, EXCH(exchange), ARED(allreduce))
    lib().cup3d_debug_host_transport.argtypes = [C.c_int, C.c_int, C.POINTER(Transport)]
    check(lib().cup3d_debug_host_transport(rank, world, C.byref(tr)))
    return tr


def relaunch_under_torchrun(n, need_devices=True):
    """`python bench.py --gpus N` (N > 1) as the driver types it: one process per GPU via torch.distributed.run, rendezvous on
    127.0.0.1; rank 0's JSON line is the only thing on stdout."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and need_devices:
        sys.stderr.write(f"bench.py --gpus {n} needs {n} devices; {have} visible on this host\n")
        sys.exit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def cpu_baseline(size_cpu, steps, threads):
    """Time the compiled reference (oracle/_ref/ref_tool = unmodified main.cpp) on the host."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    level = int(np.log2(size_cpu // 8))
    if O.have_ref_tool():
        args = O.ref_args((1, 1, 1), level + 1, level, 2 * np.pi, ("wall",) * 3, nu=0.01, cfl=0.3, extra=["-rampup", "0"])
        try:
            # `steps` separate steps of the reference's own time loop, each timed by the harness: the MEDIAN step is the sample
            recs, _ = O.run_ref(["zero chi", "set step 21", f"rep {steps}", "op steps 1"], args, threads=threads, timeout=600)
            rs = [r for r in recs if r["op"] == "steps"]
            secs, its = sorted(r["seconds"] for r in rs), [r["iters"] for r in rs]
            sec = secs[len(secs) // 2]
            return {"value": size_cpu ** 3 / sec / 1e6, "unit": "Mcell-updates/s", "cores": threads, "kind": "reference",
                    "sample": f"reference main.cpp operators, {size_cpu}^3 all-wall TGV, median of {len(rs)} steps from step 21 "
                              f"({', '.join('%.2f' % x for x in secs)} s), {np.mean(its):.1f} BiCGSTAB its/step",
                    "seconds_per_step": [round(x, 3) for x in secs], "bicgstab_iters_per_step": float(np.mean(its)), "size": size_cpu}
        except Exception as e:  # fall through to the port
            sys.stderr.write(f"bench: ref_tool failed ({e}); timing the oracle port instead\n")
    g = O.OracleGrid((1, 1, 1), level + 1, level, 2 * np.pi, ("wall",) * 3)
    vel = g.taylor_green([2 * np.pi] * 3, 1.0)
    pres = np.zeros((g.nb, 8, 8, 8))
    coef = np.array([1.5, -2.0, 0.5])
    dt, t0, its = 0.0, time.time(), 0
    for n in range(steps):
        dt = O.lib().orc_calc_dt(g.h, g.max_u(vel), 0.01, 0.3, 21 + n, 0, dt, coef)
        g.advect_diffuse(vel, np.zeros_like(vel), dt, 0.01)
        O.lib().orc_external_forcing(g.g, vel, 1.0, 0.01, 2 * np.pi, dt)
        info, _, _ = g.project(vel, pres, dt, 21 + n)
        its += info.iters
    sec = time.time() - t0
    return {"value": size_cpu ** 3 * steps / sec / 1e6, "unit": "Mcell-updates/s", "cores": threads, "kind": "port",
            "sample": f"oracle C port (block loops OpenMP, reductions serial), {size_cpu}^3 all-wall TGV, {steps} steps, "
                      f"{sec:.2f} s, {its / steps:.1f} BiCGSTAB its/step"}


def run_amr(a):
    """--amr: the same step on a multi-level mesh the device builds itself.  A compact vortex on a uniform level; a few passes of
    Simulation.adaptMesh (vorticity tags -> ValidStates -> refine/compress on the device) refine around it; then K steps on the
    frozen mesh are timed.  One GPU.  Reported beside the headline, not instead of it."""
    import cup3d_amd as cu
    from cup3d_amd.capi import ProfileEntry, lib
    cu.device_init(0)
    ext, lmax, lstart = 2 * np.pi, a.amr_levels + a.amr_base, a.amr_base
    sim = cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=lmax, levelStart=lstart, extent=ext, nu=0.002, CFL=0.3, BC_x="wall", BC_y="wall",
                            BC_z="wall", rampup=0, blockSolver=a.block_solver)
    g = sim.grid
    ax = np.arange(8) + 0.5
    X = (g.index[:, 0, None] * 8 + ax[None, :])[:, None, None, :] * g.h
    Y = (g.index[:, 1, None] * 8 + ax[None, :])[:, None, :, None] * g.h
    Z = (g.index[:, 2, None] * 8 + ax[None, :])[:, :, None, None] * g.h
    gss = np.exp(-((X - 2.6) ** 2 + (Y - 3.1) ** 2 + (Z - 3.4) ** 2) / 0.6)
    vel = np.stack([-(Y - 3.1) * gss, (X - 2.6) * gss, 0.3 * gss + 0 * X], axis=-1)
    sim.upload("vel", np.ascontiguousarray(vel))
    del vel, gss, X, Y, Z
    S = cu.Simulation(sim)
    t0 = time.perf_counter()
    history = [int(sim.nblocks)]
    for _ in range(a.amr_levels - 1):
        cu.ComputeVorticity(S.sim)(0)
        w = S.sim.download("tmpV")
        linf = np.sqrt((w ** 2).sum(axis=-1)).reshape(S.sim.nblocks, -1).max(axis=1)
        S.adaptMesh(float(np.quantile(linf, 1.0 - a.amr_fraction)), -1.0)   # refine the top fraction, never compress
        history.append(int(S.sim.nblocks))
    lib().cup3d_device_synchronize()
    adapt_s = time.perf_counter() - t0
    sim = S.sim
    sim.step = 21
    iters = []

    def one_step():
        S.advance(S.calcMaxTimestep())
        iters.append(sim.last_poisson.iterations)

    for _ in range(a.warmup):
        one_step()
    iters.clear()
    lib().cup3d_profile_enable(1)
    lib().cup3d_profile_reset()
    lib().cup3d_device_synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    lib().cup3d_device_synchronize()
    sec = time.perf_counter() - t0
    ents = (ProfileEntry * 64)()
    n = C.c_int(0)
    lib().cup3d_profile_read(ents, 64, C.byref(n))
    prof = {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n.value)}
    total_ms = sum(ms for _, ms in prof.values()) or 1.0
    cells = sim.nblocks * 512.0
    t = sim.grid.tables
    kernels = []
    for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        if not launches:
            continue
        e = {"kernel": name, "launches": launches, "avg_ms": round(ms / launches, 5), "share": round(ms / total_ms, 4)}
        if name in ALGO_BYTES:
            # the unfused AMR advect-diffuse stage reads vel and tmpV and writes tmpV only (72 B/cell); k_rk_update is its own entry
            bpc = 72.0 if name == "advdiff_stage" else ALGO_BYTES[name]
            ach = bpc * cells / (ms / launches * 1e-3) / 1e9
            e.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None})
        kernels.append(e)
    with_roof = [k for k in kernels if "achieved" in k]
    out = {"metric": "Mcell-updates/s (advect+diffuse+Poisson), multi-level AMR mesh", "value": round(cells * a.steps / sec / 1e6, 2),
           "unit": "Mcell-updates/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(sec / a.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"compact vortex in an all-wall box, uniform level {lstart} refined {a.amr_levels - 1}x around it by Simulation.adaptMesh "
                                  f"(top {a.amr_fraction:.0%} of the blocks by vorticity each pass), frozen mesh while timing",
                      "blocks": int(sim.nblocks), "cells": int(cells), "blocks_per_level": {int(l): int((t[:, 0] == l).sum()) for l in sorted(set(t[:, 0].tolist()))},
                      "interface_faces": int(lib().cup3d_grid_ninterface_faces(sim.grid.handle)), "block_history": history,
                      "finest_uniform_equivalent_cells": int((8 << (lmax - 1)) ** 3), "mesh_build_seconds": round(adapt_s, 3),
                      "bicgstab_iters_per_step": round(float(np.mean(iters)), 2),
                      "block_preconditioner": {0: "block CG (reference algorithm)", 1: "direct block solve (fast diagonalisation)",
                                               5: "geometric multigrid V-cycle over the octree's levels (NOT the reference's preconditioner)"}.get(a.block_solver, str(a.block_solver))},
           "roofline": ({k: with_roof[0][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} | {"kernel": with_roof[0]["kernel"]}) if with_roof else None,
           "kernels": kernels}
    print(json.dumps(out))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=512, help="cells per side (a multiple of 8; 512 = the BASELINE workload, 768 = the largest that leaves room on one 288 GB GPU)")
    ap.add_argument("--cpu-size", type=int, default=256, help="cells per side of the CPU-baseline sample (512 needs ~35 GB and minutes per step)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="steps of the CPU-baseline sample; the median step is reported")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="OpenMP threads of the reference; 32 is its best on the 256-thread GPU host (see report()); 0: all host cores")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--stencil-only", action="store_true", help="BASELINE configs[1]: periodic, advect-diffuse only")
    ap.add_argument("--block-solver", type=int, default=0, help="0: block CG as in the reference, 1: direct block solve, 5: multigrid V-cycle")
    ap.add_argument("--no-alt", action="store_true", help="skip the second timed region with the other block solver")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="what carries the library's exchanges over ranks.  rccl: production (one device per rank).  host: the library's "
                         "host-memory TEST transport over torch.distributed/gloo (libcup3d_hip_testing.so): lets --gpus N run on fewer than N "
                         "devices to check the multi-process path and config.checksum; its rate says nothing about scaling")
    ap.add_argument("--no-checksum", action="store_true", help="skip config.checksum (one extra AdvectionDiffusion on two fields before the timed region)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host<->device transfer measurement behind `pcie_inclusive`")
    ap.add_argument("--debug-option", action="append", help="name=value for cup3d_debug_set_option (tuning scans)")
    ap.add_argument("--no-profile", action="store_true", help="A/B: no per-kernel HIP events in the timed region (no roofline in the output)")
    ap.add_argument("--no-fuse", action="store_true", help="A/B: vector loops and block CG as separate launches (round-1 structure)")
    ap.add_argument("--implicit-diffusion", action="store_true",
                    help="-implicitDiffusion 1: AdvectionDiffusionImplicit (upwind advection + three Helmholtz solves) instead of the explicit RK3")
    ap.add_argument("--nu", type=float, default=0.01)
    ap.add_argument("--amr", action="store_true", help="time the step on a multi-level mesh built on the device (see run_amr)")
    ap.add_argument("--amr-base", type=int, default=4, help="--amr: uniform starting level (16^3 blocks at 4)")
    ap.add_argument("--amr-levels", type=int, default=3, help="--amr: number of levels of the final mesh")
    ap.add_argument("--amr-fraction", type=float, default=0.3, help="--amr: fraction of the blocks refined per pass")
    a = ap.parse_args()
    if a.no_fuse or a.debug_option or a.block_solver in (3, 4) or a.transport == "host":
        os.environ["CUP3D_HIP_FLAVOUR"] = "testing"   # A/B switches live in libcup3d_hip_testing.so only; everything else times the release build
    if a.amr:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            sys.exit("bench.py --amr runs on one GPU (multi-level meshes are single-rank this round)")
        return run_amr(a)

    # RCCL prints a version banner to STDOUT under NCCL_DEBUG=VERSION (the image's default), once per process and communicator
    # library: stdout carries the one JSON line of rank 0 and nothing else
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            return relaunch_under_torchrun(a.gpus, a.transport == "rccl")
        a.gpus = world

    import torch
    import cup3d_amd as cu
    from cup3d_amd.capi import ProfileEntry, RunStats, check, lib

    ndev = max(1, torch.cuda.device_count())
    if a.transport == "rccl" and local_rank >= ndev:
        sys.exit(f"bench.py: rank {rank} has no device (RCCL wants one per rank; {ndev} visible)")
    local_dev = local_rank % ndev   # host transport: ranks may share a device
    torch.cuda.set_device(local_dev)
    cu.device_init(local_dev)
    if a.no_fuse:
        check(lib().cup3d_debug_set_option(b"no_fuse", 1))
    for opt in (a.debug_option or []):   # tuning scans: --debug-option name=value (cup3d_debug_set_option)
        name, val = opt.split("=")
        check(lib().cup3d_debug_set_option(name.encode(), int(val)))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.transport == "host":
            dist.init_process_group(backend="gloo")
            a.host_transport = install_host_transport(dist, rank, world)   # keeps the ctypes callbacks alive
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            # bootstrap the library's own RCCL communicator with rank 0's unique id
            idbuf = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                raw = (C.c_ubyte * 128)()
                check(lib().cup3d_comm_unique_id(raw))
                idbuf = torch.tensor(list(raw), dtype=torch.uint8)
            idbuf = idbuf.cuda()
            dist.broadcast(idbuf, 0)
            raw = (C.c_ubyte * 128)(*idbuf.cpu().tolist())
            check(lib().cup3d_comm_init(rank, world, raw))
    a.tdev = "cpu" if a.transport == "host" else "cuda"   # where torch.distributed's own tensors live (gloo / nccl)

    nb1 = a.size // 8
    assert nb1 >= 2 and 8 * nb1 == a.size, "--size must be a multiple of 8"
    level = (nb1 & -nb1).bit_length() - 1  # blocks per side = bpd * 2^level with the smallest base grid (512: 1 x 2^6; 768: 3 x 2^5)
    bpd = nb1 >> level
    ext = 2 * np.pi
    bc = "periodic" if a.stencil_only else "wall"
    sim = cu.SimulationData(bpdx=bpd, bpdy=bpd, bpdz=bpd, levelMax=level + 1, levelStart=level, extent=ext, nu=a.nu, CFL=0.3,
                            BC_x=bc, BC_y=bc, BC_z=bc, uMax_forced=1.0, rampup=0, rank=rank, nranks=world, blockSolver=a.block_solver,
                            implicitDiffusion=a.implicit_diffusion)
    a.checksum = None
    if not a.stencil_only and not a.implicit_diffusion and not a.no_checksum:
        a.checksum = advdiff_checksums(sim, a, dist, world)  # the run's correctness signal at every N (before anything is timed)
        sim.dt = 0.0
    sim.upload("vel", taylor_green_blocks(sim.grid, [ext] * 3, 1.0))
    sim.step = 21
    S = cu.Simulation(sim)
    adv = S.pipeline[0]  # AdvectionDiffusion, or AdvectionDiffusionImplicit with --implicit-diffusion
    iters, diff_iters = [], []

    umax = []

    def one_step():
        dt = S.calcMaxTimestep()
        umax.append(float(sim.uMax_measured))   # findMaxU of the state this step starts from (all-reduced MAX: the same on every rank)
        if a.stencil_only:
            adv(dt)
            sim.step += 1
        else:
            S.advance(dt)
            iters.append(sim.last_poisson.iterations)
        if a.implicit_diffusion:
            diff_iters.append(sum(r.iterations for r in adv.last_diffusion))

    def fence():
        torch.cuda.synchronize()  # the library's RCCL kernels are done before torch's communicator is used: two communicators never overlap
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    iters.clear()
    lib().cup3d_profile_enable(0 if a.no_profile else 1)
    lib().cup3d_profile_reset()
    lib().cup3d_stats_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    fence()
    sec = time.perf_counter() - t0
    main_iters = list(iters)
    a.umax_by_step = umax[-a.steps:] + [float(cu.findMaxU(sim))]
    st = RunStats()
    lib().cup3d_stats_read(C.byref(st))
    nit = max(1, st.solver_iterations)
    a.comm = {"rccl_ranks": world, "halo_exchanges_per_iteration": round(st.halo_exchanges / nit, 2), "halo_bytes_sent_per_iteration (rank 0)": round(st.halo_bytes_sent / nit, 1),
              "allreduces_per_iteration": round(st.allreduces / nit, 2), "host_waits_per_iteration": round(st.host_waits / nit, 3),
              "host_wait_ms_per_step (spinning on the device's status, overlapped with queued kernels)": round(st.host_wait_seconds / a.steps * 1e3, 3),
              "host_wait_fraction": round(st.host_wait_seconds / sec, 4)}
    a.diffusion_iters = round(float(np.mean(diff_iters[-a.steps:])), 2) if diff_iters else None
    if dist is not None:
        t = torch.tensor([sec], dtype=torch.float64, device=a.tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec = float(t.item())
    ents = (ProfileEntry * 64)()
    n = C.c_int(0)
    lib().cup3d_profile_read(ents, 64, C.byref(n))
    lib().cup3d_profile_enable(0)
    prof = {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n.value)}
    tot, nblk = C.c_long(0), C.c_long(0)
    lib().cup3d_profile_block_cg_iterations(sim.handle, C.byref(tot), C.byref(nblk))
    a.cg_iters_per_block = tot.value / nblk.value if nblk.value else None

    # PCIe-inclusive rate of the C++ shim (never `value`): the boundary hands over one host pointer per block; measured here through
    # that very path (cup3d_sim_upload_blocks / _download_blocks on resident, reused host pages), then applied to the bytes the shim
    # moves per step in its two modes
    a.pcie = None
    if world == 1 and not a.stencil_only and not a.implicit_diffusion and not a.no_pcie:
        fid = cu.operators.FIELDS["vel"]
        hv = np.empty((sim.nblocks, 8, 8, 8, 3))
        ptrs = (C.c_void_p * sim.nblocks)(*[hv[i].ctypes.data for i in range(sim.nblocks)])
        rates = {}
        for name, fn in (("download", lambda: check(lib().cup3d_sim_download_blocks(sim.handle, fid, ptrs))),
                         ("upload", lambda: check(lib().cup3d_sim_upload_blocks(sim.handle, fid, ptrs)))):
            fn()
            t0 = time.perf_counter()
            fn()
            rates[name] = hv.nbytes / (time.perf_counter() - t0) / 1e9
        gb = hv.nbytes / 1e9
        step_s = sec / a.steps
        cells = float(a.size) ** 3
        modes = {}
        # device_led (CUP3D_HIP_RESIDENT=3 + cup3d_hip::calcMaxTimestep / advance in the time loop): vel and pres come down only before
        # adaptMesh, i.e. every 20th step (main.cpp:15314), and go up again only if the mesh changed
        for mode, up, down in (("round_trip", 2 * gb + gb / 3, 4 * gb + gb / 3), ("resident", gb + gb / 3, gb + gb / 3), ("resident_across_steps", 0.0, gb + gb / 3),
                               ("device_led", 0.0, (gb + gb / 3) / 20)):
            t = step_s + up / rates["upload"] + down / rates["download"]
            modes[mode] = {"GB_up_per_step": round(up, 2), "GB_down_per_step": round(down, 2), "Mcell_updates_per_s": round(cells / t / 1e6, 2)}
        a.pcie = {"upload_GBps": round(rates["upload"], 1), "download_GBps": round(rates["download"], 1), "shim_modes": modes,
                  "note": "derived: device step time of this run + the shim's per-step transfers at the measured block-pointer rates"}
        del hv, ptrs

    SOLVERS = {0: "block CG (reference algorithm)", 1: "direct block solve (fast diagonalisation)", 2: "block CG, reference association (no FMA)",
               5: "geometric multigrid V(2,2)-cycle (NOT the reference's preconditioner; same operator, stopping rule and converged pressure)"}
    alt, alts = None, {}
    if not a.stencil_only and not a.no_alt and not a.implicit_diffusion and world == 1:
        # the same workload once more with the preconditioner M^-1 evaluated / chosen differently (cup3d_poisson_params.block_solver),
        # reported NEXT to the headline, never instead of it: the direct block solve (the reference's M, exact instead of by CG) and,
        # a multigrid V-cycle in M's place (what BASELINE.json's north_star wording describes; the reference has none; over several
        # GPUs every rank cycles on its own blocks -- additive Schwarz, no message inside the preconditioner)
        for solver in ([1, 5] if a.block_solver == 0 else [1 - a.block_solver] if a.block_solver in (0, 1) else []):
            sim.blockSolver = solver
            sim.upload("vel", taylor_green_blocks(sim.grid, [ext] * 3, 1.0))
            sim.fill("pres", 0.0)
            sim.step, sim.dt = 21, 0.0
            lib().cup3d_profile_enable(0)
            one_step()
            iters.clear()
            fence()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                one_step()
            fence()
            sec2 = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([sec2], dtype=torch.float64, device=a.tdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sec2 = float(t.item())
            alts[solver] = {"block_preconditioner": SOLVERS[solver], "value": round(float(a.size) ** 3 * a.steps / sec2 / 1e6, 2), "unit": "Mcell-updates/s",
                            "ms_per_step": round(sec2 / a.steps * 1e3, 3), "bicgstab_iters_per_step": round(float(np.mean(iters)), 2), "warmup": 1, "steps": a.steps}
        alt = alts.get(1, alts.get(0))
        a.alt_multigrid = alts.get(5)
    invalid = a.checksum is not None and not a.checksum["ok"]  # the same on every rank (all-gathered)
    if rank == 0:
        report(a, sim, prof, sec, main_iters, world, alt)
    if dist is not None:
        if a.transport == "host":
            lib().cup3d_debug_host_transport(0, 1, None)
        lib().cup3d_comm_finalize()
        dist.destroy_process_group()
    if invalid:
        sys.exit(3)


def ref_iters(a, device_iters=None):
    """BiCGSTAB iterations per step of the REFERENCE on this very workload, from recorded runs of the compiled reference:
    profiles/r03/reference_window_<size>.json = per-step counts over steps 21.. of the bench's own time loop (the window bench.py
    times with --warmup W --steps K is steps 21+W .. 21+W+K-1), else the one-step record of round 2; None if never recorded."""
    if a.stencil_only or a.implicit_diffusion:
        return None
    f = os.path.join(ROOT, "profiles", "r03", f"reference_window_{a.size}.json")
    if os.path.exists(f):
        rec = json.load(open(f))
        by_step = {s["step"]: s["iters"] for s in rec["steps"]}
        win = [by_step[n] for n in range(21 + a.warmup, 21 + a.warmup + a.steps) if n in by_step]
        out = {"value": round(float(np.mean(win)), 2) if win else None, "steps_covered": len(win), "of": a.steps,
               "window": f"steps {21 + a.warmup}..{21 + a.warmup + a.steps - 1} (the timed region of this run)",
               "by_step": win, "reference_threads": rec.get("threads"),
               # the device's mean over exactly the steps the recording covers (they are the first ones of the window)
               "device_over_the_same_steps": round(float(np.mean(device_iters[:len(win)])), 2) if win and device_iters and len(device_iters) >= len(win) else None,
               "source": f"profiles/r03/reference_window_{a.size}.json (compiled reference, {len(rec['steps'])} steps from step 21; its OpenMP "
                         "reductions make the count vary from run to run by ~10-20 %)"}
        return out
    f = os.path.join(ROOT, "profiles", "r02", f"reference_step_{a.size}.json")
    if not os.path.exists(f):
        return None
    rec = json.load(open(f))
    out = {"value": rec["ref_iters_per_step"], "device_in_the_same_run": rec["device_iters_per_step"], "steps": len(rec["steps"]),
           "window": "step 21 only -- NOT the window this run times",
           "source": f"profiles/r02/reference_step_{a.size}.json (compiled reference, one step from step 21)"}
    f1 = os.path.join(ROOT, "profiles", "r02", f"reference_step_{a.size}_first_run.json")
    if os.path.exists(f1):  # the reference's count is not reproducible (OpenMP reduction order): an earlier run of the same campaign
        out["value_in_an_earlier_run"] = json.load(open(f1))["ref_iters_per_step"]
    return out


def report(a, sim, prof, sec, iters, world, alt=None):
    cells = float(a.size) ** 3
    cells_local = sim.nblocks * 512.0
    value = cells * a.steps / sec / 1e6
    kernels = []
    total_ms = sum(ms for _, ms in prof.values()) or 1.0
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(traffic_file)) if os.path.exists(traffic_file) else {}
    for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        if launches == 0:
            continue
        avg_ms = ms / launches
        e = {"kernel": name, "launches": launches, "avg_ms": round(avg_ms, 5), "share": round(ms / total_ms, 4)}
        if name in ALGO_BYTES:
            ach = ALGO_BYTES[name] * cells_local / (avg_ms * 1e-3) / 1e9
            tr = traffic.get(f"{name}@{a.size}")
            e.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": tr})
            if tr is not None:  # a recorded PMC measurement of this kernel at this size, not a quantity of this run
                e["traffic_source"] = traffic.get("_source", "profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)")
        if name == "poisson_block_cg" and getattr(a, "cg_iters_per_block", None):
            # The block CG moves exactly its 16 B/cell (PMC) and sits at < 0.15 of the HBM roof: HBM is the wrong roof.  It is
            # bound by FP64 instruction issue: algorithmic flops = 17 per cell per CG iteration x the iterations the blocks of the
            # last launch actually took (counted on the device), against the FP64 vector peak.
            flops = BLOCK_CG_FLOPS_PER_CELL_ITER * 512.0 * a.cg_iters_per_block * sim.nblocks
            tf = flops / (avg_ms * 1e-3) / 1e12
            e.update({"bound": "fp64", "achieved": round(tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP64_PEAK_TFLOPS, 4),
                      "hbm_frac": e["frac"], "cg_iterations_per_block": round(a.cg_iters_per_block, 2),
                      "note": ("one wavefront per 8^3 block, <= 100 CG iterations in registers; flops = 17/cell/CG-iteration x counted iterations "
                               "(6 FMA + 5 add: 11 issue slots, so 17/22 = 0.77 of peak is the ceiling of this instruction mix); wave sums by DPP row "
                               "reductions (the FP64 matrix pipe was measured 12-20 % slower); HBM traffic = the algorithmic 16 B/cell (hbm_frac)")})
        kernels.append(e)
    with_roof = [k for k in kernels if "achieved" in k]
    dominant = with_roof[0] if with_roof else None
    out = {
        "metric": ("Mcell-updates/s (advect+diffuse+Poisson), 512^3 uniform, 1/2/4/8 GPUs" if not a.implicit_diffusion
                   else "Mcell-updates/s (implicit-diffusion advect+diffuse + Poisson), uniform") if not a.stencil_only
        else "Mcell-updates/s (advect+diffuse only), uniform periodic",
        "value": round(value, 2), "unit": "Mcell-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(sec / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"taylor-green {a.size}^3 uniform, all-wall box (reference has no lid BC), nu=0.01, CFL=0.3, rampup=0, "
                                f"poissonTol 1e-6/1e-4, bMeanConstraint 1, steps from 21") if not a.stencil_only
                   else f"taylor-green {a.size}^3 uniform periodic, advect-diffuse RK3 only",
                   "cells": int(cells), "blocks": int(cells // 512), "block": "8^3", "partition": f"hilbert-range x{world}",
                   "bicgstab_iters_per_step": round(float(np.mean(iters)), 2) if iters else None,
                   "bicgstab_iters_by_step": [int(i) for i in iters] if iters else None,
                   # max|u| entering every timed step and after the last one: a solver-level signal that must not depend on the number
                   # of ranks beyond the stopping tolerance of the projection (the checksum below covers the stencil path bit for bit)
                   "umax_by_step": getattr(a, "umax_by_step", None),
                   "ref_iters_per_step": ref_iters(a, iters),
                   "checksum": getattr(a, "checksum", None),
                   "communication": getattr(a, "comm", None),
                   "nu": a.nu, "implicit_diffusion": bool(a.implicit_diffusion),
                   "helmholtz_iters_per_step (3 solves)": getattr(a, "diffusion_iters", None),
                   "block_preconditioner": {0: "block CG (reference algorithm)", 1: "direct block solve (fast diagonalisation)",
                                            5: "geometric multigrid V-cycle (not the reference's)"}.get(a.block_solver, str(a.block_solver)),
                   "library": os.path.basename(getattr(sys.modules.get("cup3d_amd.capi"), "LIB_PATH", "libcup3d_hip.so")),
                   "transport": "rccl" if getattr(a, "transport", "rccl") == "rccl" else
                                "host-memory TEST transport over gloo, ranks may share a device: checks the multi-process path and the checksum, NOT a scaling measurement"},
        "roofline": ({k: dominant[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} | {"kernel": dominant["kernel"]})
        if dominant else None,
        "kernels": kernels,
    }
    if alt is not None:
        out["alt"] = alt
    if getattr(a, "alt_multigrid", None):
        out["alt_multigrid"] = a.alt_multigrid
    if getattr(a, "pcie", None):
        out["pcie_inclusive"] = a.pcie
    if not a.no_cpu and world == 1:
        # SURVEY 8d asks for the reference on all host cores at 512^3, else 256^3.  A 512^3 step of the reference takes 4.5 minutes
        # (268 s for the projection alone on 64 threads, profiles/r02/reference_step_512.json) and ~35 GB: the sample is ONE step at
        # 256^3.  "All cores" would be a strawman on this host: the reference's OpenMP regions (one lab per thread, master-polled halo
        # loop, 5594-5640) ANTI-scale -- one 256^3 step takes 18 s on 32 threads, 30 s on 64 and 338 s on all 256
        # (profiles/r02/probe_reference_threads_256cubed.txt) -- so the baseline runs at the reference's best setting, 32 threads,
        # and `cores` says so.  Round 1's sample (128^3, 10 steps) stays beside it as cpu_baseline_128.
        threads = min(a.cpu_threads or (os.cpu_count() or 1), os.cpu_count() or 1)
        out["cpu_baseline"] = cpu_baseline(a.cpu_size, a.cpu_steps, threads)
        out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        # the all-core figure SURVEY 8d asks for, quoted from the recorded scan (one step takes 5.6 minutes there: not re-run here)
        out["cpu_baseline"]["all_cores_recorded"] = {"value": round(256 ** 3 / 338.49 / 1e6, 4), "unit": "Mcell-updates/s", "cores": 256, "size": 256,
                                                     "also": {"64 threads": round(256 ** 3 / 30.31 / 1e6, 3), "32 threads": round(256 ** 3 / 18.17 / 1e6, 3)},
                                                     "source": "profiles/r02/probe_reference_threads_256cubed.txt (one 256^3 step each; the reference anti-scales beyond 32 threads)"}
        if a.cpu_size != 128:
            out["cpu_baseline_128"] = cpu_baseline(128, 10, min(32, os.cpu_count() or 1))
    ck = getattr(a, "checksum", None)
    if ck is not None and not ck["ok"]:
        out["valid"] = False  # the stencil path did not reproduce the oracle's bits on this partition: the rate above measures a wrong program
    print(json.dumps(out))
    sys.stdout.flush()
    if out.get("valid") is False:
        sys.stderr.write("bench: config.checksum does not match the oracle's constant -- results INVALID\n")
    return out


if __name__ == "__main__":
    main()
