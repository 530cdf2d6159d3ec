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

A run over N > 1 ranks cannot end silently: every rank reports its stage (rendezvous, comm_init, checksum, warmup, timed, ...) to a
progress file, a watchdog thread per rank (and the self-launching parent) ends the run when a stage stalls (--stall-timeout) or the
whole run exceeds --timeout, and rank 0 (or the parent) then prints ONE JSON line {"valid": false, "error": ..., "stage": ...,
"rank_progress": [...]} and exits non-zero.  torch.distributed runs on gloo (CPU tensors: bootstrap of the RCCL unique id, checksum
gather, barriers, timing reduce), so the process holds exactly ONE RCCL communicator -- the library's own.
"""
import argparse
import ctypes as C
import json
import os
import signal
import sys
import tempfile
import threading
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
    "bicgstab_loop1_fdm": 152.0,  # the same two kernels with the direct block solve behind the loops (`alt`, block_solver 1)
    "bicgstab_loop2_fdm": 136.0,
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


def _mt19937_64(seeds, ndraws):
    """std::mt19937_64 (the C++ standard's 64-bit Mersenne twister), one engine per entry of `seeds`, vectorised over the engines:
    uint64 [len(seeds)][ndraws].  (Its 10000th output for the default seed 5489 is 9981545732273789042, the standard's check value:
    tests/test_bench_contract.py.)"""
    u = np.uint64
    with np.errstate(over="ignore"):
        s = np.empty((len(seeds), 312), dtype=np.uint64)
        s[:, 0] = seeds
        for i in range(1, 312):
            p = s[:, i - 1]
            s[:, i] = u(6364136223846793005) * (p ^ (p >> u(62))) + u(i)
        out = np.empty((len(seeds), 0), dtype=np.uint64)
        UM, LM, A = u(0xFFFFFFFF80000000), u(0x7FFFFFFF), u(0xB5026F5AA96619E9)

        def mix(hi, lo, m):
            x = (hi & UM) | (lo & LM)
            return m ^ (x >> u(1)) ^ np.where((x & u(1)) != 0, A, u(0))

        while out.shape[1] < ndraws:
            n = s.copy()
            n[:, :156] = mix(s[:, :156], s[:, 1:157], s[:, 156:312])
            n[:, 156:311] = mix(s[:, 156:311], s[:, 157:312], n[:, :155])
            n[:, 311] = mix(s[:, 311], n[:, 0], n[:, 155])
            s = n
            x = s ^ ((s >> u(29)) & u(0x5555555555555555))
            x = x ^ ((x << u(17)) & u(0x71D67FFFEDA60000))
            x = x ^ ((x << u(37)) & u(0xFFF7EEE000000000))
            out = np.concatenate([out, x ^ (x >> u(43))], axis=1)
    return out[:, :ndraws]


def random_velocity_blocks(grid, seed=12345, chunk=8192):
    """SURVEY 8(d)'s solver-stress input: a seeded uniform[-1, 1) velocity, discretely divergence-full (every cell independent).  One
    std::mt19937_64 per block, seeded with `seed + Z` (Z = the block's Hilbert index), 1536 draws in the block's memory order
    [z][y][x][component] mapped like std::uniform_real_distribution<double>(-1, 1): -1 + 2 * (x / 2^64) -- so the field does not depend
    on how the blocks are spread over ranks, and a C++ host can produce the same bits."""
    Z = grid.tables[:, 1].astype(np.uint64)
    vel = np.empty((grid.nblocks, 8, 8, 8, 3))
    for b0 in range(0, grid.nblocks, chunk):
        x = _mt19937_64(Z[b0:b0 + chunk] + np.uint64(seed), 1536)
        r = np.minimum(x.astype(np.float64) * 2.0 ** -64, np.nextafter(1.0, 0.0))   # generate_canonical<double, 53> of a 64-bit engine
        vel[b0:b0 + chunk] = (-1.0 + 2.0 * r).reshape(-1, 8, 8, 8, 3)
    return vel


def checksum_dt(size):
    return 0.3 * (2 * np.pi / size)


def gather_sum(mine, a, dist, world):
    """sum over the ranks mod 2^64 of one (or a list of) 64-bit wrapping sums: all-gathered as signed int64 over gloo"""
    import torch
    vals = list(mine) if isinstance(mine, (list, tuple)) else [mine]
    if dist is None:
        tot = [v % (1 << 64) for v in vals]
    else:
        signed = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in vals], dtype=torch.int64, device=a.tdev)
        parts = [torch.zeros_like(signed) for _ in range(world)]
        dist.all_gather(parts, signed)
        tot = [sum(int(p[i].item()) for p in parts) % (1 << 64) for i in range(len(vals))]
    return tot if isinstance(mine, (list, tuple)) else tot[0]


def read_profile(cap=160):
    """cup3d_profile_read -> {entry name: (launches, total ms)} (hipEvents on the stream each kernel is launched on)"""
    from cup3d_amd.capi import ProfileEntry, lib
    ents = (ProfileEntry * cap)()
    n = C.c_int(0)
    lib().cup3d_profile_read(ents, cap, C.byref(n))
    return {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n.value)}


# algorithmic HBM bytes per FINE cell of one launch of the multigrid option's kernels (multigrid.hip):
#   mg_smooth            two red-black sweeps with frozen ghosts: iterate in 8, right-hand side in 8, iterate out 8 (the six ghost faces,
#                        0.75 cells per cell, are neighbours' cells this or a neighbouring wavefront reads anyway: L2 hits, not counted)
#   mg_smooth_from_zero  the first launch of a level: the iterate is zero and is not read
#   mg_residual_restrict iterate in 8, right-hand side in 8, the summed residual of 8 cells out 8/8
#   mg_prolong_add       fine iterate in 8 and out 8, the parent's cell in 8/8
MG_ALGO_BYTES = {"mg_smooth": 24.0, "mg_smooth_from_zero": 16.0, "mg_residual_restrict": 17.0, "mg_prolong_add": 17.0}


def multigrid_kernels(prof, fine_blocks):
    """alt_multigrid.kernels: every kernel of the V-cycle PER LEVEL (entries "mg_smooth@L6", L0 = the coarsest level, one block) with the
    HBM roofline of its level's cell count.  Only the fine levels can be near a roof: from level L-2 down a launch is shorter than its
    own launch latency (<= 4096 blocks), which `note` says instead of pretending a fraction means something there."""
    levels = sorted({int(k.split("@L")[1]) for k in prof if "@L" in k})
    if not levels:
        return []
    top = max(levels)
    out = []
    for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        if "@L" not in name or not launches:
            continue
        base, lv = name.split("@L")
        lv = int(lv)
        blocks = max(1, fine_blocks >> (3 * (top - lv)))
        avg = ms / launches
        ach = MG_ALGO_BYTES[base] * blocks * 512.0 / (avg * 1e-3) / 1e9
        e = {"kernel": base, "level": lv, "blocks": blocks, "launches": launches, "avg_ms": round(avg, 5), "total_ms": round(ms, 3), "bound": "hbm",
             "algorithmic_bytes_per_cell": MG_ALGO_BYTES[base], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        if lv == top:
            e["traffic"] = traffic_record().get(f"{base}@{round((fine_blocks * 512) ** (1 / 3))}")   # PMC bytes per launch on the finest level (recorded)
        if blocks <= 4096:
            e["note"] = "launch-latency bound: fewer blocks than the chip has wavefront slots"
        out.append(e)
    return out


POISSON_VECTORS = ("phat", "rhat", "shat", "what", "zhat", "qhat", "s", "w", "z", "t", "v", "q", "r", "y", "x", "r0", "b", "xopt")  # poisson.hip's order


def advdiff_checksums(sim, a, dist, world, prog=None):
    """config.checksum: the run's bitwise correctness signals, the same at every N (wrapping 64-bit sums of bit patterns over each rank's
    blocks, added over the ranks mod 2^64 -- integer addition commutes, so only the BITS of every cell matter, not who owns it):
      exact_field, taylor_green   `vel` after ONE AdvectionDiffusion::operator() (dt = 0.3 h, nu = 0.01) of the exact test field / the
                                  Taylor-Green field: the width-3 VECTOR halo + the advect-diffuse kernels.  expected = the CPU oracle.
      lhs_exact_field             ComputeLHS (bMeanConstraint 0) of the exact field's x component: width-1 SCALAR halo + k_lhs.  expected = the CPU oracle.
      precond_exact_field         M^-1 of that field (cup3d_preconditioner, the block CG): block-local, so partition-independent; its
                                  bits are the DEVICE's (tree-shaped sums, FMA), hence expected = the one-GPU run's recorded value.
      fused_iteration             cup3d_poisson_path_checksum: ONE BiCGSTAB iteration's kernels as the solver launches them (LHS inside
                                  the loop kernels reading the scalar face slabs in place, inner / boundary split, block CG) on hashed
                                  vectors with alpha, beta, omega set by hand: 18 vector sums folded into one.  expected = the one-GPU run's.
    The reference's own multi-rank run equals its one-rank run the same way (tests/test_oracle_vs_ref.py)."""
    import cup3d_amd as cu
    from cup3d_amd.capi import check, lib
    ext = 2 * np.pi
    golden = os.path.join(ROOT, "tests", "golden", "advdiff_checksums.json")
    expected = json.load(open(golden)).get(str(a.size), {}) if os.path.exists(golden) else {}
    out = {"what": "wrapping uint64 sums of bit patterns, added over ranks mod 2^64; `expected`: the CPU oracle's value (exact_field, taylor_green, "
                   "lhs_exact_field) or the one-GPU device run's (precond_exact_field, fused_iteration: block-local CG, device rounding) -- "
                   "tests/golden/advdiff_checksums.json", "dt": checksum_dt(a.size)}

    def record(key, value):
        exp = expected.get(key)
        out[key] = {"value": value, "expected": exp, "ok": (value == exp) if exp is not None else None}
        if prog is not None:
            prog.beat(key)

    adv = cu.AdvectionDiffusion(sim)
    nu0, sim.nu = sim.nu, 0.01
    exact = exact_test_field_blocks(sim.grid, a.size)
    for key, field in (("exact_field", exact), ("taylor_green", taylor_green_blocks(sim.grid, [ext] * 3, 1.0))):
        sim.upload("vel", field)
        adv(checksum_dt(a.size))
        record(key, gather_sum(sim.checksum("vel"), a, dist, world))
    sim.nu = nu0
    # the Poisson path: A p, M^-1 p, one fused iteration
    pres = np.ascontiguousarray(exact[..., 0])
    sim.upload("pres", pres)
    check(lib().cup3d_compute_lhs(sim.handle, 0))
    record("lhs_exact_field", gather_sum(sim.checksum("lhs"), a, dist, world))
    sim.upload("pres", pres)
    check(lib().cup3d_preconditioner(sim.handle, 0))
    record("precond_exact_field", gather_sum(sim.checksum("pres"), a, dist, world))
    sums = (C.c_ulonglong * 18)()
    check(lib().cup3d_poisson_path_checksum(sim.handle, 0, 1, sums))
    per_vector = gather_sum([int(v) for v in sums], a, dist, world)
    folded = 0
    for i, v in enumerate(per_vector):   # order-sensitive fold: a swap of two vectors' sums does not cancel
        folded = (folded * 1099511628211 + v + i) % (1 << 64)
    record("fused_iteration", folded)
    out["fused_iteration"]["vectors"] = dict(zip(POISSON_VECTORS, per_vector))
    sim.fill("pres", 0.0)
    sim.fill("lhs", 0.0)
    keys = ("exact_field", "taylor_green", "lhs_exact_field", "precond_exact_field", "fused_iteration")
    oks = [out[k]["ok"] for k in keys]
    # three outcomes: False = some constant exists and differs (the run is INVALID); True = every signal has a constant and equals it;
    # None = nothing contradicted but some signal has no recorded constant for this --size (unchecked, not wrong)
    out["ok"] = False if any(o is False for o in oks) else (True if all(o is True for o in oks) else None)
    out["unchecked"] = [k for k in keys if out[k]["ok"] is None]
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



STAGES = ("start", "rendezvous", "comm_init", "grid", "checksum", "warmup", "timed", "alt", "alt_early", "report", "cpu_baseline", "done")


def progress_dir():
    """One directory per run, the same for every rank: handed down by the self-launching parent, else derived from the rendezvous."""
    d = os.environ.get("CUP3D_BENCH_PROGRESS_DIR")
    if not d:
        d = os.path.join(tempfile.gettempdir(), "cup3d_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none")))
    os.makedirs(d, exist_ok=True)
    return d


def read_progress(d, world, since=None):
    out = []
    for r in range(world):
        try:
            rec = json.load(open(os.path.join(d, f"rank{r}.json")))
            if since is not None and rec.get("started_unix", since) < since - 600:   # a directory reused by back-to-back runs (same port)
                rec = {"rank": r, "stage": "never reported in THIS run (the file on record is an earlier run's)", "earlier_run": rec}
            out.append(rec)
        except Exception:
            out.append({"rank": r, "stage": "never reported (the process did not get as far as bench.py's main)"})
    return out


def error_line(a_gpus, stage, error, rank_progress, steps=None, warmup=None):
    """The ONE line a failed run prints instead of the result (same leading keys, so that a reader of BENCH/SCALE records sees it)."""
    return json.dumps({"metric": "Mcell-updates/s (advect+diffuse+Poisson), 512^3 uniform, 1/2/4/8 GPUs", "value": None, "unit": "Mcell-updates/s",
                       "n_gpus": a_gpus, "steps": steps, "warmup": warmup, "valid": False, "error": error, "stage": stage, "rank_progress": rank_progress})


class Progress:
    """Stage reporting + watchdog of one rank.  The watchdog is a daemon thread: it ends the process (os._exit) when the current stage
    has made no progress for its stall limit, when the run exceeds its total limit, or when the launcher sends SIGTERM because another
    rank died (signal wake-up fd: seen at once even while the main thread sits inside an RCCL call).  Rank 0 prints the error line."""

    def __init__(self, rank, world, a):
        self.rank, self.world, self.a = rank, world, a
        self.dir = progress_dir()
        self.path = os.path.join(self.dir, f"rank{rank}.json")
        self.t0 = self.last = time.time()
        self.stage, self.detail, self.limit = "start", "", a.stall_timeout
        self.done = self.failing = False
        self.lock = threading.Lock()
        self.rd, self.wr = os.pipe()
        os.set_blocking(self.wr, False)
        if world > 1 or a.watchdog:
            try:
                signal.signal(signal.SIGTERM, lambda *args: None)   # the watchdog thread acts on it (wake-up fd), not the blocked main thread
                signal.set_wakeup_fd(self.wr, warn_on_full_buffer=False)
            except ValueError:
                pass   # not the main thread (tests importing bench): no signal hook
            threading.Thread(target=self.watch, daemon=True).start()
        # 'start' = the imports (torch pages in from a cold image: minutes on a bad day, seen in round 6) -- not a collective, so not the
        # business of a short --stall-timeout
        self.set("start", limit=max(a.stall_timeout, 600.0))

    def set(self, stage, detail="", limit=None):
        with self.lock:
            self.stage, self.detail, self.last = stage, detail, time.time()
            self.limit = limit if limit is not None else self.a.stall_timeout
            rec = {"rank": self.rank, "pid": os.getpid(), "stage": stage, "detail": detail, "t": round(self.last - self.t0, 2), "started_unix": int(self.t0)}
        tmp = self.path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(rec, f)
        os.replace(tmp, self.path)
        self.last_write = time.time()

    def beat(self, detail):
        """heartbeat inside a stage.  The progress FILE is rewritten at most four times a second: a write per step cost 0.6 ms -- nothing
        against a 1.4 s step of the headline, a third of a 1.1 ms step of --stencil-only --size 256 (round 6: 1.75 instead of 1.13 ms)"""
        now = time.time()
        if now - getattr(self, "last_write", 0.0) < 0.25:
            with self.lock:
                self.detail, self.last = detail, now     # the watchdog still sees progress
            return
        self.set(self.stage, detail, self.limit)

    def fail(self, error, code=4):
        """Called by the watchdog thread or by main()'s exception handler: rank 0 prints the line; everybody leaves."""
        with self.lock:
            second, self.failing = self.failing, True
        if second:                    # the other thread is already on its way out with its own diagnosis: one line, not two
            time.sleep(30)
            os._exit(code)
        self.set(self.stage, f"FAILED: {error}", 1e9)
        fallback = getattr(self, "fallback", None)
        if fallback is not None:   # an OPTIONAL region failed (alt_early_allreduce) after the main result existed: the result is printed, the failure noted in it
            if self.rank == 0:
                time.sleep(0.5)
                fallback(f"stage '{self.stage}': {error}", read_progress(self.dir, self.world, self.t0))
                sys.stdout.flush()   # (os._exit below does not)
            else:
                time.sleep(3.0)
            sys.stderr.write(f"bench.py rank {self.rank}: optional region failed: {error} (stage {self.stage}); the main result stands\n")
            sys.stderr.flush()
            os._exit(0)
        if self.rank == 0:
            time.sleep(0.5)   # the other ranks' last words
            sys.stdout.write(error_line(self.world, self.stage, error, read_progress(self.dir, self.world, self.t0), self.a.steps, self.a.warmup) + "\n")
            sys.stdout.flush()
        else:
            time.sleep(3.0)   # rank 0 prints before the launcher tears the group down because this rank left
        sys.stderr.write(f"bench.py rank {self.rank}: {error} (stage {self.stage})\n")
        sys.stderr.flush()
        os._exit(code)

    def watch(self):
        import select
        while not self.done:
            r, _, _ = select.select([self.rd], [], [], 1.0)
            if self.done:
                return
            now = time.time()
            if r:
                sig = os.read(self.rd, 64)
                if bytes([signal.SIGTERM]) in sig:
                    self.fail("SIGTERM from the launcher: another rank ended (see rank_progress)", 143)
            with self.lock:
                stalled, total = now - self.last > self.limit, now - self.t0 > self.a.timeout
                stage, limit = self.stage, self.limit
            if stalled:
                self.fail(f"no progress for {limit:.0f} s in stage '{stage}' (a hung rendezvous, communicator or collective)")
            if total:
                self.fail(f"run exceeded --timeout {self.a.timeout:.0f} s")

    def finish(self):
        self.set("done")
        self.done = True

def relaunch_under_torchrun(n, need_devices=True, a=None):
    """`python bench.py --gpus N` (N > 1) as the driver types it: one process per GPU via torch.distributed.run, rendezvous on
    127.0.0.1; rank 0's JSON line is the only thing on stdout."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if os.environ.get("CUP3D_BENCH_SHARE_DEVICE"):   # TEST hook (with CUP3D_RCCL_LIBRARY = the stand-in of tests/fake_rccl): ranks may share a device
        need_devices = False
    if have < n and need_devices:
        sys.stderr.write(f"bench.py --gpus {n} needs {n} devices; {have} visible on this host\n")
        sys.exit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    pdir = tempfile.mkdtemp(prefix="cup3d_bench_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n))), CUP3D_BENCH_PROGRESS_DIR=pdir)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the parent is the last line of defence: whatever happens to the ranks (a hang the per-rank watchdogs did not end, a crash before
    # bench.py's main, the launcher itself failing), ONE JSON line reaches stdout and the exit code says so
    timeout = (a.timeout if a else 1500.0) + 30.0
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    lines = []
    reader = threading.Thread(target=lambda: [lines.append(ln) for ln in proc.stdout], daemon=True)
    reader.start()
    try:
        rc = proc.wait(timeout=timeout)
        why = f"the ranks ended with exit code {rc} without a result line"
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)   # the launcher and every rank: the group this parent started, nothing else
        except ProcessLookupError:
            pass
        rc = proc.wait()
        why = f"no result within {timeout:.0f} s: process group killed"
    reader.join(timeout=5)
    got = [ln for ln in lines if ln.lstrip().startswith("{") and '"metric"' in ln]
    for ln in got[:1] or []:
        sys.stdout.write(ln if ln.endswith("\n") else ln + "\n")
    if not got:
        prog = read_progress(pdir, n)
        stages = [p.get("stage") for p in prog]
        first = min((STAGES.index(st) if st in STAGES else -1) for st in stages) if stages else -1
        sys.stdout.write(error_line(n, STAGES[first] if first >= 0 else "launch", why, prog, a.steps if a else None, a.warmup if a else None) + "\n")
        rc = rc or 4
    sys.stdout.flush()
    sys.exit(rc)


def cpu_baseline(size_cpu, steps, threads, stencil_only=False):
    """Time the compiled reference (oracle/_ref/ref_tool = unmodified main.cpp) on the host."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    level = int(np.log2(size_cpu // 8))
    if stencil_only and O.have_ref_tool():
        # BASELINE configs[1]: AdvectionDiffusion::operator() alone (main.cpp:9640-9728) on the periodic Taylor-Green box, dt = CFL h / umax
        args = O.ref_args((1, 1, 1), level + 1, level, 2 * np.pi, ("periodic",) * 3, nu=0.01, cfl=0.3, extra=["-rampup", "0"])
        dt = 0.3 * (2 * np.pi / size_cpu)
        recs, _ = O.run_ref(["zero chi", "set step 21", f"op advdiff {dt!r}", f"rep {max(3, steps)}", f"op advdiff {dt!r}"], args, threads=threads, timeout=600)
        secs = sorted(r["seconds"] for r in recs if r["op"] == "advdiff")[:-1] or [r["seconds"] for r in recs if r["op"] == "advdiff"]   # (the first call warms up: dropped as the slowest)
        sec = secs[len(secs) // 2]
        return {"value": size_cpu ** 3 / sec / 1e6, "unit": "Mcell-updates/s", "cores": threads, "kind": "reference", "size": size_cpu,
                "sample": f"reference AdvectionDiffusion::operator() alone, {size_cpu}^3 periodic TGV, median of {len(secs)} calls ({', '.join('%.3f' % x for x in secs)} s)",
                "advect_diffuse_seconds": sec, "advect_diffuse_value": round(size_cpu ** 3 / sec / 1e6, 3)}
    if O.have_ref_tool():
        args = O.ref_args((1, 1, 1), level + 1, level, 2 * np.pi, ("wall",) * 3, nu=0.01, cfl=0.3, extra=["-rampup", "0"])
        try:
            # `steps` separate steps of the reference's own time loop, each timed by the harness: the MEDIAN step is the sample
            # `timeops` wraps every pipeline entry in a wall-clock timer: each step also says what AdvectionDiffusion (metric A,
            # main.cpp:9640-9728) and PressureProjection (15061-15160) took on their own
            recs, _ = O.run_ref(["zero chi", "set step 21", "timeops", f"rep {steps}", "op steps 1"], args, threads=threads, timeout=600)
            rs = [r for r in recs if r["op"] == "steps"]
            secs, its = sorted(r["seconds"] for r in rs), [r["iters"] for r in rs]
            sec = secs[len(secs) // 2]
            med = lambda name: (lambda v: v[len(v) // 2] if v else None)(sorted(r["seconds"] for r in recs if r["op"] == "optime" and r.get("name") == name))
            adv_s, proj_s = med("AdvectionDiffusion"), med("PressureProjection")
            return {"value": size_cpu ** 3 / sec / 1e6, "unit": "Mcell-updates/s", "cores": threads, "kind": "reference",
                    "sample": f"reference main.cpp operators, {size_cpu}^3 all-wall TGV, median of {len(rs)} steps from step 21 "
                              f"({', '.join('%.2f' % x for x in secs)} s), {np.mean(its):.1f} BiCGSTAB its/step",
                    "seconds_per_step": [round(x, 3) for x in secs], "bicgstab_iters_per_step": float(np.mean(its)), "size": size_cpu,
                    # metric (A) and the projection on their own (medians over the same steps)
                    "advect_diffuse_seconds": adv_s, "projection_seconds": proj_s,
                    "advect_diffuse_value": round(size_cpu ** 3 / adv_s / 1e6, 3) if adv_s else None}
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


def run_amr(a, prog=None, dist=None, rank=0, world=1):
    """--amr: the same step on a multi-level mesh the device builds itself.  A compact vortex on a uniform level; a few passes of
    Simulation.adaptMesh (vorticity tags -> ValidStates -> refine/compress on the device) refine around it; then K steps on the
    frozen mesh are timed.  Reported beside the headline, not instead of it.  With --gpus N every rank builds the same mesh (the
    adaptation is deterministic), takes its contiguous run of the block order (GridMPI's rule) as a RANK VIEW -- ghost blocks exchanged
    in sub-boxes, face fluxes, all-reduced scalars -- and the line carries what crossed ranks per BiCGSTAB iteration."""
    import cup3d_amd as cu
    from cup3d_amd.capi import ProfileEntry, RunStats, check, lib
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
    nblocks_global = int(sim.nblocks)
    tables = sim.grid.tables.copy()
    nfaces = int(lib().cup3d_grid_ninterface_faces(sim.grid.handle))
    comm = None
    if world > 1:   # this rank's view of the mesh: its contiguous run of the block order + ghost blocks; the fields of its blocks
        if prog is not None:
            prog.set("grid", "rank views of the multi-level mesh")
        owner = (np.arange(nblocks_global) * world // nblocks_global).astype(np.int32)
        mesh = sim.grid
        view = mesh.rank_view(owner, rank, world)
        mine = view.global_slot[:view.nlocal]
        vel = sim.download("vel")[mine]
        vs = sim._like(view=view)
        vs.upload("vel", vel)
        a.view_sizes = {"local_blocks": int(view.nlocal), "ghost_blocks": int(view.nghost),
                        "cells_received_per_exchange": {"width_1_sub_boxes": int(view.recv_cells[0].sum()), "width_3_sub_boxes": int(view.recv_cells[1].sum()),
                                                        "whole_blocks": int(view.nghost) * 512}}
        del S, sim, vel
        sim = vs
        S = cu.Simulation(sim)
    sim.step = 21
    iters, umax = [], []

    def one_step():
        dt = S.calcMaxTimestep()
        umax.append(float(sim.uMax_measured))
        S.advance(dt)
        iters.append(sim.last_poisson.iterations)
        if prog is not None:
            prog.beat(f"step {len(iters)}")

    def fence():
        lib().cup3d_device_synchronize()
        if dist is not None:
            dist.barrier()

    if prog is not None:
        prog.set("warmup")
    for _ in range(a.warmup):
        one_step()
    iters.clear()
    lib().cup3d_profile_enable(1)
    lib().cup3d_profile_reset()
    lib().cup3d_stats_reset()
    fence()
    if prog is not None:
        prog.set("timed")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    fence()
    sec = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([sec], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec = float(t.item())
        st = RunStats()
        lib().cup3d_stats_read(C.byref(st))
        nit = max(1, st.solver_iterations)
        comm = {"ranks": world, "ghost_and_flux_exchanges_per_iteration": round(st.halo_exchanges / nit, 2),
                "MB_sent_per_iteration (rank 0)": round(st.halo_bytes_sent / nit / 1e6, 4), "allreduces_per_iteration": round(st.allreduces / nit, 2),
                "rank_0_view": getattr(a, "view_sizes", None),
                "ghost_blocks_travel_as": "whole 8^3 blocks (A/B)" if any(o.startswith("whole_ghost_blocks=1") for o in (a.debug_option or [])) else "sub-boxes (Grid::ghost_box)",
                "transport": "rccl" if a.transport == "rccl" else "host-memory TEST transport over gloo (a correctness run: bytes and counts are meaningful, the rate is not)"}
    if rank != 0:
        return None
    prof = read_profile()
    total_ms = sum(ms for k, (_, ms) in prof.items() if not k.startswith("comm_")) or 1.0   # shares of the compute stream's time
    cells = nblocks_global * 512.0
    cells_local = sim.nblocks * 512.0
    t = tables
    kernels = []
    for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        if not launches or name.startswith("comm_"):
            continue
        e = {"kernel": name, "launches": launches, "avg_ms": round(ms / launches, 5), "share": round(ms / total_ms, 4)}
        if name in ALGO_BYTES:
            # the unfused AMR advect-diffuse stage reads vel and tmpV and writes tmpV only (72 B/cell); k_rk_update is its own entry
            bpc = 72.0 if name == "advdiff_stage" else ALGO_BYTES[name]
            ach = bpc * cells_local / (ms / launches * 1e-3) / 1e9
            e.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None})
        kernels.append(e)
    with_roof = [k for k in kernels if "achieved" in k]
    out = {"metric": "Mcell-updates/s (advect+diffuse+Poisson), multi-level AMR mesh", "value": round(cells * a.steps / sec / 1e6, 2),
           "unit": "Mcell-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(sec / a.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"compact vortex in an all-wall box, uniform level {lstart} refined {a.amr_levels - 1}x around it by Simulation.adaptMesh "
                                  f"(top {a.amr_fraction:.0%} of the blocks by vorticity each pass), frozen mesh while timing",
                      "blocks": nblocks_global, "cells": int(cells), "blocks_per_level": {int(l): int((t[:, 0] == l).sum()) for l in sorted(set(t[:, 0].tolist()))},
                      "interface_faces": nfaces, "block_history": history, "umax_by_step": umax[-a.steps:], "communication": comm,
                      "bicgstab_iters_by_step": [int(i) for i in iters],
                      "finest_uniform_equivalent_cells": int((8 << (lmax - 1)) ** 3), "mesh_build_seconds": round(adapt_s, 3),
                      "bicgstab_iters_per_step": round(float(np.mean(iters)), 2),
                      "block_preconditioner": {0: "block CG (reference algorithm)", 1: "direct block solve (fast diagonalisation)",
                                               5: "geometric multigrid V-cycle over the octree's levels (NOT the reference's preconditioner)"}.get(a.block_solver, str(a.block_solver))},
           "roofline": ({k: with_roof[0][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} | {"kernel": with_roof[0]["kernel"]}) if with_roof else None,
           "kernels": kernels}
    print(json.dumps(out))
    sys.stdout.flush()


def run_micro(a, sim, prog):
    """--micro: SURVEY 8(d)'s micro-benchmarks on the workload's grid, one rank: ComputeLHS (main.cpp:9273-9327), the block preconditioner
    (getZImplParallel 14704-14745: the block CG and its direct evaluation) and the two fused vector updates (14453-14464 + 14502-14515 with
    the LHS and the block solve inside, as the solver launches them: cup3d_poisson_path_checksum), each timed by hipEvents over `--steps`
    repetitions on a random p (std::mt19937_64 seeded 1 + block index)."""
    import cup3d_amd as cu
    from cup3d_amd.capi import ProfileEntry, check, lib
    Z = sim.grid.tables[:, 1].astype(np.uint64)
    p = np.empty((sim.nblocks, 8, 8, 8))
    for b0 in range(0, sim.nblocks, 16384):
        x = _mt19937_64(Z[b0:b0 + 16384] + np.uint64(1), 512)
        p[b0:b0 + 16384] = (-1.0 + 2.0 * np.minimum(x.astype(np.float64) * 2.0 ** -64, np.nextafter(1.0, 0.0))).reshape(-1, 8, 8, 8)
    reps = max(3, a.steps)
    sums = (C.c_ulonglong * 18)()
    prog.set("timed", "micro-benchmarks")
    lib().cup3d_profile_enable(1)
    lib().cup3d_profile_reset()
    for i in range(reps + 1):
        if i == 1:
            lib().cup3d_profile_reset()   # the first round warms up (allocations, constant tables)
        sim.upload("pres", p)
        check(lib().cup3d_compute_lhs(sim.handle, 1))
        for bs in (0, 1):
            sim.upload("pres", p)
            check(lib().cup3d_preconditioner(sim.handle, bs))
        for bs in (0, 1):
            check(lib().cup3d_poisson_path_checksum(sim.handle, bs, 1, sums))
    lib().cup3d_device_synchronize()
    mprof = read_profile()
    lib().cup3d_profile_enable(0)
    cells = sim.nblocks * 512.0
    names = {"poisson_lhs": "one LHS apply (ComputeLHS, bMeanConstraint 1)", "poisson_block_cg": "one preconditioner apply (block CG)",
             "poisson_block_fdm": "one preconditioner apply (direct block solve)", "bicgstab_loop1_cg": "fused vector update 1 (+ LHS + block CG)",
             "bicgstab_loop2_cg": "fused vector update 2 (+ LHS + block CG)", "bicgstab_loop1_fdm": "fused vector update 1 (+ LHS + direct block solve)",
             "bicgstab_loop2_fdm": "fused vector update 2 (+ LHS + direct block solve)"}
    out = []
    for nm, (ln, ms) in mprof.items():
        if nm in names and ln:
            ach = ALGO_BYTES[nm] * cells / (ms / ln * 1e-3) / 1e9
            out.append({"kernel": nm, "what": names[nm], "launches": ln, "avg_ms": round(ms / ln, 5), "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_cell": ALGO_BYTES[nm]})
    print(json.dumps({"metric": "micro-benchmarks of the Poisson path (SURVEY 8d)", "n_gpus": 1, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": f"{a.size}^3 uniform all-wall grid, random p (std::mt19937_64, 1 + block index)", "blocks": int(sim.nblocks), "repetitions": reps,
                                 "note": "the random field makes the block CG run its full course (more CG iterations than on solver inputs)"},
                      "kernels": sorted(out, key=lambda k: k["kernel"])}))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=512, help="cells per side (a multiple of 8; 512 = the BASELINE workload, 768 = the largest that leaves room on one 288 GB GPU)")
    ap.add_argument("--cpu-size", type=int, default=256, help="cells per side of the CPU-baseline sample (512 needs ~35 GB and minutes per step)")
    ap.add_argument("--cpu-steps", type=int, default=5, help="steps of the CPU-baseline sample; the median step is reported (SURVEY 8d: >= 5 repeats)")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="OpenMP threads of the reference; 32 is its best on the 256-thread GPU host (see report()); 0: all host cores")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--stencil-only", action="store_true", help="BASELINE configs[1]: periodic, advect-diffuse only")
    ap.add_argument("--block-solver", type=int, default=0, help="0: block CG as in the reference, 1: direct block solve, 5: multigrid V-cycle")
    ap.add_argument("--no-alt", action="store_true", help="skip the second timed region with the other block solver")
    ap.add_argument("--no-alt-early", action="store_true", help="N > 1: skip the optional region that times the early all-reduce order (alt_early_allreduce)")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="what carries the library's exchanges over ranks.  rccl: production (one device per rank).  host: the library's "
                         "host-memory TEST transport over torch.distributed/gloo (libcup3d_hip_testing.so): lets --gpus N run on fewer than N "
                         "devices to check the multi-process path and config.checksum; its rate says nothing about scaling")
    ap.add_argument("--timeout", type=float, default=1500.0, help="whole-run limit in seconds; the watchdog then prints the error line and exits 4 (the driver's own limit is 1800 s)")
    ap.add_argument("--stall-timeout", type=float, default=300.0, help="limit for ONE stage step without progress (a hung rendezvous / collective), seconds")
    ap.add_argument("--watchdog", action="store_true", help="run the per-rank watchdog on one rank too (it always runs when N > 1)")
    ap.add_argument("--fail-at", default=None, help="TEST: 'stage:rank:how' with how = hang | exit | raise -- that rank misbehaves when it enters the stage")
    ap.add_argument("--input", choices=["taylor_green", "random"], default="taylor_green",
                    help="initial velocity.  taylor_green: the BASELINE workload.  random: SURVEY 8(d)'s solver-stress input, a seeded "
                         "(std::mt19937_64, 12345 + block index) uniform[-1,1) divergence-full field -- reported beside the headline, never as it")
    ap.add_argument("--micro", action="store_true",
                    help="SURVEY 8(d)'s micro-benchmarks instead of the step: one LHS apply, one preconditioner apply (block CG, direct), the two "
                         "fused vector updates, on a random p (seed 1 + block index) of the workload's grid; one JSON line")
    ap.add_argument("--no-checksum", action="store_true", help="skip config.checksum (one extra AdvectionDiffusion on two fields before the timed region)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host<->device transfer measurement behind `pcie_inclusive`")
    ap.add_argument("--debug-option", action="append", help="name=value for cup3d_debug_set_option (tuning scans)")
    ap.add_argument("--full-line", action="store_true", help="print the FULL record as the one stdout line (round 5's form, ~20 KB) instead of the compact summary")
    ap.add_argument("--detail-out", default="bench_detail.json", help="file the full record is written to (relative to the repo root; '' = nowhere)")
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
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and a.gpus > 1:
        return relaunch_under_torchrun(a.gpus, a.transport == "rccl", a)
    prog = Progress(rank, world, a)
    try:
        run(a, prog)
    except SystemExit:
        raise
    except BaseException as e:   # a rank that raises must not leave the others inside a collective without a word
        import traceback
        traceback.print_exc()
        prog.fail(f"{type(e).__name__}: {e}", 5)
    prog.finish()


def misbehave(a, prog, stage, rank):
    """--fail-at stage:rank:how (tests of the watchdog): hang = sleep for ever, exit = leave at once, raise = a Python exception"""
    if not a.fail_at:
        return
    st, r, how = a.fail_at.split(":")
    if st != stage or int(r) != rank:
        return
    if how == "hang":
        while True:
            time.sleep(3600)
    if how == "exit":
        os._exit(7)
    raise RuntimeError(f"--fail-at {a.fail_at}")


def run(a, prog):
    if a.no_fuse or a.debug_option or a.block_solver in (3, 4) or a.transport == "host":
        os.environ["CUP3D_HIP_FLAVOUR"] = "testing"   # A/B switches live in libcup3d_hip_testing.so only; everything else times the release build
    # RCCL prints a version banner to STDOUT under NCCL_DEBUG=VERSION (the image's default), once per process and communicator
    # library: stdout carries the one JSON line of rank 0 and nothing else
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL between processes needs dmabuf IPC on this driver (must precede the first HIP call)
    rank, world = prog.rank, prog.world
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world

    def stage(name, detail="", limit=None):
        prog.set(name, detail, limit)
        misbehave(a, prog, name, rank)

    import torch
    import cup3d_amd as cu
    from cup3d_amd.capi import ProfileEntry, RunStats, check, lib

    ndev = max(1, torch.cuda.device_count())
    if a.transport == "rccl" and local_rank >= ndev and not os.environ.get("CUP3D_BENCH_SHARE_DEVICE"):
        sys.exit(f"bench.py: rank {rank} has no device (RCCL wants one per rank; {ndev} visible)")
    local_dev = local_rank % ndev   # host transport: ranks may share a device
    torch.cuda.set_device(local_dev)
    cu.device_init(local_dev)
    if (a.no_fuse or a.debug_option or a.transport == "host") and not hasattr(lib(), "cup3d_debug_set_option"):
        sys.exit("bench.py: --no-fuse / --debug-option / --transport host need the testing build of the library (CUP3D_HIP_FLAVOUR=testing, "
                 "libcup3d_hip_testing.so); the loaded one is %s" % os.path.basename(cu.capi.LIB_PATH))
    if a.no_fuse:
        check(lib().cup3d_debug_set_option(b"no_fuse", 1))
    for opt in (a.debug_option or []):   # tuning scans: --debug-option name=value (cup3d_debug_set_option)
        name, val = opt.split("=")
        check(lib().cup3d_debug_set_option(name.encode(), int(val)))
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (the rendezvous waits for the slowest rank's imports: the same generous limit as 'start'; the collectives after it keep --stall-timeout)
        stage("rendezvous", "torch.distributed on gloo", limit=max(a.stall_timeout, 600.0))
        # gloo with CPU tensors carries the bootstrap, the checksum gather, the barriers and the timing reduce: the process holds ONE
        # RCCL communicator, the library's own (no second one to keep apart from it)
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=max(a.stall_timeout, 600.0)))
        stage("comm_init", "host-memory test transport" if a.transport == "host" else "ncclCommInitRank of the library's communicator")
        if a.transport == "host":
            a.host_transport = install_host_transport(dist, rank, world)   # keeps the ctypes callbacks alive
        else:
            # bootstrap the library's RCCL communicator with rank 0's unique id
            idbuf = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                raw = (C.c_ubyte * 128)()
                check(lib().cup3d_comm_unique_id(raw))
                idbuf = torch.tensor(list(raw), dtype=torch.uint8)
            dist.broadcast(idbuf, 0)
            raw = (C.c_ubyte * 128)(*idbuf.tolist())
            check(lib().cup3d_comm_init(rank, world, raw))
    elif os.environ.get("CUP3D_FORCE_COMM"):
        # MEASUREMENT SUPPORT (testing build, --debug-option force_allreduce=1): ONE process whose solver scalars travel through the
        # library's communicator all the same -- a one-rank communicator of CUP3D_RCCL_LIBRARY (tests/fake_rccl with FAKE_RCCL_ALLREDUCE_US:
        # what an iteration costs when an all-reduce takes as long as it does between devices, nothing else changed)
        raw = (C.c_ubyte * 128)()
        check(lib().cup3d_comm_unique_id(raw))
        check(lib().cup3d_comm_init(0, 1, raw))
    a.tdev = "cpu"   # torch.distributed's own tensors (gloo)
    stage("grid", "topology, allocation")
    if a.amr:
        out = run_amr(a, prog, dist, rank, world)
        if dist is not None:
            if a.transport == "host":
                lib().cup3d_debug_host_transport(0, 1, None)
            lib().cup3d_comm_finalize()
            dist.destroy_process_group()
        return out

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
    if not a.stencil_only and not a.implicit_diffusion and not a.no_checksum and not a.micro:
        stage("checksum", "first exchanges over the transport: advect-diffuse, LHS, one fused iteration")
        a.checksum = advdiff_checksums(sim, a, dist, world, prog)  # the run's correctness signal at every N (before anything is timed)
        sim.dt = 0.0
    if a.micro:
        if world > 1:
            sys.exit("bench.py --micro runs on one rank")
        return run_micro(a, sim, prog)
    initial = (lambda: random_velocity_blocks(sim.grid)) if a.input == "random" else (lambda: taylor_green_blocks(sim.grid, [ext] * 3, 1.0))
    sim.upload("vel", initial())
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
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()   # gloo: host-side only
        torch.cuda.synchronize()

    stage("warmup")
    for i in range(a.warmup):
        one_step()
        prog.beat(f"step {i + 1}/{a.warmup}")
    iters.clear()
    lib().cup3d_profile_enable(0 if a.no_profile else 1)
    lib().cup3d_profile_reset()
    lib().cup3d_stats_reset()
    fence()
    stage("timed")
    t0 = time.perf_counter()
    for i in range(a.steps):
        one_step()
        prog.beat(f"step {i + 1}/{a.steps}")   # (a 100-byte file write: nothing against a step)
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
              "host_wait_fraction": round(st.host_wait_seconds / sec, 4),
              # several ranks: do the all-reduces of a fused loop start when its last block leaves the vector phase (CUP3D_EARLY_ALLREDUCE=1 /
              # --debug-option early_allreduce=1) or when the kernel has ended (default)?
              "early_allreduce": bool(int(os.environ.get("CUP3D_EARLY_ALLREDUCE", "0") or 0)) or any(o == "early_allreduce=1" for o in (a.debug_option or [])),
              "injected_allreduce_latency_us (stand-in library only)": float(os.environ.get("FAKE_RCCL_ALLREDUCE_US", "0") or 0) if os.environ.get("CUP3D_RCCL_LIBRARY") else None}
    a.diffusion_iters = round(float(np.mean(diff_iters[-a.steps:])), 2) if diff_iters else None
    if dist is not None:
        t = torch.tensor([sec], dtype=torch.float64, device=a.tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec = float(t.item())
    prof = read_profile()
    lib().cup3d_profile_enable(0)
    # device timings of the communication stream (hipEvents on that stream, rank 0's): what one iteration spends in face-slab exchanges
    # (pack + grouped send/recv), in all-reduces (+ the recurrence step behind them), and how long the COMPUTE stream sat waiting for
    # either -- the exposed part; the rest of the first two was hidden behind the inner blocks' pass of the loop kernels
    cms = lambda name: prof.get(name, (0, 0.0))[1]
    a.comm.update({"halo_ms_per_iteration": round(cms("comm_halo") / nit, 5), "allreduce_ms_per_iteration": round(cms("comm_allreduce") / nit, 5),
                   "exposed_ms_per_iteration": round((cms("comm_exposed_halo_wait") + cms("comm_exposed_scalar_wait")) / nit, 5),
                   "exposed_halo_wait_ms_per_iteration": round(cms("comm_exposed_halo_wait") / nit, 5),
                   "exposed_scalar_wait_ms_per_iteration": round(cms("comm_exposed_scalar_wait") / nit, 5),
                   "timed_by": "hipEvents on the communication stream (comm_halo, comm_allreduce) and around the compute stream's waits for it "
                               "(comm_exposed_*), rank 0; per BiCGSTAB iteration of the timed steps (the per-step exchanges of advect-diffuse and "
                               "the projection's one-shot kernels are included in the numerators)"})
    a.comm_entries = {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in prof.items() if k.startswith("comm_")}
    prof = {k: v for k, v in prof.items() if not k.startswith("comm_")}   # kernel shares are shares of the COMPUTE stream's time
    tot, nblk = C.c_long(0), C.c_long(0)
    lib().cup3d_profile_block_cg_iterations(sim.handle, C.byref(tot), C.byref(nblk))
    a.cg_iters_per_block = tot.value / nblk.value if nblk.value else None

    # SURVEY 8(d) metric (A): the advect-diffuse RK3 operator ALONE on this grid (BASELINE configs[1] is the same thing at 256^3 periodic:
    # --stencil-only --size 256) -- five operator calls after the timed region, so `value` never sees them.  The velocity it leaves
    # behind is not used again (every `alt` region uploads the initial condition).
    a.stencil_sub = None
    if not a.stencil_only and not a.implicit_diffusion:
        stage("alt", "SURVEY 8(d) metric (A): the RK3 operator alone")
        nrk = 5
        adv(sim.dt)   # (untimed: the first call after the projection)
        lib().cup3d_profile_enable(1)
        lib().cup3d_profile_reset()
        fence()
        t0 = time.perf_counter()
        for _ in range(nrk):
            adv(sim.dt)
        fence()
        sec_a = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([sec_a], dtype=torch.float64, device=a.tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sec_a = float(t.item())
        pa = read_profile()
        lib().cup3d_profile_enable(0)
        ncell = float(a.size) ** 3
        per_op = sec_a / nrk
        stages = {k: {"launches": v[0], "avg_ms": round(v[1] / v[0], 5)} for k, v in pa.items() if k.startswith("advdiff") and v[0]}
        a.stencil_sub = {"what": "AdvectionDiffusion::operator() alone (3 fused RK stages), main.cpp:9640-9728; wall clock over %d calls, all ranks" % nrk,
                         "value": round(ncell / per_op / 1e6, 1), "unit": "Mcell-updates/s", "ms_per_operator": round(per_op * 1e3, 4), "n_gpus": world,
                         "bound": "hbm", "peak": HBM_PEAK_GBS,
                         "frac_at_288_B_per_cell (SURVEY 8d: 3 x 96)": round(288.0 * ncell / per_op / 1e9 / (HBM_PEAK_GBS * world), 4),
                         "frac_at_264_B_per_cell (stage 1 reads no tmpV: 72 + 96 + 96)": round(264.0 * ncell / per_op / 1e9 / (HBM_PEAK_GBS * world), 4),
                         "target": "north_star: >= 0.40 of the HBM roofline on the fused advect-diffuse stencil at 512^3",
                         "stages (hipEvents, rank 0)": stages}

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
        # GPUs it is ONE cycle coupled over the ranks: face slabs of every level's iterate cross ranks, multigrid.hip), and the block CG in
        # the reference's own association (block_solver 2: no FMA contraction) -- the price of the faithful rounding, five steps
        stage("alt", "the same workload with the other block preconditioners (never `value`)")
        for solver in ([1, 5, 2] if a.block_solver == 0 else [1 - a.block_solver] if a.block_solver in (0, 1) else []):
            nsteps = a.steps if solver != 2 else min(a.steps, 5)   # the reference-association block CG: five steps say what it costs
            sim.blockSolver = solver
            sim.upload("vel", initial())
            sim.fill("pres", 0.0)
            sim.step, sim.dt = 21, 0.0
            lib().cup3d_profile_enable(0)
            one_step()
            iters.clear()
            if solver in (1, 5) and not a.no_profile:   # the kernels of the direct-solve iteration (what the streams alone allow); of the V-cycle, per level
                lib().cup3d_profile_enable(1)
                lib().cup3d_profile_reset()
            fence()
            t0 = time.perf_counter()
            for _ in range(nsteps):
                one_step()
                prog.beat(f"block_solver {solver}")
            fence()
            sec2 = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([sec2], dtype=torch.float64, device=a.tdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sec2 = float(t.item())
            alts[solver] = {"block_preconditioner": SOLVERS[solver], "value": round(float(a.size) ** 3 * nsteps / sec2 / 1e6, 2), "unit": "Mcell-updates/s",
                            "ms_per_step": round(sec2 / nsteps * 1e3, 3), "bicgstab_iters_per_step": round(float(np.mean(iters)), 2),
                            "ms_per_bicgstab_iteration": round(sec2 * 1e3 / max(1, sum(iters)), 4), "warmup": 1, "steps": nsteps}
            if solver == 5 and not a.no_profile:
                p5 = read_profile()
                lib().cup3d_profile_enable(0)
                ks = multigrid_kernels(p5, sim.nblocks)
                fine = [k for k in ks if k["kernel"] == "mg_smooth" and k["level"] == max(x["level"] for x in ks)]
                alts[solver]["kernels"] = ks
                alts[solver]["roofline"] = ({k: fine[0][k] for k in ("bound", "achieved", "peak", "unit", "frac")} |
                                            {"kernel": "mg_smooth on the finest level (two red-black sweeps per launch)", "traffic": (traffic_record().get(f"mg_smooth@{a.size}"))}) if fine else None
                mg_ms = sum(k["total_ms"] for k in ks)
                alts[solver]["v_cycle_share_of_device_time"] = round(mg_ms / (sum(ms for _, ms in p5.values()) or 1.0), 4)
            if solver == 1 and not a.no_profile:
                p1 = read_profile()
                lib().cup3d_profile_enable(0)
                ks = []
                for nm, (ln, ms) in p1.items():
                    if nm in ALGO_BYTES and ln and nm.endswith("_fdm"):
                        ach = ALGO_BYTES[nm] * sim.nblocks * 512.0 / (ms / ln * 1e-3) / 1e9
                        ks.append({"kernel": nm, "launches": ln, "avg_ms": round(ms / ln, 5), "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)})
                alts[solver]["kernels"] = ks
        alt = alts.get(1, alts.get(0))
        a.alt_multigrid = alts.get(5)
        a.alt_reference_association = alts.get(2)
        sim.blockSolver = a.block_solver
    invalid = a.checksum is not None and a.checksum["ok"] is False  # the same on every rank (all-gathered); None = unchecked, not wrong
    # Several ranks over RCCL: the same workload once more with both all-reduces of an iteration started EARLY (CUP3D_EARLY_ALLREDUCE=1,
    # poisson.hip: when the last block leaves its vector phase, under the block solves of the kernel's last round) -- the reference hides its
    # two MPI_Iallreduce behind the preconditioner + LHS (main.cpp:14486-14490, 14546-14550).  `value` above comes from the DEFAULT order,
    # the one every multi-rank test has run; this region is OPTIONAL: if it fails or hangs, the watchdog prints the main result with
    # alt_early_allreduce = {"error": ...} instead of losing the run.
    a.alt_early = None
    if (world > 1 and a.transport == "rccl" and not a.stencil_only and not a.implicit_diffusion and a.block_solver == 0 and not a.no_alt_early
            and not os.environ.get("CUP3D_EARLY_ALLREDUCE") and not invalid):
        def fallback(error, rank_progress):
            a.alt_early = {"error": error, "rank_progress": rank_progress}
            report(a, sim, prof, sec, main_iters, world, alt)
        prog.fallback = fallback
        stage("alt_early", "the same steps with the all-reduces started early (optional; the main result is already in hand)", limit=min(a.stall_timeout, 120.0))
        os.environ["CUP3D_EARLY_ALLREDUCE"] = "1"
        nsteps = min(a.steps, 8)
        sim.upload("vel", initial())
        sim.fill("pres", 0.0)
        sim.step, sim.dt = 21, 0.0
        lib().cup3d_profile_enable(0)
        for _ in range(min(a.warmup, 2) or 1):
            one_step()
        iters.clear()
        lib().cup3d_profile_enable(0 if a.no_profile else 1)
        lib().cup3d_profile_reset()
        fence()
        t0 = time.perf_counter()
        for i in range(nsteps):
            one_step()
            prog.beat(f"early step {i + 1}/{nsteps}")
        fence()
        sec_e = time.perf_counter() - t0
        t = torch.tensor([sec_e], dtype=torch.float64, device=a.tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec_e = float(t.item())
        pe = read_profile()
        lib().cup3d_profile_enable(0)
        nit = max(1, sum(iters))
        cms = lambda name: pe.get(name, (0, 0.0))[1]
        a.alt_early = {"what": "CUP3D_EARLY_ALLREDUCE=1: the same workload from the same initial state, both all-reduces of an iteration started when the last block leaves its vector phase",
                       "value": round(float(a.size) ** 3 * nsteps / sec_e / 1e6, 2), "unit": "Mcell-updates/s", "steps": nsteps, "ms_per_step": round(sec_e / nsteps * 1e3, 3),
                       "bicgstab_iters_per_step": round(float(np.mean(iters)), 2), "ms_per_bicgstab_iteration": round(sec_e * 1e3 / nit, 4),
                       "exposed_scalar_wait_ms_per_iteration": round(cms("comm_exposed_scalar_wait") / nit, 5), "exposed_halo_wait_ms_per_iteration": round(cms("comm_exposed_halo_wait") / nit, 5),
                       "allreduce_ms_per_iteration": round(cms("comm_allreduce") / nit, 5)}
        os.environ["CUP3D_EARLY_ALLREDUCE"] = "0"
        prog.fallback = None
    stage("report", "rank 0: JSON line (+ the CPU baseline on one rank)", limit=max(a.stall_timeout, 900.0))
    if rank == 0:
        report(a, sim, prof, sec, main_iters, world, alt)
    if dist is not None:
        if a.transport == "host":
            lib().cup3d_debug_host_transport(0, 1, None)
        lib().cup3d_comm_finalize()
        dist.destroy_process_group()
    if invalid:
        sys.exit(3)


def traffic_record():
    """profiles/traffic.json: HBM bytes per cell of the hot kernels from the rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate
    runs, FETCH_SIZE corrected as the guide prescribes for gfx950), keyed "<profile entry>@<size>"; recorded, not measured by this run"""
    f = os.path.join(ROOT, "profiles", "traffic.json")
    return json.load(open(f)) if os.path.exists(f) else {}


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
    traffic = traffic_record()
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
        if name == "bicgstab_refresh":
            e["note"] = ("k_refresh, four forms per every-50th iteration (tile LHS of the input + the pointwise work on the result + the block CG by the wavefront "
                         "that owns the block; 24 / 80 / 32 / 56 algorithmic B/cell): bound by the block CG (FP64 issue, 2.8 ms stand-alone), not by HBM")
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
        # the stable figure of this workload: `value` moves with the (erratic, rounding-dependent) iteration count of the solver, the
        # time of ONE BiCGSTAB iteration does not -- whole timed region / iterations, so advect-diffuse and the projection's passes are inside
        "ms_per_bicgstab_iteration": round(sec * 1e3 / max(1, sum(iters)), 4) if iters else None,
        "config": {"workload": ((f"taylor-green {a.size}^3 uniform" if getattr(a, "input", "taylor_green") == "taylor_green" else
                                 f"SOLVER-STRESS INPUT (not the headline): seeded random velocity (std::mt19937_64, 12345 + block index, uniform[-1,1)), {a.size}^3 uniform") +
                                f", all-wall box (reference has no lid BC), nu=0.01, CFL=0.3, rampup=0, "
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
    if getattr(a, "stencil_sub", None):
        out["stencil_only"] = a.stencil_sub   # SURVEY 8(d) metric (A); `value` above is metric (B), the full step
    elif a.stencil_only:   # --stencil-only (BASELINE configs[1]): `value` IS metric (A); the operator's own roofline fractions beside it
        per_op = sec / a.steps
        out["stencil_only"] = {"what": "AdvectionDiffusion::operator() alone (3 fused RK stages + findMaxU per step), main.cpp:9640-9728: the timed region itself",
                               "value": round(value, 1), "unit": "Mcell-updates/s", "ms_per_operator": round(per_op * 1e3, 4), "n_gpus": world, "bound": "hbm", "peak": HBM_PEAK_GBS,
                               "frac_at_288_B_per_cell (SURVEY 8d: 3 x 96)": round(288.0 * cells / per_op / 1e9 / (HBM_PEAK_GBS * world), 4),
                               "frac_at_264_B_per_cell (stage 1 reads no tmpV: 72 + 96 + 96)": round(264.0 * cells / per_op / 1e9 / (HBM_PEAK_GBS * world), 4)}
    if getattr(a, "alt_multigrid", None):
        out["alt_multigrid"] = a.alt_multigrid
    if getattr(a, "alt_reference_association", None):
        out["alt_reference_association"] = a.alt_reference_association
    if getattr(a, "alt_early", None):
        out["alt_early_allreduce"] = a.alt_early
    if getattr(a, "comm_entries", None):
        out["communication_stream"] = a.comm_entries
    if getattr(a, "pcie", None):
        out["pcie_inclusive"] = a.pcie
    if not a.no_cpu and world == 1:
        # SURVEY 8d asks for the reference on all host cores at 512^3, else 256^3.  A 512^3 step of the reference takes 4.5 minutes
        # (268 s for the projection alone on 64 threads, profiles/r02/reference_step_512.json) and ~35 GB: the sample is the MEDIAN of
        # --cpu-steps (5) separate steps at 256^3.  "All cores" would be a strawman on this host: the reference's OpenMP regions (one lab per thread, master-polled halo
        # loop, 5594-5640) ANTI-scale -- one 256^3 step takes 18 s on 32 threads, 30 s on 64 and 338 s on all 256
        # (profiles/r02/probe_reference_threads_256cubed.txt) -- so the baseline runs at the reference's best setting, 32 threads,
        # and `cores` says so.  Round 1's sample (128^3, 10 steps) stays beside it as cpu_baseline_128.
        threads = min(a.cpu_threads or (os.cpu_count() or 1), os.cpu_count() or 1)
        out["cpu_baseline"] = cpu_baseline(a.cpu_size, a.cpu_steps, threads, stencil_only=a.stencil_only)
        out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
    if not a.no_cpu and world == 1 and not a.stencil_only:
        # the all-core figure SURVEY 8d asks for, quoted from the recorded scan (one step takes 5.6 minutes there: not re-run here)
        out["cpu_baseline"]["all_cores_recorded"] = {"value": round(256 ** 3 / 338.49 / 1e6, 4), "unit": "Mcell-updates/s", "cores": 256, "size": 256,
                                                     "also": {"64 threads": round(256 ** 3 / 30.31 / 1e6, 3), "32 threads": round(256 ** 3 / 18.17 / 1e6, 3)},
                                                     "source": "profiles/r02/probe_reference_threads_256cubed.txt (one 256^3 step each; the reference anti-scales beyond 32 threads)"}
        f512 = os.path.join(ROOT, "profiles", "r04", "reference_step_512.json")
        if os.path.exists(f512):   # the headline size itself on the GPU box's host, recorded (five minutes per step: not re-run inside a bench)
            rec = json.load(open(f512))
            out["cpu_baseline"]["recorded_512"] = {"value": round(512 ** 3 / rec["reference_seconds"] / 1e6, 4), "unit": "Mcell-updates/s", "size": 512, "cores": rec.get("threads"),
                                                   "kind": "reference", "seconds_per_step": rec["reference_seconds"], "steps": len(rec["steps"]),
                                                   "bicgstab_iters_per_step": rec["ref_iters_per_step"],
                                                   "source": "profiles/r04/reference_step_512.json (the compiled reference, one 512^3 step from step 21 on the GPU box's host; recorded, "
                                                             "not timed by this run)"}
        if a.cpu_size != 128:
            out["cpu_baseline_128"] = cpu_baseline(128, 10, min(32, os.cpu_count() or 1))
    ck = getattr(a, "checksum", None)
    if ck is not None and ck["ok"] is False:
        out["valid"] = False  # a bitwise signal did not reproduce its constant on this partition: the rate above measures a wrong program
    elif ck is not None and ck["ok"] is None:
        sys.stderr.write(f"bench: no recorded checksum constant for {ck['unchecked']} at --size {a.size}: unchecked, not invalid\n")
    detail = write_detail(a, out)
    print(json.dumps(out if getattr(a, "full_line", False) else compact_line(out, detail)))
    sys.stdout.flush()
    if out.get("valid") is False:
        sys.stderr.write("bench: config.checksum does not match the oracle's constant -- results INVALID\n")
    return out


COMPACT_LIMIT = 6144   # bytes; the driver's parser dropped round 5's 20 KB line (BENCH_r05.parsed = null)


def write_detail(a, out):
    """The FULL record (every kernel, the per-level multigrid list, per-step arrays, the checksums' values, notes) goes to a file next to
    the progress files -- `--detail-out`, default bench_detail.json at the repo root and a copy under gpurun_out/ when that exists --
    never to stdout: stdout carries the compact line only."""
    path = getattr(a, "detail_out", None)
    if not path:
        return None
    paths = [path if os.path.isabs(path) else os.path.join(ROOT, path)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", os.path.basename(path)))
    wrote = None
    for f in paths:
        try:
            with open(f, "w") as fh:
                json.dump(out, fh)
                fh.write("\n")
            wrote = wrote or os.path.relpath(f, ROOT)
        except OSError as e:
            sys.stderr.write(f"bench: could not write {f}: {e}\n")
    return wrote


def compact_line(out, detail=None):
    """The ONE stdout line: the contract's keys, `roofline`, `cpu_baseline` and one number per side record -- no per-step arrays, no
    notes, no per-level list (those are in the detail file).  Stays under COMPACT_LIMIT bytes (tests/test_bench_contract.py)."""
    pick = lambda d, keys: {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}
    r = pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data",
                   "ms_per_bicgstab_iteration"))
    r["vs_baseline"] = out.get("vs_baseline")
    if "valid" in out:
        r["valid"] = out["valid"]
    c = out.get("config") or {}
    cfg = pick(c, ("workload", "cells", "blocks", "partition", "bicgstab_iters_per_step", "block_preconditioner", "library", "transport"))
    ri = c.get("ref_iters_per_step")
    if ri:
        cfg["ref_iters_per_step"] = pick(ri, ("value", "device_over_the_same_steps", "steps_covered"))
    ck = c.get("checksum")
    if ck:
        cfg["checksum"] = {"ok": ck.get("ok"), "signals": {k: v.get("ok") for k, v in ck.items() if isinstance(v, dict) and "ok" in v}}
    cm = c.get("communication")
    if cm:
        cfg["communication"] = pick(cm, ("rccl_ranks", "halo_exchanges_per_iteration", "allreduces_per_iteration", "halo_ms_per_iteration", "allreduce_ms_per_iteration",
                                         "exposed_ms_per_iteration", "exposed_halo_wait_ms_per_iteration", "exposed_scalar_wait_ms_per_iteration", "host_wait_fraction",
                                         "early_allreduce"))
    r["config"] = cfg
    r["roofline"] = out.get("roofline")
    ks = out.get("kernels") or []
    dom = next((k for k in ks if out.get("roofline") and k["kernel"] == out["roofline"].get("kernel")), None)
    if dom and r["roofline"]:
        r["roofline"] = dict(r["roofline"], avg_ms=dom["avg_ms"], launches=dom["launches"], share=dom["share"])
    r["kernels"] = [pick(k, ("kernel", "launches", "avg_ms", "share", "bound", "frac")) for k in ks[:8]]
    cb = out.get("cpu_baseline")
    if cb:
        r["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind", "size", "bicgstab_iters_per_step", "advect_diffuse_seconds", "projection_seconds",
                                      "advect_diffuse_value", "host_cores_available"))
        r["cpu_baseline"]["sample"] = (cb.get("sample") or "")[:200]
        if cb.get("recorded_512"):
            r["cpu_baseline"]["recorded_512"] = pick(cb["recorded_512"], ("value", "cores", "seconds_per_step", "bicgstab_iters_per_step"))
    so = out.get("stencil_only")
    if so:
        r["stencil_only"] = pick(so, ("value", "unit", "ms_per_operator", "n_gpus"))
        for k, v in so.items():
            if k.startswith("frac_at_264"):
                r["stencil_only"]["frac_264"] = v
            elif k.startswith("frac_at_288"):
                r["stencil_only"]["frac_288"] = v
        if cb and cb.get("advect_diffuse_value"):
            r["stencil_only"]["cpu_reference_value"] = cb["advect_diffuse_value"]
    if out.get("alt_early_allreduce"):
        r["alt_early_allreduce"] = pick(out["alt_early_allreduce"], ("value", "steps", "ms_per_step", "bicgstab_iters_per_step", "ms_per_bicgstab_iteration",
                                                                     "exposed_scalar_wait_ms_per_iteration", "allreduce_ms_per_iteration"))
        if "error" in out["alt_early_allreduce"]:
            r["alt_early_allreduce"]["error"] = str(out["alt_early_allreduce"]["error"])[:300]
    for key in ("alt", "alt_multigrid", "alt_reference_association"):
        if out.get(key):
            r[key] = pick(out[key], ("value", "ms_per_step", "bicgstab_iters_per_step", "ms_per_bicgstab_iteration", "steps"))
            if out[key].get("roofline"):
                r[key]["roofline"] = pick(out[key]["roofline"], ("bound", "achieved", "peak", "unit", "frac", "traffic"))
    pc = out.get("pcie_inclusive")
    if pc:
        r["pcie_inclusive"] = pick(pc, ("upload_GBps", "download_GBps"))
        r["pcie_inclusive"]["Mcell_updates_per_s"] = {m: v["Mcell_updates_per_s"] for m, v in pc.get("shim_modes", {}).items()}
    if detail:
        r["detail"] = detail

    def finite(x):   # strict JSON: a NaN or an infinity (a 0/0 of an empty region) would make the whole line unparsable
        if isinstance(x, float) and (x != x or x in (float("inf"), float("-inf"))):
            return None
        if isinstance(x, dict):
            return {k: finite(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [finite(v) for v in x]
        return x
    r = finite(r)
    line = json.dumps(r)
    if len(line) > COMPACT_LIMIT:   # cannot happen with the keys above; if a future key makes it so, the contract's keys win
        for k in ("kernels", "pcie_inclusive", "alt_reference_association", "alt_multigrid", "alt"):
            r.pop(k, None)
            if len(json.dumps(r)) <= COMPACT_LIMIT:
                break
    return r


if __name__ == "__main__":
    main()
