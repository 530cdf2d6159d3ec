#!/usr/bin/env python3
"""Campaign (kept out of the suite: minutes of host time, tens of GB of host memory): the device against the COMPILED REFERENCE
(oracle/_ref/ref_tool = unmodified main.cpp) at the BASELINE sizes.

    python scripts/campaigns/baseline_sizes_vs_reference.py --size 512 [--threads 64] [--out profiles/r02/reference_step_512.json]
    python scripts/campaigns/baseline_sizes_vs_reference.py --size 256 --bc periodic --steps 3 --tight

One step of the reference's own operators (AdvectionDiffusion, ExternalForcing, PressureProjection, main.cpp:15229-15246) from the
Taylor-Green initial condition at step 21 (second-order pressure path, rampup 0), the same dt on both sides:
  * advect-diffuse: device == reference, bit for bit;
  * BiCGSTAB iteration counts of both;
  * both pressures satisfy the stopping rule, hence || A (p_dev - p_ref) || <= 2 max(tol, tolRel ||r0||) away from the mean row, with
    A applied by the device's ComputeLHS (bit-exact with the reference's, tests/test_gpu_parity.py); the velocity difference equals
    the gradient update of the pressure difference;
  * --tight: the projection repeated with poissonTol 1e-9 / poissonTolRel 1e-8 on both sides (below that the reference's BiCGSTAB
    stagnates at these sizes): max|dp| / max|p| and max|du| / max|correction|.
Writes one JSON record; bench.py quotes `ref_iters_per_step` from the 512^3 record.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cup3d_amd as cu  # noqa: E402
import oracle_lib as O  # noqa: E402
from bench import taylor_green_blocks  # noqa: E402
from cup3d_amd.capi import check, lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--bc", default="wall")
    ap.add_argument("--threads", type=int, default=min(64, os.cpu_count() or 1))
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--tight", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    assert O.have_ref_tool(), "oracle/_ref/ref_tool is missing (build it where /root/reference exists: make -C oracle ref)"
    level = int(round(np.log2(a.size // 8)))
    ext, nu, cfl = 2 * np.pi, 0.01, 0.3
    bc = (a.bc,) * 3
    cu.device_init(0)
    sim = cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=level + 1, levelStart=level, extent=ext, nu=nu, CFL=cfl, BC_x=bc[0], BC_y=bc[1], BC_z=bc[2],
                            uMax_forced=1.0, rampup=0)
    g = sim.grid
    nb = g.nblocks
    vel0 = taylor_green_blocks(g, [ext] * 3, 1.0)
    rec = {"size": a.size, "bc": a.bc, "blocks": int(nb), "threads": a.threads, "steps": []}
    args = O.ref_args((1, 1, 1), level + 1, level, ext, bc, nu=nu, cfl=cfl, extra=["-rampup", "0"])
    S = cu.Simulation(sim)
    sim.upload("vel", vel0)
    del vel0
    sim.step = 21
    # the reference side: the same sequence of dt, computed by the device's calcMaxTimestep (findMaxU is exact, tests)
    script = ["zero chi", "set step 21"]
    dts = []
    dev = []
    corner = int(np.where((g.index == 0).all(axis=1))[0][0])
    for n in range(a.steps):
        dt = S.calcMaxTimestep()
        dts.append(dt)
        S.pipeline[0](dt)
        ad = sim.download("vel")
        for op in S.pipeline[1:-1]:
            op(dt)
        r = S.pipeline[-1](dt)
        dev.append(dict(ad=ad, vel=sim.download("vel"), pres=sim.download("pres"), iters=r.iterations, norm0=r.norm0, restarts=r.restarts))
        sim.step += 1
        script += [f"set step {21 + n}", f"op advdiff {dt!r}", f"dump vel ad{n}.bin", f"op forcing {dt!r}", f"op project {dt!r}", f"dump vel v{n}.bin",
                   f"dump pres p{n}.bin"]
        if a.size >= 512:
            break  # host memory: one step's fields at a time
    t0 = time.time()
    recs, wd = O.run_ref(script, args, threads=a.threads, timeout=7200)
    rec["reference_seconds"] = round(time.time() - t0, 1)
    proj = [q for q in recs if q["op"] == "project"]
    adv = [q for q in recs if q["op"] == "advdiff"]
    h = g.h
    for n, d in enumerate(dev):
        ad_ref = O.read_blocks(os.path.join(wd, f"ad{n}.bin"), nb, 3)
        step = {"dt": dts[n], "advdiff_bit_exact": bool(np.array_equal(d["ad"], ad_ref)), "iters_device": int(d["iters"]), "iters_reference": int(proj[n]["iters"]),
                "restarts_device": int(d["restarts"]), "reference_advdiff_seconds": adv[n]["seconds"], "reference_project_seconds": proj[n]["seconds"]}
        if n > 0:  # from the second step on the inputs of the two sides already differ by the first step's solver error
            step["advdiff_max_abs_diff"] = float(np.abs(d["ad"] - ad_ref).max())
        del ad_ref
        p_ref = O.read_blocks(os.path.join(wd, f"p{n}.bin"), nb, 1)
        v_ref = O.read_blocks(os.path.join(wd, f"v{n}.bin"), nb, 3)
        tau = max(1e-6, 1e-4 * d["norm0"])
        dd = np.ascontiguousarray(d["pres"] - p_ref)
        sim.upload("pres", dd)
        sim.bMeanConstraint = 0
        cu.ComputeLHS(sim)(0)
        sim.bMeanConstraint = 1
        Ad = sim.download("lhs")
        Ad[corner, 0, 0, 0] = 0.0
        step.update({"tau": tau, "norm_A_dp": float(np.linalg.norm(Ad.ravel())), "within_2_tau": bool(np.linalg.norm(Ad.ravel()) <= 2.02 * tau),
                     "max_dp_over_max_p": float(np.abs(dd).max() / np.abs(p_ref).max()),
                     "max_dv": float(np.abs(d["vel"] - v_ref).max()), "max_correction": float(np.abs(v_ref - d["ad"]).max())})
        check(lib().cup3d_grad_p(sim.handle, dts[n]))
        want = sim.download("tmpV") / h ** 3
        step["dv_is_gradient_update_of_dp_err"] = float(np.abs((d["vel"] - v_ref) - want).max())
        rec["steps"].append(step)
        del p_ref, v_ref, dd, Ad, want
    if a.tight and a.size <= 256:
        dt = dts[0]
        sim.PoissonErrorTol, sim.PoissonErrorTolRel = 1e-9, 1e-8
        sim.upload("vel", dev[0]["ad"])
        sim.fill("pres", 0.0)
        sim.step = 21
        S.pipeline[1](dt) if len(S.pipeline) > 2 else None
        r = S.pipeline[-1](dt)
        recs2, wd2 = O.run_ref(["zero chi", "set step 21", f"op advdiff {dt!r}", f"op forcing {dt!r}", f"op project {dt!r}", "dump vel v.bin", "dump pres p.bin"],
                               args + ["-poissonTol", "1e-9", "-poissonTolRel", "1e-8"], threads=a.threads, timeout=7200)
        p_ref, v_ref = O.read_blocks(os.path.join(wd2, "p.bin"), nb, 1), O.read_blocks(os.path.join(wd2, "v.bin"), nb, 3)
        corr = np.abs(v_ref - dev[0]["ad"]).max()
        rec["tight"] = {"tol": [1e-9, 1e-8], "iters_device": int(r.iterations), "iters_reference": int([q for q in recs2 if q["op"] == "project"][0]["iters"]),
                        "max_dp_over_max_p": float(np.abs(sim.download("pres") - p_ref).max() / np.abs(p_ref).max()),
                        "max_du_over_correction": float(np.abs(sim.download("vel") - v_ref).max() / corr)}
    rec["ref_iters_per_step"] = float(np.mean([s["iters_reference"] for s in rec["steps"]]))
    rec["device_iters_per_step"] = float(np.mean([s["iters_device"] for s in rec["steps"]]))
    print(json.dumps(rec))
    if a.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, a.out)), exist_ok=True)
        with open(os.path.join(ROOT, a.out), "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
