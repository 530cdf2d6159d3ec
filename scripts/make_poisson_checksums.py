#!/usr/bin/env python3
"""Record the DEVICE-side constants of bench.py's config.checksum (needs an MI355X):

    python scripts/make_poisson_checksums.py [sizes...] [--out gpurun_out/poisson_checksums.json] [--merge]

`precond_exact_field` (the block CG of one field) and `fused_iteration` (cup3d_poisson_path_checksum: one BiCGSTAB iteration's kernels
on hashed vectors, scalars by hand) are block-local / stencil computations, so their bits do not depend on how the blocks are spread
over ranks -- but they are the DEVICE's bits (tree-shaped wave sums, FMA contraction in the block CG), not the CPU oracle's.  Their
reference value is therefore the ONE-GPU run of the release library, recorded here; `bench.py --gpus N` must reproduce it at every N
and `tests/test_gpu_rccl.py` checks N = 1, 2, 3.  The oracle-side constants (exact_field, taylor_green, lhs_exact_field) come from
tests/golden/make_checksums.py and are cross-checked here (a mismatch aborts: nothing is recorded from a device that fails them).
--merge writes into tests/golden/advdiff_checksums.json (run it where the repo is writable), else only --out is written."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", nargs="*", type=int, default=[64, 128, 256, 512])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "poisson_checksums.json"))
    ap.add_argument("--merge", action="store_true")
    a = ap.parse_args()
    import bench
    import cup3d_amd as cu
    cu.device_init(0)
    out = {}
    for size in a.sizes:
        nb1 = size // 8
        level = (nb1 & -nb1).bit_length() - 1
        bpd = nb1 >> level
        sim = cu.SimulationData(bpdx=bpd, bpdy=bpd, bpdz=bpd, levelMax=level + 1, levelStart=level, extent=2 * np.pi, nu=0.01, CFL=0.3,
                                BC_x="wall", BC_y="wall", BC_z="wall", uMax_forced=1.0, rampup=0)
        ck = bench.advdiff_checksums(sim, argparse.Namespace(size=size, tdev="cpu"), None, 1)
        for k in ("exact_field", "taylor_green", "lhs_exact_field"):
            if ck[k]["ok"] is False:
                sys.exit(f"{size}^3: {k} = {ck[k]['value']} differs from the oracle's {ck[k]['expected']}: not recording anything")
        out[str(size)] = {"precond_exact_field": ck["precond_exact_field"]["value"], "fused_iteration": ck["fused_iteration"]["value"],
                          "oracle_signals_checked": [k for k in ("exact_field", "taylor_green", "lhs_exact_field") if ck[k]["ok"] is True]}
        print(size, out[str(size)], flush=True)
        del sim
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"library": os.path.basename(cu.capi.LIB_PATH), "sizes": out}, open(a.out, "w"), indent=1, sort_keys=True)
    if a.merge:
        merge(a.out)


def merge(path):
    gold = os.path.join(ROOT, "tests", "golden", "advdiff_checksums.json")
    g = json.load(open(gold))
    for size, rec in json.load(open(path))["sizes"].items():
        g.setdefault(size, {}).update({k: rec[k] for k in ("precond_exact_field", "fused_iteration")})
    json.dump(g, open(gold, "w"), indent=1, sort_keys=True)
    print("merged into", gold)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--merge-only":
        merge(sys.argv[2])
    else:
        main()
