#!/usr/bin/env python3
"""Partition-independent checksums of bench.py's first AdvectionDiffusion, from the CPU ORACLE (oracle/cup3d_oracle.c, pinned
bit-exactly against the compiled reference by tests/test_oracle_vs_ref.py).

    python tests/golden/make_checksums.py [sizes...]      ->  tests/golden/advdiff_checksums.json

Workload = bench.py's grid (all-`wall` box of extent 2*pi, nu = 0.01, uinf = 0) with two input fields, both built by bench.py's own
functions: `exact_field` (bench.exact_test_field_blocks: polynomials evaluated with correctly rounded operations only, so the input
bits do not depend on the machine's libm) and `taylor_green` (bench.taylor_green_blocks = KernelIC_taylorGreen, main.cpp:12516-12539,
through numpy's sin / cos: portable only as far as those are).  ONE AdvectionDiffusion::operator() (main.cpp:9640-9728) of the
oracle with the fixed dt = 0.3 * h; the checksum is the wrapping 64-bit sum of the bit patterns of every value of `vel` afterwards.  The stencil operators are bit-exact
on the device under any sharding of the blocks, and integer addition commutes, so `bench.py --gpus N` must reproduce these
constants at N = 1, 2, 4, 8 (it all-gathers cup3d_sim_checksum of every rank and adds mod 2^64)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
import cup3d_amd as cu  # noqa: E402  (host-side Grid only: block order and indices, no GPU)
from bench import checksum_dt, exact_test_field_blocks, taylor_green_blocks  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512]
    path = os.path.join(HERE, "advdiff_checksums.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    ext = 2 * np.pi
    for size in sizes:
        nb1 = size // 8
        level = (nb1 & -nb1).bit_length() - 1   # blocks per side = bpd * 2^level, as bench.py builds the grid
        bpd = nb1 >> level
        g = O.OracleGrid((bpd,) * 3, level + 1, level, ext, ("wall",) * 3)
        G = cu.Grid((bpd,) * 3, level + 1, level, ext, ("wall",) * 3)
        assert np.array_equal(G.index, g.index), "library and oracle disagree about the block order"
        rec = {"dt": checksum_dt(size)}
        for key, vel in (("exact_field", exact_test_field_blocks(G, size)), ("taylor_green", taylor_green_blocks(G, [ext] * 3, 1.0))):
            g.advect_diffuse(vel, np.zeros_like(vel), checksum_dt(size), 0.01, (0.0, 0.0, 0.0))
            rec[key] = int(vel.view(np.uint64).sum(dtype=np.uint64))
        # the Poisson path's stencil: ONE ComputeLHS (KernelLHSPoisson, main.cpp:9205-9215; bMeanConstraint 0) of the x component of the
        # exact test field taken as a pressure -- bit-exact on the device under any sharding, like the advect-diffuse stage
        pres = np.ascontiguousarray(exact_test_field_blocks(G, size)[..., 0])
        rec["lhs_exact_field"] = int(g.lhs(pres, 0).view(np.uint64).sum(dtype=np.uint64))
        rec = {**out.get(str(size), {}), **rec}   # keep what other generators recorded for this size (scripts/make_poisson_checksums.py)
        out[str(size)] = rec
        print(size, rec)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
