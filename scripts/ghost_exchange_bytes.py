#!/usr/bin/env python3
"""What a ghost-block exchange of a rank view ships, whole blocks against sub-boxes, on the mesh BASELINE configs[4] names (two-fish school,
bpd 2, levelMax 7: the reference itself builds it, two steps on the CPU; no GPU needed):  python scripts/ghost_exchange_bytes.py
-> profiles/r04/ghost_exchange_bytes_configs4.txt"""
import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as O, cup3d_amd as cu
TWO_FISH = ("StefanFish L=0.4 T=1.0 xpos=0.25 ypos=0.5 zpos=0.5 planarAngle=180 heightProfile=danio widthProfile=stefan bFixFrameOfRef=1\n"
            "StefanFish L=0.4 T=1.0 xpos=0.7 ypos=0.5 zpos=0.5 heightProfile=danio widthProfile=stefan")
args = ["-bMeanConstraint", "2", "-bpdx", "2", "-bpdy", "2", "-bpdz", "2", "-CFL", "0.4", "-Ctol", "0.1", "-extentx", "1", "-levelStart", "1", "-nu", "0.001",
        "-poissonSolver", "iterative", "-Rtol", "5", "-tdump", "0", "-tend", "0", "-factory", "", "-levelMax", "7", "-factory-content", TWO_FISH]
recs, wd = O.run_ref(["op steps 2", "tables t.bin"], args, threads=8, timeout=3600)
t = O.read_tables(os.path.join(wd, "t.bin"))[0]
lv, zs = t[:, 0].astype(np.int32), t[:, 1].copy()
g = cu.operators.Grid((2, 2, 2), 7, 0, 1.0, ("freespace",) * 3, leaves=(lv, zs))
nb = g.nblocks
print("blocks", nb)
for nranks in (2, 8):
    owner = (np.arange(nb) * nranks // nb).astype(np.int32)     # contiguous runs of the m_vInfo (Hilbert) order, GridMPI's rule
    views = [g.rank_view(owner, r, nranks) for r in range(nranks)]
    ghosts = [v.nghost for v in views]
    w1 = [int(v.recv_cells[0].sum()) for v in views]; w3 = [int(v.recv_cells[1].sum()) for v in views]
    print(nranks, "ranks: ghost blocks per rank", ghosts, "\n  whole-block KB per scalar exchange per rank (max)", max(ghosts) * 4096 / 1e3,
          "sub-box KB", max(w1) * 8 / 1e3, "ratio", sum(ghosts) * 512 / sum(w1), "| vector w=3: whole", max(ghosts) * 12288 / 1e3, "sub-box", max(w3) * 24 / 1e3, "ratio", sum(ghosts) * 512 / sum(w3))
