import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, oracle_lib as O, make_golden as M
rng = np.random.default_rng(2024)
BCS = ["periodic", "wall", "freespace"]
bad = 0
for trial in range(16):
    bpd = tuple(int(v) for v in rng.integers(1, 4, 3))
    if bpd[0]*bpd[1]*bpd[2] > 12: bpd = (2, 2, 1)
    lmax = int(rng.integers(2, 5))
    bc = tuple(BCS[int(v)] for v in rng.integers(0, 3, 3))
    passes = int(rng.integers(1, lmax))
    wd = O.tempfile.mkdtemp(prefix="camp_")
    pre = M.amr_mesh_script(wd, bpd, passes, float(rng.choice([1.0, 2.0])))
    args = O.ref_args(bpd, lmax, 0, M.EXT, bc)
    _, wd = O.run_ref(pre + ["tables t1.bin"], args, threads=1, workdir=wd)
    t1, _ = O.read_tables(os.path.join(wd, "t1.bin"))
    nb = len(t1)
    if nb > 1200:
        print(trial, bpd, lmax, bc, "skip", nb); continue
    vel, pres, rhs = 0.5 * rng.uniform(-1, 1, (nb, 8, 8, 8, 3)), rng.uniform(-1, 1, (nb, 8, 8, 8)), rng.uniform(-1, 1, (nb, 8, 8, 8))
    for n, a in (("velb", vel), ("presb", pres), ("rhsb", rhs)): a.tofile(os.path.join(wd, n + ".bin"))
    dt, nu = float(rng.choice([0.05, 0.01, 0.2])), float(rng.choice([2.0, 0.05, 0.5]))
    uinf = tuple(float(v) for v in rng.uniform(-0.3, 0.3, 3))
    d = int(rng.integers(0, 3))
    script = pre + [f"set uinfx {uinf[0]!r}", f"set uinfy {uinf[1]!r}", f"set uinfz {uinf[2]!r}", f"set nu {nu}", f"set dt {dt}", "set difftol 1e-9", "set difftolrel 1e-9",
                    "loadb lhs rhsb.bin", "loadb pres presb.bin", f"op diffsolve {d}", "dump pres sol.bin", "set difftol 1e-6", "set difftolrel 1e-4",
                    "loadb vel velb.bin", "loadb pres presb.bin", f"op advdiff_implicit {dt}", "dump vel imp.bin"]
    recs, wd = O.run_ref(script, args, threads=1, workdir=wd)
    its = [int(r["value"]) for r in recs if r["op"] == "diffsolve"][0]
    m = O.OracleMesh(bpd, lmax, M.EXT, bc, t1[:, 0], t1[:, 1])
    p, info = m.diff_solve(rhs, pres, d, dt, nu, 1e-9, 1e-9)
    v, _ = m.advdiff_implicit(vel, pres, dt, nu, uinf, True)
    ok = info.iters == its and np.array_equal(p, O.read_blocks(os.path.join(wd, "sol.bin"), nb, 1)) and np.array_equal(v, O.read_blocks(os.path.join(wd, "imp.bin"), nb, 3))
    bad += not ok
    print(trial, bpd, lmax, bc, passes, nb, sorted(set(t1[:, 0].tolist())), "dt", dt, "nu", nu, "dir", d, "iters", its, "OK" if ok else "MISMATCH", flush=True)
print("bad", bad)
