import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
"""Random boxes / boundary conditions / levels / rank counts: the reference's multi-level operators on several ranks of a real MPI
(oracle/_ref/ref_tool_mpi) against the one-rank AMR oracle.  Configurations in which the reference itself fails (MPI_ERR_TRUNCATE in
its synchroniser on small boxes) are reported and skipped."""
import subprocess
import numpy as np, oracle_lib as O, make_golden as M
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
BCS = ["periodic", "wall", "freespace"]
bad = done = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    bpd = tuple(int(v) for v in rng.integers(int(os.environ.get('MINBPD', '1')), 4, 3))
    if bpd[0] * bpd[1] * bpd[2] > 12: bpd = (2, 2, 2)
    lmax = int(rng.integers(3, 5))
    bc = tuple(BCS[int(v)] for v in rng.integers(0, 3, 3))
    n = int(rng.integers(2, 6))
    passes = int(rng.integers(1, lmax))
    wd = O.tempfile.mkdtemp(prefix="camp_")
    pre = M.amr_mesh_script(wd, bpd, passes, float(rng.choice([1.0, 2.0])))
    args = O.ref_args(bpd, lmax, 0, M.EXT, bc)
    try:
        O.run_ref_mpi(pre + ["tables t.bin"], args, n, workdir=wd)
        T = [O.read_tables(os.path.join(wd, f"t.bin.r{r}"))[0] for r in range(n)]
        t = np.concatenate(T); nb = len(t)
        if min(len(x) for x in T) == 0 or nb > 1500:
            print(trial, bpd, lmax, bc, n, "skip", [len(x) for x in T]); continue
        vel, pres = rng.uniform(-1, 1, (nb, 8, 8, 8, 3)), rng.uniform(-1, 1, (nb, 8, 8, 8))
        vel.tofile(os.path.join(wd, "velb.bin")); pres.tofile(os.path.join(wd, "presb.bin"))
        dt, nu, uinf = 0.01, 0.02, (0.1, -0.2, 0.3)
        O.run_ref_mpi(pre + ["loadb vel velb.bin", "loadb pres presb.bin", f"set nu {nu}", f"set uinfx {uinf[0]}", f"set uinfy {uinf[1]}", f"set uinfz {uinf[2]}",
                             f"op advdiff {dt}", "dump vel ad.bin", "set mean 0", "op lhs", "dump lhs lhs.bin", f"set dt {dt}", "loadb vel velb.bin", "zero tmpV",
                             "op rhs", "dump lhs rhs.bin", "op gradp", "dump tmpV gp.bin", "op vorticity", "dump tmpV w.bin"], args, n, workdir=wd)
    except subprocess.CalledProcessError as e:
        print(trial, bpd, lmax, bc, n, passes, "REFERENCE FAILED:", e.stderr.decode()[-120:].replace("\n", " ")); continue
    cat = lambda f, nc: np.concatenate([O.read_blocks(os.path.join(wd, f"{f}.r{r}"), len(T[r]), nc) for r in range(n)])
    m = O.OracleMesh(bpd, lmax, M.EXT, bc, t[:, 0], t[:, 1])
    v, _ = m.advect_diffuse(vel, dt, nu, uinf)
    ok = [np.array_equal(v, cat("ad.bin", 3)), np.array_equal(m.lhs(pres, 0), cat("lhs.bin", 1)),
          np.array_equal(m.pressure_rhs(vel, np.zeros_like(vel), np.zeros_like(pres), dt), cat("rhs.bin", 1)),
          np.array_equal(m.grad_p(pres, dt), cat("gp.bin", 3)), np.array_equal(m.vorticity(vel), cat("w.bin", 3))]
    done += 1; bad += not all(ok)
    print(trial, bpd, lmax, bc, "ranks", n, "passes", passes, "blocks", nb, sorted(set(t[:, 0].tolist())), "OK" if all(ok) else ("MISMATCH", ok), flush=True)
print("configurations compared", done, "bad", bad)
