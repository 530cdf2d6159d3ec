import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, oracle_lib as O, make_golden as M, cup3d_amd as cu
from test_host_indexing import _states_from_tables
rng = np.random.default_rng(77)
BCS = ["periodic", "wall", "freespace"]
bad = tot = 0
for trial in range(14):
    bpd = tuple(int(v) for v in rng.integers(1, 4, 3))
    if bpd[0]*bpd[1]*bpd[2] > 12: bpd = (2, 1, 2)
    lmax = int(rng.integers(3, 5))
    bc = tuple(BCS[int(v)] for v in rng.integers(0, 3, 3))
    n = int(rng.integers(2, 8))
    if bpd[0]*bpd[1]*bpd[2] < n: n = 2
    passes = int(rng.integers(1, lmax))
    wd = O.tempfile.mkdtemp(prefix="camp_")
    script = M.amr_mesh_script(wd, bpd, 0, float(rng.choice([1.0, 2.0]))) + ["tables t0.bin"]
    for p in range(passes): script += ["adapt", "zero chi", f"tables t{p+1}.bin"]
    try:
        O.run_ref_mpi(script, O.ref_args(bpd, lmax, 0, M.EXT, bc), n, workdir=wd)
    except Exception as e:
        print(trial, bpd, lmax, bc, n, passes, "reference failed", getattr(e, "returncode", None), (e.stderr.decode()[-300:] if hasattr(e, "stderr") and e.stderr else "")); continue
    T = [M._mpi_tables(wd, f"t{p}.bin", n) for p in range(passes + 1)]
    bcn = tuple(O.BC[b] for b in bc)
    for p in range(passes):
        old, new = np.concatenate(T[p]), np.concatenate(T[p+1])
        if not (np.all(np.diff(old[:, 5]) > 0) and np.all(np.diff(new[:, 5]) > 0)):
            print(trial, "non-contiguous ownership in the reference", [len(t) for t in T[p]]); bad += 1; continue
        ow_old = np.concatenate([np.full(len(t), r, dtype=np.int32) for r, t in enumerate(T[p])])
        ow_new = np.concatenate([np.full(len(t), r, dtype=np.int32) for r, t in enumerate(T[p+1])])
        st = _states_from_tables(old, new)
        g_old = cu.Grid(bpd, lmax, 0, 1.0, bcn, leaves=(old[:, 0].astype(np.int32), old[:, 1].copy()))
        lv, zs = g_old.adapted_leaves(st)
        g_new = cu.Grid(bpd, lmax, 0, 1.0, bcn, leaves=(lv, zs))
        ok = np.array_equal(g_new.tables, new) and np.array_equal(g_old.adapted_owners(ow_old, st, n, g_new), ow_new)
        tot += 1; bad += not ok
        print(trial, bpd, lmax, bc, "ranks", n, "pass", p, [len(t) for t in T[p]], "->", [len(t) for t in T[p+1]], "OK" if ok else "MISMATCH", flush=True)
print("transitions", tot, "bad", bad)
