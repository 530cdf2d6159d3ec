"""The CPU oracle (oracle/cup3d_oracle.c) against the golden vectors produced by the
compiled, unmodified reference (tests/golden/make_golden.py).  Everything here is
bit-exact: the oracle keeps the reference's association order and both are built
without FMA contraction; the reference ran with one OpenMP thread."""
import os

import numpy as np
import pytest

import oracle_lib as O

FIELD_CASES = ["f16_periodic", "f16_wall", "f16_mixed", "f24x16x8_mixed"]


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    g = O.OracleGrid(z["bpd"], int(z["level_max"]), int(z["level"]), float(z["extent"]), [int(b) for b in z["bc"]])
    return z, g


@pytest.mark.parametrize("name", FIELD_CASES)
def test_block_tables(golden_dir, name):
    z, g = load(golden_dir, name)
    assert np.array_equal(z["tables"], g.tables)          # level, Z, index, blockID_2 in m_vInfo order
    assert np.array_equal(z["geom"], g.geom)              # h, origin


def test_sfc_tables(golden_dir):
    z = np.load(os.path.join(golden_dir, "sfc_tables.npz"))
    L = O.lib()
    for key in z.files:
        bx, by, bz, lmax = [int(t.lstrip("L")) for t in key.split("_")[1:]]
        ref = z[key]
        s = L.orc_sfc_create(bx, by, bz, lmax)
        bpd = np.array([bx, by, bz], dtype=np.int32)
        row = 0
        for l in range(lmax):
            for k in range(bz << l):
                for j in range(by << l):
                    for i in range(bx << l):
                        idx = np.array([i, j, k], dtype=np.int32)
                        Z = L.orc_sfc_forward(s, l, i, j, k)
                        inv = np.zeros(3, dtype=np.int32)
                        L.orc_sfc_inverse(s, Z, l, inv)
                        nei = np.zeros(27, dtype=np.int64)
                        child = np.zeros(8, dtype=np.int64)
                        par = np.zeros(1, dtype=np.int64)
                        L.orc_info_tables(s, bpd, l, idx, nei, child, par)
                        got = np.concatenate([[Z, L.orc_sfc_encode(s, l, idx), par[0]], nei, child])
                        assert np.array_equal(inv, idx)
                        assert np.array_equal(got, ref[row]), (key, l, i, j, k)
                        row += 1
        assert row == ref.shape[0]
        L.orc_sfc_destroy(s)


@pytest.mark.parametrize("name", FIELD_CASES)
def test_operators_bit_exact(golden_dir, name):
    z, g = load(golden_dir, name)
    vel, pres, rhs_in = g.to_blocks(z["vel_in"]), g.to_blocks(z["pres_in"]), g.to_blocks(z["rhs_in"])
    dt, nu, uinf, step = float(z["dt"]), float(z["nu"]), z["uinf"], int(z["step"])
    assert g.max_u(vel, uinf) == float(z["maxu"])
    v, tv = vel.copy(), np.zeros_like(vel)
    g.advect_diffuse(v, tv, dt, nu, uinf)
    assert np.array_equal(v, z["ad_vel"]) and np.array_equal(tv, z["ad_tmpV"])
    assert np.array_equal(g.lhs(pres, 1), z["lhs"])
    assert np.array_equal(g.lhs(pres, 2), z["lhs_mean2"])
    assert np.array_equal(g.lhs(pres, 0), z["lhs_mean0"])
    p = pres.copy()
    g.precond(p)
    assert np.array_equal(p, z["precond"])
    b, x = rhs_in.copy(), pres.copy()
    info = g.solve(b, x)
    assert info.iters == int(z["solve_iters"]) and np.array_equal(x, z["solve"])
    assert np.array_equal(g.pressure_rhs(vel, g.to_blocks(z["udef_in"]), g.to_blocks(z["chi_in"]), dt), z["rhs"])
    assert np.array_equal(g.div_pressure(pres)[..., 0], z["divp"])
    assert np.array_equal(g.grad_p(pres, dt), z["gradp"])
    for st, tag in ((step, "pr"), (1, "pr1")):
        v, p = vel.copy(), pres.copy()
        info, _, _ = g.project(v, p, dt, st)
        assert info.iters == int(z[tag + "_iters"])
        assert np.array_equal(v, z[tag + "_vel"]) and np.array_equal(p, z[tag + "_pres"])


def test_trajectory_bit_exact(golden_dir):
    """calcMaxTimestep + AdvectionDiffusion + ExternalForcing + PressureProjection,
    main.cpp:15254-15326 with the pipeline of main.cpp:15229-15246 (no obstacles)."""
    z = np.load(os.path.join(golden_dir, "traj16_tgv.npz"))
    g = O.OracleGrid(z["bpd"], int(z["level_max"]), int(z["level"]), float(z["extent"]), [int(b) for b in z["bc"]])
    ext = [float(z["extent"])] * 3
    vel = g.taylor_green(ext, float(z["umax_forced"]))
    assert np.array_equal(vel, z["vel"][0])
    pres = np.zeros((g.nb, 8, 8, 8))
    dt, coefU = 0.0, np.array([1.5, -2.0, 0.5])
    for n in range(len(z["dts"])):
        umax = g.max_u(vel)
        dt = O.lib().orc_calc_dt(g.h, umax, float(z["nu"]), float(z["cfl"]), n, int(z["rampup"]), dt, coefU)
        assert dt == z["dts"][n]
        tmpV = np.zeros_like(vel)
        g.advect_diffuse(vel, tmpV, dt, float(z["nu"]))
        O.lib().orc_external_forcing(g.g, vel, float(z["umax_forced"]), float(z["nu"]), ext[2], dt)
        info, _, _ = g.project(vel, pres, dt, n)
        assert info.iters == int(z["iters"][n])
        assert np.array_equal(vel, z["vel"][n + 1]) and np.array_equal(pres, z["pres"][n])


def test_partition():
    """GridMPI's contiguous Z ranges, main.cpp:2970-2980."""
    import ctypes as C
    for total in (8, 27, 64, 100):
        for size in (1, 2, 3, 8):
            covered = 0
            for r in range(size):
                a, n = C.c_longlong(), C.c_longlong()
                O.lib().orc_partition(total, r, size, C.byref(a), C.byref(n))
                assert a.value == covered
                covered += n.value
            assert covered == total


ADAPT_CASES = ["adapt16_periodic", "adapt16_mixed", "adapt16_wall"]


def clamp_tags(raw, level, level_max):
    """TagBlocksVector's clamps (main.cpp:5207-5211) applied to raw TagLoadedBlock states."""
    st = raw.copy()
    if level == level_max - 1:
        st[st == 1] = 0
    if level == 0:
        st[st == -1] = 0
    return st


@pytest.mark.parametrize("name", ADAPT_CASES)
def test_mesh_adaptation_block_operators(golden_dir, name):
    """restrict (compress), prolong (RefineBlocks) and TagLoadedBlock against the reference's own adaptMesh."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    bc = [int(b) for b in z["bc"]]
    g0 = O.OracleGrid(z["bpd"], int(z["level_max"]), 0, float(z["extent"]), bc)
    g1 = O.OracleGrid(z["bpd"], int(z["level_max"]), 1, float(z["extent"]), bc)
    assert np.array_equal(g1.tables, z["tables_fine"]) and np.array_equal(g0.tables, z["tables_coarse"])
    v0, p0 = g0.to_blocks(z["vel_in"]), g0.to_blocks(z["pres_in"])
    assert np.array_equal(O.prolong_field(g0, g1, v0), z["vel_fine"])
    assert np.array_equal(O.prolong_field(g0, g1, p0), z["pres_fine"])
    assert np.array_equal(O.restrict_field(g1, g0, z["vel_fine"]), z["vel_coarse"])
    assert np.array_equal(O.restrict_field(g1, g0, z["pres_fine"]), z["pres_coarse"])
    assert np.array_equal(O.tag_blocks(g0, v0, float(z["tag_rtol"]), float(z["tag_ctol"])), clamp_tags(z["tags"], 0, int(z["level_max"])))


def test_mean_constraint_3_bit_exact(golden_dir):
    """-bMeanConstraint > 2 (main.cpp:9316-9325, 14404-14407): LHS, solve (iteration count included) and projection of the reference."""
    z, g = load(golden_dir, "mean3_mixed")
    pres, rhs, vel = g.to_blocks(z["pres_in"]), g.to_blocks(z["rhs_in"]), g.to_blocks(z["vel_in"])
    assert np.array_equal(g.lhs(pres, 3), z["lhs"])
    corner = np.where((g.index == 0).all(axis=1))[0][0]
    assert z["lhs"][corner, 0, 0, 0] == pres[corner, 0, 0, 0]          # the pinned cell
    x = pres.copy()
    info = g.solve(rhs, x, mean_constraint=3)
    assert info.iters == int(z["solve_iters"]) and np.array_equal(x, z["solve"])
    v, p = vel.copy(), pres.copy()
    info, _, _ = g.project(v, p, float(z["dt"]), int(z["step"]), mean_constraint=3)
    assert info.iters == int(z["pr_iters"]) and np.array_equal(v, z["pr_vel"]) and np.array_equal(p, z["pr_pres"])
