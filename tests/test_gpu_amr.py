"""Multi-level (AMR) meshes on the device against the AMR oracle (oracle/cup3d_oracle_amr.c, itself pinned
bit-exactly against the reference: tests/test_oracle_amr.py).  MI355X only (-m gpu).

Tolerances: ghost slabs at coarse/fine faces (restrict + prolong), flux correction, advect-diffuse, LHS (all
mean-constraint modes), pressure RHS, divP, gradP: BIT-EXACT.  Block preconditioner / BiCGSTAB / projection: as
on uniform grids (tests/test_gpu_parity.py: reductions are summed in a different order)."""
import ctypes as C
import os

import numpy as np
import pytest

import cup3d_amd as cu
import oracle_lib as O
from cup3d_amd.capi import check, lib
from cup3d_amd.operators import FIELDS

pytestmark = pytest.mark.gpu
EXT = 2 * np.pi
BCN = {0: "freespace", 1: "periodic", 2: "wall"}


@pytest.fixture(scope="module", autouse=True)
def _device():
    cu.device_init(0)


def golden_mesh(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    t = g["tables"]
    return tuple(int(b) for b in g["bpd"]), int(g["level_max"]), tuple(BCN[int(b)] for b in g["bc"]), t[:, 0].copy(), t[:, 1].copy()


def synthetic_mesh():
    bpd, lmax, bc = (2, 2, 2), 3, ("wall", "freespace", "wall")
    lv, zs = O.build_balanced_mesh(bpd, lmax, bc, [(0, 0, 0, 0), (1, 0, 0, 0)])  # levels 0, 1, 2
    return bpd, lmax, bc, lv, zs


MESHES = ["amr_periodic_l01", "amr_mixed_l12", "synthetic_l012"]


def make(golden_dir, name, **kw):
    bpd, lmax, bc, lv, zs = synthetic_mesh() if name == "synthetic_l012" else golden_mesh(golden_dir, name)
    m = O.OracleMesh(bpd, lmax, EXT, bc, lv, zs)
    sim = cu.SimulationData(bpdx=bpd[0], bpdy=bpd[1], bpdz=bpd[2], levelMax=lmax, levelStart=0, extent=EXT, BC_x=bc[0], BC_y=bc[1], BC_z=bc[2],
                            leaves=(lv, zs), **kw)
    assert np.array_equal(sim.grid.tables, m.tables)
    rng = np.random.default_rng(5)
    nb = m.nb
    f = dict(vel=rng.uniform(-1, 1, (nb, 8, 8, 8, 3)), pres=rng.uniform(-1, 1, (nb, 8, 8, 8)), rhs=rng.uniform(-1, 1, (nb, 8, 8, 8)),
             udef=rng.uniform(-1, 1, (nb, 8, 8, 8, 3)), chi=(rng.uniform(0, 1, (nb, 8, 8, 8)) > 0.7) * rng.uniform(0, 1, (nb, 8, 8, 8)))
    return m, sim, f


def slab_from_tile(tile, s, f, gl):
    """ghost layer gl behind face f of a [L][L][L][nc] tile with stencil start s -> [nc][a2*8 + a1]"""
    d, side = f >> 1, f & 1
    n = (8 + gl if side else -1 - gl) - s
    lo, hi = -s, -s + 8
    if d == 0:
        sl = tile[lo:hi, lo:hi, n]      # [z][y] -> a1 = y, a2 = z
    elif d == 1:
        sl = tile[lo:hi, n, lo:hi]      # [z][x]
    else:
        sl = tile[n, lo:hi, lo:hi]      # [y][x]
    return np.moveaxis(sl, -1, 0).reshape(tile.shape[-1], 64)


@pytest.mark.parametrize("name", MESHES)
def test_ghost_slabs_bitexact(golden_dir, name):
    m, sim, f = make(golden_dir, name)
    faces, fine, n27 = sim.grid.interface()
    assert len(faces) > 0 and set(faces[:, 1].tolist()) == {0, 1}
    for field, key, w, s, e in (("vel", "vel", 3, -3, 4), ("pres", "pres", 1, -1, 2), ("vel", "vel", 1, -1, 2)):
        sim.upload(field, f[key])
        nc = 3 if field == "vel" else 1
        got = np.zeros((len(faces), nc, w, 64))
        check(lib().cup3d_debug_amr_slabs(sim.handle, FIELDS[field], w, got))
        labs = m.labs(f[key], s, e, False)
        for ei, (sf, kind) in enumerate(faces):
            slot, face = sf // 6, sf % 6
            for gl in range(w):
                ref = slab_from_tile(labs[slot], s, face, gl)
                assert np.array_equal(got[ei, :, gl, :], ref), (field, w, "face", ei, "kind", kind, "layer", gl)


@pytest.mark.parametrize("name", MESHES)
def test_stencil_operators_bitexact(golden_dir, name):
    m, sim, f = make(golden_dir, name)
    dt, nu, uinf = 0.01, 0.02, (0.1, -0.2, 0.3)
    sim.nu, sim.uinf = nu, np.array(uinf)
    sim.upload("vel", f["vel"])
    cu.AdvectionDiffusion(sim)(dt)
    v, _ = m.advect_diffuse(f["vel"], dt, nu, uinf)
    assert np.array_equal(sim.download("vel"), v)
    corner = int(np.where((sim.grid.index == 0).all(axis=1))[0][-1])
    vol = sum(np.abs(f["pres"][b]).sum() * m.h(b) ** 3 for b in range(m.nb))
    for mc in (0, 3, 1, 2):
        sim.bMeanConstraint = mc
        sim.upload("pres", f["pres"])
        cu.ComputeLHS(sim)(0)
        got, ref = sim.download("lhs"), m.lhs(f["pres"], mc)
        if mc == 1:  # the constraint cell holds the global sum(p h^3): summed in another order on the device
            assert abs(got[corner, 0, 0, 0] - ref[corner, 0, 0, 0]) <= 1e-12 * vol
            got[corner, 0, 0, 0] = ref[corner, 0, 0, 0]
        if mc == 2:  # every cell gets + avgP h^3
            assert np.allclose(got, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
        else:
            assert np.array_equal(got, ref), f"lhs mean constraint {mc}"
    sim.bMeanConstraint = 1
    # pressure RHS with obstacles' chi / udef (udef lives in tmpV, main.cpp:15081-15085)
    sim.upload("vel", f["vel"]); sim.upload("tmpV", f["udef"]); sim.upload("chi", f["chi"])
    check(lib().cup3d_pressure_rhs(sim.handle, dt))
    assert np.array_equal(sim.download("lhs"), m.pressure_rhs(f["vel"], f["udef"], f["chi"], dt))
    sim.fill("chi", 0.0); sim.fill("tmpV", 0.0)
    check(lib().cup3d_pressure_rhs(sim.handle, dt))
    assert np.array_equal(sim.download("lhs"), m.pressure_rhs(f["vel"], np.zeros_like(f["vel"]), np.zeros_like(f["pres"]), dt))
    sim.upload("pres", f["pres"])
    check(lib().cup3d_div_pressure(sim.handle))
    assert np.array_equal(sim.download("tmpV")[..., 0], m.div_pressure(f["pres"])[..., 0])
    check(lib().cup3d_grad_p(sim.handle, dt))
    assert np.array_equal(sim.download("tmpV"), m.grad_p(f["pres"], dt))
    assert sim_max_u(sim, uinf) == m.max_u(f["vel"], uinf)
    # adaptMesh's decision input: ComputeVorticity -> tmpV, then the per-block tags (TagLoadedBlock + level clamps)
    sim.upload("vel", f["vel"])
    cu.ComputeVorticity(sim)(0)
    w = m.vorticity(f["vel"])
    assert np.array_equal(sim.download("tmpV"), w)
    linf = np.sqrt((w ** 2).sum(axis=-1)).reshape(m.nb, -1).max(axis=1)
    rt, ct = float(np.quantile(linf, 0.6)), float(np.quantile(linf, 0.3))
    tags = cu.MeshAdaptation(rt, ct).Tag(sim, "tmpV")
    assert np.array_equal(tags, m.tag(w, rt, ct)) and len(set(tags.tolist())) >= 2


@pytest.mark.parametrize("name", ["f16_mixed", "amr_periodic_l01", "amr_mixed_l12"])
def test_grad_chi_on_tmp_equals_the_reference(golden_dir, name):
    """compute<ScalarLab>(GradChiOnTmp(sim), sim.chi) (main.cpp:8540-8600): the device edit of tmpV against the REFERENCE's own output
    (tests/golden/grad_chi.npz) on a uniform grid and two multi-level meshes, with and without the levelMaxVorticity cap -- bit for
    bit, which pins the tensorial [-2,3) tile of chi behind it (two ghost layers: copies, restriction, both interpolation modes,
    zero-gradient domain faces) as well; then the tags that follow."""
    z, vz = np.load(os.path.join(golden_dir, "grad_chi.npz")), np.load(os.path.join(golden_dir, "vorticity.npz"))
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    t = g["tables"]
    bpd, lmax, bc = tuple(int(b) for b in g["bpd"]), int(g["level_max"]), tuple(BCN[int(b)] for b in g["bc"])
    kw = dict(bpdx=bpd[0], bpdy=bpd[1], bpdz=bpd[2], levelMax=lmax, extent=float(g["extent"]), BC_x=bc[0], BC_y=bc[1], BC_z=bc[2])
    sim = cu.SimulationData(levelStart=int(g["level"]), **kw) if "level" in g.files else cu.SimulationData(levelStart=0, leaves=(t[:, 0], t[:, 1]), **kw)
    assert np.array_equal(sim.grid.tables, t)
    rt, ct = (float(v) for v in vz[name + "_tol"])
    sim.Rtol, sim.Ctol = rt, ct
    sim.upload("chi", z[name + "_chi"])
    for tag, lmv in (("", lmax), ("_capped", lmax - 1)):
        sim.levelMaxVorticity = lmv
        sim.upload("tmpV", vz[name + "_vort"])
        cu.GradChiOnTmp(sim)(0)
        got = sim.download("tmpV")
        assert np.array_equal(got, z[name + "_tmpV" + tag]), (name, tag, int((got != z[name + "_tmpV" + tag]).any(axis=(1, 2, 3, 4)).sum()))
        m = O.OracleMesh(bpd, lmax, float(g["extent"]), bc, t[:, 0], t[:, 1])
        assert np.array_equal(cu.MeshAdaptation(rt, ct).Tag(sim, "tmpV"), m.tag(got, rt, ct))


def sim_max_u(sim, uinf):
    out = C.c_double()
    check(lib().cup3d_max_u(sim.handle, np.asarray(uinf, dtype=np.float64), C.byref(out)))
    return out.value


_ORACLE_CACHE = {}


def oracle_project(m, name, f, dt, step, tol, tolrel):
    key = (name, step, tol)
    if key not in _ORACLE_CACHE:
        v, p = f["vel"].copy(), f["pres"].copy()
        info, _, _ = m.project(v, p, dt, step, tol=tol, tol_rel=tolrel)
        _ORACLE_CACHE[key] = (v, p, info.iters)
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("name", MESHES)
@pytest.mark.parametrize("block_solver", [0, 1, 2, 4])
def test_poisson_path(golden_dir, name, block_solver):
    m, sim, f = make(golden_dir, name, blockSolver=block_solver)
    dt = 0.01
    # preconditioner
    sim.upload("pres", f["pres"])
    check(lib().cup3d_preconditioner(sim.handle, block_solver))
    z = f["pres"].copy()
    m.precond(z)
    assert np.abs(sim.download("pres") - z).max() <= 2e-5 * np.abs(z).max()
    # solve at the default tolerance: the returned iterate satisfies the reference's stopping rule when the residual
    # is evaluated with the ORACLE's operator; iteration count as on uniform grids
    b = f["rhs"].copy()
    sim.upload("lhs", b); sim.upload("pres", f["pres"])
    sim.pressureSolver = cu.makePoissonSolver(sim)
    sim.pressureSolver.solve()
    x = sim.download("pres")
    corner = int(np.where((sim.grid.index == 0).all(axis=1))[0][-1])
    b[corner, 0, 0, 0] = 0.0  # main.cpp:14404-14407
    res = np.linalg.norm((b - m.lhs(x, 1)).ravel())
    res0 = np.linalg.norm((b - m.lhs(f["pres"], 1)).ravel())
    assert res <= max(1e-6, 1e-4 * res0) * (1 + 1e-6)
    rhs_o, x_o = f["rhs"].copy(), f["pres"].copy()
    info = m.solve(rhs_o, x_o)
    # (random right-hand side on a 20-80 block mesh: the iteration at which BiCGSTAB first dips below the tolerance moves a lot with
    #  the rounding of the dot products -- 72 vs 46 was seen for the no-FMA block CG on amr_mixed_l12 -- hence the wider bound there)
    assert sim.last_poisson.iterations <= (1.3 if block_solver != 2 else 2.0) * info.iters + 5, (sim.last_poisson.iterations, info.iters)
    # projection, both sides solved to 1e-12 / 1e-10: the same discrete solution
    for step in ((5, 1) if block_solver == 0 else (5,)):
        tol, tolrel = 1e-12, 1e-10
        sim.PoissonErrorTol, sim.PoissonErrorTolRel = tol, tolrel
        sim.upload("vel", f["vel"]); sim.upload("pres", f["pres"])
        sim.fill("chi", 0.0)
        sim.step = step
        cu.PressureProjection(sim)(dt)
        v, p, iters = oracle_project(m, name, f, dt, step, tol, tolrel)
        # (no iteration-count assertion this deep into the residual: 144 vs 104 and 118 vs 83 were seen for two summation orders of
        #  the same dot products; the default-tolerance solve above asserts the count)
        assert sim.last_poisson.iterations < 1000
        corr = np.abs(v - f["vel"]).max()
        assert np.abs(sim.download("pres") - p).max() <= 1e-6 * np.abs(p).max(), step
        assert np.abs(sim.download("vel") - v).max() <= 1e-6 * corr, step  # 22-block mesh, random field: cond(A) * 1e-10


@pytest.mark.parametrize("name", MESHES + ["random_3_levels"])
def test_multigrid_preconditioner_on_multilevel_meshes(golden_dir, name):
    """block_solver 5 on a multi-level mesh: the octree's own levels as the multigrid hierarchy (restriction = summed residual of the
    eight children, prolongation and coarse/fine ghosts piecewise constant, multigrid.hip mg_setup_amr).  Not the reference's
    preconditioner -- the reference has no multigrid -- so parity is on the CONVERGED pressure: both sides solved to 1e-12 / 1e-10,
    same operator, same mean constraint, same stopping rule -> the oracle's pressure to 1e-6, in at most 15 BiCGSTAB iterations where
    the block CG needs 80-150."""
    if name == "random_3_levels":
        bpd, lmax, bc = (2, 2, 2), 4, ("periodic", "wall", "freespace")
        lv, zs = O.build_balanced_mesh(bpd, lmax, bc, [(0, 1, 1, 0), (1, 2, 3, 1), (1, 3, 3, 1), (2, 5, 6, 3), (0, 0, 0, 1)])
        m = O.OracleMesh(bpd, lmax, EXT, bc, lv, zs)
        sim = cu.SimulationData(bpdx=2, bpdy=2, bpdz=2, levelMax=lmax, levelStart=0, extent=EXT, BC_x=bc[0], BC_y=bc[1], BC_z=bc[2], leaves=(lv, zs), blockSolver=5)
        rng = np.random.default_rng(11)
        f = dict(vel=rng.uniform(-1, 1, (m.nb, 8, 8, 8, 3)), pres=rng.uniform(-1, 1, (m.nb, 8, 8, 8)))
        assert len(set(lv.tolist())) >= 3
    else:
        m, sim, f = make(golden_dir, name, blockSolver=5)
    dt, step, tol, tolrel = 0.01, 5, 1e-12, 1e-10
    sim.PoissonErrorTol, sim.PoissonErrorTolRel = tol, tolrel
    sim.upload("vel", f["vel"]); sim.upload("pres", f["pres"])
    sim.fill("chi", 0.0)
    sim.step = step
    r = cu.PressureProjection(sim)(dt)
    v, p = f["vel"].copy(), f["pres"].copy()
    info, _, _ = m.project(v, p, dt, step, tol=tol, tol_rel=tolrel)
    print(f"{name}: multigrid-preconditioned BiCGSTAB {r.iterations} iterations, block CG (oracle) {info.iters}; {m.nb} blocks")
    corr = np.abs(v - f["vel"]).max()
    assert np.abs(sim.download("pres") - p).max() <= 1e-6 * np.abs(p).max()
    assert np.abs(sim.download("vel") - v).max() <= 1e-6 * corr
    assert r.iterations <= 15, (r.iterations, info.iters)


def test_time_steps_on_a_fixed_mesh(golden_dir):
    """advect-diffuse + projection for a few steps on the three-level mesh from a smooth field, both sides at tight
    Poisson tolerance: the trajectories stay together to round-off of the solve."""
    m, sim, f = make(golden_dir, "synthetic_l012")
    t = sim.grid.tables
    geom = sim.grid.geom
    nb = m.nb
    vel = np.zeros((nb, 8, 8, 8, 3))
    ax = np.arange(8) + 0.5
    for b in range(nb):
        h = geom[b, 0]
        x = geom[b, 1] + ax * h; y = geom[b, 2] + ax * h; z = geom[b, 3] + ax * h
        Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
        vel[b, ..., 0] = np.sin(X) * np.cos(Y) * np.cos(Z)
        vel[b, ..., 1] = -np.cos(X) * np.sin(Y) * np.cos(Z)
    dt, nu = 0.01, 0.01
    sim.nu, sim.uinf = nu, np.zeros(3)
    sim.PoissonErrorTol, sim.PoissonErrorTolRel = 1e-12, 1e-10
    sim.upload("vel", vel); sim.fill("pres", 0.0); sim.fill("chi", 0.0)
    v, p = vel.copy(), np.zeros((nb, 8, 8, 8))
    for step in range(4):
        sim.step = step
        cu.AdvectionDiffusion(sim)(dt)
        cu.PressureProjection(sim)(dt)
        v, _ = m.advect_diffuse(v, dt, nu, (0, 0, 0))
        m.project(v, p, dt, step, tol=1e-12, tol_rel=1e-10)
    assert np.abs(sim.download("vel") - v).max() <= 1e-9
    assert np.abs(sim.download("pres") - p).max() <= 1e-6 * max(np.abs(p).max(), 1e-30)
    del t


# ------------------------------------------------------------------ mesh adaptation (Simulation::adaptMesh)
def _octet_fields(tables, seed):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as M
    return M.octet_fields(tables, seed)


def test_adapt_mesh_golden(golden_dir):
    """One full adaptMesh of the reference on a two-level mesh (vorticity tags, ValidStates, refine + compress of vel and
    pres): every stage bit-exact on the device."""
    g = np.load(os.path.join(golden_dir, "amr_adapt_mixed.npz"))
    t = g["tables"]
    bpd, lmax, bc = tuple(int(b) for b in g["bpd"]), int(g["level_max"]), tuple(BCN[int(b)] for b in g["bc"])
    sim = cu.SimulationData(bpdx=bpd[0], bpdy=bpd[1], bpdz=bpd[2], levelMax=lmax, levelStart=0, extent=float(g["extent"]), BC_x=bc[0], BC_y=bc[1],
                            BC_z=bc[2], leaves=(t[:, 0], t[:, 1]))
    sim.upload("vel", g["vel_in"]); sim.upload("pres", g["pres_in"])
    S = cu.Simulation(sim)
    rt, ct = g["tol"]
    cu.ComputeVorticity(sim)(0)
    assert np.array_equal(cu.MeshAdaptation(rt, ct).Tag(sim, "tmpV"), g["tags"])
    st = S.adaptMesh(rt, ct)
    assert (st == 1).sum() > 0 and (st == -1).sum() >= 8
    new = S.sim
    assert new is not sim and np.array_equal(new.grid.tables, g["tables_new"])
    assert np.array_equal(new.download("vel"), g["vel_new"])
    assert np.array_equal(new.download("pres"), g["pres_new"])
    # the adapted simulation keeps running: one step on the new mesh against the oracle
    m2 = O.OracleMesh(bpd, lmax, float(g["extent"]), bc, g["tables_new"][:, 0], g["tables_new"][:, 1])
    new.nu, dt = 0.01, 0.002
    cu.AdvectionDiffusion(new)(dt)
    v, _ = m2.advect_diffuse(g["vel_new"], dt, 0.01, (0, 0, 0))
    assert np.array_equal(new.download("vel"), v)


@pytest.mark.parametrize("case", ["synthetic_l012", "uniform_l1", "amr_periodic_l01"])
def test_adapt_transfer_against_oracle(golden_dir, case):
    if case == "uniform_l1":   # a uniform grid as the source: refine a few blocks of an 8^3-block level
        bpd, lmax, bc = (2, 2, 2), 3, ("periodic", "wall", "freespace")
        g0 = cu.Grid(bpd, lmax, 1, EXT, bc)
        lv, zs = g0.tables[:, 0].astype(np.int32), g0.tables[:, 1].copy()
        kw = dict(levelStart=1)
    else:
        bpd, lmax, bc, lv, zs = synthetic_mesh() if case == "synthetic_l012" else golden_mesh(golden_dir, case)
        kw = dict(levelStart=0, leaves=(lv, zs))
    m = O.OracleMesh(bpd, lmax, EXT, bc, lv, zs)
    sim = cu.SimulationData(bpdx=bpd[0], bpdy=bpd[1], bpdz=bpd[2], levelMax=lmax, extent=EXT, BC_x=bc[0], BC_y=bc[1], BC_z=bc[2], **kw)
    assert np.array_equal(sim.grid.tables, m.tables)
    vel, pres = _octet_fields(m.tables, 21)
    sim.upload("vel", vel); sim.upload("pres", pres)
    w = m.vorticity(vel)
    linf = np.sqrt((w ** 2).sum(axis=-1)).reshape(m.nb, -1).max(axis=1)
    rt, ct = float(np.quantile(linf, 0.7)), float(np.quantile(linf, 0.4))
    S = cu.Simulation(sim)
    st = S.adaptMesh(rt, ct)
    st_o = m.valid_states(m.tag(w, rt, ct))
    assert np.array_equal(st, st_o) and (st == 1).sum() > 0
    m2 = m.adapted(st_o)
    assert np.array_equal(S.sim.grid.tables, m2.tables)
    assert np.array_equal(S.sim.download("vel"), m.transfer(m2, vel))
    assert np.array_equal(S.sim.download("pres"), m.transfer(m2, pres))


def test_adaptive_time_loop_block_lists_and_fields(golden_dir):
    """the reference's time loop WITH mesh adaptation every step (Simulation::advance 15306-15326: adaptMesh,
    then the pipeline), device vs oracle, both solving to 1e-9 / 1e-8: the block lists (level, Z) agree bit-exactly at every
    step and the fields to solver round-off."""
    bpd, lmax, bc, ext, nu, cfl = (2, 2, 2), 3, ("periodic", "wall", "freespace"), 2 * np.pi, 0.05, 0.3
    g0 = cu.Grid(bpd, lmax, 0, ext, bc)
    lv, zs = g0.tables[:, 0].astype(np.int32), g0.tables[:, 1].copy()
    m = O.OracleMesh(bpd, lmax, ext, bc, lv, zs)
    geom = g0.geom
    ax = np.arange(8) + 0.5
    vel = np.zeros((m.nb, 8, 8, 8, 3))
    for b in range(m.nb):   # a compact vortex: refinement stays local
        h = geom[b, 0]
        Z, Y, X = np.meshgrid(geom[b, 3] + ax * h, geom[b, 2] + ax * h, geom[b, 1] + ax * h, indexing="ij")
        gss = np.exp(-((X - 2.5) ** 2 + (Y - 3.0) ** 2 + (Z - 3.2) ** 2) / 0.8)
        vel[b, ..., 0], vel[b, ..., 1], vel[b, ..., 2] = -(Y - 3.0) * gss, (X - 2.5) * gss, 0.2 * gss
    pres = np.zeros((m.nb, 8, 8, 8))
    sim = cu.SimulationData(bpdx=2, bpdy=2, bpdz=2, levelMax=lmax, levelStart=0, extent=ext, nu=nu, CFL=cfl, rampup=3, BC_x=bc[0], BC_y=bc[1],
                            BC_z=bc[2], poissonTol=1e-9, poissonTolRel=1e-8)
    sim.upload("vel", vel)
    S = cu.Simulation(sim)
    w = m.vorticity(vel)
    linf = np.sqrt((w ** 2).sum(axis=-1)).reshape(m.nb, -1).max(axis=1)
    rt, ct = float(np.quantile(linf, 0.6)), float(np.quantile(linf, 0.4))   # the mesh grows 8 -> 29 -> 85 -> 92 blocks
    dt, coefU, nblocks = 0.0, np.array([1.5, -2.0, 0.5]), []
    for n in range(6):
        # oracle: adaptMesh, calcMaxTimestep, AdvectionDiffusion, PressureProjection
        st = m.valid_states(m.tag(m.vorticity(vel), rt, ct))
        if (st != 0).any():
            m2 = m.adapted(st)
            vel, pres, m = m.transfer(m2, vel), m.transfer(m2, pres), m2
        dt = O.lib().orc_calc_dt(sim.hmin, m.max_u(vel), nu, cfl, n, 3, dt, coefU)
        vel, _ = m.advect_diffuse(vel, dt, nu, (0, 0, 0))
        m.project(vel, pres, dt, n, tol=1e-9, tol_rel=1e-8)
        # device
        st_dev = S.adaptMesh(rt, ct)
        assert np.array_equal(st_dev, st), n
        assert np.array_equal(S.sim.grid.tables, m.tables), n       # the block list, bit-exactly
        dt_dev = S.calcMaxTimestep()
        assert abs(dt_dev - dt) <= 1e-9 * dt
        S.advance(dt_dev)
        nblocks.append(m.nb)
        assert np.abs(S.sim.download("vel") - vel).max() <= 1e-7, n
    assert len(set(nblocks)) >= 3 and len(set(m.tables[:, 0].tolist())) >= 2   # the mesh kept changing and is multi-level
    assert np.abs(S.sim.download("pres") - pres).max() <= 1e-5 * max(np.abs(pres).max(), 1e-12)


# ------------------------------------------------------------------ randomised meshes
def random_mesh(seed):
    """A multi-level mesh grown by the ORACLE's adaptMesh from level 0 under random octet-amplitude fields (refinement and
    compression), random box shape (incl. one block across a periodic direction), boundary conditions and depth."""
    rng = np.random.default_rng(seed)
    bpd = tuple(int(v) for v in rng.choice([1, 2, 3], 3))
    if bpd == (1, 1, 1):
        bpd = (2, 1, 2)
    lmax, bc, passes = int(rng.choice([3, 4])), tuple(str(b) for b in rng.choice(["periodic", "wall", "freespace"], 3)), int(rng.choice([2, 3]))
    g0 = cu.Grid(bpd, lmax, 0, EXT, bc)
    m = O.OracleMesh(bpd, lmax, EXT, bc, g0.tables[:, 0].astype(np.int32), g0.tables[:, 1].copy())
    for p in range(passes):
        vel, _ = _octet_fields(m.tables, seed + p)
        w = m.vorticity(vel)
        linf = np.sqrt((w ** 2).sum(axis=-1)).reshape(m.nb, -1).max(axis=1)
        st = m.valid_states(m.tag(w, float(np.quantile(linf, 0.6)), float(np.quantile(linf, 0.3))))
        if (st != 0).any():
            m = m.adapted(st)
    return bpd, lmax, bc, m


@pytest.mark.parametrize("seed", [101, 202, 303, 404, 505, 606, 707, 808])
def test_random_meshes_everything_bitexact(seed):
    bpd, lmax, bc, m = random_mesh(seed)
    if m.nb > 1200 or len(set(m.tables[:, 0].tolist())) < 2:
        pytest.skip("mesh too large / single level for this seed")
    lv, zs = m.tables[:, 0].astype(np.int32), m.tables[:, 1].copy()
    sim = cu.SimulationData(bpdx=bpd[0], bpdy=bpd[1], bpdz=bpd[2], levelMax=lmax, levelStart=0, extent=EXT, BC_x=bc[0], BC_y=bc[1], BC_z=bc[2],
                            leaves=(lv, zs), nu=0.02, uinf=(0.1, -0.2, 0.3))
    assert np.array_equal(sim.grid.tables, m.tables)
    rng = np.random.default_rng(seed)
    vel, pres = rng.uniform(-1, 1, (m.nb, 8, 8, 8, 3)), rng.uniform(-1, 1, (m.nb, 8, 8, 8))
    faces, _, _ = sim.grid.interface()
    # ghost slabs
    for field, arr, w, s, e in (("vel", vel, 3, -3, 4), ("pres", pres, 1, -1, 2)):
        sim.upload(field, arr)
        nc = 3 if field == "vel" else 1
        got = np.zeros((len(faces), nc, w, 64))
        check(lib().cup3d_debug_amr_slabs(sim.handle, FIELDS[field], w, got))
        labs = m.labs(arr, s, e, False)
        for ei, (sf, kind) in enumerate(faces):
            for gl in range(w):
                assert np.array_equal(got[ei, :, gl, :], slab_from_tile(labs[sf // 6], s, sf % 6, gl)), (field, ei, kind, gl)
    # operators
    dt = 0.01
    cu.AdvectionDiffusion(sim)(dt)
    v, _ = m.advect_diffuse(vel, dt, 0.02, (0.1, -0.2, 0.3))
    assert np.array_equal(sim.download("vel"), v)
    sim.bMeanConstraint = 0
    sim.upload("pres", pres)
    cu.ComputeLHS(sim)(0)
    assert np.array_equal(sim.download("lhs"), m.lhs(pres, 0))
    check(lib().cup3d_grad_p(sim.handle, dt))
    assert np.array_equal(sim.download("tmpV"), m.grad_p(pres, dt))
    # adaptMesh
    velo, preso = _octet_fields(m.tables, seed + 50)
    sim.upload("vel", velo); sim.upload("pres", preso)
    w = m.vorticity(velo)
    linf = np.sqrt((w ** 2).sum(axis=-1)).reshape(m.nb, -1).max(axis=1)
    rt, ct = float(np.quantile(linf, 0.7)), float(np.quantile(linf, 0.35))
    S = cu.Simulation(sim)
    st = S.adaptMesh(rt, ct)
    st_o = m.valid_states(m.tag(w, rt, ct))
    assert np.array_equal(st, st_o)
    if (st != 0).any():
        m2 = m.adapted(st_o)
        assert np.array_equal(S.sim.grid.tables, m2.tables)
        assert np.array_equal(S.sim.download("vel"), m.transfer(m2, velo))
        assert np.array_equal(S.sim.download("pres"), m.transfer(m2, preso))


def test_flux_correction_is_conservative_on_a_large_periodic_mesh():
    """Size-independent property of the multi-level operators: on a periodic mesh the flux-corrected operators are discrete
    divergences, so their sum over all cells vanishes to round-off (it is O(1) per interface cell without the correction).
    The mesh (several thousand blocks on three levels) is grown by the device's own adaptMesh."""
    ext, lmax = 2 * np.pi, 6
    sim = cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=lmax, levelStart=3, extent=ext, BC_x="periodic", BC_y="periodic", BC_z="periodic",
                            bMeanConstraint=0)
    g = sim.grid
    ax = np.arange(8) + 0.5
    X = (g.index[:, 0, None] * 8 + ax[None, :])[:, None, None, :] * g.h
    Y = (g.index[:, 1, None] * 8 + ax[None, :])[:, None, :, None] * g.h
    Z = (g.index[:, 2, None] * 8 + ax[None, :])[:, :, None, None] * g.h
    gss = np.exp(-((X - 2.6) ** 2 + (Y - 3.1) ** 2 + (Z - 3.4) ** 2) / 0.6)
    sim.upload("vel", np.ascontiguousarray(np.stack([-(Y - 3.1) * gss, (X - 2.6) * gss, 0.3 * gss + 0 * X], axis=-1)))
    S = cu.Simulation(sim)
    for _ in range(2):
        cu.ComputeVorticity(S.sim)(0)
        w = S.sim.download("tmpV")
        linf = np.sqrt((w ** 2).sum(axis=-1)).reshape(S.sim.nblocks, -1).max(axis=1)
        S.adaptMesh(float(np.quantile(linf, 0.7)), -1.0)
    sim = S.sim
    t = sim.grid.tables
    assert sim.nblocks > 1000 and len(set(t[:, 0].tolist())) == 3
    rng = np.random.default_rng(4)
    p = rng.uniform(-1, 1, (sim.nblocks, 8, 8, 8))
    v = rng.uniform(-1, 1, (sim.nblocks, 8, 8, 8, 3))
    sim.bMeanConstraint = 0
    sim.upload("pres", p)
    cu.ComputeLHS(sim)(0)
    lhs = sim.download("lhs")
    assert abs(lhs.sum()) <= 1e-11 * np.abs(lhs).sum()
    sim.upload("vel", v); sim.fill("chi", 0.0); sim.fill("tmpV", 0.0)
    check(lib().cup3d_pressure_rhs(sim.handle, 0.01))
    rhs = sim.download("lhs")
    assert abs(rhs.sum()) <= 1e-11 * np.abs(rhs).sum()
    check(lib().cup3d_div_pressure(sim.handle))
    dp = sim.download("tmpV")[..., 0]
    assert abs(dp.sum()) <= 1e-11 * np.abs(dp).sum()
    # momentum: the diffusive part of the advect-diffuse increment is conservative; switch advection off via a huge... (not separable:
    # the upwind advection term is not in conservation form in the reference either), so only the three divergences above are asserted


# ------------------------------------------------------------------ obstacle operators
@pytest.mark.parametrize("name", ["f16_mixed", "amr_periodic_l01"])
def test_obstacle_operators(golden_dir, name):
    """cup3d_penalization / cup3d_update_tmpv / projection with chi and udef against the reference's golden (synthetic obstacle):
    velocities and tmpV bit-exact, force and torque to summation order, projection to solver tolerance."""
    z = np.load(os.path.join(golden_dir, "obstacle_ops.npz"))
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    t = g["tables"]
    bpd, lmax, bc = tuple(int(b) for b in g["bpd"]), int(g["level_max"]), tuple(BCN[int(b)] for b in g["bc"])
    m = O.OracleMesh(bpd, lmax, float(g["extent"]), bc, t[:, 0], t[:, 1])
    sim = cu.SimulationData(bpdx=bpd[0], bpdy=bpd[1], bpdz=bpd[2], levelMax=lmax, levelStart=0, extent=float(g["extent"]), BC_x=bc[0], BC_y=bc[1],
                            BC_z=bc[2], leaves=(t[:, 0], t[:, 1]), poissonTol=1e-12, poissonTolRel=1e-10)
    assert np.array_equal(sim.grid.tables, t)
    dt, lam, implicit, step = z[name + "_par"]
    vel, pres, chif = z[name + "_vel_in"], z[name + "_pres_in"], z[name + "_chi_field"]
    rigid = z[name + "_rigid"]
    ob = cu.ObstacleData(z[name + "_ids"], z[name + "_ochi"], z[name + "_oudef"], rigid[0:3], rigid[3:6], rigid[6:9])
    sim.obstacles, sim.lambda_penal, sim.bImplicitPenalization = [ob], float(lam), bool(implicit)
    sim.upload("vel", vel); sim.upload("chi", chif)
    cu.Penalization(sim)(float(dt))
    assert np.array_equal(sim.download("vel"), z[name + "_pen_vel"])
    f6 = z[name + "_force6"]
    assert np.abs(np.concatenate([ob.force, ob.torque]) - f6).max() <= 1e-12 * np.abs(f6).max()
    # kernelUpdateTmpV
    obst = dict(ids=z[name + "_ids"], chi=z[name + "_ochi"], udef=z[name + "_oudef"], rigid=rigid)
    sim.fill("tmpV", 0.0)
    check(lib().cup3d_update_tmpv(sim.handle, 1, cu.operators._obstacle_array([ob])))
    assert np.array_equal(sim.download("tmpV"), m.update_tmpv(np.zeros_like(vel), chif, obst))
    # projection with the obstacle, both sides solved tightly
    sim.upload("vel", vel); sim.upload("pres", pres)
    sim.step = int(step)
    cu.PressureProjection(sim)(float(dt))
    v, p = vel.copy(), pres.copy()
    m.project_obst(v, p, float(dt), int(step), chif, obst, tol=1e-12, tol_rel=1e-10)
    corr = np.abs(v - vel).max()
    assert np.abs(sim.download("pres") - p).max() <= 1e-6 * np.abs(p).max()
    assert np.abs(sim.download("vel") - v).max() <= 1e-6 * corr


# ------------------------------------------------------------------ implicit diffusion (AdvectionDiffusionImplicit, main.cpp:10030-10119)
IMPLICIT = ["uniform_mixed", "uniform_periodic", "amr_periodic_l01", "amr_mixed_l12", "synthetic_l012"]


def make_implicit(golden_dir, name, **kw):
    if name.startswith("uniform"):
        bpd, lmax = (2, 1, 2), 3
        bc = ("freespace", "wall", "periodic") if name == "uniform_mixed" else ("periodic",) * 3
        sim = cu.SimulationData(bpdx=bpd[0], bpdy=bpd[1], bpdz=bpd[2], levelMax=lmax, levelStart=1, extent=EXT, BC_x=bc[0], BC_y=bc[1], BC_z=bc[2], **kw)
        t = sim.grid.tables
        m = O.OracleMesh(bpd, lmax, EXT, bc, t[:, 0], t[:, 1])
        rng = np.random.default_rng(6)
        f = dict(vel=rng.uniform(-1, 1, (m.nb, 8, 8, 8, 3)), pres=rng.uniform(-1, 1, (m.nb, 8, 8, 8)), rhs=rng.uniform(-1, 1, (m.nb, 8, 8, 8)))
    else:
        m, sim, f = make(golden_dir, name, **kw)
    f["vel"] = 0.5 * f["vel"]
    return m, sim, f


@pytest.mark.parametrize("name", IMPLICIT)
def test_implicit_diffusion_stencils_bitexact(golden_dir, name):
    """KernelAdvect (order-independent reading: every tile from the velocity on entry -- the oracle's sequential=False; its
    sequential=True mode is pinned against the reference), KernelDiffusionRHS and KernelLHSDiffusion for the three boundary
    directions: bit-exact, flux correction at coarse/fine faces included."""
    m, sim, f = make_implicit(golden_dir, name)
    dt, nu, uinf = 0.05, 2.0, (0.1, -0.2, 0.3)
    sim.nu, sim.uinf = nu, np.array(uinf)
    sim.upload("vel", f["vel"])
    check(lib().cup3d_advect_implicit(sim.handle, dt, nu, sim.uinf))
    v, t = m.advect_implicit(f["vel"], dt, nu, uinf, False)
    assert np.array_equal(sim.download("tmpV"), t)
    assert np.array_equal(sim.download("vel"), v)
    sim.upload("vel", f["vel"])
    check(lib().cup3d_diffusion_rhs(sim.handle))
    assert np.array_equal(sim.download("tmpV"), m.diffusion_rhs(f["vel"]))
    ds = cu.DiffusionSolver(sim)
    ds.dt = dt
    out = []
    for d in range(3):
        ds.mydirection = d
        sim.upload("pres", f["pres"])
        ds.lhs()
        out.append(sim.download("lhs"))
        assert np.array_equal(out[-1], m.diff_lhs(f["pres"], d, dt, nu)), d
    if name != "uniform_periodic" and name != "amr_periodic_l01":
        assert not np.array_equal(out[0], out[1])  # the boundary rule of the component matters


@pytest.mark.parametrize("name", IMPLICIT)
def test_implicit_diffusion_solver(golden_dir, name):
    """Helmholtz block CG, DiffusionSolver::solve and the whole AdvectionDiffusionImplicit step against the oracle: dot products and
    the block-CG inner products are summed in another order on the device, so solver tolerance (both sides solved tightly)."""
    m, sim, f = make_implicit(golden_dir, name, diffusionTol=1e-12, diffusionTolRel=1e-11)
    dt, nu, uinf = 0.05, 2.0, (0.1, -0.2, 0.3)
    sim.nu, sim.uinf = nu, np.array(uinf)
    ds = cu.DiffusionSolver(sim)
    ds.dt, ds.mydirection = dt, 1
    sim.upload("pres", f["pres"])
    ds.preconditioner()
    z = m.diff_precond(f["pres"], dt, nu)
    assert np.abs(sim.download("pres") - z).max() <= 2e-5 * np.abs(z).max()
    sim.upload("lhs", f["rhs"]); sim.upload("pres", f["pres"])
    r = ds.solve()
    x, info = m.diff_solve(f["rhs"], f["pres"], 1, dt, nu, 1e-12, 1e-11)
    assert r.iterations > 5 and info.iters > 5
    assert np.abs(sim.download("pres") - x).max() <= 1e-7 * np.abs(x).max()
    res = np.linalg.norm((f["rhs"] - m.diff_lhs(sim.download("pres"), 1, dt, nu)).ravel())
    res0 = np.linalg.norm((f["rhs"] - m.diff_lhs(f["pres"], 1, dt, nu)).ravel())
    assert res <= 1e-7 * res0   # true residual of the pipelined recurrences (refreshed every 50 iterations only)
    # default tolerances: the returned iterate satisfies the reference's stopping rule under the ORACLE's operator
    sim.DiffusionErrorTol, sim.DiffusionErrorTolRel = 1e-6, 1e-4
    sim.upload("lhs", f["rhs"]); sim.upload("pres", f["pres"])
    r = ds.solve()
    res = np.linalg.norm((f["rhs"] - m.diff_lhs(sim.download("pres"), 1, dt, nu)).ravel())
    assert res <= 1.05 * max(1e-6, 1e-4 * res0) and r.iterations >= 2
    # the whole step
    sim.DiffusionErrorTol, sim.DiffusionErrorTolRel = 1e-12, 1e-11
    sim.upload("vel", f["vel"]); sim.upload("pres", f["pres"])
    op = cu.AdvectionDiffusionImplicit(sim)
    res3 = op(dt)
    v, it = m.advdiff_implicit(f["vel"], f["pres"], dt, nu, uinf, False, 1e-12, 1e-11)
    assert np.array_equal(sim.download("pres"), f["pres"])          # pres is scratch and comes back
    change = np.abs(v - f["vel"]).max()
    assert change > 1e-3
    assert np.abs(sim.download("vel") - v).max() <= 1e-6 * change
    assert all(r.iterations > 3 for r in res3)


def test_implicit_diffusion_single_block_equals_reference_order():
    """On one periodic block the reference's in-place KernelAdvect cannot be seen by any other tile: the device result must then
    equal the oracle's sequential (= reference, one thread) mode as well."""
    bc = ("periodic",) * 3
    sim = cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=1, levelStart=0, extent=EXT, BC_x=bc[0], BC_y=bc[1], BC_z=bc[2])
    assert sim.nblocks == 1
    m = O.OracleMesh((1, 1, 1), 1, EXT, bc, [0], [0])
    rng = np.random.default_rng(3)
    vel = rng.uniform(-1, 1, (1, 8, 8, 8, 3))
    sim.upload("vel", vel)
    check(lib().cup3d_advect_implicit(sim.handle, 0.05, 2.0, np.array([0.1, 0.2, -0.3])))
    v, t = m.advect_implicit(vel, 0.05, 2.0, (0.1, 0.2, -0.3), True)
    assert np.array_equal(sim.download("vel"), v) and np.array_equal(sim.download("tmpV"), t)


def test_implicit_time_loop(golden_dir):
    """Simulation with -implicitDiffusion: the pipeline picks AdvectionDiffusionImplicit (15231-15234), calcMaxTimestep uses the
    0.1 diffusive limit after step 10 (15269-15273); three steps on a multi-level mesh against the oracle's operators."""
    m, sim, f = make_implicit(golden_dir, "amr_mixed_l12", implicitDiffusion=True, nu=0.5, CFL=0.3, diffusionTol=1e-12, diffusionTolRel=1e-11,
                              poissonTol=1e-12, poissonTolRel=1e-10, rampup=0)
    S = cu.Simulation(sim)
    assert isinstance(S.pipeline[0], cu.AdvectionDiffusionImplicit)
    sim.upload("vel", f["vel"]); sim.fill("pres", 0.0)
    v, p = f["vel"].copy(), np.zeros_like(f["pres"])
    coefU = np.zeros(3)
    dt_old = 0.0
    for step in range(3):
        dt = S.calcMaxTimestep()
        ref_dt = O.lib().orc_calc_dt2(sim.hmin, m.max_u(v, sim.uinf), sim.nu, sim.CFL, step, 0, dt_old, coefU, 1)
        assert abs(dt - ref_dt) <= 1e-6 * ref_dt   # from findMaxU of trajectories that agree to solver round-off
        S.advance(dt)
        v, _ = m.advdiff_implicit(v, p, dt, sim.nu, sim.uinf, False, 1e-12, 1e-11)
        m.project(v, p, dt, step, tol=1e-12, tol_rel=1e-10)
        dt_old = dt
        assert np.abs(sim.download("vel") - v).max() <= 1e-6 * np.abs(v).max(), step
    sim.step = 11
    assert S.calcMaxTimestep() <= 0.1


@pytest.mark.parametrize("block_solver", [0, 1])
@pytest.mark.parametrize("mc", [0, 1, 2, 3])
@pytest.mark.parametrize("bc", [("wall", "freespace", "wall"), ("periodic", "periodic", "wall")])
def test_lhs_inside_the_loop_kernels_on_a_multilevel_mesh_is_bit_identical(bc, mc, block_solver):
    """The `fuse_lhs_ml` A/B (test builds; measured no faster, so production keeps one k_lhs launch per LHS on multi-level meshes): the
    blocks none of whose six faces is a coarse/fine interface form v = A zhat and t = A what inside the fused loop kernels, like every block
    of a uniform grid; only the interface blocks keep k_lhs + ghost slabs + flux correction (launch_lhs on the interface list).  Against
    the production solve: t and v are the same bits -- same stencil association, same mean-constraint rows (9299-9326) on the same cells -- hence every iterate, the iteration
    count and the returned pressure.  512 level-2 blocks of which a few are refined: most blocks are plain, the corner block too."""
    bpd, lmax = (2, 2, 2), 4
    refine = [(0, i, j, k) for k in range(2) for j in range(2) for i in range(2)] + [(1, i, j, k) for k in range(4) for j in range(4) for i in range(4)]
    refine += [(2, 3, 3, 3), (2, 4, 4, 4), (2, 6, 1, 2)]
    lv, zs = O.build_balanced_mesh(bpd, lmax, bc, refine)
    assert len(set(lv.tolist())) >= 2 and len(lv) > 500
    rng = np.random.default_rng(17 + mc)
    res = {}
    for opt in (0, 1):
        check(lib().cup3d_debug_set_option(b"fuse_lhs_ml", opt))
        try:
            sim = cu.SimulationData(bpdx=bpd[0], bpdy=bpd[1], bpdz=bpd[2], levelMax=lmax, levelStart=0, extent=EXT, BC_x=bc[0], BC_y=bc[1], BC_z=bc[2],
                                    leaves=(lv, zs), bMeanConstraint=mc, poissonTol=1e-9, poissonTolRel=1e-7, blockSolver=block_solver)
            if opt == 0:
                rhs = rng.uniform(-1, 1, (sim.nblocks, 8, 8, 8))
            sim.upload("lhs", rhs)
            sim.fill("pres", 0.0)
            r = cu.makePoissonSolver(sim).solve()
            res[opt] = (r.iterations, r.restarts, r.norm, sim.download("pres"))
        finally:
            check(lib().cup3d_debug_set_option(b"fuse_lhs_ml", 0))
    assert res[0][0] > 3
    assert res[0][:3] == res[1][:3], (res[0][:3], res[1][:3])
    assert np.array_equal(res[0][3], res[1][3])


@pytest.mark.parametrize("name", MESHES)
def test_multigrid_smoother_forms_agree_on_multilevel_meshes(golden_dir, name):
    """The wavefront-per-block smoother (k_mg_smooth_wave: shuffles, no LDS tile) against the workgroup-per-block form on the octree's
    levels -- coarse/fine ghost slabs behind interface faces included: same bits, same iteration count."""
    res = {}
    for opt in (0, 1):
        check(lib().cup3d_debug_set_option(b"mg_smooth_workgroup", opt))
        try:
            m, sim, f = make(golden_dir, name, blockSolver=5, poissonTol=1e-10, poissonTolRel=1e-9)
            sim.upload("lhs", f["rhs"]); sim.upload("pres", f["pres"])
            r = cu.makePoissonSolver(sim).solve()
            res[opt] = (r.iterations, r.norm, sim.download("pres"))
        finally:
            check(lib().cup3d_debug_set_option(b"mg_smooth_workgroup", 0))
    assert res[0][0] >= 2 and res[0][:2] == res[1][:2], (res[0][:2], res[1][:2])
    assert np.array_equal(res[0][2], res[1][2])
