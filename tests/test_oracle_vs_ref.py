"""Pins the oracle restatement against the compiled, unmodified reference
(oracle/_ref/ref_tool) on fresh seeded inputs.  Skipped where the reference binary
is absent (it is built from /root/reference by `make -C oracle ref`)."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.skipif(not O.have_ref_tool(), reason="oracle/_ref/ref_tool not built")

CASES = [
    ((4, 4, 4), 1, 0, ("periodic", "periodic", "periodic"), 21),
    ((1, 1, 1), 3, 2, ("wall", "wall", "wall"), 22),
    ((2, 1, 1), 2, 1, ("periodic", "wall", "freespace"), 23),
    ((1, 2, 3), 2, 1, ("freespace", "freespace", "freespace"), 24),
]


@pytest.mark.parametrize("bpd,lmax,lstart,bc,seed", CASES)
def test_all_ops_bit_exact(bpd, lmax, lstart, bc, seed):
    rng = np.random.default_rng(seed)
    ext = 2 * np.pi
    g = O.OracleGrid(bpd, lmax, lstart, ext, bc)
    NX, NY, NZ = g.ncell
    velg, presg = rng.uniform(-1, 1, (NZ, NY, NX, 3)), rng.uniform(-1, 1, (NZ, NY, NX))
    wd = O.tempfile.mkdtemp(prefix="pin_")
    velg.tofile(os.path.join(wd, "vel_in.bin"))
    presg.tofile(os.path.join(wd, "pres_in.bin"))
    dt, nu, uinf = 0.02, 0.005, (0.3, 0.0, -0.1)
    script = ["tables tables.bin", "zero chi", "loadg vel vel_in.bin", f"set nu {nu}",
              f"set uinfx {uinf[0]}", f"set uinfy {uinf[1]}", f"set uinfz {uinf[2]}",
              f"op advdiff {dt}", "dump vel ad.bin",
              "loadg pres pres_in.bin", "op precond", "dump pres z.bin",
              "loadg vel vel_in.bin", "loadg pres pres_in.bin", "set step 7", f"op project {dt}",
              "dump vel prv.bin", "dump pres prp.bin"]
    recs, wd = O.run_ref(script, O.ref_args(bpd, lmax, lstart, ext, bc), threads=1, workdir=wd)
    t, geom = O.read_tables(os.path.join(wd, "tables.bin"))
    assert np.array_equal(t, g.tables) and np.array_equal(geom, g.geom)
    vel, pres = g.to_blocks(velg), g.to_blocks(presg)
    v, tv = vel.copy(), np.zeros_like(vel)
    g.advect_diffuse(v, tv, dt, nu, uinf)
    assert np.array_equal(v, O.read_blocks(os.path.join(wd, "ad.bin"), g.nb, 3))
    p = pres.copy()
    g.precond(p)
    assert np.array_equal(p, O.read_blocks(os.path.join(wd, "z.bin"), g.nb, 1))
    v, p = vel.copy(), pres.copy()
    info, _, _ = g.project(v, p, dt, 7)
    assert info.iters == int([r for r in recs if r["op"] == "project"][0]["iters"])
    assert np.array_equal(v, O.read_blocks(os.path.join(wd, "prv.bin"), g.nb, 3))
    assert np.array_equal(p, O.read_blocks(os.path.join(wd, "prp.bin"), g.nb, 1))


@pytest.mark.parametrize("bc", [("periodic", "periodic", "periodic"), ("wall", "freespace", "periodic"), ("freespace", "wall", "wall")])
def test_mesh_adaptation_operators_bit_exact(bc):
    """Whole-mesh refine / compress through the reference's adaptMesh vs the oracle's prolong / restrict."""
    bpd, lmax = (2, 1, 2), 3
    rng = np.random.default_rng(77)
    g0, g1 = O.OracleGrid(bpd, lmax, 0, 2 * np.pi, bc), O.OracleGrid(bpd, lmax, 1, 2 * np.pi, bc)
    NX, NY, NZ = g0.ncell
    velg, presg = rng.uniform(-1, 1, (NZ, NY, NX, 3)), rng.uniform(-1, 1, (NZ, NY, NX))
    wd = O.tempfile.mkdtemp(prefix="pin_")
    velg.tofile(os.path.join(wd, "vel_in.bin"))
    presg.tofile(os.path.join(wd, "pres_in.bin"))
    script = ["loadg vel vel_in.bin", "loadg pres pres_in.bin", "tagvel 1.3 1.1 tags.bin", "amrtol -1 -2", "adapt", "tables t1.bin", "dump vel v1.bin",
              "dump pres p1.bin", "amrtol 1e300 1e299", "adapt", "tables t2.bin", "dump vel v2.bin", "dump pres p2.bin"]
    O.run_ref(script, O.ref_args(bpd, lmax, 0, 2 * np.pi, bc), threads=1, workdir=wd)
    t1, _ = O.read_tables(os.path.join(wd, "t1.bin"))
    t2, _ = O.read_tables(os.path.join(wd, "t2.bin"))
    assert np.array_equal(t1, g1.tables) and np.array_equal(t2, g0.tables)
    v1, p1 = O.read_blocks(os.path.join(wd, "v1.bin"), g1.nb, 3), O.read_blocks(os.path.join(wd, "p1.bin"), g1.nb, 1)
    assert np.array_equal(O.prolong_field(g0, g1, g0.to_blocks(velg)), v1)
    assert np.array_equal(O.prolong_field(g0, g1, g0.to_blocks(presg)), p1)
    assert np.array_equal(O.restrict_field(g1, g0, v1), O.read_blocks(os.path.join(wd, "v2.bin"), g0.nb, 3))
    assert np.array_equal(O.restrict_field(g1, g0, p1), O.read_blocks(os.path.join(wd, "p2.bin"), g0.nb, 1))
    raw = np.fromfile(os.path.join(wd, "tags.bin"), dtype=np.int8)
    raw[raw == -1] = 0  # level 0: TagBlocksVector's clamp
    assert np.array_equal(O.tag_blocks(g0, g0.to_blocks(velg), 1.3, 1.1), raw)


@pytest.mark.skipif(not O.have_ref_tool_mpi(), reason="oracle/_ref/ref_tool_mpi (reference against a real MPI) not built")
@pytest.mark.parametrize("nranks,bpd,lmax,lstart,bc", [(3, (2, 2, 2), 2, 1, ("periodic", "wall", "freespace")), (2, (1, 2, 3), 2, 1, ("freespace",) * 3),
                                                       (5, (2, 2, 2), 3, 2, ("periodic",) * 3)])
def test_reference_on_several_ranks_equals_the_one_rank_oracle(nranks, bpd, lmax, lstart, bc):
    """The reference run on 2-5 ranks of a real MPI: the advect-diffuse operator (three RK stages, three halo exchanges) is bit
    for bit what the one-rank oracle computes -- sharding does not change the stencil results, which is what the virtual-rank
    device tests assert for the product -- and the projection agrees to the solver's own tolerance (the dot products are summed
    rank by rank)."""
    rng = np.random.default_rng(31)
    ext = 2 * np.pi
    g = O.OracleGrid(bpd, lmax, lstart, ext, bc)
    NX, NY, NZ = g.ncell
    velg = rng.uniform(-1, 1, (NZ, NY, NX, 3))
    wd = O.tempfile.mkdtemp(prefix="pinmpi_")
    velg.tofile(os.path.join(wd, "vel_in.bin"))
    dt, nu = 0.01, 0.02
    script = ["zero chi", "tables t.bin", "loadg vel vel_in.bin", f"set nu {nu}", f"op advdiff {dt}", "dump vel ad.bin",
              "set step 4", f"op project {dt}", "dump vel pr.bin"]
    O.run_ref_mpi(script, O.ref_args(bpd, lmax, lstart, ext, bc), nranks, workdir=wd)
    T = [O.read_tables(os.path.join(wd, f"t.bin.r{r}"))[0] for r in range(nranks)]
    assert np.array_equal(np.concatenate(T), g.tables)
    cat = lambda f: np.concatenate([O.read_blocks(os.path.join(wd, f"{f}.r{r}"), len(T[r]), 3) for r in range(nranks)])  # noqa: E731
    vel = g.to_blocks(velg)
    v, tv = vel.copy(), np.zeros_like(vel)
    g.advect_diffuse(v, tv, dt, nu, (0, 0, 0))
    assert np.array_equal(v, cat("ad.bin"))
    p = np.zeros(vel.shape[:4])
    g.project(v, p, dt, 4)
    assert np.abs(v - cat("pr.bin")).max() <= 5e-3 * np.abs(v).max()
