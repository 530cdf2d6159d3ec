"""Mesh-adaptation block operators on the device (cup3d_restrict / cup3d_prolong / cup3d_tag_blocks) against
the reference's own adaptMesh (golden vectors from tests/golden/make_golden.py: every block refined, then every
block compressed) and against the oracle on larger seeded meshes.  Bit-exact: same IEEE operations, same order."""
import os

import numpy as np
import pytest

import cup3d_amd as cu
import oracle_lib as O
from test_oracle_golden import ADAPT_CASES, clamp_tags

pytestmark = pytest.mark.gpu
BCN = {0: "freespace", 1: "periodic", 2: "wall"}


@pytest.fixture(scope="module", autouse=True)
def _device():
    cu.device_init(0)


def sims(bpd, lmax, ext, bc):
    mk = lambda lvl: cu.SimulationData(bpdx=bpd[0], bpdy=bpd[1], bpdz=bpd[2], levelMax=lmax, levelStart=lvl, extent=ext,  # noqa: E731
                                       BC_x=bc[0], BC_y=bc[1], BC_z=bc[2])
    return mk


@pytest.mark.parametrize("name", ADAPT_CASES)
def test_golden_refine_and_compress(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    bc = [BCN[int(b)] for b in z["bc"]]
    mk = sims([int(b) for b in z["bpd"]], int(z["level_max"]), float(z["extent"]), bc)
    coarse, fine = mk(0), mk(1)
    assert np.array_equal(fine.grid.tables, z["tables_fine"]) and np.array_equal(coarse.grid.tables, z["tables_coarse"])
    coarse.upload("vel", coarse.grid.to_blocks(z["vel_in"]))
    coarse.upload("pres", coarse.grid.to_blocks(z["pres_in"]))
    amr = cu.MeshAdaptation(float(z["tag_rtol"]), float(z["tag_ctol"]))
    assert np.array_equal(amr.Tag(coarse, "vel"), clamp_tags(z["tags"], 0, int(z["level_max"])))
    amr.refine(coarse, fine, "vel")
    amr.refine(coarse, fine, "pres")
    assert np.array_equal(fine.download("vel"), z["vel_fine"])
    assert np.array_equal(fine.download("pres"), z["pres_fine"])
    coarse.fill("vel", 0.0)
    coarse.fill("pres", 0.0)
    amr.compress(fine, coarse, "vel")
    amr.compress(fine, coarse, "pres")
    assert np.array_equal(coarse.download("vel"), z["vel_coarse"])
    assert np.array_equal(coarse.download("pres"), z["pres_coarse"])


@pytest.mark.parametrize("bpd,lmax,lc,bc", [((1, 1, 1), 5, 2, ("freespace", "wall", "periodic")), ((3, 2, 1), 3, 1, ("wall", "wall", "wall")),
                                            ((1, 1, 1), 6, 4, ("periodic", "periodic", "periodic"))])
def test_oracle_refine_compress_tag(bpd, lmax, lc, bc):
    ext = 2 * np.pi
    oc, of = O.OracleGrid(bpd, lmax, lc, ext, bc), O.OracleGrid(bpd, lmax, lc + 1, ext, bc)
    mk = sims(bpd, lmax, ext, bc)
    coarse, fine = mk(lc), mk(lc + 1)
    rng = np.random.default_rng(lc)
    v = rng.uniform(-1, 1, (oc.nb, 8, 8, 8, 3)) * rng.uniform(0.05, 1.0, (oc.nb, 1, 1, 1, 1))
    p = rng.uniform(-1, 1, (oc.nb, 8, 8, 8))
    coarse.upload("vel", v)
    coarse.upload("pres", p)
    for rt, ct in ((0.9, 0.3), (0.2, 0.1), (5.0, 4.0)):
        assert np.array_equal(cu.MeshAdaptation(rt, ct).Tag(coarse, "vel"), O.tag_blocks(oc, v, rt, ct))
        assert np.array_equal(cu.MeshAdaptation(rt, ct).Tag(coarse, "pres"), O.tag_blocks(oc, p, rt, ct))
    cu.MeshAdaptation.refine(coarse, fine, "vel")
    cu.MeshAdaptation.refine(coarse, fine, "pres")
    vf, pf = fine.download("vel"), fine.download("pres")
    assert np.array_equal(vf, O.prolong_field(oc, of, v)) and np.array_equal(pf, O.prolong_field(oc, of, p))
    cu.MeshAdaptation.compress(fine, coarse, "vel")
    cu.MeshAdaptation.compress(fine, coarse, "pres")
    assert np.array_equal(coarse.download("vel"), O.restrict_field(of, oc, vf))
    assert np.array_equal(coarse.download("pres"), O.restrict_field(of, oc, pf))
    # property: compress(refine(u)) reproduces u up to the second-derivative term of the Taylor expansion, exactly
    # for fields that are constant per block neighbourhood
    const = np.full_like(v, 0.75)
    coarse.upload("vel", const)
    cu.MeshAdaptation.refine(coarse, fine, "vel")
    if all(b == "periodic" for b in bc):
        assert np.array_equal(fine.download("vel"), np.full((of.nb, 8, 8, 8, 3), 0.75))
