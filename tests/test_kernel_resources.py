"""What the compiler made of the production kernels, read from the code objects inside the built libraries (no GPU): registers -> wavefronts
per SIMD, LDS -> workgroups per CU, scratch.  DESIGN.md section 4 argues with these numbers (110 / 128 registers -> 4 wavefronts for the two
fused BiCGSTAB kernels with the LHS inside, 5 for the block CG, three 48.8 KB advect-diffuse workgroups per CU); this keeps them true, and
keeps register spills from creeping into a hot loop unnoticed when a kernel is edited.  `python scripts/kernel_resources.py` prints the table."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import kernel_resources as KR  # noqa: E402

RELEASE = os.path.join(ROOT, "cup3d_amd", "libcup3d_hip.so")
TESTING = os.path.join(ROOT, "cup3d_amd", "libcup3d_hip_testing.so")
LDS_PER_CU = 160 * 1024


@pytest.fixture(scope="module")
def release():
    if not os.path.exists(RELEASE):
        import __graft_entry__ as G
        G.build()
    return {r["kernel"]: r for r in KR.kernels(RELEASE)}


def test_every_translation_unit_carries_gfx950_code(release):
    triples = {t for t, _ in KR.code_objects(RELEASE)}
    assert triples and all("gfx950" in t for t in triples), triples   # one architecture, no second code path
    assert len(release) > 90


def test_fused_solver_kernels(release):
    """k_loop1_cg / k_loop2_cg_w4 <FMA, EV = 0, FLHS = true>: the two launches of a BiCGSTAB iteration on uniform grids"""
    l1, l2 = release["k_loop1_cg<b1,i6,b1>"], release["k_loop2_cg_w4<b1,i6,b1>"]
    for k in (l1, l2):
        assert k["scratch_bytes"] == 0 and k["vgpr_spills"] == 0 and k["agpr"] == 0, k
        assert k["lds_bytes"] == 10 * 96 * 8, k              # the ghosted tile: 10 planes of pitch 96 doubles, one wavefront per workgroup
        assert k["waves_per_simd"] == 4 and k["max_workgroup"] == 64, k
    assert l1["vgpr"] <= 112 and l2["vgpr"] <= 128, (l1["vgpr"], l2["vgpr"])
    # LDS would allow 21 wavefronts per CU, the registers 16: registers bound the occupancy, as DESIGN.md says
    assert LDS_PER_CU // l1["lds_bytes"] >= 4 * 4
    # round 6: launches of >= 131072 blocks (the headline's 262144) take the first kernel held to 5 wavefronts per SIMD -- without a spill
    w5 = release["k_loop1_cg_w5<b1,i6,b1>"]
    assert w5["scratch_bytes"] == 0 and w5["vgpr_spills"] == 0 and w5["waves_per_simd"] == 5 and w5["vgpr"] <= 96, w5
    assert w5["lds_bytes"] == l1["lds_bytes"] and LDS_PER_CU // w5["lds_bytes"] >= 5 * 4


def test_fused_solver_kernels_with_the_totals_inside(release):
    """k_loop1_cg_tot / k_loop2_cg_tot <FMA, EV = 0>: the same two kernels totalling their per-block values themselves (Arrive, poisson.hip) --
    the early all-reduce over ranks (CUP3D_EARLY_ALLREDUCE=1).  The arrival code sits behind the plane loop; the second kernel is HELD to 128
    registers (the compiler would take 136 -> 3 wavefronts per SIMD): one 8-byte value is parked in scratch before the plane loop and fetched
    back when the block CG starts -- never inside a loop (the ISA's only scratch accesses: one store pair up front, one load pair behind
    the vector phase)."""
    l1, l2 = release["k_loop1_cg_tot<b1,i6>"], release["k_loop2_cg_tot<b1,i6>"]
    for k in (l1, l2):
        assert k["agpr"] == 0 and k["lds_bytes"] == 10 * 96 * 8 and k["max_workgroup"] == 64, k
    assert l1["scratch_bytes"] == 0 and l1["vgpr_spills"] == 0 and l1["vgpr"] <= 96, l1
    assert l2["waves_per_simd"] == 4 and l2["vgpr"] <= 128 and l2["vgpr_spills"] <= 2 and l2["scratch_bytes"] <= 16, l2


def test_fused_solver_kernels_with_the_direct_block_solve(release):
    """k_loop1_fdm / k_loop2_fdm <FLHS = true> (`alt`, block_solver 1): the same kernels with the fast diagonalisation behind the loops.  They
    are the HBM-bound form of the iteration (0.76 / 0.79 of the roof, profiles/r04), which only holds while nothing spills: the second one
    is held to 128 registers (amdgpu_waves_per_eu(4, 8); the compiler would take 136 -> 3 wavefronts per SIMD)."""
    for name in ("k_loop1_fdm<b1>", "k_loop2_fdm<b1>"):
        k = release[name]
        # (round 5: the first one comes out at 94 registers, 5 wavefronts per SIMD, by the compiler's own choice; it was 110 / 4)
        assert k["scratch_bytes"] == 0 and k["vgpr_spills"] == 0 and k["waves_per_simd"] in (4, 5) and k["lds_bytes"] == 10 * 96 * 8 and k["vgpr"] <= 128, k
    for name in ("k_loop1_fdm<b0>", "k_loop2_fdm<b0>"):   # multi-level meshes: no tile, the transposes' 4.5 KB only
        k = release[name]
        assert k["scratch_bytes"] == 0 and k["waves_per_simd"] >= 4 and k["lds_bytes"] == 64 * 9 * 8, k


def test_fused_solver_kernels_without_the_lhs(release):
    """<FMA, 0, FLHS = false>: multi-level meshes (the LHS needs coarse/fine ghosts there) -- 5 wavefronts per SIMD; the second kernel is HELD
    to 96 registers at the price of two spills outside the CG loop (measured faster than 122 registers at 4 wavefronts, profiles/README.md)"""
    l1, l2 = release["k_loop1_cg<b1,i6,b0>"], release["k_loop2_cg<b1,i6,b0>"]
    assert l1["waves_per_simd"] == 5 and l1["scratch_bytes"] == 0 and l1["lds_bytes"] == 5184, l1
    assert l2["waves_per_simd"] == 5 and l2["vgpr_spills"] <= 2 and l2["scratch_bytes"] <= 16 and l2["lds_bytes"] == 5184, l2


def test_block_preconditioner_and_stencils(release):
    for name, k in release.items():
        if name.startswith("k_precond<"):
            assert k["waves_per_simd"] == 5 and k["scratch_bytes"] == 0 and k["lds_bytes"] == 5184 and k["max_workgroup"] == 64, k
    for first in ("b0", "b1"):
        k = release[f"k_advdiff<{first},i2,i0,b0,b0>"]        # <FIRST_STAGE, 2 cells per thread, production variant, uniform grid, explicit>
        assert k["scratch_bytes"] == 0 and k["lds_bytes"] == 48768 and k["max_workgroup"] == 256, k
        assert LDS_PER_CU // k["lds_bytes"] == 3              # three workgroups = 12 wavefronts per CU: the LDS, not the registers, bounds it
        assert k["waves_per_simd"] >= 3, k
    assert release["k_lhs<b0>"]["scratch_bytes"] == 0 and release["k_lhs<b0>"]["waves_per_simd"] == 8


def test_no_other_kernel_uses_scratch(release):
    # the register-held forms of the tests above (FMA-contracted / reference association)
    held = {"k_loop2_cg<b1,i6,b0>", "k_loop2_cg<b0,i0,b0>", "k_loop2_cg_tot<b1,i6>", "k_loop2_cg_tot<b0,i0>"}
    bad = {n: (k["scratch_bytes"], k["vgpr_spills"]) for n, k in release.items() if (k["scratch_bytes"] or k["vgpr_spills"]) and n not in held}
    assert not bad, bad


def test_release_build_has_no_tuning_variants_of_the_block_cg():
    """the A/B evaluations of the block CG (EV != 0, two blocks per wavefront) exist in the testing build only"""
    if not os.path.exists(TESTING):
        pytest.skip("testing flavour not built")
    rel = {r["kernel"] for r in KR.kernels(RELEASE)}
    tst = {r["kernel"] for r in KR.kernels(TESTING)}
    assert rel <= tst, sorted(rel - tst)
    extra = tst - rel
    assert extra and not any(n.startswith("k_precond<") and not n.endswith((",i0>", ",i6>")) for n in rel), sorted(rel)
    assert any(n.startswith("k_precond") for n in extra), sorted(extra)[:10]
    # measured and dropped (profiles/README.md): test builds only
    assert not any(n.startswith(("k_advdiff_c", "k_advdiff_pc", "k_debug")) or n in ("k_loop2_cg<b1,i6,b1>", "k_loop2_cg_w4<b1,i6,b0>") for n in rel), sorted(rel)
