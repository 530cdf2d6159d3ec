"""Host-side integer contract of the product library: Hilbert indexing,
block order, neighbour tables, partition and halo plan — bit-exact against the golden
tables produced by the compiled reference and against the oracle.  No GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest

import cup3d_amd as cu
import oracle_lib as O

L = cu.lib()


def test_every_header_symbol_is_exported():
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "cup3d_hip.h")).read()
    names = set(re.findall(r"\b(cup3d_[a-z0-9_]+)\s*\(", hdr))
    assert names, "header parse failed"
    from cup3d_amd.capi import SIGNATURES
    assert names == set(SIGNATURES), (names ^ set(SIGNATURES))
    for n in names:
        assert getattr(L, n) is not None


def test_both_builds_export_the_boundary_and_only_the_test_build_the_test_support_symbols():
    """libcup3d_hip.so (release) and libcup3d_hip_testing.so export every symbol of include/cup3d_hip.h; the names of
    include/cup3d_hip_testing.h (capi.DEBUG_SIGNATURES lists exactly those) exist in the test build ONLY -- a release ABI does not
    carry them, not even as stubs (`nm -D`: no cup3d_debug_* at all)."""
    import re
    import subprocess
    from cup3d_amd.capi import DEBUG_SIGNATURES
    inc = os.path.join(os.path.dirname(__file__), "..", "include")
    names = set(re.findall(r"\b(cup3d_[a-z0-9_]+)\s*\(", open(os.path.join(inc, "cup3d_hip.h")).read()))
    dbg = set(re.findall(r"\b(cup3d_debug_[a-z0-9_]+)\s*\(", open(os.path.join(inc, "cup3d_hip_testing.h")).read()))
    assert dbg == set(DEBUG_SIGNATURES), dbg ^ set(DEBUG_SIGNATURES)
    assert not any(n.startswith("cup3d_debug_") for n in names)
    here = os.path.join(os.path.dirname(__file__), "..", "cup3d_amd")
    exported = {}
    for so in ("libcup3d_hip.so", "libcup3d_hip_testing.so"):
        out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(here, so)], stdout=subprocess.PIPE, check=True, text=True).stdout
        every = {ln.split()[-1] for ln in out.splitlines() if ln.split()}
        # -fvisibility=hidden + the linker's version script (csrc/exports.map): no C++ internal, no kernel stub, no libstdc++ instantiation
        assert not {n for n in every if not n.startswith("cup3d_")}, (so, sorted(n for n in every if not n.startswith("cup3d_"))[:10])
        exported[so] = every
        assert names <= exported[so], (so, names - exported[so])
    assert dbg <= exported["libcup3d_hip_testing.so"], dbg - exported["libcup3d_hip_testing.so"]
    assert not {n for n in exported["libcup3d_hip.so"] if n.startswith("cup3d_debug_")}, "the release library exports test-support symbols"
    assert exported["libcup3d_hip.so"] == names, exported["libcup3d_hip.so"] ^ names   # nothing undeclared either


def test_sfc_tables_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "sfc_tables.npz"))
    for key in z.files:
        bx, by, bz, lmax = [int(t.lstrip("L")) for t in key.split("_")[1:]]
        ref = z[key]
        h = C.c_void_p()
        assert L.cup3d_sfc_create(bx, by, bz, lmax, C.byref(h)) == 0
        row = 0
        for l in range(lmax):
            for k in range(bz << l):
                for j in range(by << l):
                    for i in range(bx << l):
                        idx = np.array([i, j, k], dtype=np.int32)
                        Z = L.cup3d_sfc_forward(h, l, i, j, k)
                        inv = np.zeros(3, dtype=np.int32)
                        L.cup3d_sfc_inverse(h, Z, l, inv)
                        nei, child, par = np.zeros(27, dtype=np.int64), np.zeros(8, dtype=np.int64), np.zeros(1, dtype=np.int64)
                        L.cup3d_sfc_info(h, l, idx, nei, child, par)
                        got = np.concatenate([[Z, L.cup3d_sfc_encode(h, l, idx), par[0]], nei, child])
                        assert np.array_equal(inv, idx)
                        assert np.array_equal(got, ref[row]), (key, l, i, j, k)
                        row += 1
        L.cup3d_sfc_destroy(h)


@pytest.mark.parametrize("name", ["f16_periodic", "f16_wall", "f16_mixed", "f24x16x8_mixed", "traj16_tgv"])
def test_block_order_matches_reference(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    g = cu.Grid(z["bpd"], int(z["level_max"]), int(z["level"]), float(z["extent"]), z["bc"])
    assert np.array_equal(g.tables, z["tables"])
    if "geom" in z.files:
        assert np.array_equal(g.geom, z["geom"])


@pytest.mark.parametrize("bpd,lmax,level", [((4, 4, 4), 1, 0), ((1, 1, 1), 4, 3), ((3, 2, 1), 2, 1), ((8, 8, 8), 1, 0)])
def test_block_order_matches_oracle(bpd, lmax, level):
    bc = ("periodic", "wall", "freespace")
    g = cu.Grid(bpd, lmax, level, 2 * np.pi, bc)
    o = O.OracleGrid(bpd, lmax, level, 2 * np.pi, bc)
    assert np.array_equal(g.tables, o.tables) and np.array_equal(g.geom, o.geom)


def expected_neighbour_index(idx, f, nbd, bc):
    d, side = f >> 1, f & 1
    c = list(idx)
    at_face = c[d] == (nbd[d] - 1 if side else 0)
    if at_face and bc[d] != 1:
        return None
    c[d] = (c[d] + (1 if side else -1)) % nbd[d]
    return tuple(c)


def test_block_ownership_equals_the_reference_under_real_mpi(golden_dir):
    """tests/golden/partition_mpi.npz: the reference run on 2-8 ranks of a real MPI (oracle/_ref/ref_tool_mpi); the block table of
    every rank must be what cup3d_grid_create_uniform(rank, nranks) and the oracle's partition rule give."""
    z = np.load(os.path.join(golden_dir, "partition_mpi.npz"))
    k = 0
    while f"case{k}" in z:
        bx, by, bz, lmax, lstart, nranks = (int(v) for v in z[f"case{k}"])
        level = int(z[f"case{k}_r0"][0, 0])
        total = (bx << level) * (by << level) * (bz << level)
        assert sum(len(z[f"case{k}_r{r}"]) for r in range(nranks)) == total
        for r in range(nranks):
            ref = z[f"case{k}_r{r}"]
            g = cu.Grid((bx, by, bz), lmax, level, 1.0, (1, 1, 1), r, nranks)
            assert np.array_equal(g.tables, ref), (k, r)
            a, n = C.c_longlong(), C.c_longlong()
            O.lib().orc_partition(total, r, nranks, C.byref(a), C.byref(n))
            assert sorted(ref[:, 1].tolist()) == list(range(a.value, a.value + n.value)), (k, r)
        k += 1
    assert k >= 5


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
@pytest.mark.parametrize("bc", [(1, 1, 1), (2, 0, 1)])
def test_neighbours_partition_and_plan(nranks, bc):
    bpd, lmax, level = (2, 2, 1), 2, 1
    grids = [cu.Grid(bpd, lmax, level, 1.0, bc, r, nranks) for r in range(nranks)]
    nbd = [b << level for b in bpd]
    total = nbd[0] * nbd[1] * nbd[2]
    # ownership = contiguous Z ranges of the GridMPI constructor (main.cpp:2970-2980)
    z0 = 0
    owner = {}
    for r, g in enumerate(grids):
        a, n = C.c_longlong(), C.c_longlong()
        O.lib().orc_partition(total, r, nranks, C.byref(a), C.byref(n))
        assert sorted(g.tables[:, 1]) == list(range(a.value, a.value + n.value))
        assert a.value == z0
        z0 += n.value
        for s, row in enumerate(g.tables):
            owner[tuple(row[2:5])] = (r, s)
        assert np.all(np.diff(g.tables[:, 5]) > 0)  # sorted by blockID_2
    assert z0 == total
    plans = [g.halo_plan() for g in grids]
    for r, g in enumerate(grids):
        nbr = g.neighbours()
        send, recv, faces = plans[r]
        # what each peer sends me must be what I expect to receive, in the same order
        recv_entries = []
        for p in range(nranks):
            ps, _, pf = plans[p]
            off = int(ps[:r].sum())
            assert ps[r] == recv[p]
            for e in range(int(ps[r])):
                sf = int(pf[off + e])
                recv_entries.append((p, sf // 6, sf % 6))
        for s in range(g.nblocks):
            for f in range(6):
                exp = expected_neighbour_index(g.index[s], f, nbd, bc)
                n = int(nbr[s, f])
                if exp is None:
                    assert n == -1 - bc[f >> 1]
                    continue
                pr, ps_ = owner[exp]
                if pr == r:
                    assert n == ps_
                else:
                    assert n >= cu.capi.NBR_HALO
                    p, slot, face = recv_entries[n - cu.capi.NBR_HALO]
                    # the slab comes from the neighbour block, through its opposite face
                    assert (p, slot, face) == (pr, ps_, f ^ 1)
        inner = L.cup3d_grid_ninner(g.handle)
        has_remote = (nbr >= cu.capi.NBR_HALO).any(axis=1).sum()
        assert inner == g.nblocks - has_remote


def test_calc_max_timestep_matches_oracle():
    coef_a, coef_b = np.array([1.5, -2.0, 0.5]), np.array([1.5, -2.0, 0.5])
    dt_a = dt_b = 0.0
    for step in range(8):
        dt_a = L.cup3d_calc_max_timestep(0.05, 0.9 + 0.01 * step, 0.01, 0.3, step, 4, dt_a, coef_a)
        dt_b = O.lib().orc_calc_dt(0.05, 0.9 + 0.01 * step, 0.01, 0.3, step, 4, dt_b, coef_b)
        assert dt_a == dt_b and np.array_equal(coef_a, coef_b)
    # -implicitDiffusion: the diffusive limit is 0.1 once step > 10 (main.cpp:15269-15273); nu large enough that it matters
    dt_a = dt_b = 1e-4
    seen = set()
    for step in range(8, 15):
        dt_a = L.cup3d_calc_max_timestep2(0.05, 0.2, 5.0, 0.3, step, 4, dt_a, coef_a, 1)
        dt_b = O.lib().orc_calc_dt2(0.05, 0.2, 5.0, 0.3, step, 4, dt_b, coef_b, 1)
        assert dt_a == dt_b and np.array_equal(coef_a, coef_b)
        seen.add(dt_a)
        assert L.cup3d_calc_max_timestep2(0.05, 0.2, 5.0, 0.3, step, 4, 0.0, np.zeros(3), 0) == L.cup3d_calc_max_timestep(0.05, 0.2, 5.0, 0.3, step, 4, 0.0, np.zeros(3))
    assert len(seen) == 2 and abs(max(seen) - 0.075) < 1e-6   # min(0.1, CFL*h/u) after step 10, h^2/6/(nu + ..) before


def test_bad_arguments_are_reported():
    h = C.c_void_p()
    bad = np.array([0, 1, 1], dtype=np.int32)
    bc = np.array([1, 1, 1], dtype=np.int32)
    assert L.cup3d_grid_create_uniform(bad, 1, 0, 1.0, bc, 0, 1, C.byref(h)) == -1
    assert b"bpd" in L.cup3d_last_error()
    ok = np.array([1, 1, 1], dtype=np.int32)
    assert L.cup3d_grid_create_uniform(ok, 1, 0, 1.0, bc, 0, 2, C.byref(h)) == -1   # fewer blocks than ranks
    # a NULL handle is rejected before anything touches the device (no GPU needed): status code, no crash
    u = np.zeros(3)
    assert L.cup3d_advect_diffuse(None, 0.1, 0.1, u) == -1
    assert L.cup3d_advect_diffuse_implicit(None, 0.1, 0.1, u, None, None) == -1
    assert L.cup3d_advect_implicit(None, 0.1, 0.1, u) == -1
    assert L.cup3d_diffusion_rhs(None) == -1 and L.cup3d_diffusion_lhs(None, 0, 0.1, 0.1) == -1
    assert L.cup3d_diffusion_preconditioner(None, 0.1, 0.1) == -1 and L.cup3d_diffusion_solve(None, 0, 0.1, 0.1, None, None) == -1
    assert L.cup3d_penalization(None, 0.1, 1.0, 1, 0, None) == -1 and L.cup3d_update_tmpv(None, 0, None) == -1
    assert L.cup3d_sim_upload_block_list(None, 0, 0, None, None) == -1 and L.cup3d_sim_download_block_list(None, 0, 0, None, None) == -1
    assert L.cup3d_grid_adapted_owners(None, None, None, 2, None, None) == -1


def test_compute_fails_loudly_without_gpu():
    if cu.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(cu.Cup3dError):
        cu.device_init(0)
    with pytest.raises(cu.Cup3dError):
        cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=1)


# ------------------------------------------------------------------ multi-level meshes (host topology only)
def _mesh_cases(golden_dir):
    import os
    for name in ("amr_periodic_l01", "amr_mixed_l12"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        t = g["tables"]
        yield name, tuple(int(b) for b in g["bpd"]), int(g["level_max"]), tuple(int(b) for b in g["bc"]), t[:, 0].copy(), t[:, 1].copy(), t
    bpd, lmax, bc = (2, 2, 2), 3, (2, 0, 2)
    lv, zs = O.build_balanced_mesh(bpd, lmax, bc, [(0, 0, 0, 0), (1, 0, 0, 0)])
    yield "synthetic_l012", bpd, lmax, bc, lv, zs, None


def test_mesh_topology_matches_reference_tables_and_oracle_states(golden_dir):
    for name, bpd, lmax, bc, lv, zs, tables in _mesh_cases(golden_dir):
        rng = np.random.default_rng(1)
        perm = rng.permutation(len(lv))  # leaves may come in any order
        g = cu.Grid(bpd, lmax, 0, 2 * np.pi, bc, leaves=(lv[perm], zs[perm]))
        m = O.OracleMesh(bpd, lmax, 2 * np.pi, bc, lv, zs)
        assert np.array_equal(g.tables, m.tables), name                       # m_vInfo order, index, blockID_2
        if tables is not None:
            assert np.array_equal(g.tables, tables), name                     # ... as the reference's adaptMesh left them
        assert np.array_equal(g.geom[:, 0], np.array([m.h(b) for b in range(m.nb)]))
        faces, fine, n27 = g.interface()
        st = m.states()
        same = st >= 0
        assert np.array_equal(n27 >= 0, (st >= 0) | (st <= -100))
        assert np.array_equal(n27[same], st[same])
        coarser = st <= -100
        assert np.array_equal(n27[coarser] - cu.capi.NBR_COARSER, -100 - st[coarser])
        assert np.array_equal(n27 == -3, st == -3) and np.array_equal(n27 == -1, st == -1)
        # face table: same-level slot, boundary code, or interface face e (numbered in (slot, face) order)
        nbr = g.neighbours()
        face_codes = [12, 14, 10, 16, 4, 22]  # x-,x+,y-,y+,z-,z+ in the 27-code numbering
        e = 0
        for s in range(g.nblocks):
            for f in range(6):
                v = st[s, face_codes[f]]
                if v >= 0:
                    assert nbr[s, f] == v
                elif v == -1:
                    assert nbr[s, f] == -1 - bc[f >> 1]
                else:
                    assert nbr[s, f] == cu.capi.NBR_HALO + e
                    assert faces[e, 0] == 6 * s + f and faces[e, 1] == (1 if v == -3 else 0)
                    e += 1
        assert e == len(faces)
        # every coarse-side face points at the four opposite fine faces, quadrant by quadrant
        lev, idx = g.tables[:, 0], g.tables[:, 2:5]
        for ei, (sf, kind) in enumerate(faces):
            if kind == 0:
                assert (fine[ei] == -1).all()
                continue
            s, f = sf // 6, sf % 6
            d, side = f >> 1, f & 1
            dfast, dslow = (1 if d == 0 else 0), (1 if d == 2 else 2)
            for B in range(4):
                fs, ff = faces[fine[ei, B], 0] // 6, faces[fine[ei, B], 0] % 6
                assert ff == f ^ 1 and faces[fine[ei, B], 1] == 0 and lev[fs] == lev[s] + 1
                exp = 2 * idx[s].copy()
                exp[d] += 2 if side else -1
                exp[dfast] += B % 2
                exp[dslow] += B // 2
                n = np.array(bpd) << int(lev[fs])
                assert np.array_equal(idx[fs], exp % n)


def test_mesh_rejects_unbalanced_and_duplicate_leaves():
    h = C.c_void_p()
    bpd = np.array([2, 2, 2], dtype=np.int32)
    bc = np.array([1, 1, 1], dtype=np.int32)
    lv, zs = O.build_balanced_mesh((2, 2, 2), 3, (1, 1, 1), [(0, 0, 0, 0)])
    dup = (np.concatenate([lv, lv[:1]]).astype(np.int32), np.concatenate([zs, zs[:1]]).astype(np.int64))
    assert L.cup3d_grid_create_mesh(bpd, 3, 1.0, bc, len(dup[0]), dup[0], dup[1], C.byref(h)) == -1
    assert b"duplicate" in L.cup3d_last_error()
    # split a level-1 leaf twice without balancing: a level-3... (levelMax 4) leaf next to level-0 blocks
    sfc = O.lib().orc_sfc_create(2, 2, 2, 4)
    leaves = {(0, i, j, k) for i in range(2) for j in range(2) for k in range(2)}
    for leaf in [(0, 0, 0, 0), (1, 1, 1, 1)]:
        leaves.remove(leaf)
        l, i, j, k = leaf
        leaves |= {(l + 1, 2 * i + (q & 1), 2 * j + ((q >> 1) & 1), 2 * k + (q >> 2)) for q in range(8)}
    out = sorted(leaves)
    lv = np.array([t[0] for t in out], dtype=np.int32)
    zs = np.array([O.lib().orc_sfc_forward(sfc, *t) for t in out], dtype=np.int64)
    O.lib().orc_sfc_destroy(sfc)
    assert L.cup3d_grid_create_mesh(bpd, 4, 1.0, bc, len(lv), lv, zs, C.byref(h)) == -1
    assert b"2:1" in L.cup3d_last_error()


def test_valid_states_and_adapted_mesh_match_reference(golden_dir):
    """Integer contract of MeshAdaptation: ValidStates (5330-5492) and the block list Adapt (5086-5159) leaves behind."""
    import os
    g = np.load(os.path.join(golden_dir, "amr_adapt_mixed.npz"))
    t = g["tables"]
    bpd, lmax, bc = tuple(int(b) for b in g["bpd"]), int(g["level_max"]), tuple(int(b) for b in g["bc"])
    grid = cu.Grid(bpd, lmax, 0, float(g["extent"]), bc, leaves=(t[:, 0], t[:, 1]))
    m = O.OracleMesh(bpd, lmax, float(g["extent"]), bc, t[:, 0], t[:, 1])
    st = grid.valid_states(g["tags"])
    assert np.array_equal(st, m.valid_states(g["tags"]))
    lv, zs = grid.adapted_leaves(st)
    new = cu.Grid(bpd, lmax, 0, float(g["extent"]), bc, leaves=(lv, zs))
    assert np.array_equal(new.tables, g["tables_new"])          # what the reference's adaptMesh produced
    # random tags on more meshes against the oracle (itself pinned to the reference by tests/test_oracle_amr.py)
    rng = np.random.default_rng(3)
    for name, bpd, lmax, bc, lv, zs, _ in _mesh_cases(golden_dir):
        grid = cu.Grid(bpd, lmax, 0, 2 * np.pi, bc, leaves=(lv, zs))
        m = O.OracleMesh(bpd, lmax, 2 * np.pi, bc, lv, zs)
        for trial in range(6):
            tags = rng.choice(np.array([-1, 0, 1], dtype=np.int8), size=grid.nblocks, p=[(0.7, 0.2, 0.1), (0.2, 0.5, 0.3)][trial % 2])
            st = grid.valid_states(tags)
            assert np.array_equal(st, m.valid_states(tags)), (name, trial)
            lv2, zs2 = grid.adapted_leaves(st)
            assert np.array_equal(cu.Grid(bpd, lmax, 0, 2 * np.pi, bc, leaves=(lv2, zs2)).tables, m.adapted(st).tables)   # also checks 2:1 balance
    # a uniform one-rank grid is accepted too (viewed as a one-level mesh)
    u = cu.Grid((2, 2, 2), 3, 1, 1.0, (1, 1, 1))
    tags = np.zeros(u.nblocks, dtype=np.int8)
    tags[5] = 1
    st = u.valid_states(tags)
    lv2, zs2 = u.adapted_leaves(st)
    assert len(lv2) == u.nblocks + 7 and set(lv2.tolist()) == {1, 2}


def test_block_tables_for_every_small_box_shape():
    """Product topology vs oracle for every box up to 4x4x3 level-0 blocks (the reference treats boxes that are a prefix of the
    enclosing cube's Hilbert curve, e.g. 1x1x2 and 1x2x2, as 'regular': main.cpp:216-234)."""
    import itertools
    for bpd in itertools.product([1, 2, 3, 4], [1, 2, 3, 4], [1, 2, 3]):
        for lmax, lev in ((1, 0), (2, 1), (3, 2)):
            g = cu.Grid(bpd, lmax, lev, 1.0, (1, 1, 1))
            o = O.OracleGrid(bpd, lmax, lev, 1.0, (1, 1, 1))
            assert np.array_equal(g.tables, o.tables), (bpd, lmax, lev)


def _states_from_tables(old, new):
    """Valid states of the old leaves implied by the new leaf set: still there -> Leave, children there -> Refine, else Compress."""
    have = {(int(r[0]), int(r[2]), int(r[3]), int(r[4])) for r in new}
    st = np.zeros(len(old), dtype=np.int8)
    for b, (l, _, i, j, k, _) in enumerate(old):
        if (int(l), int(i), int(j), int(k)) in have:
            continue
        st[b] = 1 if (int(l) + 1, 2 * int(i), 2 * int(j), 2 * int(k)) in have else -1
    return st


def test_block_ownership_after_adaptation_equals_the_reference_under_real_mpi(golden_dir):
    """tests/golden/adapt_mpi.npz: Simulation::adaptMesh of the reference on 3-7 ranks of a real MPI.  The owner rank of every block
    of the adapted mesh -- children with the refined parent, compressed octets gathered on the base block's rank, Balance_Global
    and Balance_Diffusion of the LoadBalancer (main.cpp:4660-5022) -- from cup3d_grid_adapted_owners and from the oracle."""
    z = np.load(os.path.join(golden_dir, "adapt_mpi.npz"))
    k, modes = 0, set()
    while f"t{k}_meta" in z:
        bx, by, bz, lmax, b0, b1, b2, nranks = (int(v) for v in z[f"t{k}_meta"])
        old, new, ow_old, ow_new = z[f"t{k}_old"], z[f"t{k}_new"], z[f"t{k}_old_owner"], z[f"t{k}_new_owner"]
        st = _states_from_tables(old, new)
        g_old = cu.Grid((bx, by, bz), lmax, 0, 1.0, (b0, b1, b2), leaves=(old[:, 0].astype(np.int32), old[:, 1].copy()))
        lv, zs = g_old.adapted_leaves(st)
        g_new = cu.Grid((bx, by, bz), lmax, 0, 1.0, (b0, b1, b2), leaves=(lv, zs))
        assert np.array_equal(g_new.tables, new), k              # the adapted block list itself (integer contract, one more pin)
        assert np.array_equal(g_old.adapted_owners(ow_old, st, nranks, g_new), ow_new), k
        m_old = O.OracleMesh((bx, by, bz), lmax, 1.0, (b0, b1, b2), old[:, 0], old[:, 1])
        m_new = m_old.adapted(st)
        assert np.array_equal(m_new.tables, new) and np.array_equal(m_old.adapted_owners(ow_old, st, nranks, m_new), ow_new), k
        cnt = np.bincount(ow_new, minlength=nranks)
        modes.add("even" if cnt.max() - cnt.min() <= 1 else "diffusion")
        modes.add("compress" if (st == -1).any() else "refine")
        k += 1
    assert k >= 10 and modes == {"even", "diffusion", "compress", "refine"}


# ------------------------------------------------------------------ multi-level meshes spread over ranks (host planner)
def _decode(x, to_global_slot, to_global_face):
    """a neighbour-table entry of a rank view -> the same entry in global numbering"""
    H, Cc = cu.capi.NBR_HALO, cu.capi.NBR_COARSER
    if x >= H:
        return H + int(to_global_face[x - H])
    if x >= Cc:
        return Cc + int(to_global_slot[x - Cc])
    return int(to_global_slot[x]) if x >= 0 else int(x)


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_rank_views_of_a_multilevel_mesh(golden_dir, nranks):
    """Grid.rank_view: the tables a rank sees are the global ones renumbered (so a kernel reads through them what it reads on one
    rank), every remote block / fine face they refer to is a ghost, no ghost is superfluous, and the send lists of the owners match
    the ghost order of the receivers -- checked by carrying out both exchanges on arrays of global ids."""
    for name, bpd, lmax, bc, lv, zs, _ in _mesh_cases(golden_dir):
        mesh = cu.Grid(bpd, lmax, 0, 1.0, bc, leaves=(lv, zs))
        nb = mesh.nblocks
        faces, fine, n27 = mesh.interface()
        nbr = mesh.neighbours()
        a = np.zeros(nranks + 1, dtype=np.int64)       # contiguous runs of the global order, as the reference's ranks own them
        for r in range(nranks):
            a[r + 1] = a[r] + nb // nranks + (1 if r < nb % nranks else 0)
        owner = np.repeat(np.arange(nranks), np.diff(a)).astype(np.int32)
        views = [mesh.rank_view(owner, r, nranks) for r in range(nranks)]
        assert sum(v.nlocal for v in views) == nb
        assert all(v.nghost > 0 for v in views) and any(v.nfaces_ghost > 0 for v in views)
        for r, v in enumerate(views):
            gs, gf = v.global_slot, v.global_face
            assert np.array_equal(gs[:v.nlocal], np.where(owner == r)[0])
            assert (owner[gs[v.nlocal:]] != r).all() and len(set(gs.tolist())) == len(gs)
            # tables of the local blocks == global tables, renumbered
            for i in range(v.nlocal):
                g = gs[i]
                assert [_decode(int(x), gs, gf) for x in v.nbr[i]] == nbr[g].tolist(), (name, r, i)
                assert [_decode(int(x), gs, gf) for x in v.nbr27[i]] == n27[g].tolist(), (name, r, i)
            used_slots, used_faces = set(), set()
            for i in range(v.nlocal):
                for x in v.nbr27[i].tolist():
                    if x >= cu.capi.NBR_COARSER:
                        used_slots.add(x - cu.capi.NBR_COARSER)
                    elif x >= 0:
                        used_slots.add(x)
            for e in range(len(v.faces)):
                slot, f = divmod(int(v.faces[e, 0]), 6)
                assert gs[slot] * 6 + f == faces[gf[e], 0] and v.faces[e, 1] == faces[gf[e], 1]
                if e < v.nfaces_local:
                    assert slot < v.nlocal
                    if v.faces[e, 1] == 1:   # coarse side: the four fine faces, in the view's numbering
                        assert [int(gf[x]) for x in v.fine[e]] == fine[gf[e]].tolist()
                        used_faces.update(int(x) for x in v.fine[e])
                        used_slots.update(int(v.faces[x, 0]) // 6 for x in v.fine[e])
                else:
                    assert slot >= v.nlocal and v.faces[e, 1] == 0
            # inner blocks (computed while the ghost blocks travel): nothing their tables lead to is a ghost
            waits = np.zeros(v.nlocal, dtype=bool)
            for i in range(v.nlocal):
                sl = np.where(v.nbr27[i] >= cu.capi.NBR_COARSER, v.nbr27[i] - cu.capi.NBR_COARSER, v.nbr27[i])
                waits[i] = (sl >= v.nlocal).any()
            for e in range(v.nfaces_local):
                if v.faces[e, 1] == 1 and any(x >= v.nfaces_local or v.faces[x, 0] // 6 >= v.nlocal for x in v.fine[e]):
                    waits[int(v.faces[e, 0]) // 6] = True
            assert v.ninner == int((~waits).sum()) and 0 <= v.ninner < v.nlocal
            assert set(range(v.nlocal, v.nlocal + v.nghost)) <= used_slots            # no superfluous ghost block
            assert set(range(v.nfaces_local, len(v.faces))) <= used_faces             # ... or ghost face
            assert max(used_slots) < v.nlocal + v.nghost
        # the two exchanges, carried out on global ids: every ghost slot / ghost face receives its owner's entry
        for r, v in enumerate(views):
            got_b = np.full(v.nlocal + v.nghost, -1)
            got_b[:v.nlocal] = v.global_slot[:v.nlocal]
            got_f = np.full(len(v.faces), -1)
            got_f[:v.nfaces_local] = v.global_face[:v.nfaces_local]
            pos_b, pos_f = v.nlocal, v.nfaces_local
            for p, w in enumerate(views):
                if p == r:
                    assert v.recv_block_count[p] == 0 and v.recv_flux_count[p] == 0
                    continue
                sb0 = int(w.send_block_count[:r].sum())
                sent = w.global_slot[w.send_blocks[sb0:sb0 + int(w.send_block_count[r])]]      # what p packs for r
                assert len(sent) == v.recv_block_count[p]
                got_b[pos_b:pos_b + len(sent)] = sent
                pos_b += len(sent)
                sf0 = int(w.send_flux_count[:r].sum())
                sentf = w.global_face[w.send_flux_faces[sf0:sf0 + int(w.send_flux_count[r])]]
                assert len(sentf) == v.recv_flux_count[p] and (w.send_flux_faces[sf0:sf0 + len(sentf)] < w.nfaces_local).all()
                got_f[pos_f:pos_f + len(sentf)] = sentf
                pos_f += len(sentf)
            assert np.array_equal(got_b, v.global_slot) and np.array_equal(got_f, v.global_face), (name, r)


@pytest.mark.parametrize("seed", range(6))
def test_multigrid_hierarchies_of_all_ranks_fit_together(seed):
    """The multigrid option on a multi-level mesh spread over ranks (multigrid.hip on rank views): every rank's level hierarchy -- owned
    nodes (leaves + ancestors, an ancestor living with its first child), ghost nodes, neighbour / parent / coarse-fine tables, the ghost
    exchange plan and the restriction-octant plan -- built by Grid::mg_hierarchy and cross-checked by cup3d_debug_mg_plan_check: tables
    in range, what r sends to p is node for node what p expects from r, every owned ancestor receives each of its eight octants exactly
    once.  Random balanced meshes (2-4 levels, mixed boundary conditions, non-cubic boxes) under contiguous (Hilbert-range-like) AND
    scattered ownership on 1-7 ranks."""
    rng = np.random.default_rng(1000 + seed)
    bpd = tuple(int(v) for v in rng.choice([1, 2, 3], 3))
    if bpd == (1, 1, 1):
        bpd = (2, 1, 2)
    lmax = int(rng.choice([3, 4]))
    bc = tuple(str(b) for b in rng.choice(["periodic", "wall", "freespace"], 3))
    refine = []
    for l in range(lmax - 1):          # refine a few random blocks per level (build_balanced_mesh keeps the 2:1 balance)
        n = [b << l for b in bpd]
        for _ in range(int(rng.integers(1, 4))):
            refine.append((l, int(rng.integers(0, n[0])), int(rng.integers(0, n[1])), int(rng.integers(0, n[2]))))
    try:
        lv, zs = O.build_balanced_mesh(bpd, lmax, bc, refine)
    except Exception:
        pytest.skip("the random refinement list named a block that no longer exists")
    g = cu.operators.Grid(bpd, lmax, 0, 2 * np.pi, bc, leaves=(lv, zs))
    nb = g.nblocks
    assert L.cup3d_debug_mg_plan_check(g.handle, None, 1) == 0, L.cup3d_last_error().decode()
    for nranks in (2, 3, 7):
        if nranks > nb:
            continue
        contiguous = (np.arange(nb) * nranks // nb).astype(np.int32)
        scattered = rng.integers(0, nranks, nb).astype(np.int32)
        scattered[:nranks] = np.arange(nranks)       # every rank owns something
        for owner in (contiguous, scattered):
            ow = np.ascontiguousarray(owner)
            assert L.cup3d_debug_mg_plan_check(g.handle, ow.ctypes.data_as(C.c_void_p), nranks) == 0, (nranks, L.cup3d_last_error().decode())


@pytest.mark.parametrize("seed", range(5))
def test_sub_box_exchange_plans_of_all_ranks_fit_together(seed):
    """The sub-box form of the ghost-block exchange (comm.hip; the reference ships face sub-boxes and coarse shadow cells,
    main.cpp:1832-1966, 2423-2544): for both stencil-width classes, what rank r packs for rank p -- block by block, box by box -- is what
    p expects from r; a box never leaves its block; for width 1 the message is smaller than whole 8^3 blocks, and the
    width-3 boxes contain the width-1 ones."""
    rng = np.random.default_rng(2000 + seed)
    bpd = tuple(int(v) for v in rng.choice([1, 2, 3], 3))
    if bpd == (1, 1, 1):
        bpd = (2, 1, 2)
    lmax = int(rng.choice([3, 4]))
    bc = tuple(str(b) for b in rng.choice(["periodic", "wall", "freespace"], 3))
    refine = []
    for l in range(lmax - 1):
        n = [b << l for b in bpd]
        for _ in range(int(rng.integers(1, 4))):
            refine.append((l, int(rng.integers(0, n[0])), int(rng.integers(0, n[1])), int(rng.integers(0, n[2]))))
    try:
        lv, zs = O.build_balanced_mesh(bpd, lmax, bc, refine)
    except Exception:
        pytest.skip("the random refinement list named a block that no longer exists")
    g = cu.operators.Grid(bpd, lmax, 0, 2 * np.pi, bc, leaves=(lv, zs))
    nb = g.nblocks
    for nranks in (2, 3, 5):
        if nranks > nb:
            continue
        for kind, owner in (("ranges", (np.arange(nb) * nranks // nb).astype(np.int32)), ("scattered", rng.integers(0, nranks, nb).astype(np.int32))):
            if kind == "scattered":
                owner[:nranks] = np.arange(nranks)
            views = [g.rank_view(owner, r, nranks) for r in range(nranks)]
            whole = sum(v.nghost for v in views) * 512
            for k in (0, 1):
                for r, v in enumerate(views):
                    gb = v.ghost_box[k].astype(int)
                    assert (gb[:, 3:] <= 8).all() and (gb[:, :3] <= gb[:, 3:]).all()
                    gown = owner[v.global_slot[v.nlocal:]]
                    vol = np.prod(gb[:, 3:] - gb[:, :3], axis=1)
                    for p in range(nranks):
                        assert v.recv_cells[k][p] == vol[gown == p].sum()
                        if p == r:
                            continue
                        # what p sends to r: p's send list, peer-major -- the run for r starts after the runs of the ranks before r
                        w = views[p]
                        start = int(sum(w.send_block_count[q] for q in range(r)))
                        mine = w.send_box[k][start:start + int(w.send_block_count[r])]
                        assert np.array_equal(mine, v.ghost_box[k][gown == p]), (nranks, k, r, p)
                        assert np.array_equal(w.global_slot[w.send_blocks[start:start + int(w.send_block_count[r])]], v.global_slot[v.nlocal:][gown == p])
                        assert w.send_cells[k][r] == v.recv_cells[k][p]
                    if k == 1:      # a wider stencil reads at least what the narrow one reads
                        g0 = v.ghost_box[0].astype(int)
                        nz = np.prod(g0[:, 3:] - g0[:, :3], axis=1) > 0
                        assert (gb[nz, :3] <= g0[nz, :3]).all() and (gb[nz, 3:] >= g0[nz, 3:]).all()
                cells = sum(int(v.recv_cells[k].sum()) for v in views)
                assert 0 < cells <= whole
                if k == 0:
                    assert cells < whole   # (how much less depends on the mesh: the test below pins it on a mesh of realistic shape)


def test_sub_box_exchange_saves_most_of_the_bytes_on_a_mesh_of_realistic_shape():
    """533 blocks on three levels (a uniform level-2 grid refined in three places), contiguous ownership on 2 and 8 ranks: the width-1
    exchange (the four scalar halos of every BiCGSTAB iteration) ships 5-6 times fewer cells than whole ghost blocks, the width-3 one
    (advection-diffusion) 2.4-2.9 times fewer."""
    bpd, lmax, bc = (2, 2, 2), 4, ("wall", "freespace", "wall")
    refine = [(0, i, j, k) for k in range(2) for j in range(2) for i in range(2)] + [(1, i, j, k) for k in range(4) for j in range(4) for i in range(4)]
    refine += [(2, 3, 3, 3), (2, 4, 4, 4), (2, 6, 1, 2)]
    lv, zs = O.build_balanced_mesh(bpd, lmax, bc, refine)
    g = cu.operators.Grid(bpd, lmax, 0, 2 * np.pi, bc, leaves=(lv, zs))
    nb = g.nblocks
    for nranks in (2, 8):
        owner = (np.arange(nb) * nranks // nb).astype(np.int32)
        views = [g.rank_view(owner, r, nranks) for r in range(nranks)]
        whole = sum(v.nghost for v in views) * 512
        w1, w3 = (sum(int(v.recv_cells[k].sum()) for v in views) for k in (0, 1))
        print(f"{nranks} ranks: {whole // 512} ghost blocks; cells shipped: whole blocks {whole}, width 1 {w1} (x{whole / w1:.2f}), width 3 {w3} (x{whole / w3:.2f})")
        assert whole >= 5 * w1 and whole >= 2.3 * w3, (whole, w1, w3)


@pytest.mark.parametrize("seed", range(12))
def test_sub_boxes_hold_every_cell_the_reference_pinned_oracle_reads(seed):
    """The sub-box plans against an INDEPENDENT consumer.  grid.cpp derives the boxes by replaying the index arithmetic of the DEVICE's
    consumers (tile loads, k_ghost_restrict, k_ghost_prolong); the GPU tests run those consumers with the unshipped cells poisoned.  Here
    the consumer is the CPU oracle's BlockLab::load (orc_mesh_labs: same-level copies, AverageDown of finer neighbours, the coarse shadow
    tile and CoarseFineInterpolation -- pinned bit for bit against the reference, tests/test_oracle_amr.py): for every rank, every cell
    of every block the rank does not own is NaN unless it lies in the box the rank RECEIVES for it; the star part of the ghosted tiles of
    the rank's own blocks -- [-1,2) for the scalar consumers, [-3,4) for advection-diffusion -- must come out exactly as from the intact
    field.  Random balanced meshes, mixed boundary conditions, contiguous and scattered ownership.  (Mutation run: with every box one
    layer short in z all cases fail.)"""
    rng = np.random.default_rng(3000 + seed)
    bpd = tuple(int(v) for v in rng.choice([1, 2, 3], 3))
    if bpd == (1, 1, 1):
        bpd = (2, 1, 2)
    lmax = int(rng.choice([3, 4]))
    bc = tuple(str(b) for b in rng.choice(["periodic", "wall", "freespace"], 3))
    refine = []
    for l in range(lmax - 1):
        n = [b << l for b in bpd]
        for _ in range(int(rng.integers(1, 4))):
            refine.append((l, int(rng.integers(0, n[0])), int(rng.integers(0, n[1])), int(rng.integers(0, n[2]))))
    try:
        lv, zs = O.build_balanced_mesh(bpd, lmax, bc, refine)
    except Exception:
        pytest.skip("the random refinement list named a block that no longer exists")
    g = cu.operators.Grid(bpd, lmax, 0, 2 * np.pi, bc, leaves=(lv, zs))
    m = O.OracleMesh(bpd, lmax, 2 * np.pi, bc, lv, zs)
    nb = g.nblocks
    assert nb == m.nb and np.array_equal(g.tables[:, :2], m.tables[:, :2])     # same block order on both sides
    fields = {0: rng.uniform(-1, 1, (nb, 8, 8, 8)), 1: rng.uniform(-1, 1, (nb, 8, 8, 8, 3))}
    stencil = {0: (-1, 2), 1: (-3, 4)}
    intact = {k: m.labs(fields[k], *stencil[k]) for k in (0, 1)}
    checked = 0
    for nranks in (2, 3):
        if nranks > nb:
            continue
        for kind in ("ranges", "scattered"):
            owner = (np.arange(nb) * nranks // nb).astype(np.int32) if kind == "ranges" else rng.integers(0, nranks, nb).astype(np.int32)
            if kind == "scattered":
                owner[:nranks] = np.arange(nranks)
            for r in range(nranks):
                v = g.rank_view(owner, r, nranks)
                own = np.flatnonzero(owner == r)
                assert np.array_equal(np.sort(v.global_slot[:v.nlocal]), own)
                for k in (0, 1):
                    f = fields[k].copy()
                    f[owner != r] = np.nan
                    for slot, bx in zip(v.global_slot[v.nlocal:], v.ghost_box[k].astype(int)):
                        x0, y0, z0, x1, y1, z1 = bx
                        f[slot, z0:z1, y0:y1, x0:x1] = fields[k][slot, z0:z1, y0:y1, x0:x1]
                    labs = m.labs(f, *stencil[k])
                    idx = np.arange(8 + stencil[k][1] - stencil[k][0] - 1) + stencil[k][0]
                    out = (idx < 0) | (idx >= 8)
                    mask = (out[:, None, None].astype(int) + out[None, :, None] + out[None, None, :]) <= 1   # the STAR: what the kernels read
                    got, want = labs[own][:, mask], intact[k][own][:, mask]
                    bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
                    assert not bad.any(), (bpd, lmax, bc, nranks, kind, r, k, int(bad.sum()), int(np.isnan(got).sum()))
                    checked += got.size
    assert checked > 0


@pytest.mark.parametrize("seed", range(12))
def test_tensorial_view_lists_every_block_the_reference_pinned_oracle_reads(seed):
    """The same for the TENSORIAL view (whole ghost blocks; built inside cup3d_adapt_migrate and cup3d_grad_chi_on_tmp_over_ranks): with
    every block the rank neither owns nor lists as a ghost set to NaN, the oracle's tensorial tiles of the rank's own blocks -- [-1,2)
    (RefineBlocks' parent tile, main.cpp:5493-5565) and [-2,3) (GradChiOnTmp, 8540-8600), edges and corners included -- come out as from
    the intact field.  Random balanced meshes, contiguous and scattered ownership."""
    rng = np.random.default_rng(4000 + seed)
    bpd = tuple(int(v) for v in rng.choice([1, 2, 3], 3))
    if bpd == (1, 1, 1):
        bpd = (2, 1, 2)
    lmax = int(rng.choice([3, 4]))
    bc = tuple(str(b) for b in rng.choice(["periodic", "wall", "freespace"], 3))
    refine = []
    for l in range(lmax - 1):
        n = [b << l for b in bpd]
        for _ in range(int(rng.integers(1, 4))):
            refine.append((l, int(rng.integers(0, n[0])), int(rng.integers(0, n[1])), int(rng.integers(0, n[2]))))
    try:
        lv, zs = O.build_balanced_mesh(bpd, lmax, bc, refine)
    except Exception:
        pytest.skip("the random refinement list named a block that no longer exists")
    g = cu.operators.Grid(bpd, lmax, 0, 2 * np.pi, bc, leaves=(lv, zs))
    m = O.OracleMesh(bpd, lmax, 2 * np.pi, bc, lv, zs)
    nb = g.nblocks
    field = rng.uniform(-1, 1, (nb, 8, 8, 8))
    intact = {se: m.labs(field, *se, tensorial=True) for se in ((-1, 2), (-2, 3))}
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    fewer = 0
    for nranks in (2, 3):
        if nranks > nb:
            continue
        for kind in ("ranges", "scattered"):
            owner = (np.arange(nb) * nranks // nb).astype(np.int32) if kind == "ranges" else rng.integers(0, nranks, nb).astype(np.int32)
            if kind == "scattered":
                owner[:nranks] = np.arange(nranks)
            for r in range(nranks):
                h = C.c_void_p()
                assert L.cup3d_debug_grid_rank_view_tensorial(g.handle, p(owner), r, nranks, C.byref(h)) == 0, L.cup3d_last_error().decode()
                sz = (C.c_long * 6)()
                assert L.cup3d_grid_view_sizes(h, sz) == 0
                nlocal, nghost, nfl, nfg, nsb, nsf = (int(v) for v in sz)
                slot, face = np.zeros(nlocal + nghost, dtype=np.int32), np.zeros(max(nfl + nfg, 1), dtype=np.int32)
                sb, sf = np.zeros(max(nsb, 1), dtype=np.int32), np.zeros(max(nsf, 1), dtype=np.int32)
                cnt = [np.zeros(nranks, dtype=np.int64) for _ in range(4)]
                assert L.cup3d_grid_view_plan(h, p(slot), p(face), p(sb), p(cnt[0]), p(cnt[1]), p(sf), p(cnt[2]), p(cnt[3])) == 0
                L.cup3d_grid_destroy(h)
                own = np.flatnonzero(owner == r)
                assert np.array_equal(np.sort(slot[:nlocal]), own) and (owner[slot[nlocal:]] != r).all()
                f = np.full_like(field, np.nan)
                f[slot] = field[slot]
                fewer += nb - len(slot)
                for se in intact:
                    got, want = m.labs(f, *se, tensorial=True)[own], intact[se][own]
                    bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
                    assert not bad.any(), (bpd, lmax, bc, nranks, kind, r, se, int(bad.sum()))
    assert fewer >= 0


def test_sub_boxes_and_tensorial_ghost_lists_are_tight():
    """... and not much more than that travels.  A box is the bounding box of the cells its consumers read, so each of its six outer
    layers must hold a cell the oracle's tile assembly reads: poison one layer at a time and a NaN must appear in the star part of some
    own tile.  Measured: 6 of 276 outer layers of the width-1 boxes and 0 of 276 of the width-3 ones on these meshes (24 / 3 of 720 over six
    seeds) are NOT read by the oracle -- the boxes follow the device consumers' addressing, which is a little wider there; why was not
    looked into -- asserted below 6 %.  The tensorial view's ghost list is exact: every listed block is read."""
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    tot, loose, ghosts, unread = [0, 0], [0, 0], 0, 0
    for seed in range(3):
        rng = np.random.default_rng(3000 + seed)
        bpd = tuple(int(v) for v in rng.choice([1, 2, 3], 3))
        if bpd == (1, 1, 1):
            bpd = (2, 1, 2)
        lmax = int(rng.choice([3, 4]))
        bc = tuple(str(b) for b in rng.choice(["periodic", "wall", "freespace"], 3))
        refine = []
        for l in range(lmax - 1):
            n = [b << l for b in bpd]
            for _ in range(int(rng.integers(1, 4))):
                refine.append((l, int(rng.integers(0, n[0])), int(rng.integers(0, n[1])), int(rng.integers(0, n[2]))))
        try:
            lv, zs = O.build_balanced_mesh(bpd, lmax, bc, refine)
        except Exception:
            continue
        g = cu.operators.Grid(bpd, lmax, 0, 2 * np.pi, bc, leaves=(lv, zs))
        m = O.OracleMesh(bpd, lmax, 2 * np.pi, bc, lv, zs)
        nb, nranks = g.nblocks, 2
        fields = {0: rng.uniform(-1, 1, (nb, 8, 8, 8)), 1: rng.uniform(-1, 1, (nb, 8, 8, 8, 3))}
        stencil = {0: (-1, 2), 1: (-3, 4)}
        owner = (np.arange(nb) * nranks // nb).astype(np.int32)
        for r in range(nranks):
            v = g.rank_view(owner, r, nranks)
            own = np.flatnonzero(owner == r)
            for k in (0, 1):
                idx = np.arange(8 + stencil[k][1] - stencil[k][0] - 1) + stencil[k][0]
                out = (idx < 0) | (idx >= 8)
                mask = (out[:, None, None].astype(int) + out[None, :, None] + out[None, None, :]) <= 1
                for slot, bx in zip(v.global_slot[v.nlocal:], v.ghost_box[k].astype(int)):
                    x0, y0, z0, x1, y1, z1 = bx
                    if x1 <= x0 or y1 <= y0 or z1 <= z0:
                        continue
                    X, Y, Z = slice(x0, x1), slice(y0, y1), slice(z0, z1)
                    for layer in ((Z, Y, slice(x0, x0 + 1)), (Z, Y, slice(x1 - 1, x1)), (Z, slice(y0, y0 + 1), X), (Z, slice(y1 - 1, y1), X),
                                  (slice(z0, z0 + 1), Y, X), (slice(z1 - 1, z1), Y, X)):
                        f = fields[k].copy()
                        f[(slot,) + layer] = np.nan
                        tot[k] += 1
                        loose[k] += not np.isnan(m.labs(f, *stencil[k])[own][:, mask]).any()
            # the tensorial view: every ghost block is read by the [-2,3) tensorial tiles of the rank's own blocks
            h = C.c_void_p()
            assert L.cup3d_debug_grid_rank_view_tensorial(g.handle, p(owner), r, nranks, C.byref(h)) == 0
            sz = (C.c_long * 6)()
            assert L.cup3d_grid_view_sizes(h, sz) == 0
            nlocal, nghost, nfl, nfg, nsb, nsf = (int(t) for t in sz)
            slot, face = np.zeros(nlocal + nghost, dtype=np.int32), np.zeros(max(nfl + nfg, 1), dtype=np.int32)
            sb, sf = np.zeros(max(nsb, 1), dtype=np.int32), np.zeros(max(nsf, 1), dtype=np.int32)
            cnt = [np.zeros(nranks, dtype=np.int64) for _ in range(4)]
            assert L.cup3d_grid_view_plan(h, p(slot), p(face), p(sb), p(cnt[0]), p(cnt[1]), p(sf), p(cnt[2]), p(cnt[3])) == 0
            L.cup3d_grid_destroy(h)
            for gi in range(nlocal, nlocal + nghost):
                f = fields[0].copy()
                f[slot[gi]] = np.nan
                ghosts += 1
                unread += not np.isnan(m.labs(f, -2, 3, tensorial=True)[own]).any()
    print(f"box layers checked {tot}, not read by the oracle {loose}; tensorial ghost blocks {ghosts}, not read {unread}")
    assert min(tot) > 100 and ghosts > 20
    assert loose[0] <= 0.06 * tot[0] and loose[1] <= 0.06 * tot[1], (tot, loose)
    assert unread == 0


@pytest.mark.parametrize("seed", list(range(12)) + ["533 blocks"])
def test_inner_blocks_of_a_rank_view_read_nothing_remote(seed):
    """The overlap of the ghost-block exchange with the interior (halo_begin / halo_finish on rank views; the reference's inner_blocks,
    main.cpp:2196-2199, 5598-5618): the stencil kernels run on the INNER blocks while the ghost blocks are still travelling, so a block
    wrongly classified inner would read stale ghost slots -- a race the device tests can only catch by luck.  Here every block the rank
    does not own is NaN and the oracle's star tiles ([-1,2) and [-3,4)) of the inner blocks must come out as from the intact field.
    The split is conservative (any ghost among the 27 neighbours makes a block a boundary block): the fraction of boundary blocks that
    do read something remote is printed."""
    big = not isinstance(seed, int)
    rng = np.random.default_rng(5000 + (99 if big else seed))
    bpd = tuple(int(v) for v in rng.choice([1, 2, 3], 3))
    if bpd == (1, 1, 1):
        bpd = (2, 1, 2)
    lmax = int(rng.choice([3, 4]))
    bc = tuple(str(b) for b in rng.choice(["periodic", "wall", "freespace"], 3))
    refine = []
    for l in range(lmax - 1):
        n = [b << l for b in bpd]
        for _ in range(int(rng.integers(1, 4))):
            refine.append((l, int(rng.integers(0, n[0])), int(rng.integers(0, n[1])), int(rng.integers(0, n[2]))))
    if big:   # a mesh with an interior: a uniform level-2 grid refined in three places (the mesh of the byte-count test above)
        bpd, lmax, bc = (2, 2, 2), 4, ("wall", "freespace", "periodic")
        refine = [(0, i, j, k) for k in range(2) for j in range(2) for i in range(2)] + [(1, i, j, k) for k in range(4) for j in range(4) for i in range(4)]
        refine += [(2, 3, 3, 3), (2, 4, 4, 4), (2, 6, 1, 2)]
    try:
        lv, zs = O.build_balanced_mesh(bpd, lmax, bc, refine)
    except Exception:
        pytest.skip("the random refinement list named a block that no longer exists")
    g = cu.operators.Grid(bpd, lmax, 0, 2 * np.pi, bc, leaves=(lv, zs))
    m = O.OracleMesh(bpd, lmax, 2 * np.pi, bc, lv, zs)
    nb = g.nblocks
    fields = {(-1, 2): rng.uniform(-1, 1, (nb, 8, 8, 8)), (-3, 4): rng.uniform(-1, 1, (nb, 8, 8, 8, 3))}
    intact = {se: m.labs(f, *se) for se, f in fields.items()}
    ninner_total = nbound = nbound_reading = 0
    for nranks in (2, 3):
        if nranks > nb:
            continue
        for kind in ("ranges", "scattered"):
            owner = (np.arange(nb) * nranks // nb).astype(np.int32) if kind == "ranges" else rng.integers(0, nranks, nb).astype(np.int32)
            if kind == "scattered":
                owner[:nranks] = np.arange(nranks)
            for r in range(nranks):
                v = g.rank_view(owner, r, nranks)
                inner = np.zeros(max(v.ninner, 1), dtype=np.int32)
                assert L.cup3d_debug_grid_inner_blocks(v.handle, inner.ctypes.data_as(C.c_void_p)) == 0
                inner = inner[:v.ninner]
                assert len(set(inner.tolist())) == v.ninner and (inner < v.nlocal).all()
                reads_remote = np.zeros(v.nlocal, bool)
                for se, fld in fields.items():
                    f = fld.copy()
                    f[owner != r] = np.nan
                    idx = np.arange(8 + se[1] - se[0] - 1) + se[0]
                    out = (idx < 0) | (idx >= 8)
                    mask = (out[:, None, None].astype(int) + out[None, :, None] + out[None, None, :]) <= 1
                    loc = v.global_slot[:v.nlocal]
                    got, want = m.labs(f, *se)[loc][:, mask], intact[se][loc][:, mask]
                    same = ((got == want) | (np.isnan(got) & np.isnan(want))).reshape(v.nlocal, -1).all(axis=1)
                    reads_remote |= ~same
                assert not reads_remote[inner].any(), (bpd, lmax, bc, nranks, kind, r, np.flatnonzero(reads_remote[inner]))
                ninner_total += v.ninner
                nbound += v.nlocal - v.ninner
                nbound_reading += int(reads_remote.sum())
    print(f"inner blocks {ninner_total}; boundary blocks {nbound}, of which {nbound_reading} read something remote")
    if big:
        assert ninner_total > 500   # the case that does test something: hundreds of inner blocks per ownership


@pytest.mark.parametrize("bpd,level,bc,nranks", [((2, 2, 2), 1, ("periodic", "periodic", "periodic"), 3), ((1, 1, 1), 3, ("wall", "wall", "wall"), 8),
                                                 ((3, 2, 1), 2, ("periodic", "wall", "freespace"), 5), ((1, 1, 1), 2, ("freespace", "periodic", "wall"), 2)])
def test_inner_blocks_of_a_uniform_grid_read_nothing_remote(bpd, level, bc, nranks):
    """The same on uniform grids sharded by Hilbert ranges (face-slab exchange overlapped with the inner blocks' pass, comm.hip
    halo_begin / halo_finish): with every block outside the rank's range NaN, the [-3,4) star tiles of the rank's inner blocks are intact;
    and every boundary block does read something remote (the split is exact here: a face neighbour is read or it is not)."""
    g0 = cu.operators.Grid(bpd, level + 1, level, 2 * np.pi, bc)
    m = O.OracleMesh(bpd, level + 1, 2 * np.pi, bc, g0.tables[:, 0].astype(np.int32), g0.tables[:, 1].copy())
    nb = g0.nblocks
    assert np.array_equal(g0.tables[:, :2], m.tables[:, :2])
    rng = np.random.default_rng(7)
    fld = rng.uniform(-1, 1, (nb, 8, 8, 8, 3))
    intact = m.labs(fld, -3, 4)
    idx = np.arange(14) - 3
    out = (idx < 0) | (idx >= 8)
    mask = (out[:, None, None].astype(int) + out[None, :, None] + out[None, None, :]) <= 1
    start, total_inner = 0, 0
    for r in range(nranks):
        g = cu.operators.Grid(bpd, level + 1, level, 2 * np.pi, bc, rank=r, nranks=nranks)
        n = g.nblocks
        assert np.array_equal(g.tables[:, :2], g0.tables[start:start + n, :2])     # the rank's contiguous range of the global order
        ninner = int(L.cup3d_grid_ninner(g.handle))
        inner = np.zeros(max(ninner, 1), dtype=np.int32)
        assert L.cup3d_debug_grid_inner_blocks(g.handle, inner.ctypes.data_as(C.c_void_p)) == 0
        inner = inner[:ninner]
        f = np.full_like(fld, np.nan)
        f[start:start + n] = fld[start:start + n]
        got, want = m.labs(f, -3, 4)[start:start + n][:, mask], intact[start:start + n][:, mask]
        reads_remote = ~((got == want) | (np.isnan(got) & np.isnan(want))).reshape(n, -1).all(axis=1)
        assert not reads_remote[inner].any(), (r, np.flatnonzero(reads_remote[inner]))
        boundary = np.setdiff1d(np.arange(n), inner)
        assert reads_remote[boundary].all(), (r, boundary[~reads_remote[boundary]])
        total_inner += ninner
        start += n
    assert start == nb
    print(f"{nranks} ranks, {nb} blocks: {total_inner} inner blocks")
