"""ctypes binding of the CPU oracle (oracle/libcup3d_oracle.so) and a driver for the
compiled reference (oracle/_ref/ref_tool).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

Field layout everywhere in this module = the reference's block memory:
scalar [nb,8,8,8] (z,y,x), vector [nb,8,8,8,3]; block order = reference m_vInfo order.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libcup3d_oracle.so")
REF_TOOL = os.path.join(ORACLE_DIR, "_ref", "ref_tool")

BC = {"freespace": 0, "periodic": 1, "wall": 2}
BC_NAMES = {v: k for k, v in BC.items()}

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class SolveInfo(C.Structure):
    _fields_ = [("tol", C.c_double), ("tol_rel", C.c_double), ("mean_constraint", C.c_int),
                ("iters", C.c_int), ("restarts", C.c_int), ("norm0", C.c_double), ("norm", C.c_double)]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libcup3d_oracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    L = C.CDLL(ORACLE_SO)
    vp = C.c_void_p
    L.orc_sfc_create.restype = vp
    L.orc_sfc_create.argtypes = [C.c_int] * 4
    L.orc_sfc_destroy.argtypes = [vp]
    L.orc_sfc_forward.restype = C.c_longlong
    L.orc_sfc_forward.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_sfc_inverse.argtypes = [vp, C.c_longlong, C.c_int, _ip]
    L.orc_sfc_encode.restype = C.c_longlong
    L.orc_sfc_encode.argtypes = [vp, C.c_int, _ip]
    L.orc_info_tables.argtypes = [vp, _ip, C.c_int, _ip, _lp, _lp, _lp]
    L.orc_grid_create.restype = vp
    L.orc_grid_create.argtypes = [C.c_int] * 5 + [C.c_double, _ip]
    L.orc_grid_destroy.argtypes = [vp]
    L.orc_grid_nblocks.restype = C.c_long
    L.orc_grid_nblocks.argtypes = [vp]
    L.orc_grid_h.restype = C.c_double
    L.orc_grid_h.argtypes = [vp]
    L.orc_grid_tables.argtypes = [vp, _lp, _dp]
    L.orc_partition.argtypes = [C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.orc_ic_taylor_green.argtypes = [vp, _dp, _dp, C.c_double]
    L.orc_max_u.restype = C.c_double
    L.orc_max_u.argtypes = [vp, _dp, _dp]
    L.orc_calc_dt.restype = C.c_double
    L.orc_calc_dt.argtypes = [C.c_double] * 4 + [C.c_int, C.c_int, C.c_double, _dp]
    L.orc_calc_dt2.restype = C.c_double
    L.orc_calc_dt2.argtypes = [C.c_double] * 4 + [C.c_int, C.c_int, C.c_double, _dp, C.c_int]
    L.orc_external_forcing.argtypes = [vp, _dp] + [C.c_double] * 4
    L.orc_advect_diffuse.argtypes = [vp, _dp, _dp, C.c_double, C.c_double, _dp]
    L.orc_advdiff_stage_rhs.argtypes = [vp, _dp, _dp, C.c_double, C.c_double, _dp]
    L.orc_lhs.argtypes = [vp, _dp, _dp, C.c_int]
    L.orc_precond.argtypes = [vp, _dp]
    L.orc_solve.argtypes = [vp, _dp, _dp, C.POINTER(SolveInfo)]
    L.orc_pressure_rhs.argtypes = [vp, _dp, _dp, _dp, _dp, C.c_double]
    L.orc_div_pressure.argtypes = [vp, _dp, _dp]
    L.orc_grad_p.argtypes = [vp, _dp, _dp, C.c_double]
    L.orc_project.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.POINTER(SolveInfo)]
    L.orc_restrict.argtypes = [vp, vp, _dp, _dp, C.c_int]
    L.orc_prolong.argtypes = [vp, vp, _dp, _dp, C.c_int, C.c_int]
    L.orc_tag.argtypes = [vp, _dp, C.c_int, C.c_double, C.c_double, np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")]
    L.orc_mesh_create.restype = vp
    L.orc_mesh_create.argtypes = [C.c_int] * 4 + [C.c_double, _ip, C.c_long, _ip, _lp]
    L.orc_mesh_destroy.argtypes = [vp]
    L.orc_mesh_nblocks.restype = C.c_long
    L.orc_mesh_nblocks.argtypes = [vp]
    L.orc_mesh_tables.argtypes = [vp, _lp]
    L.orc_mesh_h.restype = C.c_double
    L.orc_mesh_h.argtypes = [vp, C.c_long]
    L.orc_mesh_labs.argtypes = [vp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
    L.orc_mesh_advdiff_stage_rhs.argtypes = [vp, _dp, _dp, C.c_double, C.c_double, _dp]
    L.orc_mesh_advect_diffuse.argtypes = [vp, _dp, _dp, C.c_double, C.c_double, _dp]
    L.orc_mesh_lhs.argtypes = [vp, _dp, _dp, C.c_int]
    L.orc_mesh_precond.argtypes = [vp, _dp]
    L.orc_mesh_solve.argtypes = [vp, _dp, _dp, C.POINTER(SolveInfo)]
    L.orc_mesh_pressure_rhs.argtypes = [vp, _dp, _dp, _dp, _dp, C.c_double]
    L.orc_mesh_div_pressure.argtypes = [vp, _dp, _dp]
    L.orc_mesh_grad_p.argtypes = [vp, _dp, _dp, C.c_double]
    L.orc_mesh_project.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.POINTER(SolveInfo)]
    L.orc_mesh_states.argtypes = [vp, _ip]
    L.orc_mesh_penalize.argtypes = [vp, _dp, _dp, C.c_long, _lp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_int, _dp]
    L.orc_mesh_update_tmpv.argtypes = [vp, _dp, _dp, C.c_long, _lp, _dp, _dp]
    _bp = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
    L.orc_mesh_valid_states.argtypes = [vp, _bp]
    L.orc_mesh_adapted_leaves.restype = C.c_long
    L.orc_mesh_adapted_leaves.argtypes = [vp, _bp, _ip, _lp]
    L.orc_mesh_transfer.argtypes = [vp, vp, _dp, _dp, C.c_int, C.c_int]
    L.orc_mesh_adapted_owners.argtypes = [vp, _ip, _bp, C.c_int, vp, _ip]
    L.orc_mesh_vorticity.argtypes = [vp, _dp, _dp]
    L.orc_mesh_tag.argtypes = [vp, _dp, C.c_int, C.c_double, C.c_double, np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")]
    L.orc_mesh_project_obst.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.POINTER(SolveInfo), C.c_long, _lp, _dp, _dp]
    L.orc_mesh_max_u.restype = C.c_double
    L.orc_mesh_max_u.argtypes = [vp, _dp, _dp]
    L.orc_mesh_advect_implicit.argtypes = [vp, _dp, _dp, C.c_double, C.c_double, _dp, C.c_int]
    L.orc_mesh_diffusion_rhs.argtypes = [vp, _dp, _dp]
    L.orc_mesh_grad_chi_on_tmp.argtypes = [vp, _dp, _dp, C.c_double, C.c_double, C.c_int]
    L.orc_mesh_diff_lhs.argtypes = [vp, _dp, _dp, C.c_int, C.c_double, C.c_double]
    L.orc_mesh_diff_precond.argtypes = [vp, _dp, C.c_double, C.c_double]
    L.orc_mesh_diff_solve.argtypes = [vp, _dp, _dp, C.c_int, C.c_double, C.c_double, C.POINTER(SolveInfo)]
    L.orc_mesh_advdiff_implicit.argtypes = [vp, _dp, _dp, _dp, _dp, C.c_double, C.c_double, _dp, C.c_double, C.c_double, C.c_int, _ip]
    _lib = L
    return L


class OracleGrid:
    """Uniform single-level block grid in the reference's block order."""

    def __init__(self, bpd, level_max, level, maxextent, bc):
        self.bpd = tuple(int(b) for b in bpd)
        self.level_max, self.level, self.maxextent = int(level_max), int(level), float(maxextent)
        self.bc = tuple(BC[b] if isinstance(b, str) else int(b) for b in bc)
        self._bc = np.array(self.bc, dtype=np.int32)
        self.g = lib().orc_grid_create(*self.bpd, self.level_max, self.level, self.maxextent, self._bc)
        self.nb = lib().orc_grid_nblocks(self.g)
        self.h = lib().orc_grid_h(self.g)
        t = np.zeros((self.nb, 6), dtype=np.int64)
        geom = np.zeros((self.nb, 4), dtype=np.float64)
        lib().orc_grid_tables(self.g, t, geom)
        self.tables, self.geom = t, geom
        self.index = t[:, 2:5].copy()
        self.ncell = tuple((b << self.level) * 8 for b in self.bpd)  # (NX, NY, NZ)

    def __del__(self):
        try:
            lib().orc_grid_destroy(self.g)
        except Exception:
            pass

    # ---- layout helpers: global [NZ,NY,NX(,3)] <-> block order ----
    def to_blocks(self, glob):
        glob = np.asarray(glob, dtype=np.float64)
        nc = glob.shape[3:]
        out = np.empty((self.nb, 8, 8, 8) + nc)
        for s, (i, j, k) in enumerate(self.index):
            out[s] = glob[8 * k:8 * k + 8, 8 * j:8 * j + 8, 8 * i:8 * i + 8]
        return np.ascontiguousarray(out)

    def to_global(self, blocks):
        nc = blocks.shape[4:]
        NX, NY, NZ = self.ncell
        out = np.empty((NZ, NY, NX) + nc)
        for s, (i, j, k) in enumerate(self.index):
            out[8 * k:8 * k + 8, 8 * j:8 * j + 8, 8 * i:8 * i + 8] = blocks[s]
        return out

    # ---- operators (all in place on contiguous float64 arrays) ----
    def taylor_green(self, ext, umax):
        v = np.zeros((self.nb, 8, 8, 8, 3))
        lib().orc_ic_taylor_green(self.g, v, np.asarray(ext, dtype=np.float64), umax)
        return v

    def max_u(self, vel, uinf=(0, 0, 0)):
        return lib().orc_max_u(self.g, vel, np.asarray(uinf, dtype=np.float64))

    def advect_diffuse(self, vel, tmpV, dt, nu, uinf=(0, 0, 0)):
        lib().orc_advect_diffuse(self.g, vel, tmpV, dt, nu, np.asarray(uinf, dtype=np.float64))

    def advdiff_stage_rhs(self, vel, tmpV, dt, nu, uinf=(0, 0, 0)):
        lib().orc_advdiff_stage_rhs(self.g, vel, tmpV, dt, nu, np.asarray(uinf, dtype=np.float64))

    def lhs(self, pres, mean_constraint=1):
        out = np.zeros_like(pres)
        lib().orc_lhs(self.g, pres, out, mean_constraint)
        return out

    def precond(self, pres):
        lib().orc_precond(self.g, pres)

    def solve(self, lhs, pres, tol=1e-6, tol_rel=1e-4, mean_constraint=1):
        info = SolveInfo(tol, tol_rel, mean_constraint, 0, 0, 0.0, 0.0)
        lib().orc_solve(self.g, lhs, pres, C.byref(info))
        return info

    def pressure_rhs(self, vel, udef, chi, dt):
        out = np.zeros(vel.shape[:4])
        lib().orc_pressure_rhs(self.g, vel, udef, chi, out, dt)
        return out

    def div_pressure(self, pres):
        out = np.zeros(pres.shape + (3,))
        lib().orc_div_pressure(self.g, pres, out)
        return out

    def grad_p(self, pres, dt):
        out = np.zeros(pres.shape + (3,))
        lib().orc_grad_p(self.g, pres, out, dt)
        return out

    def project(self, vel, pres, dt, step, tol=1e-6, tol_rel=1e-4, mean_constraint=1, chi=None):
        tmpV = np.zeros_like(vel)
        lhs = np.zeros_like(pres)
        chi = np.zeros_like(pres) if chi is None else chi
        info = SolveInfo(tol, tol_rel, mean_constraint, 0, 0, 0.0, 0.0)
        lib().orc_project(self.g, vel, pres, tmpV, lhs, chi, dt, step, C.byref(info))
        return info, tmpV, lhs


class OracleMesh:
    """Multi-level leaf-block mesh (levels, Zs in any order; stored in the reference's m_vInfo order)."""

    def __init__(self, bpd, level_max, maxextent, bc, levels, Zs):
        self.bpd = tuple(int(b) for b in bpd)
        self.level_max, self.maxextent = int(level_max), float(maxextent)
        self.bc = tuple(BC[b] if isinstance(b, str) else int(b) for b in bc)
        lv = np.ascontiguousarray(levels, dtype=np.int32)
        zs = np.ascontiguousarray(Zs, dtype=np.int64)
        self.m = lib().orc_mesh_create(*self.bpd, self.level_max, self.maxextent, np.array(self.bc, dtype=np.int32), len(lv), lv, zs)
        self.nb = lib().orc_mesh_nblocks(self.m)
        self.tables = np.zeros((self.nb, 6), dtype=np.int64)
        lib().orc_mesh_tables(self.m, self.tables)

    def __del__(self):
        try:
            lib().orc_mesh_destroy(self.m)
        except Exception:
            pass

    def labs(self, field, s, e, tensorial=False):
        nc = 3 if field.ndim == 5 else 1
        L = 8 + e - s - 1
        out = np.zeros((self.nb, L, L, L, nc))
        lib().orc_mesh_labs(self.m, np.ascontiguousarray(field), nc, 1 if nc == 3 else 0, s, e, 1 if tensorial else 0, out)
        return out

    def h(self, b):
        return lib().orc_mesh_h(self.m, b)

    def advect_diffuse(self, vel, dt, nu, uinf):
        vel = np.ascontiguousarray(vel).copy()
        tmpV = np.zeros_like(vel)
        lib().orc_mesh_advect_diffuse(self.m, vel, tmpV, dt, nu, np.asarray(uinf, dtype=np.float64))
        return vel, tmpV

    def lhs(self, pres, mean_constraint=1):
        out = np.zeros_like(pres)
        lib().orc_mesh_lhs(self.m, np.ascontiguousarray(pres), out, mean_constraint)
        return out

    def precond(self, pres):
        lib().orc_mesh_precond(self.m, pres)

    def solve(self, lhs, pres, tol=1e-6, tol_rel=1e-4, mean_constraint=1):
        info = SolveInfo(tol, tol_rel, mean_constraint, 0, 0, 0.0, 0.0)
        lib().orc_mesh_solve(self.m, lhs, pres, C.byref(info))
        return info

    def pressure_rhs(self, vel, udef, chi, dt):
        out = np.zeros(vel.shape[:4])
        lib().orc_mesh_pressure_rhs(self.m, vel, udef, chi, out, dt)
        return out

    def div_pressure(self, pres):
        out = np.zeros(pres.shape + (3,))
        lib().orc_mesh_div_pressure(self.m, pres, out)
        return out

    def grad_p(self, pres, dt):
        out = np.zeros(pres.shape + (3,))
        lib().orc_mesh_grad_p(self.m, pres, out, dt)
        return out

    def grad_chi_on_tmp(self, chi, tmpV, Rtol, Ctol, level_max_vorticity=None):
        """GradChiOnTmp (main.cpp:8540-8600): returns tmpV (the vorticity) edited from chi."""
        out = np.ascontiguousarray(tmpV).copy()
        lib().orc_mesh_grad_chi_on_tmp(self.m, np.ascontiguousarray(chi), out, Rtol, Ctol, self.level_max if level_max_vorticity is None else level_max_vorticity)
        return out

    def project(self, vel, pres, dt, step, tol=1e-6, tol_rel=1e-4, mean_constraint=1, chi=None):
        tmpV = np.zeros_like(vel)
        lhs = np.zeros_like(pres)
        chi = np.zeros_like(pres) if chi is None else chi
        info = SolveInfo(tol, tol_rel, mean_constraint, 0, 0, 0.0, 0.0)
        lib().orc_mesh_project(self.m, vel, pres, tmpV, lhs, chi, dt, step, C.byref(info))
        return info, tmpV, lhs

    def project_obst(self, vel, pres, dt, step, chi_field, obst, tol=1e-6, tol_rel=1e-4, mean_constraint=1):
        """PressureProjection with one obstacle (chi in the RHS, udef through tmpV); vel and pres are updated in place."""
        tmpV, lhs = np.zeros_like(vel), np.zeros_like(pres)
        info = SolveInfo(tol, tol_rel, mean_constraint, 0, 0, 0.0, 0.0)
        lib().orc_mesh_project_obst(self.m, vel, pres, tmpV, lhs, np.ascontiguousarray(chi_field), dt, step, C.byref(info), len(obst["ids"]),
                                    np.ascontiguousarray(obst["ids"], dtype=np.int64), np.ascontiguousarray(obst["chi"]), np.ascontiguousarray(obst["udef"]))
        return info

    def max_u(self, vel, uinf=(0, 0, 0)):
        return lib().orc_mesh_max_u(self.m, vel, np.asarray(uinf, dtype=np.float64))

    def vorticity(self, vel):
        out = np.zeros_like(vel)
        lib().orc_mesh_vorticity(self.m, np.ascontiguousarray(vel), out)
        return out

    def tag(self, field, rtol, ctol):
        nc = 3 if field.ndim == 5 else 1
        st = np.zeros(self.nb, dtype=np.int8)
        lib().orc_mesh_tag(self.m, np.ascontiguousarray(field), nc, rtol, ctol, st)
        return st

    def valid_states(self, tags):
        st = np.ascontiguousarray(tags, dtype=np.int8).copy()
        lib().orc_mesh_valid_states(self.m, st)
        return st

    def adapted(self, states):
        """The mesh MeshAdaptation::Adapt produces from (valid) states."""
        st = np.ascontiguousarray(states, dtype=np.int8)
        n = self.nb + 7 * int((st == 1).sum())
        lv, zs = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int64)
        n = lib().orc_mesh_adapted_leaves(self.m, st, lv, zs)
        return OracleMesh(self.bpd, self.level_max, self.maxextent, self.bc, lv[:n], zs[:n])

    def adapted_owners(self, owner, states, nranks, new_mesh):
        """Rank of every leaf of `new_mesh` (= self.adapted(states)) after MeshAdaptation::Adapt + LoadBalancer on `nranks` ranks."""
        out = np.zeros(new_mesh.nb, dtype=np.int32)
        lib().orc_mesh_adapted_owners(self.m, np.ascontiguousarray(owner, dtype=np.int32), np.ascontiguousarray(states, dtype=np.int8), nranks, new_mesh.m, out)
        return out

    def transfer(self, new_mesh, field):
        nc = 3 if field.ndim == 5 else 1
        out = np.zeros((new_mesh.nb, 8, 8, 8, 3) if nc == 3 else (new_mesh.nb, 8, 8, 8))
        lib().orc_mesh_transfer(self.m, new_mesh.m, np.ascontiguousarray(field), out, nc, 1 if nc == 3 else 0)
        return out

    def penalize(self, vel, chi_field, obst, dt, lam, implicit):
        """obst = dict(ids, chi[n,8,8,8], udef[n,8,8,8,3], rigid[9]); returns (vel', force6)."""
        v = np.ascontiguousarray(vel).copy()
        f6 = np.zeros(6)
        lib().orc_mesh_penalize(self.m, v, np.ascontiguousarray(chi_field), len(obst["ids"]), np.ascontiguousarray(obst["ids"], dtype=np.int64),
                                np.ascontiguousarray(obst["chi"]), np.ascontiguousarray(obst["udef"]), np.ascontiguousarray(obst["rigid"]),
                                dt, lam, 1 if implicit else 0, f6)
        return v, f6

    def update_tmpv(self, tmpV, chi_field, obst):
        t = np.ascontiguousarray(tmpV).copy()
        lib().orc_mesh_update_tmpv(self.m, t, np.ascontiguousarray(chi_field), len(obst["ids"]), np.ascontiguousarray(obst["ids"], dtype=np.int64),
                                   np.ascontiguousarray(obst["chi"]), np.ascontiguousarray(obst["udef"]))
        return t

    # ---- implicit diffusion (AdvectionDiffusionImplicit, main.cpp:10030-10118)
    def advect_implicit(self, vel, dt, nu, uinf, sequential):
        """KernelAdvect: returns (vel', tmpV).  sequential=True is the reference with one thread (in-place update seen by
        later blocks), False the order-independent reading the device implements."""
        v = np.ascontiguousarray(vel).copy()
        t = np.zeros_like(v)
        lib().orc_mesh_advect_implicit(self.m, v, t, dt, nu, np.asarray(uinf, dtype=np.float64), 1 if sequential else 0)
        return v, t

    def diffusion_rhs(self, vel):
        out = np.zeros_like(vel)
        lib().orc_mesh_diffusion_rhs(self.m, np.ascontiguousarray(vel), out)
        return out

    def diff_lhs(self, pres, direction, dt, nu):
        out = np.zeros_like(pres)
        lib().orc_mesh_diff_lhs(self.m, np.ascontiguousarray(pres), out, direction, dt, nu)
        return out

    def diff_precond(self, pres, dt, nu):
        p = np.ascontiguousarray(pres).copy()
        lib().orc_mesh_diff_precond(self.m, p, dt, nu)
        return p

    def diff_solve(self, rhs, x0, direction, dt, nu, tol=1e-6, tol_rel=1e-4):
        lhs, p = np.ascontiguousarray(rhs).copy(), np.ascontiguousarray(x0).copy()
        info = SolveInfo(tol, tol_rel, 0, 0, 0, 0.0, 0.0)
        lib().orc_mesh_diff_solve(self.m, lhs, p, direction, dt, nu, C.byref(info))
        return p, info

    def advdiff_implicit(self, vel, pres, dt, nu, uinf, sequential, tol=1e-6, tol_rel=1e-4):
        """AdvectionDiffusionImplicit::euler: returns (vel', iterations[3]); pres is scratch and comes back unchanged."""
        v, p = np.ascontiguousarray(vel).copy(), np.ascontiguousarray(pres).copy()
        t, l = np.zeros_like(v), np.zeros_like(p)
        it = np.zeros(3, dtype=np.int32)
        lib().orc_mesh_advdiff_implicit(self.m, v, p, t, l, dt, nu, np.asarray(uinf, dtype=np.float64), tol, tol_rel, 1 if sequential else 0, it)
        assert np.array_equal(p, pres)
        return v, it

    def states(self):
        out = np.zeros((self.nb, 27), dtype=np.int32)
        lib().orc_mesh_states(self.m, out)
        return out


def restrict_field(fine, coarse, field):
    """MeshAdaptation::compress of every sibling octet (fine level -> coarse level)."""
    nc = 3 if field.ndim == 5 else 1
    out = np.zeros((coarse.nb, 8, 8, 8, 3) if nc == 3 else (coarse.nb, 8, 8, 8))
    lib().orc_restrict(fine.g, coarse.g, np.ascontiguousarray(field), out, nc)
    return out


def prolong_field(coarse, fine, field):
    """refine_1 + RefineBlocks of every block (coarse level -> fine level)."""
    nc = 3 if field.ndim == 5 else 1
    out = np.zeros((fine.nb, 8, 8, 8, 3) if nc == 3 else (fine.nb, 8, 8, 8))
    lib().orc_prolong(coarse.g, fine.g, np.ascontiguousarray(field), out, nc, 1 if nc == 3 else 0)
    return out


def tag_blocks(grid, field, rtol, ctol):
    nc = 3 if field.ndim == 5 else 1
    st = np.zeros(grid.nb, dtype=np.int8)
    lib().orc_tag(grid.g, np.ascontiguousarray(field), nc, rtol, ctol, st)
    return st


# ------------------------------ compiled reference ------------------------------
def have_ref_tool():
    return os.path.exists(REF_TOOL) and os.access(REF_TOOL, os.X_OK)


def ref_args(bpd, level_max, level_start, extent, bc, nu=0.01, cfl=0.3, ic="taylorGreen", umax_forced=1.0,
             extra=()):
    a = ["-bpdx", bpd[0], "-bpdy", bpd[1], "-bpdz", bpd[2], "-levelMax", level_max, "-levelStart", level_start,
         "-extentx", repr(float(extent)), "-CFL", cfl, "-nu", nu, "-initCond", ic, "-uMax_forced", umax_forced,
         "-BC_x", BC_NAMES[BC[bc[0]]] if isinstance(bc[0], str) else BC_NAMES[bc[0]],
         "-BC_y", BC_NAMES[BC[bc[1]]] if isinstance(bc[1], str) else BC_NAMES[bc[1]],
         "-BC_z", BC_NAMES[BC[bc[2]]] if isinstance(bc[2], str) else BC_NAMES[bc[2]],
         "-Rtol", "1e9", "-Ctol", "0", "-factory", ""]
    return [str(x) for x in a] + [str(x) for x in extra]


def run_ref(script_lines, args, threads=1, workdir=None, timeout=3600):
    """Run oracle/_ref/ref_tool; returns (stdout REF records, workdir)."""
    wd = workdir or tempfile.mkdtemp(prefix="cup3d_ref_")
    with open(os.path.join(wd, "script.txt"), "w") as f:
        f.write("\n".join(script_lines) + "\n")
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    out = subprocess.run([REF_TOOL, "script.txt", "--"] + list(args), cwd=wd, env=env, check=True,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout).stdout.decode()
    recs = []
    for line in out.splitlines():
        if line.startswith("REF "):
            p = line.split()
            d = {"op": p[1]}
            for kv in p[2:]:
                k, v = kv.split("=")
                try:
                    d[k] = float(v)
                except ValueError:   # `REF optime name=<class> ...`
                    d[k] = v
            recs.append(d)
    return recs, wd


REF_TOOL_MPI = os.path.join(ORACLE_DIR, "_ref", "ref_tool_mpi")
MPIEXEC = "/opt/conda/bin/mpiexec"


def have_ref_tool_mpi():
    return os.path.exists(REF_TOOL_MPI) and os.path.exists(MPIEXEC)


def run_ref_mpi(script_lines, args, nranks, workdir=None, timeout=3600):
    """Run oracle/_ref/ref_tool_mpi (the reference TU against a real MPI) on `nranks` ranks; output files get `.r<rank>`."""
    wd = workdir or tempfile.mkdtemp(prefix="cup3d_refmpi_")
    with open(os.path.join(wd, "script.txt"), "w") as f:
        f.write("\n".join(script_lines) + "\n")
    env = dict(os.environ, OMP_NUM_THREADS="1", LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib")
    subprocess.run([MPIEXEC, "-n", str(nranks), REF_TOOL_MPI, "script.txt", "--"] + list(args), cwd=wd, env=env, check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return wd


def lab_mask(s, e, tens):
    """Cells of the [s,e) tile the reference defines: all of them when the tile is built tensorially
    (use_averages), else only faces (edge/corner ghosts are never read by the kernels and hold stale memory)."""
    L = 8 + e - s - 1
    if tens or s < -2 or e > 3:
        return np.ones((L, L, L), bool)
    idx = np.arange(L) + s
    out = ~((idx >= 0) & (idx < 8))
    cnt = out[:, None, None].astype(int) + out[None, :, None].astype(int) + out[None, None, :].astype(int)
    return cnt <= 1


def read_blocks(path, nb, ncomp):
    a = np.fromfile(path, dtype=np.float64)
    shape = (nb, 8, 8, 8, 3) if ncomp == 3 else (nb, 8, 8, 8)
    return a.reshape(shape)


def read_tables(path):
    raw = np.fromfile(path, dtype=np.uint8)
    nb = raw.size // (6 * 8 + 4 * 8)
    t = raw[:nb * 48].view(np.int64).reshape(nb, 6)
    g = raw[nb * 48:].view(np.float64).reshape(nb, 4)
    return t, g


def build_balanced_mesh(bpd, level_max, bc, refine):
    """Leaves (levels, Zs) of a 2:1-balanced multi-level mesh made without the reference: start from the level-0
    blocks, split the leaves listed in `refine` = [(level, i, j, k), ...] in that order, and after each split
    keep splitting coarser leaves until no leaf touches (26-neighbourhood, periodic wrap where the BC is periodic)
    a leaf more than one level away -- the invariant MeshAdaptation::ValidStates (main.cpp:5330-5492) maintains."""
    bcs = [BC[b] if isinstance(b, str) else int(b) for b in bc]
    leaves = {(0, i, j, k) for i in range(bpd[0]) for j in range(bpd[1]) for k in range(bpd[2])}

    def split(leaf):
        l, i, j, k = leaf
        assert l + 1 < level_max
        leaves.remove(leaf)
        for q in range(8):
            leaves.add((l + 1, 2 * i + (q & 1), 2 * j + ((q >> 1) & 1), 2 * k + (q >> 2)))

    def covering(l, c):
        for up in range(l + 1):
            cand = (l - up, c[0] >> up, c[1] >> up, c[2] >> up)
            if cand in leaves:
                return cand
        return None  # finer leaves there

    def balance():
        changed = True
        while changed:
            changed = False
            for leaf in sorted(leaves, key=lambda t: -t[0]):
                l = leaf[0]
                for code in range(27):
                    d = (code % 3 - 1, (code // 3) % 3 - 1, code // 9 - 1)
                    c, ok = [], True
                    for a in range(3):
                        n = bpd[a] << l
                        v = leaf[1 + a] + d[a]
                        if v < 0 or v >= n:
                            if bcs[a] != 1:
                                ok = False
                            v %= n
                        c.append(v)
                    if not ok:
                        continue
                    cov = covering(l, c)
                    if cov is not None and cov[0] < l - 1:
                        split(cov)
                        changed = True
                        break
                if changed:
                    break

    for leaf in refine:
        if tuple(leaf) in leaves:  # may already have been split by the balancing
            split(tuple(leaf))
            balance()
    sfc = lib().orc_sfc_create(int(bpd[0]), int(bpd[1]), int(bpd[2]), int(level_max))
    out = sorted(leaves)
    levels = np.array([t[0] for t in out], dtype=np.int32)
    Zs = np.array([lib().orc_sfc_forward(sfc, t[0], t[1], t[2], t[3]) for t in out], dtype=np.int64)
    lib().orc_sfc_destroy(sfc)
    return levels, Zs


def synthetic_obstacle(tables_h, nb, seed, frac=0.4):
    """A synthetic obstacle for the harness's `obstacle` command and the oracle: ObstacleBlocks on a random subset of the blocks
    with a smooth-ish chi in [0,1] (zeros included) and a small deformation velocity, plus a rigid motion.
    Returns (obst dict, chi_field [nb,8,8,8] = what CreateObstacles would leave in sim.chi for a single obstacle)."""
    rng = np.random.default_rng(seed)
    ids = np.sort(rng.choice(nb, size=max(1, int(frac * nb)), replace=False)).astype(np.int64)
    chi = rng.uniform(-0.4, 1.2, (len(ids), 8, 8, 8)).clip(0.0, 1.0)
    udef = 0.1 * rng.uniform(-1, 1, (len(ids), 8, 8, 8, 3))
    rigid = np.concatenate([rng.uniform(1.0, 4.0, 3), rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.2, 0.2, 3)])
    chi_field = np.zeros((nb, 8, 8, 8))
    chi_field[ids] = chi
    # a few cells where the grid's chi exceeds the obstacle's (another obstacle would own them): exercised `continue` branch
    mask = rng.uniform(size=chi_field.shape) < 0.05
    chi_field[mask] = np.minimum(1.0, chi_field[mask] + 0.3)
    return dict(ids=ids, chi=chi, udef=udef, rigid=rigid), chi_field


def write_obstacle_file(path, obst):
    with open(path, "wb") as f:
        np.array([len(obst["ids"])], dtype=np.int64).tofile(f)
        np.ascontiguousarray(obst["ids"], dtype=np.int64).tofile(f)
        np.ascontiguousarray(obst["chi"], dtype=np.float64).tofile(f)
        np.ascontiguousarray(obst["udef"], dtype=np.float64).tofile(f)
        np.ascontiguousarray(obst["rigid"], dtype=np.float64).tofile(f)
