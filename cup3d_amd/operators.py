"""Host-side mirror of the reference's Operator / PoissonSolverBase surface for the hot
path, driving the C ABI (include/cup3d_hip.h).  Names, argument meaning and call order
follow slitvinov/CUP3D main.cpp so that tests read like the reference:

    sim = SimulationData(bpdx=.., levelMax=.., BC_x="periodic", nu=.., ...)   # 15330-15387
    pipeline = [AdvectionDiffusion(sim), ExternalForcing(sim), PressureProjection(sim)]  # 15229-15246
    dt = Simulation(sim).calcMaxTimestep(); for op in pipeline: op(dt)                   # 15254-15326

All field data lives on the GPU; host arrays cross the boundary only through
upload()/download() in the reference's block memory layout.  No CPU fallback exists.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import (BC, FIELD_CHI, FIELD_LHS, FIELD_NCOMP, FIELD_PRES, FIELD_TMPV, FIELD_VEL, Cup3dError,
                   PoissonParams, PoissonResult, check, lib)

FIELDS = {"chi": FIELD_CHI, "pres": FIELD_PRES, "vel": FIELD_VEL, "tmpV": FIELD_TMPV, "lhs": FIELD_LHS}


class Grid:
    """Block topology of one rank (host only): GridMPI ownership + m_vInfo order + face
    neighbours + halo plan.  Works without a GPU."""

    def __init__(self, bpd, levelMax, level, maxextent, bc, rank=0, nranks=1, leaves=None):
        """Uniform grid at `level`, or -- with leaves=(levels, Zs) -- the multi-level mesh made of those leaf blocks
        (what MeshAdaptation leaves in m_vInfo; one rank)."""
        self.bpd = np.array(bpd, dtype=np.int32)
        self.bc = np.array([BC[b] if isinstance(b, str) else int(b) for b in bc], dtype=np.int32)
        self.levelMax, self.level, self.maxextent = int(levelMax), int(level), float(maxextent)
        self.rank, self.nranks = int(rank), int(nranks)
        self.multilevel = leaves is not None
        h = C.c_void_p()
        if self.multilevel:
            lv = np.ascontiguousarray(leaves[0], dtype=np.int32)
            zs = np.ascontiguousarray(leaves[1], dtype=np.int64)
            check(lib().cup3d_grid_create_mesh(self.bpd, self.levelMax, self.maxextent, self.bc, len(lv), lv, zs, C.byref(h)))
        else:
            check(lib().cup3d_grid_create_uniform(self.bpd, self.levelMax, self.level, self.maxextent, self.bc,
                                                  self.rank, self.nranks, C.byref(h)))
        self.handle = h
        self.nblocks = lib().cup3d_grid_nblocks(h)
        self.nblocks_global = lib().cup3d_grid_nblocks_global(h)
        self.tables = np.zeros((self.nblocks, 6), dtype=np.int64)
        self.geom = np.zeros((self.nblocks, 4), dtype=np.float64)
        check(lib().cup3d_grid_tables(h, self.tables, self.geom))
        self.index = self.tables[:, 2:5]
        self.h = float(self.geom[0, 0])
        self.ncell = tuple(int(b << self.level) * 8 for b in self.bpd)

    def __del__(self):
        try:
            lib().cup3d_grid_destroy(self.handle)
        except Exception:
            pass

    def valid_states(self, tags):
        """MeshAdaptation::ValidStates (main.cpp:5330-5492) of per-block tags -> the states Adapt will act on."""
        st = np.ascontiguousarray(tags, dtype=np.int8).copy()
        check(lib().cup3d_grid_valid_states(self.handle, st))
        return st

    def adapted_leaves(self, states):
        """(levels, Zs) of the mesh MeshAdaptation::Adapt (5086-5159) produces from valid states."""
        h = C.c_void_p()
        check(lib().cup3d_grid_adapted(self.handle, np.ascontiguousarray(states, dtype=np.int8), C.byref(h)))
        n = lib().cup3d_grid_nblocks(h)
        t, geom = np.zeros((n, 6), dtype=np.int64), np.zeros((n, 4))
        check(lib().cup3d_grid_tables(h, t, geom))
        lib().cup3d_grid_destroy(h)
        return t[:, 0].astype(np.int32), t[:, 1].copy()

    def adapted_owners(self, owner, states, nranks, adapted):
        """Rank of every leaf of `adapted` (a Grid built from self.adapted_leaves(states)) after MeshAdaptation::Adapt and the
        LoadBalancer on `nranks` ranks, given the rank of every leaf of this mesh (cup3d_grid_adapted_owners)."""
        ow = np.ascontiguousarray(owner, dtype=np.int32)
        st = np.ascontiguousarray(states, dtype=np.int8)
        out = np.zeros(adapted.nblocks, dtype=np.int32)
        check(lib().cup3d_grid_adapted_owners(self.handle, ow.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), int(nranks),
                                              adapted.handle, out.ctypes.data_as(C.c_void_p)))
        return out

    def rank_view(self, owner, rank, nranks):
        """One rank's view of this multi-level mesh when its leaves are spread over `nranks` ranks as owner[nblocks] says
        (cup3d_grid_rank_view): a RankView with the renumbered tables and the ghost-block / face-flux exchange plans."""
        return RankView(self, owner, rank, nranks)

    def interface(self):
        """Multi-level meshes: (faces[ne,2] = 6*slot+face, kind; fine4[ne,4]; nbr27[nb,27]), see cup3d_grid_interface."""
        ne = lib().cup3d_grid_ninterface_faces(self.handle)
        faces = np.zeros((max(ne, 1), 2), dtype=np.int32)
        fine = np.zeros((max(ne, 1), 4), dtype=np.int32)
        n27 = np.zeros((self.nblocks, 27), dtype=np.int32)
        check(lib().cup3d_grid_interface(self.handle, faces.ctypes.data_as(C.c_void_p), fine.ctypes.data_as(C.c_void_p),
                                         n27.ctypes.data_as(C.c_void_p)))
        return faces[:ne], fine[:ne], n27

    def neighbours(self):
        nbr = np.zeros((self.nblocks, 6), dtype=np.int32)
        check(lib().cup3d_grid_neighbours(self.handle, nbr))
        return nbr

    def halo_plan(self):
        send = np.zeros(self.nranks, dtype=np.int64)
        recv = np.zeros(self.nranks, dtype=np.int64)
        faces = np.zeros(max(1, lib().cup3d_grid_nsend_faces(self.handle)), dtype=np.int32)
        check(lib().cup3d_grid_halo_plan(self.handle, send, recv, faces.ctypes.data_as(C.c_void_p)))
        return send, recv, faces[:lib().cup3d_grid_nsend_faces(self.handle)]

    # global [NZ,NY,NX(,3)] array <-> this rank's blocks in the reference layout
    def to_blocks(self, glob):
        glob = np.asarray(glob, dtype=np.float64)
        out = np.empty((self.nblocks, 8, 8, 8) + glob.shape[3:])
        for s, (i, j, k) in enumerate(self.index):
            out[s] = glob[8 * k:8 * k + 8, 8 * j:8 * j + 8, 8 * i:8 * i + 8]
        return np.ascontiguousarray(out)

    def scatter_to_global(self, blocks, glob):
        for s, (i, j, k) in enumerate(self.index):
            glob[8 * k:8 * k + 8, 8 * j:8 * j + 8, 8 * i:8 * i + 8] = blocks[s]


class RankView:
    """One rank's view of a multi-level mesh whose leaves are spread over ranks (Grid.rank_view): local blocks, ghost blocks, interface
    faces, neighbour tables in the view's slot numbering, and the two exchange plans (whole ghost blocks before a stencil kernel,
    face-flux arrays after a flux-corrected one).  SimulationData(view=...) runs the operators on it."""

    def __init__(self, mesh, owner, rank, nranks):
        ow = np.ascontiguousarray(owner, dtype=np.int32)
        h = C.c_void_p()
        check(lib().cup3d_grid_rank_view(mesh.handle, ow.ctypes.data_as(C.c_void_p), int(rank), int(nranks), C.byref(h)))
        self.handle, self.rank, self.nranks = h, int(rank), int(nranks)
        self.bpd, self.bc, self.levelMax, self.level, self.maxextent, self.multilevel = mesh.bpd, mesh.bc, mesh.levelMax, mesh.level, mesh.maxextent, True
        self.nblocks = lib().cup3d_grid_nblocks(h)
        self.nblocks_global = lib().cup3d_grid_nblocks_global(h)
        self.tables = np.zeros((self.nblocks, 6), dtype=np.int64)
        self.geom = np.zeros((self.nblocks, 4), dtype=np.float64)
        check(lib().cup3d_grid_tables(h, self.tables, self.geom))
        self.index = self.tables[:, 2:5]
        self.h = mesh.h
        self.ninner = int(lib().cup3d_grid_ninner(h))  # local blocks whose tables lead to no ghost: computed while the ghost blocks travel
        sz = (C.c_long * 6)()
        check(lib().cup3d_grid_view_sizes(h, sz))
        self.nlocal, self.nghost, self.nfaces_local, self.nfaces_ghost, nsb, nsf = (int(v) for v in sz)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        self.global_slot = np.zeros(self.nlocal + self.nghost, dtype=np.int32)
        self.global_face = np.zeros(max(self.nfaces_local + self.nfaces_ghost, 1), dtype=np.int32)
        self.send_blocks, self.send_flux_faces = np.zeros(max(nsb, 1), dtype=np.int32), np.zeros(max(nsf, 1), dtype=np.int32)
        counts = [np.zeros(nranks, dtype=np.int64) for _ in range(4)]
        check(lib().cup3d_grid_view_plan(h, p(self.global_slot), p(self.global_face), p(self.send_blocks), p(counts[0]), p(counts[1]),
                                         p(self.send_flux_faces), p(counts[2]), p(counts[3])))
        self.global_face = self.global_face[:self.nfaces_local + self.nfaces_ghost]
        self.send_blocks, self.send_flux_faces = self.send_blocks[:nsb], self.send_flux_faces[:nsf]
        self.send_block_count, self.recv_block_count, self.send_flux_count, self.recv_flux_count = counts
        self.nbr = np.zeros((self.nlocal, 6), dtype=np.int32)
        check(lib().cup3d_grid_neighbours(h, self.nbr.reshape(-1)))
        nf = self.nfaces_local + self.nfaces_ghost
        self.faces, self.fine = np.zeros((max(nf, 1), 2), dtype=np.int32), np.zeros((max(nf, 1), 4), dtype=np.int32)
        self.nbr27 = np.zeros((self.nlocal, 27), dtype=np.int32)
        check(lib().cup3d_grid_interface(h, p(self.faces), p(self.fine), p(self.nbr27)))
        self.faces, self.fine = self.faces[:nf], self.fine[:nf]
        # sub-box form of the ghost-block exchange, per stencil-width class (0: w = 1, 1: w = 3): boxes lo x,y,z / hi x,y,z and cells per rank
        self.ghost_box, self.send_box, self.send_cells, self.recv_cells = [], [], [], []
        for k in (0, 1):
            gb, sb = np.zeros((max(self.nghost, 1), 6), dtype=np.uint8), np.zeros((max(nsb, 1), 6), dtype=np.uint8)
            sc, rc = np.zeros(nranks, dtype=np.int64), np.zeros(nranks, dtype=np.int64)
            check(lib().cup3d_grid_view_boxes(h, k, p(gb), p(sb), p(sc), p(rc)))
            self.ghost_box.append(gb[:self.nghost]); self.send_box.append(sb[:nsb]); self.send_cells.append(sc); self.recv_cells.append(rc)

    def __del__(self):
        try:
            lib().cup3d_grid_destroy(self.handle)
        except Exception:
            pass


class SimulationData:
    """Device-resident counterpart of struct SimulationData (main.cpp:6600-6677): the five
    block grids chi, pres, vel, tmpV, lhs plus the run parameters the hot path reads."""

    def __init__(self, bpdx=1, bpdy=1, bpdz=1, levelMax=1, levelStart=None, extent=1.0, nu=0.0, CFL=0.1,
                 BC_x="freespace", BC_y="freespace", BC_z="freespace", uinf=(0.0, 0.0, 0.0), uMax_forced=0.0,
                 poissonTol=1e-6, poissonTolRel=1e-4, bMeanConstraint=1, poissonSolver="hip_iterative", rampup=100,
                 blockSolver=0, leaves=None, implicitDiffusion=False, diffusionTol=1e-6, diffusionTolRel=1e-4,
                 rank=0, nranks=1, device=None, view=None):
        if device is not None or not capi._device_ready:
            capi.device_init(0 if device is None else device)
        self.bpdx, self.bpdy, self.bpdz = bpdx, bpdy, bpdz
        self.levelMax = levelMax
        self.levelStart = levelMax - 1 if levelStart is None else levelStart
        self.maxextent = float(extent)
        # leaves=(levels, Zs): run on that multi-level mesh instead of the uniform grid at levelStart
        # view=RankView: this rank's share of a multi-level mesh spread over several ranks (ghost blocks + exchange plans)
        self.grid = view if view is not None else Grid((bpdx, bpdy, bpdz), levelMax, self.levelStart, self.maxextent, (BC_x, BC_y, BC_z), rank, nranks,
                                                       leaves=leaves)
        # extents / hmin as in _preprocessArguments, main.cpp:15394-15415
        aux = 1 << (levelMax - 1)
        nfe = [bpdx * aux * 8, bpdy * aux * 8, bpdz * aux * 8]
        self.extents = [n / max(nfe) * self.maxextent for n in nfe]
        self.hmin = self.extents[0] / nfe[0]
        self.BCx_flag, self.BCy_flag, self.BCz_flag = BC_x, BC_y, BC_z
        self.nu, self.CFL, self.uinf = float(nu), float(CFL), np.array(uinf, dtype=np.float64)
        self.uMax_forced, self.rampup = float(uMax_forced), int(rampup)
        self.PoissonErrorTol, self.PoissonErrorTolRel, self.bMeanConstraint = poissonTol, poissonTolRel, bMeanConstraint
        self.poissonSolver = poissonSolver
        # -implicitDiffusion / -diffusionTol / diffusionTolRel (main.cpp:15368-15370)
        self.implicitDiffusion, self.DiffusionErrorTol, self.DiffusionErrorTolRel = bool(implicitDiffusion), diffusionTol, diffusionTolRel
        self.blockSolver = int(blockSolver)  # 0: block CG as in the reference, 1: direct block solve (fast diagonalisation)
        self.dt, self.dt_old, self.time, self.step, self.step_2nd_start = 0.0, 0.0, 0.0, 0, 2
        self.coefU = np.array([1.5, -2.0, 0.5])
        self.uMax_measured = 0.0
        h = C.c_void_p()
        check(lib().cup3d_sim_create(self.grid.handle, C.byref(h)))
        self.handle = h
        self.pressureSolver = None
        self.last_poisson = None
        self.Rtol, self.Ctol, self.levelMaxVorticity = 1e9, 0.0, levelMax   # -Rtol / -Ctol / -levelMaxVorticity (15340-15342)
        self.obstacles = []            # ObstacleData list (geometry and motion come from the host)
        self.lambda_penal = 1e6        # sim.lambda (-lambda)
        self.bImplicitPenalization = True

    def __del__(self):
        try:
            lib().cup3d_sim_destroy(self.handle)
        except Exception:
            pass

    @property
    def nblocks(self):
        return self.grid.nblocks

    def upload(self, field, blocks):
        fid = FIELDS[field]
        nc = FIELD_NCOMP[fid]
        a = np.ascontiguousarray(blocks, dtype=np.float64)
        want = (self.nblocks, 8, 8, 8, 3) if nc == 3 else (self.nblocks, 8, 8, 8)
        if a.shape != want:
            raise ValueError(f"{field}: expected block array of shape {want}, got {a.shape}")
        check(lib().cup3d_sim_upload(self.handle, fid, a))
        if field == "chi":
            self._chi_uploaded = True   # chi set by hand (tests): PressureProjection lets the library decide which right-hand side to use

    def download(self, field):
        fid = FIELDS[field]
        nc = FIELD_NCOMP[fid]
        out = np.empty((self.nblocks, 8, 8, 8, 3) if nc == 3 else (self.nblocks, 8, 8, 8))
        check(lib().cup3d_sim_download(self.handle, fid, out))
        return out

    def fill(self, field, value):
        check(lib().cup3d_sim_fill(self.handle, FIELDS[field], float(value)))
        if field == "chi":
            self._chi_uploaded = float(value) != 0.0   # chi set (or cleared) by hand, as upload() notes

    def _block_ptrs(self, a):
        return (C.c_void_p * len(a))(*[a[i].ctypes.data for i in range(len(a))])

    def upload_block_list(self, field, slots, blocks):
        """Partial upload: blocks[i] ([8][8][8][(3)], the reference's block layout) -> block slot slots[i]."""
        fid = FIELDS[field]
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        a = np.ascontiguousarray(blocks, dtype=np.float64)
        if a.shape != ((len(sl), 8, 8, 8, 3) if FIELD_NCOMP[fid] == 3 else (len(sl), 8, 8, 8)):
            raise ValueError(f"{field}: block list of shape {a.shape} for {len(sl)} slots")
        check(lib().cup3d_sim_upload_block_list(self.handle, fid, len(sl), sl.ctypes.data_as(C.c_void_p), self._block_ptrs(a)))

    def download_block_list(self, field, slots):
        fid = FIELDS[field]
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        out = np.empty((len(sl), 8, 8, 8, 3) if FIELD_NCOMP[fid] == 3 else (len(sl), 8, 8, 8))
        check(lib().cup3d_sim_download_block_list(self.handle, fid, len(sl), sl.ctypes.data_as(C.c_void_p), self._block_ptrs(out)))
        return out

    def _like(self, **kw):
        """A SimulationData with this one's run parameters on another mesh (leaves=... or view=...), run state carried along."""
        new = SimulationData(bpdx=self.bpdx, bpdy=self.bpdy, bpdz=self.bpdz, levelMax=self.levelMax, levelStart=self.levelStart,
                             extent=self.maxextent, nu=self.nu, CFL=self.CFL, BC_x=self.BCx_flag, BC_y=self.BCy_flag, BC_z=self.BCz_flag,
                             uinf=self.uinf, uMax_forced=self.uMax_forced, poissonTol=self.PoissonErrorTol, poissonTolRel=self.PoissonErrorTolRel,
                             bMeanConstraint=self.bMeanConstraint, poissonSolver=self.poissonSolver, rampup=self.rampup,
                             blockSolver=self.blockSolver, implicitDiffusion=self.implicitDiffusion,
                             diffusionTol=self.DiffusionErrorTol, diffusionTolRel=self.DiffusionErrorTolRel, **kw)
        new.dt, new.dt_old, new.time, new.step, new.coefU = self.dt, self.dt_old, self.time, self.step, self.coefU.copy()
        new.uMax_measured = self.uMax_measured
        return new

    def adapted(self, states):
        """A new SimulationData on the mesh that MeshAdaptation::Adapt produces from (valid) `states`, with vel and pres
        moved over on the device (refine / compress / copy; chi, lhs and tmpV are adapted without data in the reference
        too, 15188-15190) and the run state carried along.  Simulation::adaptMesh's second half (15184-15193)."""
        lv, zs = self.grid.adapted_leaves(states)
        new = self._like(leaves=(lv, zs))
        for f in ("vel", "pres"):
            check(lib().cup3d_adapt_transfer(self.handle, new.handle, FIELDS[f]))
        return new

    def adapted_over_ranks(self, mesh, owner, states, rank, nranks):
        """The same when the leaves of `mesh` (the GLOBAL mesh object) are spread over ranks as `owner` says and this SimulationData
        lives on rank `rank`'s view of it: MeshAdaptation::Adapt + the LoadBalancer (main.cpp:4660-5022, 5086-5159).  Collective.
        Returns (new SimulationData on this rank's view of the adapted mesh, the adapted global mesh, its owners)."""
        lv, zs = mesh.adapted_leaves(states)
        new_mesh = Grid(mesh.bpd, mesh.levelMax, 0, mesh.maxextent, mesh.bc, leaves=(lv, zs))
        new_owner = mesh.adapted_owners(owner, states, nranks, new_mesh)
        new = self._like(view=new_mesh.rank_view(new_owner, rank, nranks))
        ow = np.ascontiguousarray(owner, dtype=np.int32)
        for f in ("vel", "pres"):
            check(lib().cup3d_adapt_migrate(mesh.handle, ow.ctypes.data_as(C.c_void_p), self.handle, new_mesh.handle,
                                            new_owner.ctypes.data_as(C.c_void_p), new.handle, FIELDS[f]))
        return new, new_mesh, new_owner

    def poisson_params(self):
        p = PoissonParams()
        lib().cup3d_poisson_default_params(C.byref(p))
        p.tol, p.tol_rel, p.mean_constraint = self.PoissonErrorTol, self.PoissonErrorTolRel, self.bMeanConstraint
        p.block_solver = self.blockSolver
        return p

    def checksum(self, field):
        """Wrapping 64-bit sum of the bit patterns of this rank's blocks of `field` (cup3d_sim_checksum): sums over ranks mod 2^64
        are independent of the partition."""
        out = C.c_ulonglong(0)
        check(lib().cup3d_sim_checksum(self.handle, FIELDS[field], C.byref(out)))
        return int(out.value)

    def device_bytes(self):
        return lib().cup3d_sim_device_bytes(self.handle)


class Operator:
    """class Operator, main.cpp:6678-6684."""

    def __init__(self, sim):
        self.sim = sim

    def __call__(self, dt):
        raise NotImplementedError


class AdvectionDiffusion(Operator):
    """AdvectionDiffusion::operator()(dt), main.cpp:9640-9728 (uses sim.dt like KernelAdvectDiffuse, 9465)."""

    def __call__(self, dt):
        s = self.sim
        s.dt = dt
        check(lib().cup3d_advect_diffuse(s.handle, dt, s.nu, s.uinf))


class AdvectionDiffusionImplicit(Operator):
    """AdvectionDiffusionImplicit::operator()(dt), main.cpp:10030-10119 (calls euler(sim.dt)): upwind advection + one Helmholtz
    solve per velocity component (DiffusionSolver, 6719-7147).  `last_diffusion` holds the three solver results."""

    def diffusion_params(self):
        s = self.sim
        p = PoissonParams()
        lib().cup3d_poisson_default_params(C.byref(p))
        p.tol, p.tol_rel = s.DiffusionErrorTol, s.DiffusionErrorTolRel
        return p

    def __call__(self, dt):
        s = self.sim
        s.dt = dt
        res = (PoissonResult * 3)()
        check(lib().cup3d_advect_diffuse_implicit(s.handle, dt, s.nu, s.uinf, C.byref(self.diffusion_params()), res))
        self.last_diffusion = [res[0], res[1], res[2]]
        return self.last_diffusion


class DiffusionSolver:
    """class DiffusionSolver (main.cpp:6719-7147) on the device: right-hand side in sim.lhs, initial guess and result in sim.pres;
    `mydirection` selects the velocity component whose boundary rule the ghosts follow, `dt` the Helmholtz coefficient."""

    def __init__(self, sim):
        self.sim, self.mydirection, self.dt = sim, 0, sim.dt

    def _params(self):
        p = PoissonParams()
        lib().cup3d_poisson_default_params(C.byref(p))
        p.tol, p.tol_rel = self.sim.DiffusionErrorTol, self.sim.DiffusionErrorTolRel
        return p

    def solve(self):
        r = PoissonResult()
        check(lib().cup3d_diffusion_solve(self.sim.handle, self.mydirection, self.dt, self.sim.nu, C.byref(self._params()), C.byref(r)))
        return r

    def lhs(self):
        """_lhs: sim.lhs <- A sim.pres."""
        check(lib().cup3d_diffusion_lhs(self.sim.handle, self.mydirection, self.dt, self.sim.nu))

    def preconditioner(self):
        """_preconditioner on sim.pres in place."""
        check(lib().cup3d_diffusion_preconditioner(self.sim.handle, self.dt, self.sim.nu))


class ComputeVorticity(Operator):
    """ComputeVorticity::operator()(dt), main.cpp:8726-8746: tmpV <- curl(vel) (the input of adaptMesh's tagging)."""

    def __call__(self, dt=0):
        check(lib().cup3d_compute_vorticity(self.sim.handle))


class GradChiOnTmp(Operator):
    """compute<ScalarLab>(GradChiOnTmp(sim), sim.chi), main.cpp:15182 / 8540-8600: the chi-driven edit of the vorticity in tmpV that
    precedes the tagging in adaptMesh (needs sim.Rtol, sim.Ctol, sim.levelMaxVorticity; chi resident)."""

    def __call__(self, dt=0, mesh=None, owner=None):
        """mesh / owner (the global mesh object and the rank of every leaf): the mesh is spread over ranks and this is a collective."""
        s = self.sim
        lmv = int(getattr(s, "levelMaxVorticity", s.levelMax))
        if mesh is None:
            check(lib().cup3d_grad_chi_on_tmp(s.handle, float(s.Rtol), float(s.Ctol), lmv))
        else:
            ow = np.ascontiguousarray(owner, dtype=np.int32)
            check(lib().cup3d_grad_chi_on_tmp_over_ranks(s.handle, mesh.handle, ow.ctypes.data_as(C.c_void_p), float(s.Rtol), float(s.Ctol), lmv))


class ObstacleData:
    """Host-side description of one obstacle for the device operators: the non-null ObstacleBlocks (block slots, chi, udef in the
    reference's layout, main.cpp:7256-7263) and the rigid motion (centre of mass, translation and angular velocity)."""

    def __init__(self, slots, chi, udef, cm, vel, omega):
        self.slots = np.ascontiguousarray(slots, dtype=np.int32)
        self.chi = np.ascontiguousarray(chi, dtype=np.float64).reshape(len(self.slots), 8, 8, 8)
        self.udef = np.ascontiguousarray(udef, dtype=np.float64).reshape(len(self.slots), 8, 8, 8, 3)
        self.cm, self.vel, self.omega = (np.array(v, dtype=np.float64) for v in (cm, vel, omega))
        self.force, self.torque = np.zeros(3), np.zeros(3)


def _obstacle_array(obstacles):
    from .capi import Obstacle
    arr = (Obstacle * max(1, len(obstacles)))()
    for o, a in zip(obstacles, arr):
        a.nblocks = len(o.slots)
        a.slots, a.chi, a.udef = o.slots.ctypes.data, o.chi.ctypes.data, o.udef.ctypes.data
        for d in range(3):
            a.cm[d], a.vel[d], a.omega[d] = o.cm[d], o.vel[d], o.omega[d]
    return arr


class Penalization(Operator):
    """Penalization::operator()(dt) without the collision model (main.cpp:14326-14341) for sim.obstacles (ObstacleData list):
    KernelPenalization on the resident vel / chi, then the obstacles' force and torque."""

    def __call__(self, dt):
        s = self.sim
        if not s.obstacles:
            return
        arr = _obstacle_array(s.obstacles)
        check(lib().cup3d_penalization(s.handle, dt, s.lambda_penal, 1 if s.bImplicitPenalization else 0, len(s.obstacles), arr))
        for o, a in zip(s.obstacles, arr):
            o.force, o.torque = np.array(a.force[:]), np.array(a.torque[:])


class ExternalForcing(Operator):
    """ExternalForcing::operator()(dt), main.cpp:10581-10596."""

    def __call__(self, dt):
        s = self.sim
        d = 1 if s.BCy_flag == "wall" else 2
        check(lib().cup3d_external_forcing(s.handle, s.uMax_forced, s.nu, s.extents[d], dt))


class ComputeLHS(Operator):
    """ComputeLHS::operator()(dt), main.cpp:9273-9327: lhs <- A(pres)."""

    def __call__(self, dt=0):
        check(lib().cup3d_compute_lhs(self.sim.handle, self.sim.bMeanConstraint))


class PoissonSolverBase:
    """class PoissonSolverBase, main.cpp:8921-8928."""

    def solve(self):
        raise NotImplementedError


class PoissonSolverHIP(PoissonSolverBase):
    """PoissonSolverAMR (main.cpp:9329-9435, solve 14363-14616) on the device: RHS in sim.lhs,
    initial guess and result in sim.pres."""

    def __init__(self, sim):
        self.sim = sim

    def solve(self):
        p, r = self.sim.poisson_params(), PoissonResult()
        check(lib().cup3d_poisson_solve(self.sim.handle, C.byref(p), C.byref(r)))
        self.sim.last_poisson = r
        return r

    def preconditioner(self):
        """_preconditioner on sim.pres in place (getZImplParallel, main.cpp:14704-14745)."""
        check(lib().cup3d_preconditioner(self.sim.handle, self.sim.blockSolver))


def makePoissonSolver(sim):
    """makePoissonSolver, main.cpp:14747-14758.  "iterative" is the reference's CPU solver and
    is not provided here; "cuda_iterative" is the slot the reference reserves for a GPU solver
    (14750) and is accepted as an alias of "hip_iterative"."""
    if sim.poissonSolver in ("hip_iterative", "cuda_iterative"):
        return PoissonSolverHIP(sim)
    if sim.poissonSolver == "iterative":
        raise RuntimeError('Poisson solver: "iterative" is the CPU reference; this library has no CPU path')
    raise ValueError(f'Poisson solver: "{sim.poissonSolver}" unrecognized!')


class PressureProjection(Operator):
    """PressureProjection::operator()(dt), main.cpp:15061-15160."""

    def __init__(self, sim):
        super().__init__(sim)
        self.pressureSolver = makePoissonSolver(sim)  # 15058-15059
        sim.pressureSolver = self.pressureSolver

    def __call__(self, dt):
        s = self.sim
        s.dt = dt
        if s.obstacles:  # tmpV = 0; kernelUpdateTmpV (15066-15082): chi must be resident (sim.upload("chi", ...))
            s.fill("tmpV", 0.0)
            check(lib().cup3d_update_tmpv(s.handle, len(s.obstacles), _obstacle_array(s.obstacles)))
        # every rank knows whether obstacles exist (the list is replicated, like the reference's obstacle_vector); chi placed by hand
        # without an ObstacleData list (tests) leaves the decision to the library
        by_hand = getattr(s, "_chi_uploaded", False) or getattr(s, "chi_resident", False)
        check(lib().cup3d_sim_set_obstacles(s.handle, 1 if s.obstacles else (-1 if by_hand else 0)))
        p, r = s.poisson_params(), PoissonResult()
        check(lib().cup3d_pressure_project(s.handle, dt, s.step, C.byref(p), C.byref(r)))
        s.last_poisson = r
        return r


class MeshAdaptation:
    """Data-movement half of class MeshAdaptation (main.cpp:5023-5583) for whole-mesh transitions between
    two uniform levels; the tagging decision is returned to the host, which owns the tree."""

    def __init__(self, Rtol, Ctol):
        self.tolerance_for_refinement, self.tolerance_for_compression = float(Rtol), float(Ctol)

    def Tag(self, sim, field="tmpV"):
        """TagLoadedBlock on every block (5566-5582) -> int8 states: 1 Refine, -1 Compress, 0 Leave."""
        st = np.zeros(sim.nblocks, dtype=np.int8)
        check(lib().cup3d_tag_blocks(sim.handle, FIELDS[field], self.tolerance_for_refinement, self.tolerance_for_compression, st))
        return st

    @staticmethod
    def refine(coarse, fine, field):
        """refine_1 + RefineBlocks (5227-5249, 5493-5565) of every block of `coarse` into `fine`."""
        check(lib().cup3d_prolong(coarse.handle, fine.handle, FIELDS[field]))

    @staticmethod
    def compress(fine, coarse, field):
        """compress (5272-5329) of every sibling octet of `fine` into `coarse`."""
        check(lib().cup3d_restrict(fine.handle, coarse.handle, FIELDS[field]))


def findMaxU(sim):
    """findMaxU, main.cpp:8603-8623."""
    out = C.c_double(0.0)
    check(lib().cup3d_max_u(sim.handle, sim.uinf, C.byref(out)))
    return out.value


class Simulation:
    """The time loop of struct Simulation restricted to the hot path (no obstacles, frozen mesh):
    calcMaxTimestep 15254-15305, advance 15306-15326, pipeline order of setupOperators 15229-15246."""

    def __init__(self, sim):
        self.sim = sim
        self.pipeline = [AdvectionDiffusionImplicit(sim) if sim.implicitDiffusion else AdvectionDiffusion(sim)]  # 15231-15234
        if sim.uMax_forced > 0:
            self.pipeline.append(ExternalForcing(sim))
        self.pipeline.append(PressureProjection(sim))

    def adaptMesh(self, Rtol, Ctol):
        """Simulation::adaptMesh (15179-15194) without obstacles: vorticity -> tags -> ValidStates -> Adapt.  Replaces
        self.sim (and the operators bound to it) when the mesh changes; returns the valid states."""
        s = self.sim
        ComputeVorticity(s)(0)
        if s.obstacles or getattr(s, "chi_resident", False):   # compute<ScalarLab>(GradChiOnTmp(sim), sim.chi), 15182
            s.Rtol, s.Ctol = Rtol, Ctol
            GradChiOnTmp(s)(0)
        st = s.grid.valid_states(MeshAdaptation(Rtol, Ctol).Tag(s, "tmpV"))
        if (st != 0).any():
            self.__init__(s.adapted(st))
        return st

    def adaptMeshOverRanks(self, mesh, owner, rank, nranks, Rtol, Ctol, allgather):
        """Simulation::adaptMesh on a mesh spread over ranks.  mesh / owner: the global mesh object and the rank of every leaf (every
        rank holds them, like the reference's Octree); allgather(int8 array of this rank's tags) -> list of every rank's tags (the
        host's collective: MPI in the shim, torch.distributed or a thread barrier in the tests).  Returns (states, mesh, owner) -- the
        new ones when the mesh changed; self.sim is then this rank's SimulationData on the adapted mesh."""
        s = self.sim
        ComputeVorticity(s)(0)
        if s.obstacles or getattr(s, "chi_resident", False):   # compute<ScalarLab>(GradChiOnTmp(sim), sim.chi), 15182
            s.Rtol, s.Ctol = Rtol, Ctol
            GradChiOnTmp(s)(0, mesh=mesh, owner=owner)
        parts = allgather(MeshAdaptation(Rtol, Ctol).Tag(s, "tmpV"))
        tags = np.zeros(mesh.nblocks, dtype=np.int8)
        ow = np.asarray(owner)
        for r in range(nranks):
            tags[ow == r] = parts[r]          # a rank's blocks appear in the global order inside its view
        st = mesh.valid_states(tags)
        if not (st != 0).any():
            return st, mesh, owner
        new, new_mesh, new_owner = s.adapted_over_ranks(mesh, owner, st, rank, nranks)
        self.__init__(new)
        return st, new_mesh, new_owner

    def calcMaxTimestep(self):
        s = self.sim
        s.dt_old = s.dt
        s.uMax_measured = findMaxU(s)
        s.dt = lib().cup3d_calc_max_timestep2(s.hmin, s.uMax_measured, s.nu, s.CFL, s.step, s.rampup, s.dt_old, s.coefU, int(s.implicitDiffusion))
        if s.dt <= 0:
            raise Cup3dError(f"dt <= 0. CFL={s.CFL}, hMin={s.hmin}, sim.uMax_measured={s.uMax_measured}")
        return s.dt

    def advance(self, dt):
        for op in self.pipeline:
            op(dt)
        self.sim.step += 1
        self.sim.time += dt
