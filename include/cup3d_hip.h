/* cup3d_hip.h — C ABI of the MI355X-native (HIP, gfx950) implementation of CUP3D's
 * per-block stencil + pressure-Poisson hot path.
 *
 * Every entry point names the reference interface it replaces (slitvinov/CUP3D,
 * main.cpp:LINE).  Plain pointers and sizes only; all functions return 0 on success
 * and a negative CUP3D_E* code on failure (the reference has no return codes: it
 * abort()s / MPI_Abort()s, main.cpp:1098, 15265 — the C++ shim in
 * cup3d_amd/host/cup3d_hip_operators.h turns a non-zero status into the same).
 * cup3d_last_error() returns a human readable message for the calling thread.
 *
 * Host block memory at this boundary is the reference's own:
 *   ScalarBlock = double[8][8][8]      (z,y,x; 4096 B,  main.cpp:5882-5919, 6586)
 *   VectorBlock = double[8][8][8][3]   (AoS u,v,w; 12288 B, main.cpp:6590)
 * in m_vInfo order (sorted by blockID_2, main.cpp:943-964).  On the device every
 * field is a slab [block][component][z][y][x] (SoA per block).
 */
#ifndef CUP3D_HIP_H
#define CUP3D_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: these entry points -- and nothing else -- are exported */
#ifndef CUP3D_API
#define CUP3D_API __attribute__((visibility("default")))
#endif

#define CUP3D_OK 0
#define CUP3D_EINVAL (-1)   /* bad argument */
#define CUP3D_EDEVICE (-2)  /* HIP error / no usable GPU */
#define CUP3D_ENOMEM (-3)
#define CUP3D_ECOMM (-4)    /* RCCL error */
#define CUP3D_ESTATE (-5)   /* call out of order */

CUP3D_API const char *cup3d_last_error(void);
CUP3D_API const char *cup3d_version(void);

/* enum BCflag { freespace, periodic, wall }  (main.cpp:6081) */
enum { CUP3D_BC_FREESPACE = 0, CUP3D_BC_PERIODIC = 1, CUP3D_BC_WALL = 2 };
/* fields of SimulationData (main.cpp:6603-6607) */
enum { CUP3D_FIELD_CHI = 0, CUP3D_FIELD_PRES = 1, CUP3D_FIELD_VEL = 2, CUP3D_FIELD_TMPV = 3, CUP3D_FIELD_LHS = 4 };

/* ---------------------------------------------------------------------------
 * Indexing (host, integer, bit-exact contract).
 * Replaces class SpaceFillingCurve (main.cpp:95-319) and Info::setup (384-420).
 * ------------------------------------------------------------------------- */
typedef struct cup3d_sfc cup3d_sfc_t;
CUP3D_API int cup3d_sfc_create(int bpdx, int bpdy, int bpdz, int level_max, cup3d_sfc_t **out);   /* ctor 196-236 */
CUP3D_API void cup3d_sfc_destroy(cup3d_sfc_t *);
CUP3D_API long long cup3d_sfc_forward(const cup3d_sfc_t *, int level, int i, int j, int k);       /* 237-255 */
CUP3D_API void cup3d_sfc_inverse(const cup3d_sfc_t *, long long Z, int level, int ijk[3]);        /* 256-276 */
CUP3D_API long long cup3d_sfc_encode(const cup3d_sfc_t *, int level, const int index[3]);         /* 287-318 (blockID_2) */
/* Info::setup neighbour / child / parent ids: nei[27] as Znei[i+1][j+1][k+1] */
CUP3D_API void cup3d_sfc_info(const cup3d_sfc_t *, int level, const int index[3], long long nei[27], long long child[8],
                    long long *parent);

/* ---------------------------------------------------------------------------
 * Block topology of one rank (host).  Replaces the parts of Grid / GridMPI the hot
 * path needs: block ownership by contiguous Hilbert ranges (GridMPI ctor,
 * main.cpp:2959-2986), m_vInfo ordering (FillPos 943-964), neighbour lookup
 * (Info::Znei + Tree().rank(), 384-420, 840-855) and the inner/halo block split of
 * SynchronizerMPI_AMR::_Setup (1979-2286) reduced to what face-only stencils need.
 * ------------------------------------------------------------------------- */
typedef struct cup3d_grid cup3d_grid_t;
/* uniform grid at `level` (< level_max) of a bpd[0] x bpd[1] x bpd[2] level-0 block box;
 * maxextent = SimulationData::maxextent (15401); bc[d] per axis; this rank owns the
 * Z range of GridMPI's constructor. */
CUP3D_API int cup3d_grid_create_uniform(const int bpd[3], int level_max, int level, double maxextent, const int bc[3],
                              int rank, int nranks, cup3d_grid_t **out);
/* multi-level (AMR) mesh on one rank from its leaf blocks (level, Z) in any order: the block set Grid::m_vInfo holds
 * after MeshAdaptation::Adapt (main.cpp:5086-5159).  Blocks are ordered by blockID_2; the octree states
 * (Exists / CheckCoarser / CheckFiner, 321-330) follow from the leaf set; the mesh must be 2:1 balanced. */
CUP3D_API int cup3d_grid_create_mesh(const int bpd[3], int level_max, double maxextent, const int bc[3], long nleaves,
                           const int32_t *levels, const int64_t *Zs, cup3d_grid_t **out);
/* interface faces of a multi-level mesh (faces whose same-level neighbour does not exist, FluxCorrection::prepare
 * 676-711): faces2[e] = {6*slot + face, kind} with kind 0 = neighbour coarser, 1 = neighbour finer; fine4[e] = interface-face
 * indices of the four finer blocks' opposite faces (FillCase 601-661), -1 for kind 0; nbr27[slot][27] = neighbour states
 * (same-level slot, CUP3D_NBR_COARSER + slot, CUP3D_NBR_SKIPPED, CUP3D_NBR_FINER).  Any output may be NULL. */
#define CUP3D_NBR_COARSER 0x20000000
#define CUP3D_NBR_SKIPPED (-1)
#define CUP3D_NBR_FINER (-3)
CUP3D_API long cup3d_grid_ninterface_faces(const cup3d_grid_t *);
CUP3D_API int cup3d_grid_interface(const cup3d_grid_t *, int32_t *faces2, int32_t *fine4, int32_t *nbr27);
/* Mesh adaptation decisions (integer contract; one rank).  cup3d_grid_valid_states = MeshAdaptation::ValidStates
 * (main.cpp:5330-5492): in/out states[nblocks] in {-1 Compress, 0 Leave, 1 Refine} (e.g. from cup3d_tag_blocks):
 * refinement propagates to coarser neighbours to keep the mesh 2:1 balanced, unbalanced or partial compressions are
 * dropped.  cup3d_grid_adapted = the mesh MeshAdaptation::Adapt (5086-5159) produces from valid states (a new grid object,
 * always of the multi-level kind); cup3d_adapt_transfer (below) moves the field data. */
CUP3D_API int cup3d_grid_valid_states(const cup3d_grid_t *, signed char *states);
CUP3D_API int cup3d_grid_adapted(const cup3d_grid_t *, const signed char *states, cup3d_grid_t **out);
/* Ownership of the adapted mesh's leaves on several ranks = what MeshAdaptation::Adapt + LoadBalancer (main.cpp:4660-5022, 5086-5159)
 * leave behind: children on the refined parent's rank, a compressed octet's parent on the rank of its base block (even indices;
 * PrepareCompression 4729-4804), then Balance_Diffusion (4805-4905: (my - neighbour)/4 blocks to each neighbour) or, when
 * max/min > 1.01 or a rank is empty, Balance_Global (4906-5021: even cut of the rank-major order).  `mesh`: all leaves of all ranks
 * (cup3d_grid_create_mesh), owner[nblocks] their ranks, states the valid states, adapted = cup3d_grid_adapted(mesh, states);
 * new_owner[nblocks of adapted].  cup3d_adapt_migrate (below) moves the field data accordingly. */
CUP3D_API int cup3d_grid_adapted_owners(const cup3d_grid_t *mesh, const int32_t *owner, const signed char *states, int nranks,
                              const cup3d_grid_t *adapted, int32_t *new_owner);
/* One rank's view of a multi-level mesh whose leaves are spread over several ranks (what SynchronizerMPI_AMR::_Setup 1979-2286 and
 * FluxCorrectionMPI::prepare 2680-2824 derive from the shared octree): block slots [0, nlocal) are the rank's own leaves, then GHOST
 * leaves -- every remote leaf the neighbour tables of a local block refer to -- ordered by (owner, global order); interface faces:
 * the local blocks' first, then the fine faces of ghost blocks whose fluxes / cells the local coarse-side faces need.  The tables
 * (cup3d_grid_neighbours, cup3d_grid_interface) are the global ones renumbered, so a kernel reads through them what it would read on
 * one rank once the two exchanges of cup3d_grid_view_plan have run; cup3d_sim_create on a view allocates the ghost slots and the
 * operators run those exchanges over RCCL (ghost blocks before a stencil kernel -- of each only the sub-box the stencil's consumers read,
 * cup3d_grid_view_boxes; whole blocks for the tensorial tiles of mesh adaptation --, face fluxes after a flux-corrected one). */
CUP3D_API int cup3d_grid_rank_view(const cup3d_grid_t *mesh, const int32_t *owner, int rank, int nranks, cup3d_grid_t **view);
/* out: local blocks, ghost blocks, local interface faces, ghost faces, blocks sent per exchange, face-flux arrays sent per exchange */
CUP3D_API int cup3d_grid_view_sizes(const cup3d_grid_t *view, long out[6]);
/* any pointer may be NULL.  global_slot[nlocal+nghost], global_face[nfaces]: position in the global mesh; send_blocks[]: local slots in
 * peer-major order (each peer's run in that peer's ghost order), counts per rank; the same for the face-flux arrays */
CUP3D_API int cup3d_grid_view_plan(const cup3d_grid_t *view, int32_t *global_slot, int32_t *global_face, int32_t *send_blocks, long *send_block_count,
                         long *recv_block_count, int32_t *send_flux_faces, long *send_flux_count, long *recv_flux_count);
/* The sub-box form of the ghost-block exchange (what SynchronizerMPI_AMR ships: face sub-boxes and coarse shadow cells, main.cpp:1832-1966,
 * 2423-2544): of every ghost block only the bounding box of the cells the rank's star-stencil consumers read travels -- the w layers behind
 * a shared face, the 2w layers a restriction averages, the cells of the coarse shadow patch of an interpolation.  width_class 0: stencil
 * width 1, 1: width 3.  ghost_box[nghost][6], send_box[blocks sent][6] = lo x, y, z, hi x, y, z (hi exclusive; all zero: nothing of the
 * block is read); send_cells[nranks] / recv_cells[nranks]: cells per component to / from every rank.  Any pointer may be NULL. */
CUP3D_API int cup3d_grid_view_boxes(const cup3d_grid_t *view, int width_class, unsigned char *ghost_box, unsigned char *send_box, long *send_cells, long *recv_cells);
CUP3D_API void cup3d_grid_destroy(cup3d_grid_t *);
CUP3D_API long cup3d_grid_nblocks(const cup3d_grid_t *);        /* local blocks = m_vInfo.size() */
CUP3D_API long cup3d_grid_nblocks_global(const cup3d_grid_t *);
CUP3D_API long cup3d_grid_nhalo_faces(const cup3d_grid_t *);    /* remote face slabs this rank receives */
CUP3D_API long cup3d_grid_ninner(const cup3d_grid_t *);         /* blocks with no remote neighbour (Synchronizer inner_blocks) */
/* per local block, m_vInfo order: tab6 = level, Z, index[3], blockID_2 ; geom4 = h, origin[3] (Info 331-346) */
CUP3D_API int cup3d_grid_tables(const cup3d_grid_t *, long long *tab6, double *geom4);
/* face neighbour table nbr[nblocks][6] (faces x-,x+,y-,y+,z-,z+): >=0 local slot,
 * CUP3D_NBR_HALO + e = remote slab e, CUP3D_NBR_BC - bc = domain face with that BC */
#define CUP3D_NBR_HALO 0x40000000
#define CUP3D_NBR_BC (-1)
CUP3D_API int cup3d_grid_neighbours(const cup3d_grid_t *, int32_t *nbr);
/* halo-exchange plan (what SynchronizerMPI_AMR::_Setup computes, 1979-2286): for peer
 * p in [0,nranks): number of face slabs sent to / received from p; send_faces[s] =
 * local_slot*6+face of the s-th slab sent (peer-major, then (Z,face) of the sender's
 * block); received slabs are numbered in the same peer-major order. */
CUP3D_API int cup3d_grid_halo_plan(const cup3d_grid_t *, long *send_count, long *recv_count, int32_t *send_faces);
CUP3D_API long cup3d_grid_nsend_faces(const cup3d_grid_t *);

/* Simulation::calcMaxTimestep (main.cpp:15254-15305), explicit diffusion, CFL > 0;
 * updates coefU when step > step_2nd_start. */
CUP3D_API double cup3d_calc_max_timestep(double hmin, double umax, double nu, double cfl, int step, int rampup, double dt_old,
                               double coefU[3]);
/* the same with sim.implicitDiffusion: the diffusive limit becomes 0.1 once step > 10 (15269-15273) */
CUP3D_API double cup3d_calc_max_timestep2(double hmin, double umax, double nu, double cfl, int step, int rampup, double dt_old,
                                double coefU[3], int implicit_diffusion);

/* ---------------------------------------------------------------------------
 * Device side.
 * ------------------------------------------------------------------------- */
CUP3D_API int cup3d_device_count(int *n);
CUP3D_API int cup3d_device_init(int device);            /* hipSetDevice; fails loudly unless the device is gfx950 */
CUP3D_API int cup3d_set_stream(void *hip_stream);       /* compute stream for all launches (NULL = default stream) */
CUP3D_API int cup3d_device_synchronize(void);

/* RCCL communicator owned by the library (replaces sim.comm's role on the hot path:
 * halo Isend/Irecv 2370/2402 and the Allreduce/Iallreduce sites 8620, 9295, 14442,
 * 14486, 14546, 14584, 15123).  id is the 128-byte ncclUniqueId from rank 0. */
/* COLLECTIVES AND ERRORS.  On a grid that spans several ranks every operator entry point is a collective: all ranks call it, in the
 * same order (the library issues its RCCL calls from one stream per rank in that order).  A non-zero status on ONE rank -- a bad
 * argument, a HIP error -- leaves the others inside the collective, exactly as a failing rank does under the reference's MPI: treat any
 * non-zero status as fatal for the job, as the reference does (MPI_Abort at 8444, 15265, 15289; the C++ shim's CUP3D_HIP_CALL and
 * torch.distributed.run's process group do that for their hosts).  There is no rank-local error recovery. */
/* ENVIRONMENT (read once per process, the same on every rank).  CUP3D_RCCL_LIBRARY=<path>: the RCCL to dlopen instead of the one the
 * process has loaded / the loader finds.  CUP3D_EARLY_ALLREDUCE=1: the two all-reduces of a BiCGSTAB iteration (main.cpp:14486, 14546) start
 * when the loop kernel's last block has left its vector phase instead of when the kernel has ended -- they then run under its block solves,
 * as MPI_Iallreduce runs under the preconditioner in the reference; uniform grids, block CG, one process per device; the dot products are
 * added in another (deterministic) order than by default.  Off unless set. */
CUP3D_API int cup3d_comm_unique_id(void *id128);
CUP3D_API int cup3d_comm_init(int rank, int nranks, const void *id128);
CUP3D_API int cup3d_comm_finalize(void);

typedef struct cup3d_sim cup3d_sim_t; /* device mirror of SimulationData's five grids + solver vectors */
CUP3D_API int cup3d_sim_create(const cup3d_grid_t *, cup3d_sim_t **out);
CUP3D_API void cup3d_sim_destroy(cup3d_sim_t *);
CUP3D_API size_t cup3d_sim_device_bytes(const cup3d_sim_t *);
/* host <-> device; host side is the reference's block memory: either one pointer per
 * block (Info::block, main.cpp:343, 877-884) or one contiguous array [nb][8][8][8][nc] */
CUP3D_API int cup3d_sim_upload_blocks(cup3d_sim_t *, int field, const void *const *block_ptrs);
CUP3D_API int cup3d_sim_download_blocks(cup3d_sim_t *, int field, void *const *block_ptrs);
/* the same for a subset: only the n listed block slots move (slots[i] <-> block_ptrs[i]).  Lets the host-side obstacle operators
 * (UpdateObstacles / Penalization, which touch only the blocks an obstacle covers, 13841-13912) run between two device operators
 * without a full-field round trip */
CUP3D_API int cup3d_sim_upload_block_list(cup3d_sim_t *, int field, long n, const int32_t *slots, const void *const *block_ptrs);
CUP3D_API int cup3d_sim_download_block_list(cup3d_sim_t *, int field, long n, const int32_t *slots, void *const *block_ptrs);
CUP3D_API int cup3d_sim_upload(cup3d_sim_t *, int field, const double *blocks);
CUP3D_API int cup3d_sim_download(cup3d_sim_t *, int field, double *blocks);
CUP3D_API int cup3d_sim_fill(cup3d_sim_t *, int field, double value);
/* raw device pointer of a field slab [nb][nc][512] (for zero-copy hosts).  The VEL pointer is valid until the next operator that
 * advects (cup3d_advect_diffuse, cup3d_advect_diffuse_implicit, cup3d_advect_implicit): those write the new velocity into a second
 * buffer and swap the two, so re-query it after every such call; the other fields never move. */
CUP3D_API int cup3d_sim_device_ptr(cup3d_sim_t *, int field, void **ptr);
/* A host that WRITES a field through that pointer must say so afterwards: the library tracks two facts the operators branch on --
 * "chi is non-zero" (KernelPressureRHS then reads chi and udef, main.cpp:14858-14871) and "tmpV holds the udef of the next
 * projection" (otherwise cup3d_pressure_project clears tmpV as the reference does at 15076-15078) -- and upload / fill /
 * cup3d_update_tmpv set them, stores through a raw pointer cannot.  CHI and TMPV are the fields that matter; others are accepted. */
CUP3D_API int cup3d_sim_mark_written(cup3d_sim_t *, int field);
/* Does ANY rank hold an obstacle?  The reference's obstacle_vector is replicated on every rank (sim.obstacle_vector->nObstacles(),
 * main.cpp:15081), so the host knows without communicating; the same value on every rank (it decides whether the collective udef
 * exchange of the pressure right-hand side runs).  1: chi / udef path (KernelPressureRHS 14858-14871); 0: obstacle-free path, no udef
 * exchange, tmpV not cleared; -1 (default): not told -- one rank decides by "chi was written", several ranks always take the chi path.
 * A chi written (upload / fill / mark_written) is never dropped silently: on one rank it takes the chi path even after 0; on several
 * ranks cup3d_pressure_project refuses the combination on EVERY rank: CUP3D_ESTATE on all of them together, agreed inside the
 * mean-pressure all-reduce the operator performs anyway (15123) -- i.e. AFTER the Poisson solve.  The simulation state is then
 * UNDEFINED (pres, the previous-pressure copy, lhs and tmpV have been overwritten; vel has not been projected): the host must end the
 * run or re-upload its fields, exactly as after the MPI_Abort the reference answers an inconsistent run with (15265, 15289). */
CUP3D_API int cup3d_sim_set_obstacles(cup3d_sim_t *, int any_rank_has_obstacles);
/* Wrapping 64-bit sum of the bit patterns of every FP64 value of the rank's own blocks of `field` (ghost blocks of a rank view
 * excluded).  Integer addition commutes, so the sum of the ranks' values is independent of the partition: bench.py all-gathers it
 * after the first AdvectionDiffusion and compares it with the constant the CPU oracle produced for the same step (the stencil
 * operators are bit-exact under any sharding; the reference's own partition is main.cpp:2970-2986). */
CUP3D_API int cup3d_sim_checksum(cup3d_sim_t *, int field, unsigned long long *sum);

/* AdvectionDiffusion::operator()(dt) (main.cpp:9640-9728): low-storage RK3 of
 * KernelAdvectDiffuse (9461-9549) on vel, scratch tmpV; fused into 3 launches. */
CUP3D_API int cup3d_advect_diffuse(cup3d_sim_t *, double dt, double nu, const double uinf[3]);
/* findMaxU (main.cpp:8603-8623) incl. the MAX all-reduce */
CUP3D_API int cup3d_max_u(cup3d_sim_t *, const double uinf[3], double *umax);
/* ExternalForcing::operator() (main.cpp:10581-10596): vel.u[0] += 8*uMax*nu/H/H*dt */
CUP3D_API int cup3d_external_forcing(cup3d_sim_t *, double umax_forced, double nu, double H, double dt);

typedef struct {
  double tol;          /* sim.PoissonErrorTol     (-poissonTol, 1e-6)    */
  double tol_rel;      /* sim.PoissonErrorTolRel  (-poissonTolRel, 1e-4) */
  int mean_constraint; /* sim.bMeanConstraint     (-bMeanConstraint, 1)  */
  int max_iter;        /* 1000 (main.cpp:14449) */
  int max_restarts;    /* 100  (main.cpp:14374) */
  int block_solver;    /* how the block preconditioner M^-1 (getZImplParallel, 14704-14745) is evaluated:
                          0 = the reference's block-local CG, iteration for iteration, as the device evaluates it fastest: a*b+c
                              contracted to FMA (wave-wide sums by DPP reductions, IEEE divisions).  Differs from the reference's z
                              by less than the CG's own 1e-7 truncation (tests: measured against block_solver 2 and the reference);
                          1 = direct block solve by fast diagonalisation (same operator, exact to rounding);
                          2 = the block CG in the reference's association: no FMA contraction (only the ORDER of the 512-term sums
                              differs from the CPU); slower, for parity checks;
                          3, 4 = A/B timing variants (3: alias of 0; 4: two blocks per wavefront), libcup3d_hip_testing.so only;
                          5 = NOT the reference's preconditioner: one geometric-multigrid V(2,2)-cycle (red-black Gauss-Seidel in LDS,
                              summed-residual restriction, piecewise-constant prolongation) on the hierarchy of uniform block grids --
                              same operator, same stopping rule, same converged pressure to solver tolerance, O(10) instead of O(150)
                              iterations; uniform grids (over several ranks ONE cycle coupled over the ranks: every level is partitioned
                              like the solver's grid and its iterate crosses ranks as face slabs before each launch that reads
                              ghosts) and multi-level meshes (the octree's levels; over ranks: rank views, ghost nodes exchanged before
                              every launch that reads them); bench.py: `alt_multigrid` only */
} cup3d_poisson_params;
typedef struct {
  int iterations; /* BiCGSTAB iterations performed (= 7-double reductions, main.cpp:14546) */
  int restarts;
  double norm0, norm; /* ||r0||, last ||r|| */
  int used_xopt;
} cup3d_poisson_result;
CUP3D_API void cup3d_poisson_default_params(cup3d_poisson_params *);

/* ComputeLHS::operator() (main.cpp:9273-9327): lhs = h*(sum6 - 6p) of pres + mean constraint */
CUP3D_API int cup3d_compute_lhs(cup3d_sim_t *, int mean_constraint);
/* poisson_kernels::getZImplParallel (main.cpp:14704-14745): block preconditioner on pres, in place;
 * block_solver as in cup3d_poisson_params */
CUP3D_API int cup3d_preconditioner(cup3d_sim_t *, int block_solver);
/* PoissonSolverBase::solve() (main.cpp:8921-8928; PoissonSolverAMR::solve 14363-14616):
 * RHS in lhs, initial guess and result in pres; lhs is clobbered. */
CUP3D_API int cup3d_poisson_solve(cup3d_sim_t *, const cup3d_poisson_params *, cup3d_poisson_result *);
/* KernelPressureRHS via compute<> (main.cpp:15083-15085): lhs from vel, tmpV(=udef), chi */
CUP3D_API int cup3d_pressure_rhs(cup3d_sim_t *, double dt);
/* KernelDivPressure (main.cpp:15088): tmpV.u[0] = h * lap(pres) */
CUP3D_API int cup3d_div_pressure(cup3d_sim_t *);
/* KernelGradP (main.cpp:15146): tmpV = -0.5*dt*h^2 * central grad(pres) */
CUP3D_API int cup3d_grad_p(cup3d_sim_t *, double dt);
/* PressureProjection::operator()(dt) (main.cpp:15061-15160), obstacle-free or with
 * chi/udef already resident: tmpV must hold udef (upload / fill / cup3d_update_tmpv after the previous projection); if it was not
 * touched since, it is zeroed as at 15076-15078. */
CUP3D_API int cup3d_pressure_project(cup3d_sim_t *, double dt, int step, const cup3d_poisson_params *, cup3d_poisson_result *);

/* ---------------------------------------------------------------------------
 * Mesh-adaptation block operators (data movement of MeshAdaptation, main.cpp:5023-5583) for
 * whole-mesh transitions between two uniform levels l (coarse) and l+1 (fine) of the same box.
 * The integer decisions (which blocks to refine, 2:1 balancing, load balancing) stay on the host.
 * ------------------------------------------------------------------------- */
/* "restrict": MeshAdaptation::compress (5272-5329): every sibling octet of `fine` -> its parent block of `coarse` */
CUP3D_API int cup3d_restrict(cup3d_sim_t *fine, cup3d_sim_t *coarse, int field);
/* "prolong": refine_1 + RefineBlocks (5227-5249, 5493-5565): every block of `coarse` -> its eight children in `fine` */
CUP3D_API int cup3d_prolong(cup3d_sim_t *coarse, cup3d_sim_t *fine, int field);
/* TagLoadedBlock (5566-5582) + level clamps (5207-5211): states[nblocks] in {-1 Compress, 0 Leave, 1 Refine} (enum State, 320);
 * any mesh (uniform or multi-level).  With ComputeVorticity this is the device half of Simulation::adaptMesh's decision input
 * (15180-15183, obstacle-free: GradChiOnTmp only reads chi); ValidStates' 2:1 balancing of the tags stays on the host. */
CUP3D_API int cup3d_tag_blocks(cup3d_sim_t *, int field, double rtol, double ctol, signed char *states);
/* field `field` of `src` onto the mesh of `dst` (MeshAdaptation::Adapt for one grid, basic = false): blocks present in both
 * are copied, children of a refined block come from refine_1 + RefineBlocks (5227-5249, 5493-5565: 2nd-order Taylor
 * expansion from the parent's tensorial [-1,2) tile on the OLD mesh, coarse/fine ghosts included), the parent of a
 * compressed octet from compress (5272-5329).  Every block of dst must be a block, a child or the parent of blocks of src. */
CUP3D_API int cup3d_adapt_transfer(cup3d_sim_t *src, cup3d_sim_t *dst, int field);
/* The same over ranks: MeshAdaptation::Adapt plus the block traffic of the LoadBalancer (PrepareCompression 4729-4804, Balance_Diffusion
 * 4805-4905, Balance_Global 4906-5021) for one field.  old_mesh / new_mesh: the GLOBAL mesh objects before and after (cup3d_grid_adapted),
 * old_owner / new_owner: the rank of every leaf (new_owner from cup3d_grid_adapted_owners); src / dst: this rank's sims on its views of
 * the two (cup3d_grid_rank_view).  Collective over the ranks of the communicator.  Each block of the new mesh is built by the rank that
 * owns its origin in the old mesh (the leaf itself, the refined parent, or the base block of a compressed octet) and sent straight to its
 * new owner; the result equals the one-rank cup3d_adapt_transfer bit for bit, block by block. */
CUP3D_API int cup3d_adapt_migrate(const cup3d_grid_t *old_mesh, const int32_t *old_owner, cup3d_sim_t *src, const cup3d_grid_t *new_mesh,
                        const int32_t *new_owner, cup3d_sim_t *dst, int field);
/* Obstacle operators (the obstacles themselves -- geometry, chi/udef rasterisation, rigid-body integration -- stay on the host).
 * One cup3d_obstacle = the ObstacleBlocks of one Obstacle on this rank in the reference's own layout (struct ObstacleBlock,
 * main.cpp:7256-7263) plus its rigid motion. */
typedef struct {
  long nblocks;          /* number of non-null entries of Obstacle::obstacleBlocks */
  const int32_t *slots;  /* [nblocks] their block slots (= Info::blockID) */
  const double *chi;     /* [nblocks][8][8][8]     ObstacleBlock::chi */
  const double *udef;    /* [nblocks][8][8][8][3]  ObstacleBlock::udef */
  double cm[3], vel[3], omega[3]; /* getCenterOfMass(), getTranslationVelocity(), getAngularVelocity() */
  double force[3], torque[3];     /* out of cup3d_penalization: Obstacle::force / torque (13932-13937) */
} cup3d_obstacle;
/* Penalization::operator() without the collision model (14330-14340): KernelPenalization (13841-13912) on the resident vel with
 * the resident chi, obstacle after obstacle, then kernelFinalizePenalizationForce (13913-13938) */
CUP3D_API int cup3d_penalization(cup3d_sim_t *, double dt, double lambda, int implicit_penalization, int nobstacles, cup3d_obstacle *obstacles);
/* kernelUpdateTmpV (14948-14979): tmpV += udef where chi <= the obstacle's chi; call after clearing tmpV and before
 * cup3d_pressure_rhs / cup3d_pressure_project (15066-15085) */
CUP3D_API int cup3d_update_tmpv(cup3d_sim_t *, int nobstacles, const cup3d_obstacle *obstacles);
/* Implicit diffusion: AdvectionDiffusionImplicit (main.cpp:7148-7157, 10030-10119), selected by -implicitDiffusion (15231-15232).
 * One call = euler(dt): KernelAdvect, the explicit-diffusion guess, KernelDiffusionRHS and one DiffusionSolver::solve per velocity
 * component.  `params`: tol / tol_rel = sim.DiffusionErrorTol / DiffusionErrorTolRel (15369-15370), max_iter; mean_constraint,
 * max_restarts and block_solver are ignored (DiffusionSolver has no mean constraint, no restart cap and its own block CG).
 * results[3] (may be NULL): per-component solver statistics.  pres is used as scratch and restored; tmpV and lhs are clobbered.
 * KernelAdvect (9849-10029) updates vel IN PLACE in the reference while other blocks still build their ghosted tiles from it, so
 * the reference's own result depends on block order and thread timing; this implementation reads every tile from the velocity
 * on entry (the order-independent reading).  Everything else is the reference's arithmetic. */
CUP3D_API int cup3d_advect_diffuse_implicit(cup3d_sim_t *, double dt, double nu, const double uinf[3], const cup3d_poisson_params *params,
                                  cup3d_poisson_result results[3]);
/* its parts, for tests and for callers that interleave their own operators:
 * compute<VectorLab>(KernelAdvect(sim, dt), vel, tmpV) (10038): tmpV <- facD*lap(vel), vel <- vel + facA*(u.grad)u/h^3 */
CUP3D_API int cup3d_advect_implicit(cup3d_sim_t *, double dt, double nu, const double uinf[3]);
/* compute<VectorLab>(KernelDiffusionRHS(sim), vel, tmpV) (10057, 9729-9848): tmpV <- h*lap(vel) */
CUP3D_API int cup3d_diffusion_rhs(cup3d_sim_t *);
/* DiffusionSolver::_lhs (6836-6875) with mydirection = direction: lhs <- h*(sum6 - 6 pres) - h^3/(dt nu)*pres on the
 * BlockLabBC<ScalarGrid, .., direction> tile (wall: ghost = -face cell; freespace: negated behind the faces normal to direction) */
CUP3D_API int cup3d_diffusion_lhs(cup3d_sim_t *, int direction, double dt, double nu);
/* diffusion_kernels::getZImplParallel (10534-10579): pres <- block-local CG solve with centre coefficient -6 - h^2/nu/dt, in place */
CUP3D_API int cup3d_diffusion_preconditioner(cup3d_sim_t *, double dt, double nu);
/* DiffusionSolver::solve (6896-7146): right-hand side in lhs (clobbered), initial guess and result in pres */
CUP3D_API int cup3d_diffusion_solve(cup3d_sim_t *, int direction, double dt, double nu, const cup3d_poisson_params *params, cup3d_poisson_result *result);
/* ComputeVorticity::operator() (8726-8746, KernelVorticity 8624-8645): tmpV <- curl(vel); any mesh */
CUP3D_API int cup3d_compute_vorticity(cup3d_sim_t *);

/* compute<ScalarLab>(GradChiOnTmp(sim), sim.chi) (main.cpp:15182, 8540-8600): tmpV (the vorticity left by cup3d_compute_vorticity) edited
 * from the resident chi on its tensorial [-2,3) tile -- blocks with an obstacle surface within reach are flagged (1e10), cells deep
 * inside a body cleared, vorticity capped on level levelMaxVorticity - 1.  With cup3d_compute_vorticity before and cup3d_tag_blocks
 * after, this is the decision input of Simulation::adaptMesh (15180-15183) for runs with obstacles.  One rank. */
CUP3D_API int cup3d_grad_chi_on_tmp(cup3d_sim_t *, double Rtol, double Ctol, int level_max_vorticity);
/* ... on a mesh spread over ranks (collective; mesh / owner as for cup3d_adapt_migrate): the chi blocks behind edges, corners and finer
 * neighbours that other ranks own arrive first, by the plan of the rank's tensorial view (SynchronizerMPI_AMR with a tensorial stencil) */
CUP3D_API int cup3d_grad_chi_on_tmp_over_ranks(cup3d_sim_t *, const cup3d_grid_t *mesh, const int32_t *owner, double Rtol, double Ctol, int level_max_vorticity);

/* per-kernel device time accounting (hipEvents on the compute stream) */
CUP3D_API int cup3d_profile_enable(int on);
CUP3D_API int cup3d_profile_reset(void);
/* fills up to max entries; returns the number of distinct kernels in *n */
typedef struct { char name[48]; long launches; double total_ms; } cup3d_profile_entry;
/* run statistics since the last reset, this process: what the rank handed to RCCL (the payload of the reference's MPI_Isend at
 * main.cpp:2402 and of its MPI_(I)allreduce sites) and how long the host thread sat waiting for device results (the reference's
 * MPI_Waitall / MPI_Wait, 2324, 14490, 14550) */
typedef struct {
  long halo_exchanges;      /* face-slab / ghost-block / face-flux exchanges started */
  double halo_bytes_sent;   /* bytes of those this rank sent */
  long allreduces;          /* scalar all-reduces issued */
  long host_waits;          /* times the host waited for scalars of the device */
  double host_wait_seconds; /* ... and for how long in total */
  long solver_iterations;   /* BiCGSTAB iterations (Poisson and Helmholtz solves) */
  double field_bytes_uploaded;   /* field data that crossed the host boundary (cup3d_sim_upload*, PCIe host -> device) ... */
  double field_bytes_downloaded; /* ... and back (cup3d_sim_download*): what the drop-in adds to a step of the host's time loop */
} cup3d_run_stats;
CUP3D_API int cup3d_stats_reset(void);
CUP3D_API int cup3d_stats_read(cup3d_run_stats *);
CUP3D_API int cup3d_profile_read(cup3d_profile_entry *entries, int max, int *n);
/* VERIFICATION SUPPORT: the bits of the Poisson path under any sharding of the blocks (bench.py's config.checksum at every N).
 * Fills PoissonSolverAMR's 18 work vectors (main.cpp:14382-14399) with a function of (vector, level, global cell index) only, sets
 * alpha, beta, omega (and the mean-constraint total) BY HAND and runs the kernels of ONE BiCGSTAB iteration exactly as
 * cup3d_poisson_solve launches them -- t = A what (14549), loop 1 (14453-14464), zhat = M^-1 z (14488, getZImplParallel 14704-14745),
 * v = A zhat (14489), loop 2 (14502-14515), what = M^-1 w (14548): the fused kernels, the width-1 scalar halo exchanges, the
 * inner / boundary split -- with no dot product feeding back.  The stencil and the block-local solve do not see the partition, so
 * sums[18] (wrapping 64-bit sums of each vector's bit patterns over the rank's blocks, vector order of poisson.hip), added over the
 * ranks mod 2^64, are the same at every N.  block_solver 0, 1 or 2 (the solvers with fused kernels); clobbers the work vectors only. */
CUP3D_API int cup3d_poisson_path_checksum(cup3d_sim_t *, int block_solver, int mean_constraint, unsigned long long *sums18);
/* CG iterations of the last block-CG launch made while cup3d_profile_enable(1) was on, summed over the rank's blocks (the flop count
 * behind bench.py's FP64 roofline of the block preconditioner, getZImplParallel main.cpp:14704-14745) */
CUP3D_API int cup3d_profile_block_cg_iterations(cup3d_sim_t *, long *total, long *nblocks);

#ifdef __cplusplus
}
#endif
#endif
