/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (cup3d_amd/, include/) never does.
 *
 * Plain C restatement of slitvinov/CUP3D's per-block stencil + Poisson path
 * for single-level (uniform) block grids, one rank.  Every function cites the
 * reference lines (main.cpp:NNNN) it follows and keeps the reference's
 * floating-point association so that, built without FMA contraction, it is
 * bit-identical to the reference run with OMP_NUM_THREADS=1.  Pinned against
 * oracle/_ref/ref_tool (the unmodified reference TU) by tests/test_oracle_vs_ref.py
 * and against the committed vectors in tests/golden/.
 *
 * Field layout = the reference's own block memory: scalar [nb][8][8][8],
 * vector [nb][8][8][8][3] (AoS), block order = m_vInfo order (sorted by
 * blockID_2, main.cpp:943-964).
 */
#ifndef CUP3D_ORACLE_H
#define CUP3D_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BC_FREESPACE = 0, ORC_BC_PERIODIC = 1, ORC_BC_WALL = 2 }; /* enum BCflag, main.cpp:6081 */

typedef struct orc_sfc orc_sfc;
typedef struct orc_grid orc_grid;

/* --- indexing (integer, bit-exact contract) ------------------ */
orc_sfc *orc_sfc_create(int bx, int by, int bz, int level_max);             /* main.cpp:196-236 */
void orc_sfc_destroy(orc_sfc *);
long long orc_sfc_forward(const orc_sfc *, int l, int i, int j, int k);       /* main.cpp:237-255 */
void orc_sfc_inverse(const orc_sfc *, long long Z, int l, int ijk[3]);        /* main.cpp:256-276 */
long long orc_sfc_encode(const orc_sfc *, int level, const int index[3]);     /* main.cpp:287-318 */
/* Info::setup tables (main.cpp:384-420): nei[27] (x slowest as Znei[i][j][k]), child[8], parent */
void orc_info_tables(const orc_sfc *, const int bpd[3], int level, const int index[3],
                     long long nei[27], long long child[8], long long *parent);

/* --- uniform grid (GridMPI ctor main.cpp:2959-2986 + FillPos 943-964) -------- */
orc_grid *orc_grid_create(int bx, int by, int bz, int level_max, int level,
                          double maxextent, const int bc[3]);
void orc_grid_destroy(orc_grid *);
long orc_grid_nblocks(const orc_grid *);
double orc_grid_h(const orc_grid *);
/* tables in block order: out6 = level,Z,ix,iy,iz,blockID_2 ; geom4 = h,origin[3] */
void orc_grid_tables(const orc_grid *, long long *out6, double *geom4);
/* contiguous Z-range ownership of rank r of n (main.cpp:2970-2986): first Z and count */
void orc_partition(long long total_blocks, int rank, int size, long long *z_start, long long *count);

/* --- operators ------------------------------------------------------------ */
void orc_ic_taylor_green(const orc_grid *, double *vel, const double ext[3], double umax); /* 12516-12539 */
double orc_max_u(const orc_grid *, const double *vel, const double uinf[3]);           /* 8603-8623 */
/* Simulation::calcMaxTimestep, main.cpp:15254-15305; returns dt, updates coefU when step>2 */
double orc_calc_dt(double hmin, double umax, double nu, double cfl, int step, int rampup,
                   double dt_old, double coefU[3]);
double orc_calc_dt2(double hmin, double umax, double nu, double cfl, int step, int rampup, double dt_old, double coefU[3], int implicitDiffusion);
void orc_external_forcing(const orc_grid *, double *vel, double umax_forced, double nu, double H, double dt); /* 10581-10596 */
/* AdvectionDiffusion::operator(), main.cpp:9640-9728 (RK3, KernelAdvectDiffuse 9461-9549) */
void orc_advect_diffuse(const orc_grid *, double *vel, double *tmpV, double dt, double nu, const double uinf[3]);
/* one application of KernelAdvectDiffuse: tmpV += rhs(vel) (main.cpp:9484-9549) */
void orc_advdiff_stage_rhs(const orc_grid *, const double *vel, double *tmpV, double dt, double nu, const double uinf[3]);
/* ComputeLHS::operator(), main.cpp:9273-9327 */
void orc_lhs(const orc_grid *, const double *pres, double *lhs, int mean_constraint);
/* poisson_kernels::getZImplParallel, main.cpp:14704-14745 (in place on pres) */
void orc_precond(const orc_grid *, double *pres);
/* PoissonSolverAMR::solve, main.cpp:14363-14616: rhs in lhs (clobbered), x0/result in pres */
typedef struct { double tol, tol_rel; int mean_constraint; int iters; int restarts; double norm0, norm; } orc_solve_info;
void orc_solve(const orc_grid *, double *lhs, double *pres, orc_solve_info *);
long orc_precond_block(double *blk, double h); /* one 8^3 block of getZImplParallel; returns its CG iterations */
long orc_precond_block_coef(double *blk, double h, double coefficient); /* diffusion_kernels::getZImplParallel, 10534-10579 */
void orc_solve_generic2(void *mesh, long N, long corner, void (*op_lhs)(void *, const double *, double *, int),
                        void (*op_precond)(void *, double *), double *lhs, double *pres, orc_solve_info *, int max_restarts);
void orc_solve_generic(void *mesh, long N, long corner, void (*op_lhs)(void *, const double *, double *, int),
                       void (*op_precond)(void *, double *), double *lhs, double *pres, orc_solve_info *);
void orc_pressure_rhs(const orc_grid *, const double *vel, const double *udef, const double *chi, double *lhs, double dt); /* 14849-14875 */
void orc_div_pressure(const orc_grid *, const double *pres, double *tmpV);   /* 14769-14778 */
void orc_grad_p(const orc_grid *, const double *pres, double *tmpV, double dt); /* 14990-14999 */
/* PressureProjection::operator(), main.cpp:15061-15160 (no obstacles: chi=0, udef=0) */
void orc_project(const orc_grid *, double *vel, double *pres, double *tmpV, double *lhs, const double *chi,
                 double dt, int step, orc_solve_info *);
/* --- mesh-adaptation block operators (uniform level l <-> l+1, whole mesh) ----------------- */
/* "restrict": MeshAdaptation::compress, main.cpp:5290-5303 (8-cell mean, its association) */
void orc_restrict(const orc_grid *fine, const orc_grid *coarse, const double *ffield, double *cfield, int nc);
/* "prolong": refine_1 + RefineBlocks, main.cpp:5227-5249, 5493-5565: 2nd-order Taylor expansion from the tensorial
 * [-1,2) lab (all 26 neighbours + domain-face rules of BlockLabBC / BlockLabNeumann3D incl. edge fills) */
void orc_prolong(const orc_grid *coarse, const orc_grid *fine, const double *cfield, double *ffield, int nc, int is_vector);
/* TagLoadedBlock (5566-5582) + the level clamps of TagBlocksVector (5207-5211): states[b] in {-1 Compress, 0 Leave, 1 Refine} */
void orc_tag(const orc_grid *, const double *field, int nc, double rtol, double ctol, signed char *states);
/* --- multi-level (AMR) meshes: cup3d_oracle_amr.c ------------------------------------------- */
typedef struct orc_mesh orc_mesh;
orc_mesh *orc_mesh_create(int bx, int by, int bz, int level_max, double maxextent, const int bc[3], long nblocks,
                          const int *levels, const long long *Zs);
void orc_mesh_destroy(orc_mesh *);
long orc_mesh_nblocks(const orc_mesh *);
void orc_mesh_tables(const orc_mesh *, long long *out6); /* level,Z,index[3],blockID_2 in m_vInfo order */
double orc_mesh_h(const orc_mesh *, long block);
/* ghosted tiles of every block for stencil [s,e)^3 as BlockLab::load builds them (main.cpp:3623-4614):
 * out [nb][L][L][L][nc], L = 8 + e - s - 1 */
void orc_mesh_labs(const orc_mesh *, const double *field, int nc, int is_vector, int s, int e, int tensorial, double *out);
/* the operators on multi-level meshes, each followed by the reference's flux correction at coarse/fine faces
 * (compute<Lab>(kernel, g, g_corr), main.cpp:5584-5644; FluxCorrection 588-802) */
void orc_mesh_advdiff_stage_rhs(const orc_mesh *, const double *vel, double *tmpV, double dt, double nu, const double uinf[3]);
void orc_mesh_advect_diffuse(const orc_mesh *, double *vel, double *tmpV, double dt, double nu, const double uinf[3]);
void orc_mesh_lhs(const orc_mesh *, const double *pres, double *lhs, int mean_constraint);
void orc_mesh_precond(const orc_mesh *, double *pres);
void orc_mesh_solve(const orc_mesh *, double *lhs, double *pres, orc_solve_info *);
void orc_mesh_pressure_rhs(const orc_mesh *, const double *vel, const double *udef, const double *chi, double *lhs, double dt);
void orc_mesh_div_pressure(const orc_mesh *, const double *pres, double *tmpV);
void orc_mesh_grad_p(const orc_mesh *, const double *pres, double *tmpV, double dt);
void orc_mesh_project(const orc_mesh *, double *vel, double *pres, double *tmpV, double *lhs, const double *chi, double dt, int step,
                      orc_solve_info *);
void orc_mesh_project_obst(const orc_mesh *, double *vel, double *pres, double *tmpV, double *lhs, const double *chi, double dt, int step,
                           orc_solve_info *, long n, const long long *ids, const double *ochi, const double *oudef);
double orc_mesh_max_u(const orc_mesh *, const double *vel, const double uinf[3]);
/* implicit diffusion (AdvectionDiffusionImplicit, main.cpp:10030-10118); `sequential`: see cup3d_oracle_amr.c */
void orc_mesh_advect_implicit(const orc_mesh *, double *vel, double *tmpV, double dt, double nu, const double uinf[3], int sequential);
void orc_mesh_diffusion_rhs(const orc_mesh *, const double *vel, double *tmpV);
void orc_mesh_diff_lhs(const orc_mesh *, const double *pres, double *lhs, int direction, double dt, double nu);
void orc_mesh_diff_precond(const orc_mesh *, double *pres, double dt, double nu);
void orc_mesh_diff_solve(const orc_mesh *, double *lhs, double *pres, int direction, double dt, double nu, orc_solve_info *);
void orc_mesh_advdiff_implicit(const orc_mesh *, double *vel, double *pres, double *tmpV, double *lhs, double dt, double nu,
                               const double uinf[3], double tol, double tol_rel, int sequential, int iters[3]);
void orc_mesh_vorticity(const orc_mesh *, const double *vel, double *tmpV); /* ComputeVorticity, main.cpp:8624-8746 */
void orc_mesh_tag(const orc_mesh *, const double *field, int nc, double rtol, double ctol, signed char *states);
/* compute<ScalarLab>(GradChiOnTmp(sim), sim.chi), main.cpp:8540-8600: tmpV (the vorticity) edited in place from chi */
void orc_mesh_grad_chi_on_tmp(const orc_mesh *, const double *chi, double *tmpV, double Rtol, double Ctol, int level_max_vorticity);
void orc_mesh_states(const orc_mesh *, int *out27);
/* obstacle operators for one obstacle given by its ObstacleBlocks (ids, chi[n][512], udef[n][512][3]) and rigid = cm[3], vel[3],
 * omega[3]: KernelPenalization + kernelFinalizePenalizationForce (13841-13938), kernelUpdateTmpV (14948-14979) */
void orc_mesh_penalize(const orc_mesh *, double *vel, const double *chi_field, long n, const long long *ids, const double *chi,
                       const double *udef, const double rigid[9], double dt, double lambda, int implicit, double force6[6]);
void orc_mesh_update_tmpv(const orc_mesh *, double *tmpV, const double *chi_field, long n, const long long *ids, const double *chi,
                          const double *udef);
/* mesh adaptation (one rank): ValidStates 5330-5492 (in/out states), the leaf set after Adapt 5086-5159, and the field data
 * on the adapted mesh (RefineBlocks 5493-5565 from the old mesh's tensorial tiles, compress 5272-5329, copies) */
void orc_mesh_valid_states(const orc_mesh *, signed char *states);
long orc_mesh_adapted_leaves(const orc_mesh *, const signed char *states, int *levels, long long *Zs);
/* ownership of the adapted mesh's leaves on several ranks (LoadBalancer, main.cpp:4660-5022) */
void orc_mesh_adapted_owners(const orc_mesh *old_mesh, const int *owner, const signed char *states, int nranks, const orc_mesh *new_mesh, int *new_owner);
void orc_mesh_transfer(const orc_mesh *old_mesh, const orc_mesh *new_mesh, const double *f_old, double *f_new, int nc, int is_vector); /* octree states of the 27 neighbour positions of every block */
#ifdef __cplusplus
}
#endif
#endif
