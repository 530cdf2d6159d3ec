/* cup3d_hip_testing.h -- TEST / TUNING SUPPORT of libcup3d_hip_testing.so (built with -DCUP3D_TESTING).  NOT part of the drop-in
 * boundary (include/cup3d_hip.h): nothing here replaces a reference interface.  The release library (libcup3d_hip.so) does not export any of these
 * names (tests/test_host_indexing.py checks their absence).
 */
#ifndef CUP3D_HIP_TESTING_H
#define CUP3D_HIP_TESTING_H
#include "cup3d_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* kernel variants / ablations by name (0 = production behaviour) */
CUP3D_API int cup3d_debug_set_option(const char *name, int value);
/* several ranks' sims in one process on one GPU: exchanges become no-ops filled by cup3d_debug_halo_pull */
CUP3D_API int cup3d_debug_virtual_ranks(int on);
CUP3D_API int cup3d_debug_halo_pull(cup3d_sim_t *dst, cup3d_sim_t *const *peers, int npeers, int field, int nc, int w);
/* in-process communicator over `nranks` host threads (one Sim per thread-rank, ordinary entry points); 0 tears it down */
CUP3D_API int cup3d_debug_virtual_comm(int nranks);
/* a single Runge-Kutta stage of cup3d_advect_diffuse; the ghost slabs of the interface faces of a multi-level mesh */
CUP3D_API int cup3d_debug_advdiff_stage(cup3d_sim_t *, int rk, double dt, double nu, const double uinf[3]);
CUP3D_API int cup3d_debug_amr_slabs(cup3d_sim_t *, int field, int w, double *out);
CUP3D_API int cup3d_debug_wave_sum(const double *in64, double *out128);
/* the solver's scalar recurrences (SolverCtl, poisson.hip) stepped on the host -- the same functions the device runs; no GPU needed.
 * io[16] = alpha, beta, omega, r0r_prev, norm, init_norm, min_norm, tol, tol_rel, state (0 run, 1 done, 2 restart), restarts,
 * max_restarts, xcur, xopt, iter; step 1 takes totals[2] (main.cpp:14493), step 2 totals[7] (14558-14601) */
CUP3D_API int cup3d_debug_ctl_step(int step, double *io, const double *totals);
/* the multigrid option's level hierarchies of all `nranks` ranks for the leaf ownership `owner[nblocks]` of a global multi-level mesh,
 * checked against each other (tables in range, exchange plans symmetric node for node, every ancestor's octants complete); no GPU */
CUP3D_API int cup3d_debug_mg_plan_check(const cup3d_grid_t *mesh, const int32_t *owner, int nranks);
/* the local slots whose kernels run before the halo exchange has completed (cup3d_grid_ninner of them); no GPU */
CUP3D_API int cup3d_debug_grid_inner_blocks(const cup3d_grid_t *grid_or_view, int32_t *slots);
/* a rank's TENSORIAL view of a mesh (cup3d_grid_rank_view gives the star-stencil one): what cup3d_adapt_migrate and
 * cup3d_grad_chi_on_tmp_over_ranks build internally -- edge / corner neighbours are ghosts too, whole blocks travel; no GPU */
CUP3D_API int cup3d_debug_grid_rank_view_tensorial(const cup3d_grid_t *mesh, const int32_t *owner, int rank, int nranks, cup3d_grid_t **view);

/* HOST-MEMORY TRANSPORT in RCCL's place: one process per rank as in production, but every exchange of the library (face slabs, ghost
 * blocks, face fluxes, block migration, scalar all-reduces) is staged through host memory and carried by the CALLER's transport --
 * the reference's own MPI in the test harness (oracle/ref_harness.cpp).  RCCL refuses two ranks on one device; this lets the
 * multi-rank code of the C++ shim and of the library run as real processes on a one-GPU box.  Blocking and slow by design.
 *   exchange:  for every peer p: send_bytes[p] bytes at sendbuf + send_off[p] go to p, recv_bytes[p] bytes from p land at
 *              recvbuf + recv_off[p] (the caller's rank has zero bytes both ways); host memory
 *   allreduce: n doubles in place, sum (is_max = 0) or max */
typedef struct {
  void *ctx;
  int (*exchange)(void *ctx, const void *sendbuf, const long *send_off, const long *send_bytes, void *recvbuf, const long *recv_off,
                  const long *recv_bytes);
  int (*allreduce)(void *ctx, double *buf, int n, int is_max);
} cup3d_host_transport;
CUP3D_API int cup3d_debug_host_transport(int rank, int nranks, const cup3d_host_transport *t); /* t = NULL: remove it */

#ifdef __cplusplus
}
#endif
#endif
