// Device mirror of SimulationData's five block grids (main.cpp:6603-6607) plus the
// solver work vectors of PoissonSolverAMR (main.cpp:9394-9411), for one rank.
//
// HBM layout: every field is one slab  [block slot][component][z][y][x]  of FP64
// (SoA per block: 4 KiB per component per block), slots in m_vInfo order.  Ghost
// cells are never materialised in HBM: kernels stage the ghosted tile of a block in
// LDS straight from the neighbour slots (this replaces BlockLab::load, 3623-3743).
// Face slabs owned by other ranks live in `halo_*` buffers filled by RCCL.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/cup3d_hip.h"
#ifdef CUP3D_TESTING
#include "../../include/cup3d_hip_testing.h"  // the test-support entry points are exported (CUP3D_API) from the testing build only
#endif
#include "grid.hpp"

namespace cup3d {

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);
#define CUP3D_HIP(call)                                                        \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) return ::cup3d::hip_fail(e_, #call, __FILE__, __LINE__); \
  } while (0)

hipStream_t stream();  // compute stream (cup3d_set_stream)
// Two flavours of the library are built from these sources (Makefile): libcup3d_hip.so, the release build, and
// libcup3d_hip_testing.so (-DCUP3D_TESTING), which adds what tests and tuning scans need and a deployment does not: the debug-option
// map read on launch paths, the in-process "virtual" communicator (comm.hip), the evaluation variants of the block CG.  In the
// release build debug_option() is a constant 0 and the variant branches fold away.
#ifdef CUP3D_TESTING
int debug_option(const char *name);  // cup3d_debug_set_option; 0 when unset (production behaviour)
#else
constexpr int debug_option(const char *) { return 0; }
#endif
int not_in_release(const char *what);  // sets the error text, returns CUP3D_ESTATE
bool profile_on();                   // cup3d_profile_enable

// ---- per-kernel timing (cup3d_profile_*) ----
struct ProfileScope {
  explicit ProfileScope(const char *name);        // events on the compute stream
  ProfileScope(const char *name, hipStream_t st);  // ... on `st` (the communication stream: "comm_*" entries)
  ~ProfileScope();
  int idx;
  hipEvent_t start;
  hipStream_t st;
};

// ---- run statistics (cup3d_stats_*): what crossed ranks and how long the host waited for the device
void stats_host_wait(double seconds);
void stats_solver_iterations(long n);
void stats_halo(size_t bytes_sent);
void stats_allreduce();

struct Comm;  // RCCL state (comm.cpp)
Comm *comm();  // nullptr when single rank
bool virtual_ranks();  // test mode (cup3d_debug_virtual_ranks)
bool host_transport();  // test mode (cup3d_debug_host_transport): exchanges staged through host memory and the caller's transport

// Device-side description of the topology, passed by value to kernels.
struct GridDev {
  const int32_t *nbr;   // [nb][6]
  const int32_t *list;  // optional block list (inner / boundary); nullptr = all blocks
  int nblocks;          // number of blocks this launch covers
  int chunk;            // ceil(nblocks / 8): blocks per XCD
  double h;
  // multi-level meshes: per-block spacing (nullptr on uniform grids) and the face-flux arrays of the
  // interface faces [(e*nc + c)][64] the kernel must fill (e = nbr - kNbrHalo); see amr.hip
  const double *hb;
  double *flux;
  // multi-level meshes: 1 for the blocks that have a coarse-side interface face, i.e. whose result the flux correction still edits;
  // the advect-diffuse stage leaves their raw increment in tmpV and fuses the Runge-Kutta update for all the others
  const unsigned char *raw;
};
__device__ __forceinline__ double block_h(const GridDev &g, int slot) { return g.hb ? g.hb[slot] : g.h; }

// The 16 doubles of Sim::d_red, by user.  The ranges must stay disjoint: the solver's scalars at kRedDots are live from one loop
// kernel to the next, and a checksum or a multigrid cycle may run in between.
enum RedSlot : int {
  kRedDots = 0,      // [0, 8): totals of the 2 / 7 dot products of a BiCGSTAB loop, the mean-constraint total riding along behind them
                     //         ([2] after loop 1, [7] after loop 2); outside a solve: max|u| ([0]), the penalisation's 6 sums + status ([0, 7))
  kRedDotsEnd = 8,
  kRedMeanLhs = 8,   // [8]: sum(p h^3) of a stand-alone LHS application (k_mean_finish, stencil.hip)
  kRedEarlyMean = 9, // [9], [10]: several ranks, early all-reduce: sum(zhat h^3) / sum(what h^3) of a fused loop, all-reduced on their own behind a flag
  kRedChecksum = 12, // [12]: cup3d_sim_checksum's 64-bit accumulator
  kRedMg = 14,       // [14]: coarsest-level sum of the multigrid cycle over ranks (multigrid.hip)
  kRedSize = 16
};
static_assert(kRedDotsEnd <= kRedMeanLhs && kRedMeanLhs < kRedEarlyMean && kRedEarlyMean + 2 <= kRedChecksum && kRedChecksum < kRedMg && kRedMg < kRedSize, "Sim::d_red slots overlap");

struct Sim {
  const Grid *grid = nullptr;
  int64_t nb = 0;    // local blocks: what the kernels sweep and what crosses the host boundary
  int64_t nvis = 0;  // block slots of every field array: nb + the ghost blocks of a rank view (Grid::rank_view), else nb
  // topology on device
  int32_t *d_nbr = nullptr, *d_inner = nullptr, *d_boundary = nullptr, *d_send_faces = nullptr;
  // fields
  double *vel = nullptr, *vel2 = nullptr, *tmpV = nullptr;    // [nb][3][512]
  double *pres = nullptr, *lhs = nullptr, *chi = nullptr;     // [nb][512]
  double *pold = nullptr;
  bool chi_nonzero = false;   // chi was uploaded / filled non-zero: the pressure RHS reads chi and udef (= tmpV)
  bool udef_nonzero = false;  // tmpV holds the udef of the NEXT projection (uploaded, filled or cup3d_update_tmpv since the last one)
  // cup3d_sim_set_obstacles: does ANY rank hold an obstacle (the reference's obstacle_vector is replicated, every rank knows)?
  // -1 = not told: several ranks then always take the chi / udef path of the pressure right-hand side, because its udef exchange is a
  // collective and chi_nonzero is per-rank state
  int obstacles_global = -1;
  // one rank: told "obstacles" or chi was written (a chi filled by hand after set_obstacles(0) still counts -- never silently dropped);
  // several ranks: as told, the same on every rank (told 0 with a written chi is refused by cup3d_pressure_project, chi_conflict())
  bool chi_path() const { return grid->nranks == 1 ? (obstacles_global == 1 || chi_nonzero) : obstacles_global != 0; }
  bool chi_conflict() const { return grid->nranks > 1 && obstacles_global == 0 && chi_nonzero; }
  int block_solver = 0;  // cup3d_poisson_params.block_solver of the running solve
  int scalar_bc_dir = -1;  // >= 0 while a Helmholtz solve of the implicit diffusion runs: domain-face rule of the scalar tiles
  // solver vectors (allocated on first solve), each [nb][512]
  double *sv[18] = {nullptr};
  // reductions
  double *d_partials = nullptr;  // [max_groups][8]
  double *d_red = nullptr;       // [kRedSize] final reduced scalars, see RedSlot
  const double *sums_of = nullptr;  // vector whose per-block sums (mean constraint) are current in d_partials' tail
  double *h_red = nullptr;       // pinned host mirror
  double *h_red_dev = nullptr;   // the same memory as the device sees it (kernels store the reduced scalars there directly)
  unsigned *d_counters = nullptr;  // [4] tickets of grid_sum_finish (tile.hpp), zero between launches
  int *d_cg_iters = nullptr;       // [nb] CG iterations per block of the last block-CG launch (only while profiling)
  double *d_block_dots = nullptr;  // [7][nb] per-block dot products of the fused loop + block-CG kernels (poisson.hip)
  // ... and what totals them inside those kernels (Arrive, poisson.hip): group / super-group sums, arrival counters, the two flags the
  // communication stream (dots) and the corner block's wavefront (mean-constraint total) wait for when the all-reduce starts early
  double *d_arrive_sums = nullptr;
  unsigned *d_arrive = nullptr;
  struct LoopSums *d_loop_sums = nullptr;  // [2] (poisson.hip)
  unsigned *h_early_fail = nullptr, *h_early_fail_dev = nullptr;  // pinned: a bounded device-side wait of the early all-reduce gave up
  void *mg = nullptr;              // level hierarchy of the multigrid preconditioner (multigrid.hip), built on first use
  int max_groups = 0;
  // staging for host transfers
  double *d_stage = nullptr;
  double *h_stage = nullptr;  // pinned; two buffers of stage_blocks vector blocks each, like d_stage
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  size_t stage_blocks = 0;
  bool stage_ready = false;  // every staging resource exists (ensure_stage is all-or-nothing)
  int32_t *d_stage_slots = nullptr, *h_stage_slots = nullptr;  // block lists of the partial transfers (cup3d_sim_*_block_list)
  // multi-level mesh tables (amr.hip); all nullptr / 0 on uniform grids
  int32_t *d_amr_faces = nullptr, *d_amr_fine = nullptr, *d_nbr27 = nullptr, *d_index = nullptr;
  int32_t *d_restrict_list = nullptr, *d_prolong_list = nullptr, *d_fix_list[3] = {nullptr, nullptr, nullptr};
  const double *mean_total_of = nullptr, *mean_total = nullptr;  // the vector whose sum(p h^3) total is already in *mean_total (device)
  unsigned red_seq = 0;  // last sequence number handed to a reduction kernel (Reducer, poisson.hip)
  // BiCGSTAB's scalar recurrences on the device (SolverCtl, poisson.hip): the struct, and the pinned ring its outcome reaches the host through
  void *d_ctl = nullptr, *h_ctl = nullptr, *h_ctl_dev = nullptr;
  unsigned ctl_seq = 0;
  unsigned n_restrict = 0, n_prolong = 0;
  int32_t *d_fix_blocks = nullptr;  // [n_fix_blocks][6]: per block with a corrected (coarse-side) face, the interface face behind each of its six faces or -1
  unsigned n_fix_blocks = 0;
  unsigned n_restrict_inner = 0, n_prolong_inner = 0;  // leading entries of the two lists that belong to inner blocks (rank views)
  struct { const double *field = nullptr; int nc = 0, w = 0, bc_dir = -1; double *slabs = nullptr; bool open = false; } pending_fill;  // halo_begin -> halo_finish
  int32_t *d_send_blocks = nullptr, *d_send_flux = nullptr;  // rank views: exchange plans (comm.hip)
  // ... and the sub-box form of the ghost-block exchange (Grid::ghost_box / send_box), per stencil-width class: boxes, the offset (in cells
  // per component) of every block's cells in the packed message, the staging buffer the messages arrive in before they are scattered
  unsigned char *d_send_box[2] = {nullptr, nullptr}, *d_ghost_box[2] = {nullptr, nullptr};
  long long *d_send_off[2] = {nullptr, nullptr}, *d_ghost_off[2] = {nullptr, nullptr};
  double *box_recv = nullptr;
  unsigned char *d_raw_mask = nullptr;  // [nb], see GridDev::raw
  int32_t *d_raw_list = nullptr;        // the blocks with raw_mask set
  // multi-level mesh on one rank: blocks with an interface face (coarse/fine: ghost slabs, face fluxes, flux correction) and the rest,
  // whose six faces are same-level blocks or domain faces -- for those the LHS application lives inside the fused loop kernels (poisson.hip)
  int32_t *d_iface_list = nullptr, *d_plain_list = nullptr;
  unsigned n_iface = 0, n_plain = 0;
  bool corner_is_plain = false;  // the block of the mean-constraint row (grid->corner_slot) is in the plain list
  unsigned n_raw = 0;
  double *d_hb = nullptr, *d_flux = nullptr;
  // halo buffers (multi-rank)
  double *halo_recv = nullptr, *halo_send = nullptr;  // n faces x 3 comps x 3 layers x 64
  size_t bytes = 0;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_h1 = nullptr, ev_h2 = nullptr, ev_m = nullptr;
  hipEvent_t ev_vc_pack = nullptr, ev_vc_done = nullptr;  // virtual communicator (tests): "my send buffer is packed" / "my copies are enqueued"

  GridDev gdev(bool boundary_only = false, bool inner_only = false) const;
  GridDev gdev_list(const int32_t *list, unsigned n) const;  // the same over an explicit block list
  double *field(int id, int *ncomp) const;
};

int sim_alloc(double **p, size_t n_doubles, Sim *s);
// wrapping 64-bit sum of the bit patterns of n doubles (independent of launch geometry and of the sharding); synchronises
int checksum_array(Sim *s, const double *p, long n, unsigned long long *sum);
Sim *sim_comm_only(const Grid *g, hipStream_t comm_stream);  // exchange-only Sim of a coarse multigrid level (sim.hip)
void sim_comm_only_destroy(Sim *s);

// halo exchange of the face slabs of `field` (ncomp components, w ghost layers); no-op on one rank
int halo_exchange(Sim *s, const double *field, int ncomp, int w);
// overlapped form: begin on the communication stream, finish = compute stream waits for the slabs
int halo_begin(Sim *s, const double *field, int ncomp, int w);
int halo_finish(Sim *s);
// sum / max all-reduce of n doubles resident in device memory; no-op on one rank
int allreduce(Sim *s, double *d_buf, int n, bool is_max, hipStream_t st);
// collective status agreement before the first exchange of a collective entry point: every rank passes its local outcome, all get the
// worst one back (comm.hip); no-op on one rank
int agree(Sim *s, int rc, const char *where);
bool scalars_cross_ranks(const Sim *s);   // does allreduce() do anything for this sim?
hipStream_t scalar_stream(const Sim *s);  // the stream all-reduces are enqueued on (communication stream where there is one)
// rank views of a multi-level mesh: face-flux arrays of remote fine faces -> ghost face range of d_flux (before k_flux_fix)
int view_exchange_flux(Sim *s, int nfc);
// ... and whole blocks of `field` -> ghost slot range, before the ghost slabs of a stencil kernel are built (no-op elsewhere)
int view_exchange_blocks(Sim *s, double *field, int nc, int w);  // w: width of the star stencil that will read them (1 or 3)
// generic exchange of `per`-double items between ranks (own rank included), peer-major buffers; compute stream
int exchange_items(Sim *s, const double *sendbuf, const std::vector<int64_t> &send_count, double *recvbuf, const std::vector<int64_t> &recv_count, size_t per);
void vcomm_register(Sim *s);    // in-process test communicator (comm.hip)
void vcomm_unregister(Sim *s);

// multi-level meshes: ghost slabs of every interface face of `field` for a w-deep star stencil -> slabs;
// flux correction of `out` (out_nc components per block, the first nfc corrected) from s->d_flux
int amr_fill_ghosts(Sim *s, const double *field, int ncomp, int w, double *slabs, int part = 0);  // part 0: every face, 1: inner blocks' faces, 2: boundary blocks' faces
int amr_flux_fix(Sim *s, int nfc, double *out, int out_nc);

// kernels' launchers shared across translation units
// list != nullptr: only the `nlist` blocks listed (and the mean-constraint fix-ups of those blocks only; needs the total of p to be known
// already, Sim::mean_total_of == p) -- the interface blocks of a multi-level mesh, while the loop kernels form the LHS of all others
int launch_lhs(Sim *s, const double *p, double *out, int mean_constraint, const int32_t *list = nullptr, unsigned nlist = 0);
int launch_precond(Sim *s, const double *in, double *out, bool want_sums);
int launch_mean_total(Sim *s);  // total of the block sums in d_partials' tail -> d_red[kRedMeanLhs] (stencil.hip)
// block_solver 5: one multigrid V-cycle from a zero guess as M^-1 (multigrid.hip; an alternative, not the reference's algorithm)
int mg_vcycle(Sim *s, const double *in, double *out);
void mg_destroy(Sim *s);
// implicit diffusion (DiffusionSolver, main.cpp:6719-7147): Helmholtz operator of velocity component `direction`
struct HelmholtzOp { int direction; double dt, nu; };
int launch_lhs_diffusion(Sim *s, const double *p, double *out, const HelmholtzOp &op);
int launch_precond_diffusion(Sim *s, const double *in, double *out, const HelmholtzOp &op);
int solve_helmholtz(Sim *s, const cup3d_poisson_params &P, cup3d_poisson_result *res, const HelmholtzOp &op);
int launch_diffusion_rhs(Sim *s);
int launch_advect_implicit(Sim *s, double dt, double nu, const double uinf[3]);

}  // namespace cup3d
