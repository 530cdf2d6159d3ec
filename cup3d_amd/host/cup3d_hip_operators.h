// cup3d_hip_operators.h — C++ host side of the drop-in: the reference's plugin surface
// (class Operator, main.cpp:6678-6684; class PoissonSolverBase, 8921-8928) implemented on
// top of the C ABI of include/cup3d_hip.h.
//
// Include this header in a translation unit AFTER the reference's declarations (e.g. after
// `#include "main.cpp"` with `main` renamed).  Nothing in main.cpp is
// modified: cup3d_hip::install(simulation) swaps the AdvectionDiffusion and
// PressureProjection entries of sim.pipeline (built by Simulation::setupOperators,
// 15229-15246) and sim.pressureSolver for the HIP-backed ones below.
//
// Data ownership stays with the reference: the five GridMPI objects own the block memory
// (one posix_memalign per block, 877-884).  Each operator uploads the blocks it reads
// (Info::block pointers, m_vInfo order), runs on the GPU and downloads what the reference
// would have written, so every CPU-side operator in between (obstacles, penalisation,
// mesh adaptation, dump) keeps working unchanged.  The device mirror is rebuilt whenever
// the block list changes.  Errors follow the reference's convention: print and MPI_Abort.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <map>
#include <utility>
#include <vector>

#include "../../include/cup3d_hip.h"
#ifdef CUP3D_HIP_SHIM_TEST_TRANSPORT  // test builds only (oracle/Makefile: ref_tool_hip_mpi_testing, linked against libcup3d_hip_testing.so)
#include "../../include/cup3d_hip_testing.h"
#endif

namespace cup3d_hip {

#ifdef CUP3D_HIP_SHIM_TEST_TRANSPORT
// The library's exchanges carried by the host's own MPI through host memory instead of RCCL (cup3d_debug_host_transport): RCCL
// refuses two ranks on one device, so this is how the multi-rank branch below runs as real MPI processes on a one-GPU box.
// Selected at run time with CUP3D_HIP_HOST_TRANSPORT=1; a production build of the shim does not contain it.
struct HostTransport {
  MPI_Comm comm;
  static int exchange(void *ctx, const void *sb, const long *so, const long *sn, void *rb, const long *ro, const long *rn) {
    MPI_Comm c = static_cast<HostTransport *>(ctx)->comm;
    int size = 1;
    MPI_Comm_size(c, &size);
    std::vector<MPI_Request> rq;
    for (int p = 0; p < size; ++p)
      if (rn[p]) { rq.emplace_back(); MPI_Irecv((char *)rb + ro[p], (int)rn[p], MPI_BYTE, p, 4711, c, &rq.back()); }
    for (int p = 0; p < size; ++p)
      if (sn[p]) { rq.emplace_back(); MPI_Isend((const char *)sb + so[p], (int)sn[p], MPI_BYTE, p, 4711, c, &rq.back()); }
    return MPI_Waitall((int)rq.size(), rq.data(), MPI_STATUSES_IGNORE) == MPI_SUCCESS ? 0 : 1;
  }
  static int allreduce(void *ctx, double *buf, int n, int is_max) {
    MPI_Comm c = static_cast<HostTransport *>(ctx)->comm;
    return MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE, is_max ? MPI_MAX : MPI_SUM, c) == MPI_SUCCESS ? 0 : 1;
  }
};
#endif

inline void die(const char *what, int rc) {
  fprintf(stderr, "cup3d_hip: %s failed (%d): %s\n", what, rc, cup3d_last_error());
  fflush(0);
  MPI_Abort(MPI_COMM_WORLD, 1);
}
#define CUP3D_HIP_CALL(x)                       \
  do {                                          \
    const int rc_ = (x);                        \
    if (rc_ != CUP3D_OK) ::cup3d_hip::die(#x, rc_); \
  } while (0)

// Device mirror of one SimulationData: topology + the five fields.
class DeviceMirror {
public:
  explicit DeviceMirror(SimulationData &s) : sim(s) {}
  ~DeviceMirror() { release(); }
  DeviceMirror(const DeviceMirror &) = delete;

  cup3d_sim_t *handle() {
    ensure();
    return dsim;
  }
  // resident mode (install(sim, true) / CUP3D_HIP_RESIDENT=1): between AdvectionDiffusionHIP and PressureProjectionHIP the
  // velocity stays in HBM (no download / upload), see install()
  bool resident = false;      // allowed at all: decided by install() from the pipeline
  bool vel_on_device = false; // the device copy of vel is newer than the host's
  // resident ACROSS steps (install(sim, 2) / CUP3D_HIP_RESIDENT=2): after PressureProjectionHIP the host holds a copy of the
  // device's vel and pres; as long as nobody but the intercepted operators writes those two fields on the host, the next step need
  // not send them up again (3.2 + 1.1 GB per step at 512^3).  That "as long as" is a CONTRACT with the host code: the reference's own
  // step writes vel / pres only inside the operators this header replaces or wraps and in adaptMesh (which changes the block list:
  // ensure() notices and drops the flags); any other writer -- an initial condition, a restart, a test harness loading a field --
  // must call invalidate().  Off by default for that reason.
  bool across_steps = false;
  bool host_vel_clean = false, host_pres_clean = false;  // host copy == device copy
  // DEVICE-LED (install(sim, 3) / CUP3D_HIP_RESIDENT=3): vel and pres live in HBM from one step to the next and the host copy is only
  // refreshed on demand (sync_host(): before adaptMesh, a dump, or anything else on the host that reads them).  Needs the ONE edit of
  // the reference this shim cannot make from outside: Simulation::calcMaxTimestep (15254-15305) reads the host velocity through
  // findMaxU (15259) every step, so the time loop must call cup3d_hip::calcMaxTimestep(simulation, mirror) instead -- the same
  // function with the maximum taken on the device.  Obstacle-free pipelines only: ComputeForces reads the
  // host fields every step (12250-12495), so with obstacles the mode falls back to `across_steps`.
  bool device_led = false;
  bool dev_current = false;  // the device copies of vel and pres are newer than the host's
  void invalidate() { host_vel_clean = host_pres_clean = false; dev_current = false; }
  void sync_host() {
    if (!dev_current) return;
    download(CUP3D_FIELD_VEL);
    download(CUP3D_FIELD_PRES);
    dev_current = false;
    host_vel_clean = host_pres_clean = true;
  }
  void upload(int field) {
    ensure();
    const std::vector<Info> &I = infos(field);
    ptrs.resize(I.size());
    for (size_t i = 0; i < I.size(); ++i) ptrs[i] = I[i].block;
    CUP3D_HIP_CALL(cup3d_sim_upload_blocks(dsim, field, (const void *const *)ptrs.data()));
  }
  void download(int field) {
    const std::vector<Info> &I = infos(field);
    ptrs.resize(I.size());
    for (size_t i = 0; i < I.size(); ++i) ptrs[i] = I[i].block;
    CUP3D_HIP_CALL(cup3d_sim_download_blocks(dsim, field, (void *const *)ptrs.data()));
  }
  // Partial transfers of the blocks an obstacle covers (non-null ObstacleBlock of any obstacle): the host-side obstacle operators
  // between AdvectionDiffusion and PressureProjection read and write the velocity in exactly those blocks
  // (KernelIntegrateFluidMomenta 13637-13641, KernelPenalization 13853-13861), so in resident mode only they cross PCIe.
  void download_obstacle_blocks(int field) {
    obstacle_blocks(field);
    CUP3D_HIP_CALL(cup3d_sim_download_block_list(dsim, field, (long)oslots.size(), oslots.data(), (void *const *)ptrs.data()));
  }
  void upload_obstacle_blocks(int field) {
    obstacle_blocks(field);
    CUP3D_HIP_CALL(cup3d_sim_upload_block_list(dsim, field, (long)oslots.size(), oslots.data(), (const void *const *)ptrs.data()));
  }

private:
  SimulationData &sim;
  cup3d_grid_t *grid = nullptr;
  cup3d_sim_t *dsim = nullptr;
  std::vector<long long> signature;  // (level, Z) of every local block the mirror was built for
  std::vector<void *> ptrs;
  std::vector<int32_t> oslots;
  bool device_ready = false;

  void obstacle_blocks(int field) {
    ensure();
    const std::vector<Info> &I = infos(field);
    oslots.clear();
    ptrs.clear();
    for (size_t i = 0; i < I.size(); ++i)
      for (const auto &ob : sim.obstacle_vector->getObstacleVector())
        if (ob->getObstacleBlocks()[I[i].blockID] != nullptr) {
          oslots.push_back((int32_t)i);
          ptrs.push_back(I[i].block);
          break;
        }
  }

  const std::vector<Info> &infos(int field) {
    switch (field) {
      case CUP3D_FIELD_CHI: return sim.chiInfo();
      case CUP3D_FIELD_PRES: return sim.presInfo();
      case CUP3D_FIELD_VEL: return sim.velInfo();
      case CUP3D_FIELD_TMPV: return sim.tmpVInfo();
      default: return sim.lhsInfo();
    }
  }
  void release() {
    if (dsim) cup3d_sim_destroy(dsim);
    if (grid) cup3d_grid_destroy(grid);
    dsim = nullptr;
    grid = nullptr;
  }
  void ensure() {
    const std::vector<Info> &I = sim.velInfo();
    bool same = dsim != nullptr && signature.size() == 2 * I.size();
    for (size_t i = 0; same && i < I.size(); ++i) same = signature[2 * i] == I[i].level && signature[2 * i + 1] == I[i].Z;
    if (same) return;
    release();
    invalidate();  // a new block list: nothing of the old mirror is current
    int rank = 0, size = 1;
    MPI_Comm_rank(sim.comm, &rank);
    MPI_Comm_size(sim.comm, &size);
    if (!device_ready) {
      int ndev = 0;
      CUP3D_HIP_CALL(cup3d_device_count(&ndev));
      CUP3D_HIP_CALL(cup3d_device_init(ndev > 0 ? rank % ndev : 0));  // one process per GPU
#ifdef CUP3D_HIP_SHIM_TEST_TRANSPORT
      if (size > 1 && getenv("CUP3D_HIP_HOST_TRANSPORT")) {
        static HostTransport ht;
        ht.comm = sim.comm;
        const cup3d_host_transport t = {&ht, &HostTransport::exchange, &HostTransport::allreduce};
        CUP3D_HIP_CALL(cup3d_debug_host_transport(rank, size, &t));
      } else
#endif
      if (size > 1) {  // bootstrap the library's RCCL communicator over the host MPI
        unsigned char id[128] = {0};
        if (rank == 0) CUP3D_HIP_CALL(cup3d_comm_unique_id(id));
        MPI_Bcast(id, 128, MPI_BYTE, 0, sim.comm);
        CUP3D_HIP_CALL(cup3d_comm_init(rank, size, id));
      }
      device_ready = true;
    }
    const int level = I.empty() ? sim.levelStart : I[0].level;
    const int bpd[3] = {sim.bpdx, sim.bpdy, sim.bpdz};
    const int bc[3] = {(int)sim.BCx_flag, (int)sim.BCy_flag, (int)sim.BCz_flag};  // enum BCflag == CUP3D_BC_*
    // the device topology must be the reference's own: same blocks, same order, same h
    auto matches = [&](cup3d_grid_t *g) {
      const long nb = cup3d_grid_nblocks(g);
      if ((size_t)nb != I.size()) return false;
      std::vector<long long> tab(6 * (size_t)nb);
      std::vector<double> geom(4 * (size_t)nb);
      CUP3D_HIP_CALL(cup3d_grid_tables(g, tab.data(), geom.data()));
      for (long i = 0; i < nb; ++i)
        if (!(tab[6 * i] == I[i].level && tab[6 * i + 1] == I[i].Z && tab[6 * i + 5] == I[i].blockID_2 && geom[4 * i] == I[i].h)) return false;
      return true;
    };
    // (1) one level, every rank still owning its contiguous Hilbert range (GridMPI's initial partition, 2970-2986): face-slab halos
    int not_uniform = 0;
    for (const Info &b : I) not_uniform |= b.level != level;
    if (!not_uniform) {
      CUP3D_HIP_CALL(cup3d_grid_create_uniform(bpd, sim.levelMax, level, sim.maxextent, bc, rank, size, &grid));
      if (!matches(grid)) { not_uniform = 1; cup3d_grid_destroy(grid); grid = nullptr; }
    }
    if (size > 1) {  // the choice of transport is collective
      int any = 0;
      MPI_Allreduce(&not_uniform, &any, 1, MPI_INT, MPI_MAX, sim.comm);
      if (any && grid) { cup3d_grid_destroy(grid); grid = nullptr; }
      not_uniform = any;
    }
    if (not_uniform && size == 1) {
      // (2) multi-level mesh after MeshAdaptation on one rank: the leaves of m_vInfo; coarse/fine ghosts and flux correction run on
      // the device (amr.hip)
      std::vector<int32_t> lv(I.size());
      std::vector<int64_t> zs(I.size());
      for (size_t i = 0; i < I.size(); ++i) { lv[i] = I[i].level; zs[i] = I[i].Z; }
      CUP3D_HIP_CALL(cup3d_grid_create_mesh(bpd, sim.levelMax, sim.maxextent, bc, (long)I.size(), lv.data(), zs.data(), &grid));
    } else if (not_uniform) {
      // (3) multi-level mesh (or a one-level mesh the LoadBalancer has re-dealt) spread over ranks: every rank assembles the global
      // leaf list with its owners -- the information the reference keeps in Grid::Octree on every rank (815-855) -- and takes its view
      // of it: ghost blocks + the two exchange plans (whole ghost blocks before a stencil kernel, face fluxes after a corrected one;
      // what SynchronizerMPI_AMR::_Setup 1979-2286 and FluxCorrectionMPI::prepare 2680-2824 derive from the octree)
      int nloc = (int)I.size();
      std::vector<int> counts(size), displs(size);
      MPI_Allgather(&nloc, 1, MPI_INT, counts.data(), 1, MPI_INT, sim.comm);
      long total = 0;
      for (int r = 0; r < size; ++r) { displs[r] = (int)(2 * total); total += counts[r]; counts[r] *= 2; }
      std::vector<long long> mine(2 * (size_t)nloc), all(2 * (size_t)total);
      for (int i = 0; i < nloc; ++i) { mine[2 * i] = I[i].level; mine[2 * i + 1] = I[i].Z; }
      MPI_Allgatherv(mine.data(), 2 * nloc, MPI_LONG_LONG, all.data(), counts.data(), displs.data(), MPI_LONG_LONG, sim.comm);
      std::vector<int32_t> lv((size_t)total), own_in((size_t)total);
      std::vector<int64_t> zs((size_t)total);
      for (int r = 0, k = 0; r < size; ++r)
        for (int i = 0; i < counts[r] / 2; ++i, ++k) { lv[k] = (int32_t)all[2 * k]; zs[k] = all[2 * k + 1]; own_in[k] = r; }
      cup3d_grid_t *mesh = nullptr;
      CUP3D_HIP_CALL(cup3d_grid_create_mesh(bpd, sim.levelMax, sim.maxextent, bc, total, lv.data(), zs.data(), &mesh));
      // the mesh object orders the leaves by blockID_2: owners follow through (level, Z)
      std::vector<long long> tab(6 * (size_t)total);
      std::vector<double> geom(4 * (size_t)total);
      CUP3D_HIP_CALL(cup3d_grid_tables(mesh, tab.data(), geom.data()));
      std::map<std::pair<long long, long long>, int32_t> owner_of;
      for (long k = 0; k < total; ++k) owner_of[{lv[k], zs[k]}] = own_in[k];
      std::vector<int32_t> owner((size_t)total);
      for (long k = 0; k < total; ++k) owner[k] = owner_of[{tab[6 * k], tab[6 * k + 1]}];
      CUP3D_HIP_CALL(cup3d_grid_rank_view(mesh, owner.data(), rank, size, &grid));
      cup3d_grid_destroy(mesh);
    }
    if (!matches(grid)) {
      fprintf(stderr, "cup3d_hip: device topology differs from the host grid (blocks %ld vs %zu)\n", cup3d_grid_nblocks(grid), I.size());
      fflush(0);
      MPI_Abort(sim.comm, 1);
    }
    CUP3D_HIP_CALL(cup3d_sim_create(grid, &dsim));
    signature.resize(2 * I.size());
    for (size_t i = 0; i < I.size(); ++i) {
      signature[2 * i] = I[i].level;
      signature[2 * i + 1] = I[i].Z;
    }
  }
};

inline cup3d_poisson_params poisson_params(const SimulationData &sim) {
  cup3d_poisson_params p;
  cup3d_poisson_default_params(&p);
  p.tol = sim.PoissonErrorTol;
  p.tol_rel = sim.PoissonErrorTolRel;
  p.mean_constraint = sim.bMeanConstraint;
  if (const char *e = getenv("CUP3D_HIP_BLOCK_SOLVER")) p.block_solver = atoi(e);  // 0 block CG (default), 1 direct
  return p;
}

// AdvectionDiffusion::operator()(dt), main.cpp:9640-9728
class AdvectionDiffusionHIP : public Operator {
  std::shared_ptr<DeviceMirror> devp;
  DeviceMirror &dev;

public:
  AdvectionDiffusionHIP(SimulationData &s, std::shared_ptr<DeviceMirror> d) : Operator(s), devp(d), dev(*d) {}
  void operator()(const Real dt) override {
    (void)dt;  // KernelAdvectDiffuse reads sim.dt (9465), which advance() passes as dt
    dev.handle();  // (re)builds the mirror when the block list changed, which drops the clean flags
    if (!(dev.dev_current || (dev.across_steps && dev.host_vel_clean))) dev.upload(CUP3D_FIELD_VEL);
    dev.host_vel_clean = false;  // the device is about to move on
    const double uinf[3] = {sim.uinf[0], sim.uinf[1], sim.uinf[2]};
    CUP3D_HIP_CALL(cup3d_advect_diffuse(dev.handle(), sim.dt, sim.nu, uinf));
    if (dev.resident) {
      // nobody on the host reads vel / tmpV before the projection except the obstacle operators, which get the blocks they touch
      // from UpdateObstaclesHIP (install() checked the pipeline)
      dev.vel_on_device = true;
      return;
    }
    dev.download(CUP3D_FIELD_VEL);
    dev.download(CUP3D_FIELD_TMPV);
  }
};

// AdvectionDiffusionImplicit::operator()(dt), main.cpp:10030-10119 (-implicitDiffusion 1): one upwind-advection kernel, one
// 7-point kernel and three Helmholtz solves (DiffusionSolver) on the device.  sim.pres is only scratch there and is restored
// (10108-10117), so the host copy is simply left alone.  The reference's KernelAdvect updates vel in place while other blocks
// still read it; the device reads every tile from the velocity on entry (include/cup3d_hip.h), so on more than one block the two
// agree to O(dt) of the advective increment, not to round-off.
class AdvectionDiffusionImplicitHIP : public Operator {
  std::shared_ptr<DeviceMirror> devp;
  DeviceMirror &dev;

public:
  cup3d_poisson_result last[3];
  AdvectionDiffusionImplicitHIP(SimulationData &s, std::shared_ptr<DeviceMirror> d) : Operator(s), devp(d), dev(*d) {}
  void operator()(const Real dt) override {
    (void)dt;  // euler(sim.dt), 10119
    dev.handle();
    if (!(dev.across_steps && dev.host_vel_clean)) dev.upload(CUP3D_FIELD_VEL);
    dev.host_vel_clean = false;
    const double uinf[3] = {sim.uinf[0], sim.uinf[1], sim.uinf[2]};
    cup3d_poisson_params p;
    cup3d_poisson_default_params(&p);
    p.tol = sim.DiffusionErrorTol;
    p.tol_rel = sim.DiffusionErrorTolRel;
    CUP3D_HIP_CALL(cup3d_advect_diffuse_implicit(dev.handle(), sim.dt, sim.nu, uinf, &p, last));
    if (dev.resident) {
      dev.vel_on_device = true;
      return;
    }
    dev.download(CUP3D_FIELD_VEL);
    dev.download(CUP3D_FIELD_TMPV);
  }
};

// ExternalForcing::operator()(dt), main.cpp:10581-10596: on the device while the velocity is resident there, else the
// reference's own operator
class ExternalForcingHIP : public Operator {
  std::shared_ptr<DeviceMirror> devp;
  std::shared_ptr<Operator> cpu;

public:
  ExternalForcingHIP(SimulationData &s, std::shared_ptr<DeviceMirror> d, std::shared_ptr<Operator> original) : Operator(s), devp(d), cpu(original) {}
  void operator()(const Real dt) override {
    if (!devp->vel_on_device) { (*cpu)(dt); return; }
    const int dir = sim.BCy_flag == wall ? 1 : 2;  // 10582-10583
    CUP3D_HIP_CALL(cup3d_external_forcing(devp->handle(), sim.uMax_forced, sim.nu, sim.extents[dir], dt));
  }
};

// UpdateObstacles::operator()(dt), main.cpp:13812-13837, is the first host operator after the advection step that reads the
// velocity -- in the blocks an obstacle covers only.  While the velocity is resident on the device this wrapper fetches exactly
// those blocks, then runs the reference's own operator; Penalization (next in the pipeline) updates the same blocks on the host and
// PressureProjectionHIP sends them back.
class UpdateObstaclesHIP : public Operator {
  std::shared_ptr<DeviceMirror> devp;
  std::shared_ptr<Operator> cpu;

public:
  UpdateObstaclesHIP(SimulationData &s, std::shared_ptr<DeviceMirror> d, std::shared_ptr<Operator> original) : Operator(s), devp(d), cpu(original) {}
  void operator()(const Real dt) override {
    if (devp->vel_on_device && sim.obstacle_vector->nObstacles() > 0) devp->download_obstacle_blocks(CUP3D_FIELD_VEL);
    (*cpu)(dt);
  }
};

// PoissonSolverBase::solve(), main.cpp:8921-8928 (contract of PoissonSolverAMR::solve, 14363-14616:
// RHS in sim.lhs, initial guess and result in sim.pres)
class PoissonSolverHIP : public PoissonSolverBase {
  SimulationData &sim;
  std::shared_ptr<DeviceMirror> devp;
  DeviceMirror &dev;

public:
  cup3d_poisson_result last{};
  PoissonSolverHIP(SimulationData &s, std::shared_ptr<DeviceMirror> d) : sim(s), devp(d), dev(*d) {}
  void solve() override {
    dev.upload(CUP3D_FIELD_LHS);
    dev.upload(CUP3D_FIELD_PRES);
    const cup3d_poisson_params p = poisson_params(sim);
    CUP3D_HIP_CALL(cup3d_poisson_solve(dev.handle(), &p, &last));
    dev.download(CUP3D_FIELD_PRES);
    dev.host_pres_clean = true;
  }
};

// PressureProjection::operator()(dt), main.cpp:15061-15160
class PressureProjectionHIP : public Operator {
  std::shared_ptr<DeviceMirror> devp;
  DeviceMirror &dev;
  std::shared_ptr<PoissonSolverHIP> solver;
  bool had_obstacles = false;

public:
  cup3d_poisson_result last{};
  PressureProjectionHIP(SimulationData &s, std::shared_ptr<DeviceMirror> d)
      : Operator(s), devp(d), dev(*d), solver(std::make_shared<PoissonSolverHIP>(s, d)) {
    sim.pressureSolver = solver;  // as at main.cpp:15058-15059
  }
  void operator()(const Real dt) override {
    const bool obstacles = sim.obstacle_vector->nObstacles() > 0;
    if (obstacles) {
      // tmpV = 0; kernelUpdateTmpV(sim) (15066-15082) with the reference's own host code -- the ObstacleBlocks live there -- then chi
      // and tmpV (= udef) go up: the device right-hand side reads both (KernelPressureRHS 14849-14875)
      for (auto &info : sim.tmpVInfo()) ((VectorBlock *)info.block)->clear();
      kernelUpdateTmpV(sim);
      dev.upload(CUP3D_FIELD_TMPV);
      dev.upload(CUP3D_FIELD_CHI);
      had_obstacles = true;
    } else if (had_obstacles) {
      CUP3D_HIP_CALL(cup3d_sim_fill(dev.handle(), CUP3D_FIELD_CHI, 0.0));
      had_obstacles = false;
    }
    CUP3D_HIP_CALL(cup3d_sim_set_obstacles(dev.handle(), obstacles ? 1 : 0));  // obstacle_vector is replicated: the same on every rank
    if (!dev.vel_on_device) dev.upload(CUP3D_FIELD_VEL);
    else if (obstacles) dev.upload_obstacle_blocks(CUP3D_FIELD_VEL);  // what UpdateObstacles / Penalization changed on the host
    dev.vel_on_device = false;
    if (!(dev.dev_current || (dev.across_steps && dev.host_pres_clean))) dev.upload(CUP3D_FIELD_PRES);
    const cup3d_poisson_params p = poisson_params(sim);
    CUP3D_HIP_CALL(cup3d_pressure_project(dev.handle(), dt, sim.step, &p, &last));
    if (dev.device_led && !obstacles) {  // nothing comes down: the next step starts from the device copies (see DeviceMirror::device_led)
      dev.dev_current = true;
      dev.host_vel_clean = dev.host_pres_clean = false;
      return;
    }
    dev.dev_current = false;
    dev.download(CUP3D_FIELD_VEL);
    dev.download(CUP3D_FIELD_PRES);
    dev.host_vel_clean = dev.host_pres_clean = true;
    // gradP scratch, as the reference leaves it (15146).  Nothing reads it on the host before it is overwritten (AdvectionDiffusion
    // clears tmpV first, 9702-9706; adaptMesh's ComputeVorticity and the projection's own prologue fill it) -- the resident modes
    // leave the host copy stale (3.2 GB per step at 512^3)
    if (!dev.resident) dev.download(CUP3D_FIELD_TMPV);
  }
};

// One DeviceMirror per SimulationData, shared by every HIP-backed operator / solver built for it.
inline std::shared_ptr<DeviceMirror> mirror_of(SimulationData &sim) {
  static std::map<SimulationData *, std::weak_ptr<DeviceMirror>> reg;
  std::shared_ptr<DeviceMirror> m = reg[&sim].lock();
  if (!m) { m = std::make_shared<DeviceMirror>(sim); reg[&sim] = m; }
  return m;
}

// makePoissonSolver (main.cpp:14747-14758) with the slot the reference reserves for a GPU solver filled in: -poissonSolver
// "cuda_iterative" (the reference's own name for it, 14750) or "hip_iterative" gives the device solver behind PoissonSolverBase;
// every other value is handed to the reference's factory unchanged ("iterative" -> PoissonSolverAMR, unknown -> its error).
inline std::shared_ptr<PoissonSolverBase> makePoissonSolver(SimulationData &s) {
  if (s.poissonSolver == "cuda_iterative" || s.poissonSolver == "hip_iterative") return std::make_shared<PoissonSolverHIP>(s, mirror_of(s));
  return ::makePoissonSolver(s);
}

struct Installed {
  std::shared_ptr<DeviceMirror> mirror;
  std::shared_ptr<Operator> advdiff;  // AdvectionDiffusionHIP, or AdvectionDiffusionImplicitHIP with -implicitDiffusion 1
  std::shared_ptr<ExternalForcingHIP> forcing;
  std::shared_ptr<UpdateObstaclesHIP> update_obstacles;
  std::shared_ptr<PressureProjectionHIP> projection;
};

// Swap the hot-path operators of an initialised Simulation for the HIP-backed ones.
// resident (default: environment CUP3D_HIP_RESIDENT, 0 / 1 / 2): 1 = keep the velocity in HBM between AdvectionDiffusion and
// PressureProjection; 2 = also skip the uploads of vel and pres at the next step while the host copies are known to be current
// (DeviceMirror::across_steps: read its contract).  Allowed only if every operator the reference put between the two (setupOperators 15229-15246) is
// ExternalForcing (then run on the device too) or one of the obstacle operators: UpdateObstacles and Penalization touch the
// velocity only in the blocks an obstacle covers, so those blocks alone make the round trip (UpdateObstaclesHIP fetches them,
// PressureProjectionHIP sends them back); without obstacles both return immediately (13813-13814, 14327-14328).  With
// FixMassFlux in between, every operator round-trips in full as before.
inline Installed install(SimulationData &sim, int resident = -1) {
  Installed r;
  r.mirror = mirror_of(sim);
  if (resident < 0) {
    const char *e = getenv("CUP3D_HIP_RESIDENT");
    resident = e ? atoi(e) : 0;
  }
  bool between = false, safe = true;
  for (auto &op : sim.pipeline) {
    if (std::dynamic_pointer_cast<AdvectionDiffusion>(op) || std::dynamic_pointer_cast<AdvectionDiffusionImplicit>(op)) between = true;
    else if (std::dynamic_pointer_cast<PressureProjection>(op)) between = false;
    else if (between && !(std::dynamic_pointer_cast<ExternalForcing>(op) || std::dynamic_pointer_cast<UpdateObstacles>(op) ||
                          std::dynamic_pointer_cast<Penalization>(op)))
      safe = false;
  }
  r.mirror->resident = resident != 0 && safe;
  r.mirror->across_steps = resident >= 2 && safe;
  r.mirror->device_led = resident >= 3 && safe;
  for (auto &op : sim.pipeline) {
    if (std::dynamic_pointer_cast<AdvectionDiffusion>(op)) {
      r.advdiff = std::make_shared<AdvectionDiffusionHIP>(sim, r.mirror);
      op = r.advdiff;
    } else if (std::dynamic_pointer_cast<AdvectionDiffusionImplicit>(op)) {  // setupOperators 15231-15232
      r.advdiff = std::make_shared<AdvectionDiffusionImplicitHIP>(sim, r.mirror);
      op = r.advdiff;
    } else if (r.mirror->resident && std::dynamic_pointer_cast<ExternalForcing>(op)) {
      r.forcing = std::make_shared<ExternalForcingHIP>(sim, r.mirror, op);
      op = r.forcing;
    } else if (r.mirror->resident && std::dynamic_pointer_cast<UpdateObstacles>(op)) {
      r.update_obstacles = std::make_shared<UpdateObstaclesHIP>(sim, r.mirror, op);
      op = r.update_obstacles;
    } else if (std::dynamic_pointer_cast<PressureProjection>(op)) {
      r.projection = std::make_shared<PressureProjectionHIP>(sim, r.mirror);
      op = r.projection;
    }
  }
  return r;
}

// Device-led replacement for simulation.calcMaxTimestep() (main.cpp:15254-15305; DeviceMirror::device_led).  Nothing of the
// reference's function is restated here: the maximum comes from cup3d_max_u (findMaxU incl. its MAX all-reduce, 8603-8623, exact) and
// the time-step rule and the pressure-extrapolation coefficients from the library (cup3d_calc_max_timestep2, pinned against the
// reference by tests/test_host_indexing.py); this function only moves the results into SimulationData.
inline Real calcMaxTimestep(Simulation &S, DeviceMirror &dev) {
  SimulationData &sim = S.sim;
  // host copy current (first step, after an adaptation, obstacle runs), or a fixed-dt run (CFL <= 0: nothing for the library to choose)
  if (!dev.dev_current) return S.calcMaxTimestep();
  if (!(sim.CFL > 0)) { dev.sync_host(); return S.calcMaxTimestep(); }
  auto stop = [&](const char *why, double value) {
    if (sim.rank == 0) fprintf(stderr, "cup3d_hip::calcMaxTimestep: %s (%g) at step %d -- stopping the run\n", why, value, sim.step);
    MPI_Abort(sim.comm, 1);
  };
  const double uinf[3] = {sim.uinf[0], sim.uinf[1], sim.uinf[2]};
  double umax = 0, coef[3] = {sim.coefU[0], sim.coefU[1], sim.coefU[2]};
  CUP3D_HIP_CALL(cup3d_max_u(dev.handle(), uinf, &umax));
  if (umax > sim.uMax_allowed) stop("velocity maximum above -uMax_allowed", umax);
  const double previous = sim.dt;
  const double dt = cup3d_calc_max_timestep2(sim.hmin, umax, sim.nu, sim.CFL, sim.step, sim.rampup, previous, coef, sim.implicitDiffusion ? 1 : 0);
  if (!(dt > 0)) stop("non-positive time step", dt);
  sim.uMax_measured = umax;
  sim.dt_old = previous;
  sim.dt = dt;
  for (int i = 0; i < 3; ++i) sim.coefU[i] = coef[i];
  if (sim.DLM > 0) sim.lambda = sim.DLM / dt;
  if (sim.rank == 0) printf("main.cpp: step: %d, time: %f\n", sim.step, sim.time);  // the progress line the reference's loop prints from here
  return sim.dt;
}
// ... and the companion of Simulation::advance (15306-15326) in that mode: adaptMesh (15314) and dump (15307-15313) read the host fields
inline bool advance(Simulation &S, DeviceMirror &dev, const Real dt) {
  SimulationData &sim = S.sim;
  if ((sim.dumpTime > 0 && sim.time >= sim.nextDumpTime) || sim.step % 20 == 0 || sim.step < 10) dev.sync_host();
  return S.advance(dt);
}

}  // namespace cup3d_hip
