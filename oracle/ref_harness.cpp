/* TEST INFRASTRUCTURE — never linked into or called by the product path.
 *
 * Builds the UNMODIFIED reference translation unit (by #include from where it
 * lies, -DCUP3D_REFERENCE_MAIN="\"/root/reference/main.cpp\"") into
 * oracle/_ref/ref_tool, a small command interpreter that drives the
 * reference's own objects so that
 *   (1) the C restatement in oracle/cup3d_oracle.c can be pinned against the
 *       reference itself, and
 *   (2) golden vectors under tests/golden/ can be generated
 *       (tests/golden/make_golden.py), and
 *   (3) bench.py can time the reference's CPU operators ("cpu_baseline",
 *       kind "reference") on the GPU box's host cores.
 *
 * _ref/ref_tool_mpi is the same file linked against a real MPI (conda MPICH in the build container) instead of the single-rank
 * stub, for pinning what only exists on several ranks: block ownership (GridMPI partition 2970-2986) and, after adaptMesh,
 * the LoadBalancer's block moves (4660-5022).  Run under mpiexec -n N; every output file gets the suffix .r<rank>.
 *
 * Usage:  ref_tool <script> -- <reference command line>
 * Script lines (whitespace separated):
 *   tables <file>          int64[nb][6] = level,Z,ix,iy,iz,blockID_2 then
 *                          double[nb][4] = h,origin[3]   (m_vInfo order)
 *   sfc <file>             forward/Encode/neighbour tables for every level
 *   loadg <field> <file>   fill a field from a global x-fastest array
 *                          [NZ][NY][NX][ncomp] (uniform single-level grids)
 *   zero <field>           clear a field (chi is uninitialised memory without obstacles)
 *   dump <field> <file>    block-order dump [nb][8][8][8][ncomp] (the
 *                          reference's own memory layout per block)
 *   set step|dt|time|nu|uinfx|uinfy|uinfz|mean <value>
 *   op advdiff <dt> | lhs | precond | solve | project <dt> | maxu |
 *      steps <n> | forcing <dt> | rhs | divp | gradp   (the last three use `set dt`)
 *      midstep <dt>        pipeline entries from AdvectionDiffusion through PressureProjection (works with `obstacle`)
 *      advdiff_implicit <dt> | advect | diffrhs | diffprecond | difflhs <dir> | diffsolve <dir>   (implicit diffusion;
 *      `set dt`, `set nu`, `set difftol`, `set difftolrel`; value = number of 6-double reductions = solver iterations)
 *   hip on                 (ref_tool_hip only) route advdiff/project/steps through the HIP drop-in
 *   hipsolver <key>        (ref_tool_hip only) sim.pressureSolver = cup3d_hip::makePoissonSolver(sim) with sim.poissonSolver = key
 *   lab <field> <s> <e> <tensorial> <file>   ghosted tiles of every block (BlockLab::load)
 *   loadb <field> <file>   block-order load (multi-level meshes)
 *   amrtol <rt> <ct> | adapt | tagvel <rt> <ct> <file>   mesh-adaptation hooks (refine/compress everything)
 *   timeops                wrap every entry of sim.pipeline in a wall-clock timer (after `hip ...`, if any); `op steps` then prints
 *                          `REF optime name=<class> calls=<n> seconds=<t>` per operator, plus calcMaxTimestep / adaptMesh (= the rest
 *                          of the step) -- the Amdahl split of a step with and without the drop-in (scripts/configs4_measure.py)
 *   rep <n>                repeat every following `op` n times when timing
 * Every `op` prints one line `REF <op> seconds=<t> iters=<k> value=<v>`.
 */
#include <chrono>
#include <cxxabi.h>
/* every header the reference TU includes is pulled in FIRST, so that the access-specifier
   override below (needed to reach MeshAdaptation's tolerances, main.cpp:5037-5038) touches the
   reference's own classes only, not the standard library */
#include <algorithm>
#include <array>
#include <cassert>
#include <cctype>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <gsl/gsl_bspline.h>
#include <gsl/gsl_linalg.h>
#include <gsl/gsl_statistics.h>
#include <iomanip>
#include <ios>
#include <iosfwd>
#include <iostream>
#include <limits>
#include <list>
#include <locale>
#include <map>
#include <math.h>
#include <memory>
#include <numeric>
#include <omp.h>
#include <queue>
#include <random>
#include <set>
#include <sstream>
#include <stack>
#include <stdio.h>
#include <string>
#include <sys/stat.h>
#include <sys/time.h>
#include <thread>
#include <type_traits>
#include <unistd.h>
#include <unordered_map>
#include <utility>
#include <vector>
static long cup3d_stub_iallreduce6 = 0;
static long cup3d_stub_iallreduce7 = 0; /* one 7-double Iallreduce per BiCGSTAB iteration (main.cpp:14546) */
#define CUP3D_STUB_COUNT_IALLREDUCE(n)                                         \
  do {                                                                         \
    if ((n) == 7) cup3d_stub_iallreduce7++;                                    \
    if ((n) == 6) cup3d_stub_iallreduce6++; /* DiffusionSolver, main.cpp:7083 */ \
  } while (0)
#include <mpi.h>
#define main cup3d_reference_main
#define protected public
#define private public
#include CUP3D_REFERENCE_MAIN
#undef private
#undef protected
#undef main
#ifdef CUP3D_WITH_HIP
/* the drop-in under test: HIP-backed operators behind the reference's own plugin surface */
#include "../cup3d_amd/host/cup3d_hip_operators.h"
#endif

namespace {
struct Field {
  int ncomp;
  std::vector<Info> *infos;
};
Field field_of(SimulationData &s, const std::string &name) {
  if (name == "vel") return {3, &s.velInfo()};
  if (name == "tmpV") return {3, &s.tmpVInfo()};
  if (name == "pres") return {1, &s.presInfo()};
  if (name == "lhs") return {1, &s.lhsInfo()};
  if (name == "chi") return {1, &s.chiInfo()};
  fprintf(stderr, "ref_tool: unknown field %s\n", name.c_str());
  exit(2);
}
double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
/* `timeops`: a pipeline entry that times the operator it wraps (the HIP operators enqueue work and return: the device is drained
   before the clock is read, so that the time lands on the operator that caused it) */
struct TimedOp : public Operator {
  std::shared_ptr<Operator> inner;
  std::string name;
  double seconds = 0;
  long calls = 0;
  bool drain_device;
  TimedOp(SimulationData &s, std::shared_ptr<Operator> in, bool drain) : Operator(s), inner(in), drain_device(drain) {
    Operator &ref = *in;
    int status = 0;
    char *d = abi::__cxa_demangle(typeid(ref).name(), nullptr, nullptr, &status);
    name = (status == 0 && d) ? d : typeid(ref).name();
    free(d);
    for (char &c : name) if (c == ' ') c = '_';
  }
  void operator()(Real dt) override {
    const double t0 = now();
    (*inner)(dt);
#ifdef CUP3D_WITH_HIP
    if (drain_device) cup3d_device_synchronize();
#endif
    seconds += now() - t0;
    calls++;
  }
};
void write_file(const std::string &path0, const void *p, size_t bytes) {
  /* several ranks (ref_tool_mpi, linked against a real MPI): one file per rank, <path>.r<rank> */
  const std::string path = ::sim.size > 1 ? path0 + ".r" + std::to_string(::sim.rank) : path0;
  FILE *f = fopen(path.c_str(), "wb");
  if (!f || fwrite(p, 1, bytes, f) != bytes) { perror(path.c_str()); exit(2); }
  fclose(f);
}
std::vector<char> read_file(const std::string &path) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) { perror(path.c_str()); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> b(n);
  if (fread(b.data(), 1, n, f) != (size_t)n) { perror(path.c_str()); exit(2); }
  fclose(f);
  return b;
}
} // namespace

int main(int argc, char **argv) {
  int split = -1;
  for (int i = 1; i < argc; i++)
    if (std::string(argv[i]) == "--") { split = i; break; }
  if (argc < 3 || split != 2) {
    fprintf(stderr, "usage: ref_tool <script> -- <reference args>\n");
    return 2;
  }
  setvbuf(stdout, nullptr, _IOLBF, 0); /* the reference's progress line per step reaches a pipe at once: tests tell slow from hung by it */
  int provided;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_FUNNELED, &provided);
  MPI_Comm_rank(MPI_COMM_WORLD, &::sim.rank);
  MPI_Comm_size(MPI_COMM_WORLD, &::sim.size);
#ifdef CUP3D_WITH_HIP
  /* sign of life between two of the reference's per-step lines (a step is hundreds of BiCGSTAB iterations; with eight ranks and a test
     process time-slicing ONE device it can take minutes): every 5 s, if the library's exchange / all-reduce counters have moved, one
     line.  A launch whose counters stand still prints nothing and is hung; one that prints is slow (tests/test_gpu_00_dropin_mpi.py) */
  {
    const int my_rank = ::sim.rank;
    std::thread([my_rank] {
      long seen = -1;
      const auto t0 = std::chrono::steady_clock::now();
      auto moved = t0;
      char name[64];
      snprintf(name, sizeof name, "alive.r%d", my_rank);
      for (;;) {
        std::this_thread::sleep_for(std::chrono::seconds(5));
        cup3d_run_stats st;
        if (cup3d_stats_read(&st) != 0) continue;
        const long now = st.halo_exchanges + st.allreduces + st.host_waits;
        const auto t = std::chrono::steady_clock::now();
        if (now != seen) moved = t;
        if (my_rank == 0 && now != seen && seen != -1) printf("REF alive exchanges=%ld allreduces=%ld iterations=%ld\n", st.halo_exchanges, st.allreduces, st.solver_iterations);
        /* every rank's last word, for the test to quote when a launch fails or is cut off: where did each rank stand, and since when */
        if (FILE *f = fopen(name, "w")) {
          fprintf(f, "rank %d after %.0f s: exchanges=%ld allreduces=%ld iterations=%ld host_waits=%ld, counters last moved %.0f s ago\n", my_rank,
                  std::chrono::duration<double>(t - t0).count(), st.halo_exchanges, st.allreduces, st.solver_iterations, st.host_waits,
                  std::chrono::duration<double>(t - moved).count());
          fclose(f);
        }
        seen = now;
      }
    }).detach();
  }
#endif
  int rargc = argc - split;
  char **rargv = argv + split; /* rargv[0] = "--" plays the role of argv[0] */
  Simulation *S = new Simulation(rargc, rargv, MPI_COMM_WORLD);
  S->init();
  SimulationData &sd = S->sim;
  AdvectionDiffusion advdiff(sd);
  ComputeLHS lhsop(sd);
  std::shared_ptr<PressureProjection> proj;
  for (auto &op : sd.pipeline)
    if (auto p = std::dynamic_pointer_cast<PressureProjection>(op)) proj = p;
  std::shared_ptr<Operator> hip_adv, hip_proj;
#ifdef CUP3D_WITH_HIP
  std::shared_ptr<cup3d_hip::DeviceMirror> hip_mirror;
#define HIP_INVALIDATE() do { if (hip_mirror) hip_mirror->invalidate(); } while (0)
#else
#define HIP_INVALIDATE() do { } while (0)
#endif
  std::ifstream script(argv[1]);
  std::string cmd;
  int rep = 1;
  while (script >> cmd) {
    if (cmd == "tables") {
      std::string path; script >> path;
      auto &I = sd.velInfo();
      std::vector<long long> t(I.size() * 6);
      std::vector<double> g(I.size() * 4);
      for (size_t i = 0; i < I.size(); i++) {
        t[6 * i + 0] = I[i].level; t[6 * i + 1] = I[i].Z;
        t[6 * i + 2] = I[i].index[0]; t[6 * i + 3] = I[i].index[1]; t[6 * i + 4] = I[i].index[2];
        t[6 * i + 5] = I[i].blockID_2;
        g[4 * i] = I[i].h; g[4 * i + 1] = I[i].origin[0]; g[4 * i + 2] = I[i].origin[1]; g[4 * i + 3] = I[i].origin[2];
      }
      std::vector<char> out(t.size() * 8 + g.size() * 8);
      memcpy(out.data(), t.data(), t.size() * 8);
      memcpy(out.data() + t.size() * 8, g.data(), g.size() * 8);
      write_file(path, out.data(), out.size());
    } else if (cmd == "sfc") {
      /* per level l < levelMax, per (k,j,i) x-fastest: forward, Encode, Zparent,
         27 Znei, 8 Zchild  (int64 each) — Info::setup main.cpp:384-420 */
      std::string path; script >> path;
      std::vector<long long> out;
      for (int l = 0; l < sd.levelMax; l++) {
        const int nx = sd.bpdx << l, ny = sd.bpdy << l, nz = sd.bpdz << l;
        for (int k = 0; k < nz; k++)
          for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
              const long long Z = Info::forward(l, i, j, k);
              Info &inf = sd.vel->getInfoAll(l, Z);
              int ii, jj, kk;
              Info::inverse(Z, l, ii, jj, kk);
              if (ii != i || jj != j || kk != k) { fprintf(stderr, "sfc inverse mismatch\n"); exit(3); }
              out.push_back(Z);
              out.push_back(inf.blockID_2);
              out.push_back(inf.Zparent);
              for (int a = 0; a < 27; a++) out.push_back((&inf.Znei[0][0][0])[a]);
              for (int a = 0; a < 8; a++) out.push_back((&inf.Zchild[0][0][0])[a]);
            }
      }
      write_file(path, out.data(), out.size() * 8);
    } else if (cmd == "loadg") {
      std::string fname, path; script >> fname >> path;
      HIP_INVALIDATE();  /* the script writes a host field behind the operators' back */
      Field F = field_of(sd, fname);
      auto buf = read_file(path);
      const double *g = (const double *)buf.data();
      const int lvl = (*F.infos)[0].level;
      const long NXc = (long)(sd.bpdx << lvl) * 8, NYc = (long)(sd.bpdy << lvl) * 8, NZc = (long)(sd.bpdz << lvl) * 8;
      if (buf.size() != (size_t)(NXc * NYc * NZc * F.ncomp * 8)) { fprintf(stderr, "loadg %s: size mismatch\n", fname.c_str()); exit(2); }
      for (auto &inf : *F.infos) {
        if (inf.level != lvl) { fprintf(stderr, "loadg needs a uniform grid\n"); exit(2); }
        double *b = (double *)inf.block;
        for (int z = 0; z < 8; z++)
          for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++)
              for (int c = 0; c < F.ncomp; c++) {
                const long gx = inf.index[0] * 8 + x, gy = inf.index[1] * 8 + y, gz = inf.index[2] * 8 + z;
                b[((z * 8 + y) * 8 + x) * F.ncomp + c] = g[((gz * NYc + gy) * NXc + gx) * F.ncomp + c];
              }
      }
    } else if (cmd == "zero") {
      /* the reference never initialises chi when there are no obstacles (posix_memalign'd
         blocks, main.cpp:877-884): scripts zero it so that runs are deterministic */
      std::string fname; script >> fname;
      Field F = field_of(sd, fname);
      for (auto &inf : *F.infos) memset(inf.block, 0, 512 * F.ncomp * 8);
      HIP_INVALIDATE();  /* the script writes a host field behind the operators' back */
    } else if (cmd == "dump") {
      std::string fname, path; script >> fname >> path;
      Field F = field_of(sd, fname);
      const size_t per = 512 * F.ncomp;
      std::vector<double> out(F.infos->size() * per);
      for (size_t i = 0; i < F.infos->size(); i++)
        memcpy(&out[i * per], (*F.infos)[i].block, per * 8);
      write_file(path, out.data(), out.size() * 8);
    } else if (cmd == "set") {
      std::string k; double v; script >> k >> v;
      if (k == "step") sd.step = (int)v;
      else if (k == "dt") sd.dt = v;
      else if (k == "time") sd.time = v;
      else if (k == "nu") sd.nu = v;
      else if (k == "uinfx") sd.uinf[0] = v;
      else if (k == "uinfy") sd.uinf[1] = v;
      else if (k == "uinfz") sd.uinf[2] = v;
      else if (k == "mean") sd.bMeanConstraint = (int)v;
      else if (k == "lambda") sd.lambda = v;
      else if (k == "implicit") sd.bImplicitPenalization = v != 0;
      else if (k == "rtol") sd.Rtol = v;
      else if (k == "ctol") sd.Ctol = v;
      else if (k == "lmaxvort") sd.levelMaxVorticity = (int)v;
      else if (k == "difftol") sd.DiffusionErrorTol = v;
      else if (k == "difftolrel") sd.DiffusionErrorTolRel = v;
      else { fprintf(stderr, "ref_tool: unknown set key %s\n", k.c_str()); exit(2); }
    } else if (cmd == "hip") {
      /* `hip on` | `hip resident` | `hip resident2`: swap AdvectionDiffusion / PressureProjection in sim.pipeline for the HIP-backed
         operators (cup3d_hip::install); every later op/steps command runs through them */
      std::string v; script >> v;
#ifdef CUP3D_WITH_HIP
      static cup3d_hip::Installed inst;
      /* `hip resident`: vel stays in HBM between the two operators; `hip resident2`: also across steps (DeviceMirror::across_steps) */
      /* `hip resident3`: device-led -- vel / pres never come down between steps; `op steps` then uses cup3d_hip::calcMaxTimestep / advance */
      inst = cup3d_hip::install(sd, v == "resident3" ? 3 : (v == "resident2" ? 2 : (v == "resident" ? 1 : -1)));
      hip_adv = inst.advdiff; hip_proj = inst.projection;
      hip_mirror = inst.mirror;
#else
      fprintf(stderr, "ref_tool: built without CUP3D_WITH_HIP\n"); exit(2);
#endif
    } else if (cmd == "timeops") {
      bool drain = false;
#ifdef CUP3D_WITH_HIP
      drain = (bool)hip_mirror;
#endif
      for (auto &op : sd.pipeline)
        if (!std::dynamic_pointer_cast<TimedOp>(op)) op = std::make_shared<TimedOp>(sd, op, drain);
    } else if (cmd == "hipsolver") {
      /* `hipsolver <key>`: sim.poissonSolver = key; sim.pressureSolver = cup3d_hip::makePoissonSolver(sim) -- the reference's factory
         (main.cpp:14747-14758) with the GPU slot it reserves ("cuda_iterative") filled in; `op solve` then runs through it */
      std::string key; script >> key;
#ifdef CUP3D_WITH_HIP
      sd.poissonSolver = key;
      sd.pressureSolver = cup3d_hip::makePoissonSolver(sd);
#else
      fprintf(stderr, "ref_tool: built without CUP3D_WITH_HIP\n"); exit(2);
#endif
    } else if (cmd == "lab") {
      /* `lab <vel|tmpV|pres|lhs|chi> <s> <e> <tensorial> <file>`: the ghosted tile BlockLab::load assembles
         (main.cpp:3623-3787) for stencil [s,e)^3 of every block, [nb][L][L][L][nc], L = 8 + e - s - 1.
         Cells the reference does not fill keep whatever the previous block left there. */
      std::string fname, path; int ss, ee, tens; script >> fname >> ss >> ee >> tens >> path;
      Field F = field_of(sd, fname);
      const int L = 8 + ee - ss - 1;
      std::vector<double> out;
      auto dump_lab = [&](auto &lab, auto *grid, const std::vector<int> &comps) {
        StencilInfo st(ss, ss, ss, ee, ee, ee, tens != 0, comps);
        lab.prepare(*grid, st);
        for (auto &inf : *F.infos) {
          lab.load(inf, 0);
          for (int z = ss; z < 8 + ee - 1; z++)
            for (int y = ss; y < 8 + ee - 1; y++)
              for (int x = ss; x < 8 + ee - 1; x++)
                for (int c = 0; c < F.ncomp; c++) out.push_back(lab(x, y, z).member(c));
        }
      };
      if (F.ncomp == 3) { VectorLab lab; dump_lab(lab, fname == "vel" ? sd.vel : sd.tmpV, {0, 1, 2}); }
      else { ScalarLab lab; dump_lab(lab, fname == "pres" ? sd.pres : (fname == "lhs" ? sd.lhs : sd.chi), {0}); }
      (void)L;
      write_file(path, out.data(), out.size() * 8);
    } else if (cmd == "loadb") {
      /* `loadb <field> <file>`: block-order load [nb][8][8][8][nc] (works on multi-level meshes) */
      std::string fname, path; script >> fname >> path;
      HIP_INVALIDATE();  /* the script writes a host field behind the operators' back */
      Field F = field_of(sd, fname);
      auto buf = read_file(path);
      const size_t per = 512 * F.ncomp;
      /* several ranks: the file holds all blocks in rank-major (= global blockID_2) order; this rank's slice starts after the
         blocks of the lower ranks */
      std::vector<long long> counts(::sim.size, 0);
      long long mine = (long long)F.infos->size();
      MPI_Allgather(&mine, 1, MPI_LONG_LONG, counts.data(), 1, MPI_LONG_LONG, MPI_COMM_WORLD);
      size_t first = 0, total = 0;
      for (int r = 0; r < ::sim.size; r++) { if (r < ::sim.rank) first += counts[r]; total += counts[r]; }
      if (buf.size() != total * per * 8) { fprintf(stderr, "loadb %s: size mismatch\n", fname.c_str()); exit(2); }
      for (size_t i = 0; i < F.infos->size(); i++) memcpy((*F.infos)[i].block, buf.data() + (first + i) * per * 8, per * 8);
    } else if (cmd == "obstacle") {
      /* `obstacle <file>`: add ONE synthetic obstacle (the reference's own obstacles are fish whose geometry needs GSL): a
         plain Obstacle whose ObstacleBlocks (7256-7263: chi[8][8][8], udef[8][8][8][3]) are read from the file together with its
         rigid motion.  File: int64 n, int64 blockID[n], double chi[n][512], double udef[n][512][3], double cm[3], vel[3], omega[3].
         Everything downstream (KernelPenalization 13841-13912, kernelUpdateTmpV 14948-14979, KernelPressureRHS) is the
         reference's own code. */
      std::string path; script >> path;
      auto buf = read_file(path);
      const char *q = buf.data();
      long long n; memcpy(&n, q, 8); q += 8;
      std::vector<long long> ids(n); memcpy(ids.data(), q, 8 * n); q += 8 * n;
      auto ob = std::make_shared<Obstacle>(sd);
      ob->obstacleBlocks.assign(sd.chiInfo().size(), nullptr);
      for (long long i = 0; i < n; i++) {
        ObstacleBlock *b = new ObstacleBlock();
        memcpy(&b->chi[0][0][0], q + (size_t)i * 512 * 8, 512 * 8);
        ob->obstacleBlocks[ids[i]] = b;
      }
      q += (size_t)n * 512 * 8;
      for (long long i = 0; i < n; i++) memcpy(&ob->obstacleBlocks[ids[i]]->udef[0][0][0][0], q + (size_t)i * 1536 * 8, 1536 * 8);
      q += (size_t)n * 1536 * 8;
      double rm[9]; memcpy(rm, q, 72);
      for (int d = 0; d < 3; d++) { ob->centerOfMass[d] = rm[d]; ob->transVel[d] = rm[3 + d]; ob->angVel[d] = rm[6 + d]; }
      sd.obstacle_vector->addObstacle(ob);
    } else if (cmd == "forces") {
      /* `forces <file>`: force[3], torque[3] of every obstacle as kernelFinalizePenalizationForce left them (13913-13938) */
      std::string path; script >> path;
      std::vector<double> out;
      for (const auto &o : sd.obstacle_vector->getObstacleVector()) {
        for (int d = 0; d < 3; d++) out.push_back(o->force[d]);
        for (int d = 0; d < 3; d++) out.push_back(o->torque[d]);
      }
      write_file(path, out.data(), out.size() * 8);
    } else if (cmd == "amrtol") {
      /* tolerance_for_refinement / tolerance_for_compression of the five MeshAdaptation objects
         (main.cpp:5037-5038); `amrtol -1 -2` makes every block refine, `amrtol 1e300 1e299` compress */
      double rt, ct; script >> rt >> ct;
      sd.chi_amr->tolerance_for_refinement = sd.lhs_amr->tolerance_for_refinement = sd.pres_amr->tolerance_for_refinement = rt;
      sd.vel_amr->tolerance_for_refinement = sd.tmpV_amr->tolerance_for_refinement = rt;
      sd.chi_amr->tolerance_for_compression = sd.lhs_amr->tolerance_for_compression = sd.pres_amr->tolerance_for_compression = ct;
      sd.vel_amr->tolerance_for_compression = sd.tmpV_amr->tolerance_for_compression = ct;
    } else if (cmd == "adapt") {
      S->adaptMesh(); /* main.cpp:15179-15194 */
    } else if (cmd == "tagvel") {
      /* MeshAdaptation::TagLoadedBlock on every vel block (main.cpp:5566-5582) with given tolerances -> int8 file */
      double rt, ct; std::string path; script >> rt >> ct >> path;
      const double r0 = sd.vel_amr->tolerance_for_refinement, c0 = sd.vel_amr->tolerance_for_compression;
      sd.vel_amr->tolerance_for_refinement = rt; sd.vel_amr->tolerance_for_compression = ct;
      std::vector<signed char> st;
      for (auto &inf : sd.velInfo()) st.push_back((signed char)sd.vel_amr->TagLoadedBlock(sd.vel->getInfoAll(inf.level, inf.Z)));
      sd.vel_amr->tolerance_for_refinement = r0; sd.vel_amr->tolerance_for_compression = c0;
      write_file(path, st.data(), st.size());
    } else if (cmd == "tagtmp") {
      /* what tmpV_amr->Tag() decides per block (TagBlocksVector 5196-5226: TagLoadedBlock + the level clamps), before
         ValidStates balances the result -> int8 file */
      double rt, ct; std::string path; script >> rt >> ct >> path;
      const double r0 = sd.tmpV_amr->tolerance_for_refinement, c0 = sd.tmpV_amr->tolerance_for_compression;
      sd.tmpV_amr->tolerance_for_refinement = rt; sd.tmpV_amr->tolerance_for_compression = ct;
      std::vector<signed char> st;
      for (auto &inf : sd.tmpVInfo()) {
        int s = (int)sd.tmpV_amr->TagLoadedBlock(sd.tmpV->getInfoAll(inf.level, inf.Z));
        if (s == (int)Refine && inf.level == sd.levelMax - 1) s = (int)Leave;
        if (s == (int)Compress && inf.level == 0) s = (int)Leave;
        st.push_back((signed char)(s == (int)Refine ? 1 : (s == (int)Compress ? -1 : 0)));
      }
      sd.tmpV_amr->tolerance_for_refinement = r0; sd.tmpV_amr->tolerance_for_compression = c0;
      write_file(path, st.data(), st.size());
    } else if (cmd == "rep") {
      script >> rep;
    } else if (cmd == "op") {
      std::string op; script >> op;
      double arg = 0;
      if (op == "advdiff" || op == "project" || op == "steps" || op == "forcing" || op == "penalize" || op == "advdiff_implicit" || op == "midstep" ||
          op == "difflhs" || op == "diffsolve") script >> arg;
      for (int r = 0; r < rep; r++) {
        cup3d_stub_iallreduce7 = 0;
        cup3d_stub_iallreduce6 = 0;
        double value = 0;
        const double t0 = now();
        if (op == "advdiff") { sd.dt = arg; if (hip_adv && !sd.implicitDiffusion) (*hip_adv)(arg); else advdiff(arg); }
        else if (op == "lhs") lhsop(0);
        else if (op == "penalize") { /* Penalization::operator() without the collision model, main.cpp:14330-14340 */
          const std::vector<Info> &ci = sd.chiInfo(), &vi = sd.velInfo();
          KernelPenalization K(arg, sd.lambda, sd.bImplicitPenalization, sd.obstacle_vector);
          for (size_t i = 0; i < ci.size(); ++i) K(vi[i], ci[i]);
          kernelFinalizePenalizationForce(sd);
        }
        else if (op == "vorticity") { ComputeVorticity w(sd); w(0); } /* first half of adaptMesh, main.cpp:15180-15181 */
        else if (op == "gradchi") compute<ScalarLab>(GradChiOnTmp(sd), sd.chi);                      /* main.cpp:15182 */
        else if (op == "precond") {
#pragma omp parallel
          { poisson_kernels::getZImplParallel(sd.presInfo()); }
        }
        else if (op == "solve") sd.pressureSolver->solve();
        else if (op == "project") { sd.dt = arg; if (hip_proj) (*hip_proj)(arg); else (*proj)(arg); }
        else if (op == "rhs") { /* the call at main.cpp:15083-15085 */
          KernelPressureRHS K(sd, sd.dt);
          compute<KernelPressureRHS, VectorGrid, VectorLab, VectorGrid, VectorLab, ScalarGrid>(K, *sd.vel, *sd.tmpV, true, sd.lhs);
        }
        else if (op == "divp") compute<ScalarLab>(KernelDivPressure(sd), sd.pres, sd.tmpV);   /* main.cpp:15088 */
        else if (op == "gradp") compute<ScalarLab>(KernelGradP(sd, sd.dt), sd.pres, sd.tmpV); /* main.cpp:15146 */
        /* implicit diffusion (AdvectionDiffusionImplicit::euler, main.cpp:10030-10118) and its parts; all use `set dt`, `set nu` */
        else if (op == "advdiff_implicit") { /* with `hip on` and -implicitDiffusion 1 on the command line: the drop-in */
          sd.dt = arg;
          if (hip_adv && sd.implicitDiffusion) (*hip_adv)(arg);
          else { AdvectionDiffusionImplicit a(sd); a(arg); }
        }
        else if (op == "advect") compute<VectorLab>(KernelAdvect(sd, sd.dt), sd.vel, sd.tmpV);       /* 10038 */
        else if (op == "diffrhs") compute<VectorLab>(KernelDiffusionRHS(sd), sd.vel, sd.tmpV);        /* 10057 */
        else if (op == "diffprecond") {                                                               /* 6823-6824 */
#pragma omp parallel
          { diffusion_kernels::getZImplParallel(sd.presInfo(), sd.nu, sd.dt); }
        }
        else if (op == "difflhs" || op == "diffsolve") { /* <direction>: DiffusionSolver::_lhs (pres -> lhs) / ::solve */
          DiffusionSolver ds(sd);
          ds.mydirection = (int)arg;
          ds.dt = sd.dt;
          if (op == "diffsolve") ds.solve();
          else {
            const size_t nb = sd.presInfo().size();
            std::vector<Real> in(nb * 512), out(nb * 512);
            for (size_t i = 0; i < nb; i++) memcpy(&in[i * 512], sd.presInfo()[i].block, 4096);
            ds._lhs(in, out);
          }
          value = (double)cup3d_stub_iallreduce6;
        }
        else if (op == "midstep") {
          /* the operators of Simulation::advance (15316-15318) from the advection-diffusion step through PressureProjection, as
             they stand in sim.pipeline (setupOperators 15229-15246: [1] = AdvectionDiffusion(Implicit), then forcing, UpdateObstacles,
             Penalization, PressureProjection) -- CreateObstacles before and the force diagnostics after need real obstacles */
          sd.dt = arg;
          for (size_t c = 1; c < sd.pipeline.size(); c++) {
            (*sd.pipeline[c])(arg);
            if (sd.pipeline[c] == std::shared_ptr<Operator>(proj) || (hip_proj && sd.pipeline[c] == hip_proj)) break;
          }
        }
        else if (op == "maxu") value = findMaxU(sd);
        else if (op == "forcing") { ExternalForcing f(sd); f(arg); }
        else if (op == "steps") {
          double t_dt = 0, t_adv = 0;
          for (int n = 0; n < (int)arg; n++) {
#ifdef CUP3D_WITH_HIP
            if (hip_mirror && hip_mirror->device_led) {  /* the time loop of the one-edit integration */
              const double a0 = now();
              const Real dt = cup3d_hip::calcMaxTimestep(*S, *hip_mirror);
              const double a1 = now();
              cup3d_hip::advance(*S, *hip_mirror, dt);
              t_dt += a1 - a0; t_adv += now() - a1;
              value = dt;
              continue;
            }
#endif
            const double a0 = now();
            const Real dt = S->calcMaxTimestep();
            const double a1 = now();
            S->advance(dt);
            t_dt += a1 - a0; t_adv += now() - a1;
            value = dt;
          }
          { /* `timeops`: where the steps' time went, operator by operator; what is left of advance() is adaptMesh (+ dump) */
            double in_ops = 0;
            bool any = false;
            for (auto &pop : sd.pipeline)
              if (auto t = std::dynamic_pointer_cast<TimedOp>(pop)) {
                printf("REF optime name=%s calls=%ld seconds=%.6f\n", t->name.c_str(), t->calls, t->seconds);
                in_ops += t->seconds; t->seconds = 0; t->calls = 0; any = true;
              }
            if (any) {
              printf("REF optime name=calcMaxTimestep calls=%d seconds=%.6f\n", (int)arg, t_dt);
              printf("REF optime name=adaptMesh_and_rest calls=%d seconds=%.6f\n", (int)arg, t_adv - in_ops);
              printf("REF optime name=blocks calls=%ld seconds=0\n", (long)sd.velInfo().size());
#ifdef CUP3D_WITH_HIP
              cup3d_run_stats st;
              if (hip_mirror && cup3d_stats_read(&st) == 0) {  /* PCIe traffic of the drop-in over these steps (then reset) */
                printf("REF optime name=pcie_MB_up calls=%ld seconds=0\n", (long)(st.field_bytes_uploaded / 1e6));
                printf("REF optime name=pcie_MB_down calls=%ld seconds=0\n", (long)(st.field_bytes_downloaded / 1e6));
                printf("REF optime name=bicgstab_iterations calls=%ld seconds=0\n", st.solver_iterations);
                cup3d_stats_reset();
              }
#endif
            }
          }
#ifdef CUP3D_WITH_HIP
          if (hip_mirror) hip_mirror->sync_host();  /* whatever the script does next (dump, tables, another op) sees the host fields */
#endif
        } else { fprintf(stderr, "ref_tool: unknown op %s\n", op.c_str()); exit(2); }
        const double t1 = now();
        printf("REF %s seconds=%.6f iters=%ld value=%.17g\n", op.c_str(), t1 - t0, cup3d_stub_iallreduce7, value);
        fflush(stdout);
      }
    } else {
      fprintf(stderr, "ref_tool: unknown command %s\n", cmd.c_str());
      return 2;
    }
  }
  fflush(0);
  MPI_Barrier(MPI_COMM_WORLD); /* real MPI: a rank that leaves early makes mpiexec kill the others before they have written */
  MPI_Finalize();
  _exit(0); /* skip the reference's destructors: nothing left to verify */
}
