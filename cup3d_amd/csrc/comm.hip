// RCCL transport for the two exchange patterns of the hot path (one process per GPU):
//   * face-slab halo exchange  <- SynchronizerMPI_AMR::sync, MPI_Irecv/Isend per neighbour
//     rank (main.cpp:2356-2405): device pack kernel, one grouped ncclSend/ncclRecv per
//     peer straight into the receiver's halo-slab buffer (no unpack pass: kernels read the
//     packed slabs in place);
//   * small all-reduces        <- MPI_Allreduce / MPI_Iallreduce of 1..7 doubles
//     (main.cpp:8620, 9295, 14442, 14486, 14546, 14584, 15123).
// RCCL is resolved with dlopen at cup3d_comm_init time only, so single-GPU use never
// touches it.  xGMI is point-to-point: every peer pair is one link, each halo message is
// a single contiguous buffer per peer per exchange.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/cup3d_hip_testing.h"
#include "sim.hpp"

namespace cup3d {

int launch_pack(Sim *src, const double *field, int nc, int w, hipStream_t st);  // advdiff.hip
#ifdef CUP3D_TESTING
static bool g_virtual_ranks = false;  // test mode: halos are pre-filled by cup3d_debug_halo_pull
#else
static constexpr bool g_virtual_ranks = false;  // release build: no test transport, every `if (g_vcomm)` / `g_virtual_ranks` branch folds away
#endif
bool virtual_ranks() { return g_virtual_ranks; }

struct Comm {
  void *dl = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  double *d_flag = nullptr;  // one double: the operand of agree()
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Comm g_comm;
static bool g_comm_ready = false;
static bool g_comm_broken = false;  // an RCCL call failed and the communicator was aborted (rccl_fail)
Comm *comm() { return g_comm_ready ? &g_comm : nullptr; }
static int no_comm(const char *what) {
  if (g_comm_broken) { set_error("%s: the RCCL communicator was aborted after an earlier error", what); return CUP3D_ECOMM; }
  set_error("%s without cup3d_comm_init", what);
  return CUP3D_ESTATE;
}

static int load_rccl() {
  if (g_comm.dl) return CUP3D_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  // CUP3D_RCCL_LIBRARY=<path>: this library and no other (a site's own RCCL build; tests: the stand-in of tests/fake_rccl, which lets
  // every line below run on a box where RCCL itself cannot -- two ranks on one device).  RTLD_LOCAL + dlsym on the handle: its symbols,
  // not those of an RCCL the process has loaded already.
  if (const char *path = getenv("CUP3D_RCCL_LIBRARY")) {
    if (!(g_comm.dl = dlopen(path, RTLD_NOW | RTLD_LOCAL))) { set_error("CUP3D_RCCL_LIBRARY: cannot dlopen %s: %s", path, dlerror()); return CUP3D_ECOMM; }
  }
  // An RCCL the process has loaded already (torch.distributed brings its own copy, soname "librccl.so") is the one to use.  Otherwise the
  // system's is loaded -- RTLD_LOCAL, never RTLD_GLOBAL: every entry point is taken from the handle by dlsym, nothing needs its symbols
  // in the global scope, and putting them there is what made a python host abort at exit with glibc's "double free or corruption"
  // when it imported torch AFTER cup3d_comm_init (round 5's open defect, bisected in round 6: scripts/exit_repro.py "rccl,torch" = 134,
  // "torch,rccl" = 0, "rccl" alone = 0): torch's own librccl.so, loaded second, bound its references to the first copy's global
  // symbols, and two copies of one library then shared -- and twice destroyed -- one set of objects.
  for (const char *n : names) {
    if (g_comm.dl) break;
    g_comm.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
  }
  for (const char *n : names) {
    if (g_comm.dl) break;
    g_comm.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!g_comm.dl) { set_error("cannot dlopen librccl: %s", dlerror()); return CUP3D_ECOMM; }
#define SYM(field, name)                                                      \
  *(void **)(&g_comm.field) = dlsym(g_comm.dl, name);                         \
  if (!g_comm.field) { set_error("librccl lacks %s", name); return CUP3D_ECOMM; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy") SYM(CommAbort, "ncclCommAbort")
  SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
  SYM(AllReduce, "ncclAllReduce") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  return CUP3D_OK;
}
// An RCCL call that fails leaves the communicator unusable and the peers possibly inside the matching call: abort the communicator
// (ncclCommAbort: pending operations of THIS rank end instead of spinning) and refuse every later exchange with CUP3D_ECOMM -- the
// host then ends the run on all ranks (the C++ shim: MPI_Abort, as the reference does, main.cpp:15265, 15289; bench.py: its watchdog).
static thread_local int g_group_depth = 0;  // ncclGroupStart calls of this thread not yet closed (RCCL keeps the group per thread)
static int rccl_fail(ncclResult_t r, const char *what) {
  // a Send / Recv that failed between GroupStart and GroupEnd would leave this thread's RCCL group open: a later ncclCommInitRank (the
  // recovery cup3d_comm_init offers) would then be deferred into that stale group.  Close it first; its own result is of no interest.
  while (g_group_depth > 0 && g_comm.GroupEnd) { --g_group_depth; (void)g_comm.GroupEnd(); }
  g_group_depth = 0;
  set_error("RCCL error %d (%s) in %s; communicator aborted", (int)r, g_comm.GetErrorString ? g_comm.GetErrorString(r) : "?", what);
  if (g_comm_ready && g_comm.comm && g_comm.CommAbort) g_comm.CommAbort(g_comm.comm);
  g_comm.comm = nullptr;
  g_comm_ready = false;
  g_comm_broken = true;
  return CUP3D_ECOMM;
}
#define CUP3D_NCCL(call)                                                      \
  do {                                                                        \
    ncclResult_t r_ = (call);                                                 \
    if (r_ != ncclSuccess) return rccl_fail(r_, #call);                       \
  } while (0)

// ------------------------------------------------------------------ in-process transport (TEST SUPPORT)
// "Virtual communicator": the ranks of a run are host THREADS of one process on one GPU, each driving its own Sim through the
// ordinary entry points; the two exchange patterns and the all-reduce are carried out by device copies between the ranks' buffers,
// ordered by host barriers (all ranks enqueue on the same stream, so enqueue order is execution order).  Everything else -- the
// partition, the plans, the pack kernels, the inner/boundary split, which rank owns the corner cell, the order of the collectives
// in solve() -- is the code the RCCL path runs.  The driver has 8-GPU nodes, the builder's boxes have one GPU: this is how the
// multi-rank control flow is executed before it meets RCCL.
struct VComm {
  int n = 0;
  std::vector<Sim *> sims;   // registry: the sim each rank created last
  std::vector<Sim *> cur;    // the sim each rank entered the CURRENT collective with (a rank can hold two: old and adapted mesh)
  std::vector<double *> ptr;
  std::vector<const std::vector<int64_t> *> counts;
  double *d_tmp = nullptr;  // [n][16]
  std::vector<int> codes;   // agree(): the status every rank arrived with
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  long gen = 0;
  bool failed = false;
  // false: a rank did not arrive within 30 s (it failed or diverged in control flow) -- every waiter then gives up, so
  // a broken run ends with an error instead of a hang
  bool barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (failed) return false;
    const long g0 = gen;
    if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); return true; }
    if (!cv.wait_for(lk, std::chrono::seconds(30), [&] { return gen != g0 || failed; }) || failed) {
      failed = true;
      cv.notify_all();
      return false;
    }
    return true;
  }
};
#ifdef CUP3D_TESTING
static VComm *g_vcomm = nullptr;
#else
static constexpr VComm *g_vcomm = nullptr;
#endif
VComm *vcomm() { return g_vcomm; }

// ------------------------------------------------------------------ host-memory transport (TEST SUPPORT, cup3d_hip_testing.h)
// One process per rank as under RCCL, but the bytes travel device -> host -> the caller's transport (MPI in the harness) -> host ->
// device, synchronously.  Everything else -- plans, pack kernels, streams, the order of the collectives -- is the production code.
#ifdef CUP3D_TESTING
static cup3d_host_transport g_ht;
static bool g_ht_on = false;
#else
static constexpr bool g_ht_on = false;
static constexpr cup3d_host_transport g_ht{nullptr, nullptr, nullptr};
#endif
bool host_transport() { return g_ht_on; }
// send_count / recv_count items of `per` doubles per peer, peer-major buffers on the device; `skip` = a rank whose share is not sent
// (exchange_items handles the caller's own share itself) but still occupies its place in both buffers
static int ht_exchange(const double *d_send, const std::vector<int64_t> &send_count, double *d_recv, const std::vector<int64_t> &recv_count, size_t per, int skip,
                       hipStream_t st) {
  const int n = (int)send_count.size();
  std::vector<long> so(n), sb(n), ro(n), rb(n);
  size_t ts = 0, tr = 0;
  for (int p = 0; p < n; ++p) {
    so[p] = (long)(ts * sizeof(double)); ro[p] = (long)(tr * sizeof(double));
    const size_t ns = (size_t)send_count[p] * per, nr = (size_t)recv_count[p] * per;
    sb[p] = p == skip ? 0 : (long)(ns * sizeof(double)); rb[p] = p == skip ? 0 : (long)(nr * sizeof(double));
    ts += ns; tr += nr;
  }
  std::vector<double> hs(ts ? ts : 1), hr(tr ? tr : 1);
  CUP3D_HIP(hipStreamSynchronize(st));  // the pack kernel
  if (ts) CUP3D_HIP(hipMemcpy(hs.data(), d_send, ts * sizeof(double), hipMemcpyDeviceToHost));
  if (g_ht.exchange(g_ht.ctx, hs.data(), so.data(), sb.data(), hr.data(), ro.data(), rb.data())) { set_error("host transport: exchange failed"); return CUP3D_ECOMM; }
  for (int p = 0; p < n; ++p)  // only what arrived: the skipped rank's place in d_recv belongs to the caller
    if (rb[p]) CUP3D_HIP(hipMemcpyAsync((char *)d_recv + ro[p], (const char *)hr.data() + ro[p], (size_t)rb[p], hipMemcpyHostToDevice, st));
  CUP3D_HIP(hipStreamSynchronize(st));
  return CUP3D_OK;
}
void vcomm_register(Sim *s) {
  if (g_vcomm && s->grid->nranks == g_vcomm->n) g_vcomm->sims[s->grid->rank] = s;
}
void vcomm_unregister(Sim *s) {
  if (!g_vcomm) return;
  for (auto &p : g_vcomm->sims) if (p == s) p = nullptr;
  for (auto &p : g_vcomm->cur) if (p == s) p = nullptr;
}

struct VPtrs { const double *p[16]; };
__global__ void k_vreduce(VPtrs v, int nranks, int n, int is_max, double *__restrict__ out) {
  const int i = threadIdx.x;
  if (i >= n) return;
  double a = v.p[0][i];
  for (int r = 1; r < nranks; ++r) a = is_max ? fmax(a, v.p[r][i]) : a + v.p[r][i];  // rank order: identical bits on every rank
  out[i] = a;
}

// The stream exchanges are enqueued on.  RCCL: the rank's communication stream.  Virtual communicator: the same by default, so that
// the event hand-offs between a rank's compute stream and its communication stream are executed (and can fail) in the tests exactly
// as under RCCL; cup3d_debug_set_option("vcomm_one_stream", 1) puts everything on the compute stream instead.
static bool vcomm_two_streams(const Sim *s) { return g_vcomm && s->comm_stream && !debug_option("vcomm_one_stream"); }
static hipStream_t exchange_stream(const Sim *s) {
  if (g_vcomm) return vcomm_two_streams(s) ? s->comm_stream : stream();
  return s->comm_stream;
}

bool scalars_cross_ranks(const Sim *s) {
  if (g_vcomm) return s->grid->nranks > 1;
  if (g_virtual_ranks) return false;
  return s->grid->nranks > 1 || (debug_option("force_allreduce") && comm());
}
// the stream the scalar all-reduces are enqueued on: the communication stream where there is one (every RCCL call of the library is
// issued from that stream, in the same order on all ranks), else the compute stream
hipStream_t scalar_stream(const Sim *s) {
  if (g_virtual_ranks && !g_vcomm) return stream();
  if (!s->comm_stream || (g_vcomm && !vcomm_two_streams(s))) return stream();
  return s->comm_stream;
}

// copy `count[p]` items of `per` doubles from every peer's send buffer (peer-major, so my share of p's buffer starts after what p
// sends to the ranks before me) to dst, in rank order
template <class CountOf>
static int vcomm_pull(Sim *s, double *dst, size_t per, const std::vector<int64_t> &recv_count, CountOf send_count_of, hipStream_t st) {
  VComm *vc = g_vcomm;
  const Grid *g = s->grid;
  // my pack is enqueued on st: the event lets the receivers' streams wait for it (what a matched ncclSend / ncclRecv does)
  vc->cur[g->rank] = s;
  CUP3D_HIP(hipEventRecord(s->ev_vc_pack, st));
  if (!vc->barrier()) { set_error("virtual communicator: a rank is missing at the exchange"); return CUP3D_ECOMM; }  // every rank has enqueued its pack
  size_t ro = 0;
  for (int p = 0; p < g->nranks; ++p) {
    const size_t nr = (size_t)recv_count[p] * per;
    if (!nr) continue;
    Sim *src = vc->cur[p];
    if (!src) { set_error("virtual communicator: rank %d has no sim", p); return CUP3D_ESTATE; }
    const std::vector<int64_t> &sc = send_count_of(src->grid);
    size_t so = 0;
    for (int q = 0; q < g->rank; ++q) so += (size_t)sc[q] * per;
    if ((size_t)sc[g->rank] * per != nr) { set_error("plan mismatch between ranks %d and %d", p, g->rank); return CUP3D_ESTATE; }
    CUP3D_HIP(hipStreamWaitEvent(st, src->ev_vc_pack, 0));
    CUP3D_HIP(hipMemcpyAsync(dst + ro, src->halo_send + so, nr * sizeof(double), hipMemcpyDeviceToDevice, st));
    ro += nr;
  }
  CUP3D_HIP(hipEventRecord(s->ev_vc_done, st));
  if (!vc->barrier()) { set_error("virtual communicator: a rank is missing at the exchange"); return CUP3D_ECOMM; }  // every copy is enqueued
  // a send completes when the data has left: my stream (hence my next pack into halo_send) waits for the copies of my receivers
  const std::vector<int64_t> &mine = send_count_of(g);
  for (int p = 0; p < g->nranks; ++p)
    if (mine[p] && vc->cur[p]) CUP3D_HIP(hipStreamWaitEvent(st, vc->cur[p]->ev_vc_done, 0));
  return CUP3D_OK;
}

// ------------------------------------------------------------------ rank views of a multi-level mesh (Grid::rank_view)
// Before a stencil kernel: the blocks of `field` other ranks' tables refer to travel into the ghost slot range
// [n_local, n_local + nghost) (SynchronizerMPI_AMR::sync / fetch, main.cpp:2356-2544, which ships sub-boxes and coarse shadow cells).
// Every consumer -- same-level copies, restriction, the coarse shadow tile of the interpolation -- addresses whole ghost block slots
// through the renumbered tables, so there is one code path downstream; what TRAVELS is, by default, only the sub-box of each block
// those consumers read (k_pack_boxes / k_unpack_boxes, plan: Grid::ghost_box / send_box), whole blocks for tensorial views.  After a flux-corrected kernel: the face-flux arrays of fine
// faces whose coarse neighbour lives elsewhere (FluxCorrectionMPI::FillBlockCases, 2848-2945).
__global__ void __launch_bounds__(256) k_pack_blocks(const double *__restrict__ field, const int32_t *__restrict__ slots, int nc, double *__restrict__ out) {
  const double *src = field + (size_t)slots[blockIdx.x] * nc * 512;
  double *dst = out + (size_t)blockIdx.x * nc * 512;
  for (int i = threadIdx.x; i < nc * 512; i += 256) dst[i] = src[i];
}
__global__ void __launch_bounds__(64) k_pack_flux(const double *__restrict__ flux, const int32_t *__restrict__ faces, int nfc, double *__restrict__ out) {
  const double *src = flux + (size_t)faces[blockIdx.x] * nfc * 64;
  double *dst = out + (size_t)blockIdx.x * nfc * 64;
  for (int c = 0; c < nfc; ++c) dst[c * 64 + threadIdx.x] = src[c * 64 + threadIdx.x];
}

// sub-box form: block i of the list contributes the cells of its box ([c][z][y][x] over the box) at offset off[i] * nc of the message
__global__ void __launch_bounds__(256) k_pack_boxes(const double *__restrict__ field, const int32_t *__restrict__ slots, const unsigned char *__restrict__ box,
                                                    const long long *__restrict__ off, int nc, double *__restrict__ out) {
  const int i = blockIdx.x;
  const unsigned char *b = box + 6 * i;
  const int nx = b[3] - b[0], ny = b[4] - b[1], nz = b[5] - b[2], vol = nx * ny * nz;
  if (vol == 0) return;
  const double *src = field + (size_t)slots[i] * nc * 512;
  double *dst = out + (size_t)off[i] * nc;
  for (int j = threadIdx.x; j < vol * nc; j += 256) {
    const int c = j / vol, q = j - c * vol, x = q % nx, y = (q / nx) % ny, z = q / (nx * ny);
    dst[j] = src[c * 512 + (b[2] + z) * 64 + (b[1] + y) * 8 + (b[0] + x)];
  }
}
// ... and arrives in the box of ghost block i (slots first_ghost + i of the field)
__global__ void __launch_bounds__(256) k_unpack_boxes(double *__restrict__ ghosts, const unsigned char *__restrict__ box, const long long *__restrict__ off, int nc,
                                                      const double *__restrict__ in) {
  const int i = blockIdx.x;
  const unsigned char *b = box + 6 * i;
  const int nx = b[3] - b[0], ny = b[4] - b[1], nz = b[5] - b[2], vol = nx * ny * nz;
  if (vol == 0) return;
  double *dst = ghosts + (size_t)i * nc * 512;
  const double *src = in + (size_t)off[i] * nc;
  for (int j = threadIdx.x; j < vol * nc; j += 256) {
    const int c = j / vol, q = j - c * vol, x = q % nx, y = (q / nx) % ny, z = q / (nx * ny);
    dst[c * 512 + (b[2] + z) * 64 + (b[1] + y) * 8 + (b[0] + x)] = src[j];
  }
}
__global__ void __launch_bounds__(256) k_poison(double *__restrict__ p, size_t n) {  // TEST SUPPORT ("poison_ghosts"): quiet NaNs
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = __builtin_nan("");
}

// items of `per` doubles: packed send buffer -> peers; received runs land contiguously at dst.  RCCL on the communication stream
// (compute stream <-> communication stream hand-off by events), or the in-process transport.
static size_t sent_bytes(const std::vector<int64_t> &send_count, size_t per, int skip = -1) {
  size_t n = 0;
  for (size_t p = 0; p < send_count.size(); ++p) if ((int)p != skip) n += (size_t)send_count[p];
  return n * per * sizeof(double);
}
// kind: which of the view's plans the counts belong to (the in-process test transport looks the SENDER's counts up by it)
enum ViewPlan { kPlanBlocks = 0, kPlanFlux = 1, kPlanBoxes1 = 2, kPlanBoxes3 = 3 };
static int view_transfer(Sim *s, double *dst, size_t per, const std::vector<int64_t> &send_count, const std::vector<int64_t> &recv_count, ViewPlan kind) {
  const Grid *g = s->grid;
  stats_halo(sent_bytes(send_count, per));
  if (g_vcomm) {
    if (kind == kPlanFlux) return vcomm_pull(s, dst, per, recv_count, [](const Grid *q) -> const std::vector<int64_t> & { return q->send_flux_count; }, exchange_stream(s));
    if (kind == kPlanBoxes1) return vcomm_pull(s, dst, per, recv_count, [](const Grid *q) -> const std::vector<int64_t> & { return q->send_cells[0]; }, exchange_stream(s));
    if (kind == kPlanBoxes3) return vcomm_pull(s, dst, per, recv_count, [](const Grid *q) -> const std::vector<int64_t> & { return q->send_cells[1]; }, exchange_stream(s));
    return vcomm_pull(s, dst, per, recv_count, [](const Grid *q) -> const std::vector<int64_t> & { return q->send_block_count; }, exchange_stream(s));
  }
  if (g_ht_on) return ht_exchange(s->halo_send, send_count, dst, recv_count, per, -1, exchange_stream(s));
  Comm *c = comm();
  if (!c) return no_comm("multi-rank mesh");
  CUP3D_NCCL(c->GroupStart()); ++g_group_depth;
  size_t so = 0, ro = 0;
  for (int p = 0; p < g->nranks; ++p) {
    const size_t ns = (size_t)send_count[p] * per, nr = (size_t)recv_count[p] * per;
    if (ns) CUP3D_NCCL(c->Send(s->halo_send + so, ns, ncclDouble, p, c->comm, s->comm_stream));
    if (nr) CUP3D_NCCL(c->Recv(dst + ro, nr, ncclDouble, p, c->comm, s->comm_stream));
    so += ns;
    ro += nr;
  }
  --g_group_depth; CUP3D_NCCL(c->GroupEnd());
  return CUP3D_OK;
}

// begin: pack + transfer enqueued (on the communication stream with RCCL; ev_h2 marks the arrival); finish: the compute stream waits
static int view_exchange_blocks_begin(Sim *s, double *field, int nc, int w) {
  const Grid *g = s->grid;
  hipStream_t st = exchange_stream(s);
  if (st != stream()) {
    CUP3D_HIP(hipEventRecord(s->ev_h1, stream()));
    CUP3D_HIP(hipStreamWaitEvent(st, s->ev_h1, 0));
  }
  ProfileScope ps("comm_ghost_blocks", st);  // pack + transfer as the communication stream sees them
  const unsigned nsend = (unsigned)g->send_blocks.size(), nghost = (unsigned)g->nghost();
  // the sub-box plans exist for the two stencil widths of the path (1: scalar stencils, 3: advect-diffuse) and box_recv holds three
  // components: anything else would leave ghost layers stale without a sign -- refused
  if ((w != 1 && w != 3) || nc < 1 || nc > 3) { set_error("ghost-block exchange: width %d / %d components (supported: width 1 or 3, 1..3 components)", w, nc); return CUP3D_EINVAL; }
  const int k = w == 3 ? 1 : 0;
  int rc;
  if (!g->send_cells[k].empty() && s->d_send_box[k] && !debug_option("whole_ghost_blocks")) {
    // sub-box form (what the reference ships: SynchronizerMPI_AMR's face sub-boxes and coarse shadow cells, main.cpp:1832-1966,
    // 2423-2544): of every ghost block only the box of cells the width-w star consumers read (Grid::ghost_box) -- packed, sent, scattered
    if (nsend) hipLaunchKernelGGL(k_pack_boxes, dim3(nsend), dim3(256), 0, st, (const double *)field, (const int32_t *)s->d_send_blocks, (const unsigned char *)s->d_send_box[k],
                                  (const long long *)s->d_send_off[k], nc, s->halo_send);
    CUP3D_HIP(hipGetLastError());
    if ((rc = view_transfer(s, s->box_recv, (size_t)nc, g->send_cells[k], g->recv_cells[k], k ? kPlanBoxes3 : kPlanBoxes1))) return rc;
    double *ghosts = field + (size_t)g->n_local * nc * 512;
    if (debug_option("poison_ghosts") && nghost)  // tests: whatever is NOT shipped reads as NaN, so a consumer outside its box cannot go unnoticed
      hipLaunchKernelGGL(k_poison, dim3(256), dim3(256), 0, st, ghosts, (size_t)nghost * nc * 512);
    if (nghost) hipLaunchKernelGGL(k_unpack_boxes, dim3(nghost), dim3(256), 0, st, ghosts, (const unsigned char *)s->d_ghost_box[k], (const long long *)s->d_ghost_off[k], nc,
                                   (const double *)s->box_recv);
    CUP3D_HIP(hipGetLastError());
  } else {
    if (nsend) hipLaunchKernelGGL(k_pack_blocks, dim3(nsend), dim3(256), 0, st, field, s->d_send_blocks, nc, s->halo_send);
    CUP3D_HIP(hipGetLastError());
    if ((rc = view_transfer(s, field + (size_t)g->n_local * nc * 512, (size_t)nc * 512, g->send_block_count, g->recv_block_count, kPlanBlocks))) return rc;
  }
  if (st != stream()) CUP3D_HIP(hipEventRecord(s->ev_h2, st));
  return CUP3D_OK;
}
static int view_exchange_blocks_finish(Sim *s) {
  if (exchange_stream(s) != stream()) {
    ProfileScope ps("comm_exposed_halo_wait");  // what the compute stream idles for (zero when the inner blocks hid the transfer)
    CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_h2, 0));
  }
  return CUP3D_OK;
}
int view_exchange_blocks(Sim *s, double *field, int nc, int w) {
  const Grid *g = s->grid;
  if (g->n_local < 0 || g->nranks == 1) return CUP3D_OK;
  int rc = view_exchange_blocks_begin(s, field, nc, w);
  if (rc) return rc;
  return view_exchange_blocks_finish(s);
}

int view_exchange_flux(Sim *s, int nfc) {
  const Grid *g = s->grid;
  if (g->n_local < 0 || g->nranks == 1) return CUP3D_OK;
  hipStream_t st = exchange_stream(s);
  if (st != stream()) {
    CUP3D_HIP(hipEventRecord(s->ev_h1, stream()));
    CUP3D_HIP(hipStreamWaitEvent(st, s->ev_h1, 0));
  }
  ProfileScope ps("comm_face_flux", st);
  const unsigned nsend = (unsigned)g->send_flux_faces.size();
  if (nsend) hipLaunchKernelGGL(k_pack_flux, dim3(nsend), dim3(64), 0, st, s->d_flux, s->d_send_flux, nfc, s->halo_send);
  CUP3D_HIP(hipGetLastError());
  int rc = view_transfer(s, s->d_flux + (size_t)g->n_local_faces * nfc * 64, (size_t)nfc * 64, g->send_flux_count, g->recv_flux_count, kPlanFlux);
  if (rc) return rc;
  if (st != stream()) {
    CUP3D_HIP(hipEventRecord(s->ev_h2, st));
    CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_h2, 0));
  }
  return CUP3D_OK;
}

// Generic peer-to-peer exchange of `per`-double items (mesh adaptation: ghost blocks of the tensorial view, then the produced blocks
// on their way to their new owners).  send_count / recv_count per rank, own rank included (a device copy); buffers are peer-major.
// Enqueued on the compute stream: adaptation is not overlapped with anything.
int exchange_items(Sim *s, const double *sendbuf, const std::vector<int64_t> &send_count, double *recvbuf, const std::vector<int64_t> &recv_count, size_t per) {
  const Grid *g = s->grid;
  const int me = g->rank, n = g->nranks;
  std::vector<size_t> so(n + 1, 0), ro(n + 1, 0);
  for (int p = 0; p < n; ++p) { so[p + 1] = so[p] + (size_t)send_count[p] * per; ro[p + 1] = ro[p] + (size_t)recv_count[p] * per; }
  if (send_count[me] != recv_count[me]) { set_error("exchange_items: inconsistent self count"); return CUP3D_EINVAL; }
  if (send_count[me]) CUP3D_HIP(hipMemcpyAsync(recvbuf + ro[me], sendbuf + so[me], (size_t)send_count[me] * per * sizeof(double), hipMemcpyDeviceToDevice, stream()));
  if (n == 1) return CUP3D_OK;
  stats_halo(sent_bytes(send_count, per, me));
  Comm *c = (g_vcomm || g_ht_on) ? nullptr : comm();
  if (!g_vcomm && !g_ht_on && !c) return no_comm("multi-rank mesh");
  hipStream_t st = g_vcomm ? exchange_stream(s) : (s->comm_stream ? s->comm_stream : stream());
  if (st != stream()) {
    CUP3D_HIP(hipEventRecord(s->ev_h1, stream()));
    CUP3D_HIP(hipStreamWaitEvent(st, s->ev_h1, 0));
  }
  if (g_vcomm) {  // the in-process transport, on the same stream and behind the same hand-offs as the RCCL calls below
    VComm *vc = g_vcomm;
    vc->cur[me] = s;
    vc->ptr[me] = const_cast<double *>(sendbuf);
    vc->counts[me] = &send_count;
    CUP3D_HIP(hipEventRecord(s->ev_vc_pack, st));
    if (!vc->barrier()) { set_error("virtual communicator: a rank is missing at the block exchange"); return CUP3D_ECOMM; }
    for (int p = 0; p < n; ++p) {
      if (p == me || !recv_count[p]) continue;
      const std::vector<int64_t> &sc = *vc->counts[p];
      size_t off = 0;
      for (int q = 0; q < me; ++q) off += (size_t)sc[q] * per;
      if (sc[me] != recv_count[p]) { set_error("exchange_items: rank %d sends %ld items, rank %d expects %ld", p, (long)sc[me], me, (long)recv_count[p]); return CUP3D_ESTATE; }
      if (vc->cur[p]) CUP3D_HIP(hipStreamWaitEvent(st, vc->cur[p]->ev_vc_pack, 0));
      CUP3D_HIP(hipMemcpyAsync(recvbuf + ro[p], vc->ptr[p] + off, (size_t)recv_count[p] * per * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    CUP3D_HIP(hipEventRecord(s->ev_vc_done, st));
    if (!vc->barrier()) { set_error("virtual communicator: a rank is missing at the block exchange"); return CUP3D_ECOMM; }
    for (int p = 0; p < n; ++p)  // a send completes when the data has left: the callers free their send buffers after the compute stream drains
      if (p != me && send_count[p] && vc->cur[p]) CUP3D_HIP(hipStreamWaitEvent(st, vc->cur[p]->ev_vc_done, 0));
  } else if (g_ht_on) {
    int rc = ht_exchange(sendbuf, send_count, recvbuf, recv_count, per, me, st);
    if (rc) return rc;
  } else {
    CUP3D_NCCL(c->GroupStart()); ++g_group_depth;
    for (int p = 0; p < n; ++p) {
      if (p == me) continue;
      if (send_count[p]) CUP3D_NCCL(c->Send(sendbuf + so[p], (size_t)send_count[p] * per, ncclDouble, p, c->comm, st));
      if (recv_count[p]) CUP3D_NCCL(c->Recv(recvbuf + ro[p], (size_t)recv_count[p] * per, ncclDouble, p, c->comm, st));
    }
    --g_group_depth; CUP3D_NCCL(c->GroupEnd());
  }
  if (st != stream()) {
    CUP3D_HIP(hipEventRecord(s->ev_h2, st));
    CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_h2, 0));
  }
  return CUP3D_OK;
}

// ------------------------------------------------------------------ face-slab halo exchange of uniform grids
static int slab_transfer(Sim *s, size_t per_face, hipStream_t st) {
  const Grid *g = s->grid;
  stats_halo(sent_bytes(g->send_count, per_face));
  if (g_vcomm) return vcomm_pull(s, s->halo_recv, per_face, g->recv_count, [](const Grid *q) -> const std::vector<int64_t> & { return q->send_count; }, st);
  if (g_ht_on) return ht_exchange(s->halo_send, g->send_count, s->halo_recv, g->recv_count, per_face, -1, st);
  Comm *c = comm();
  if (!c) return no_comm("multi-rank grid");
  CUP3D_NCCL(c->GroupStart()); ++g_group_depth;
  size_t so = 0, ro = 0;
  for (int p = 0; p < g->nranks; ++p) {
    const size_t ns = (size_t)g->send_count[p] * per_face, nr = (size_t)g->recv_count[p] * per_face;
    if (ns) CUP3D_NCCL(c->Send(s->halo_send + so, ns, ncclDouble, p, c->comm, st));
    if (nr) CUP3D_NCCL(c->Recv(s->halo_recv + ro, nr, ncclDouble, p, c->comm, st));
    so += ns;
    ro += nr;
  }
  --g_group_depth; CUP3D_NCCL(c->GroupEnd());
  return CUP3D_OK;
}

// Overlapped form (the reference's inner/halo split, compute<>() main.cpp:5598-5618): the
// exchange runs on the communication stream while the caller launches the blocks that have
// no remote neighbour on the compute stream; halo_finish() then makes the compute stream wait
// for the slabs before the boundary blocks are launched.
#ifdef CUP3D_TESTING
// MEASUREMENT SUPPORT ("halo_delay_us", scripts/halo_overlap_probe.py): a one-thread kernel that holds the exchange stream for the given
// time behind the transfer -- what a slab exchange costs when the bytes take that long to arrive.  The compute stream sees it only in
// halo_finish's wait: the part of it that the inner blocks' pass did not cover is the EXPOSED halo time.
__global__ void k_hold_stream(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
#endif
int halo_begin(Sim *s, const double *field, int nc, int w) {
  const Grid *g = s->grid;
  if (g->multilevel) {  // coarse/fine ghost slabs take the halo slabs' place
    if (g->n_local < 0 || g->nranks == 1) return amr_fill_ghosts(s, field, nc, w, s->halo_recv);
    // a rank view: the ghost blocks travel on the communication stream while the slabs of the inner blocks' faces are produced and
    // the caller launches its kernel on the inner blocks (gdev(inner_only)); halo_finish() waits and produces the rest
    int rc = view_exchange_blocks_begin(s, const_cast<double *>(field), nc, w);
    if (rc) return rc;
    s->pending_fill.field = field; s->pending_fill.nc = nc; s->pending_fill.w = w; s->pending_fill.slabs = s->halo_recv; s->pending_fill.bc_dir = s->scalar_bc_dir; s->pending_fill.open = true;
    return amr_fill_ghosts(s, field, nc, w, s->halo_recv, 1);
  }
  if (g->nranks == 1 || (g_virtual_ranks && !g_vcomm)) return CUP3D_OK;
  const size_t per_face = (size_t)nc * w * 64;
  hipStream_t st = exchange_stream(s);
  // the field (and the previous consumers of the slab buffers) live on the compute stream
  if (st != stream()) {
    CUP3D_HIP(hipEventRecord(s->ev_h1, stream()));
    CUP3D_HIP(hipStreamWaitEvent(st, s->ev_h1, 0));
  }
  {
    ProfileScope ps("comm_halo", st);  // pack kernel + grouped send/recv, timed on the stream they run on
    int rc = launch_pack(s, field, nc, w, st);
    if (rc) return rc;
    if ((rc = slab_transfer(s, per_face, st))) return rc;
#ifdef CUP3D_TESTING
    if (const int us = debug_option("halo_delay_us")) hipLaunchKernelGGL(k_hold_stream, dim3(1), dim3(1), 0, st, (long long)us * 100);  // wall_clock64: 100 MHz
#endif
  }
  if (st != stream()) CUP3D_HIP(hipEventRecord(s->ev_h2, st));
  return CUP3D_OK;
}
int halo_finish(Sim *s) {
  if (s->grid->multilevel) {
    if (!s->pending_fill.open) return CUP3D_OK;
    s->pending_fill.open = false;
    int rc = view_exchange_blocks_finish(s);
    if (rc) return rc;
    const int dir = s->scalar_bc_dir;
    s->scalar_bc_dir = s->pending_fill.bc_dir;  // the Helmholtz operators set it around halo_begin only
    rc = amr_fill_ghosts(s, s->pending_fill.field, s->pending_fill.nc, s->pending_fill.w, s->pending_fill.slabs, 2);
    s->scalar_bc_dir = dir;
    return rc;
  }
  if (s->grid->nranks == 1 || (g_virtual_ranks && !g_vcomm)) return CUP3D_OK;
  if (exchange_stream(s) != stream()) {
    ProfileScope ps("comm_exposed_halo_wait");
    CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_h2, 0));
  }
  return CUP3D_OK;
}
int halo_exchange(Sim *s, const double *field, int nc, int w) {
  int rc = halo_begin(s, field, nc, w);
  if (rc) return rc;
  return halo_finish(s);
}

int allreduce(Sim *s, double *d_buf, int n, bool is_max, hipStream_t st) {
  if (!scalars_cross_ranks(s)) return CUP3D_OK;
  stats_allreduce();
  if (g_vcomm) {
    VComm *vc = g_vcomm;
    const int r = s->grid->rank;
    if (n > 16 || vc->n > 16) { set_error("virtual communicator: at most 16 ranks / 16 values"); return CUP3D_EINVAL; }
    vc->cur[r] = s;
    vc->ptr[r] = d_buf;
    CUP3D_HIP(hipEventRecord(s->ev_vc_pack, st));  // my operand is final at this point of st
    if (!vc->barrier()) { set_error("virtual communicator: a rank is missing at the all-reduce"); return CUP3D_ECOMM; }  // every rank's operand is enqueued
    VPtrs v;
    for (int q = 0; q < vc->n; ++q) {
      v.p[q] = vc->ptr[q];
      if (q != r && vc->cur[q]) CUP3D_HIP(hipStreamWaitEvent(st, vc->cur[q]->ev_vc_pack, 0));
    }
    hipLaunchKernelGGL(k_vreduce, dim3(1), dim3(64), 0, st, v, vc->n, n, is_max ? 1 : 0, vc->d_tmp + 16 * r);
    CUP3D_HIP(hipGetLastError());
    CUP3D_HIP(hipEventRecord(s->ev_vc_done, st));
    if (!vc->barrier()) { set_error("virtual communicator: a rank is missing at the all-reduce"); return CUP3D_ECOMM; }  // every rank has enqueued its read
    for (int q = 0; q < vc->n; ++q)
      if (q != r && vc->cur[q]) CUP3D_HIP(hipStreamWaitEvent(st, vc->cur[q]->ev_vc_done, 0));  // ... and has read my operand before I overwrite it
    CUP3D_HIP(hipMemcpyAsync(d_buf, vc->d_tmp + 16 * r, n * sizeof(double), hipMemcpyDeviceToDevice, st));
    return CUP3D_OK;
  }
  if (g_ht_on) {
    double h[16];
    if (n > 16) { set_error("host transport: at most 16 values per all-reduce"); return CUP3D_EINVAL; }
    CUP3D_HIP(hipStreamSynchronize(st));
    CUP3D_HIP(hipMemcpy(h, d_buf, n * sizeof(double), hipMemcpyDeviceToHost));
    if (g_ht.allreduce(g_ht.ctx, h, n, is_max ? 1 : 0)) { set_error("host transport: all-reduce failed"); return CUP3D_ECOMM; }
    CUP3D_HIP(hipMemcpyAsync(d_buf, h, n * sizeof(double), hipMemcpyHostToDevice, st));
    CUP3D_HIP(hipStreamSynchronize(st));
    return CUP3D_OK;
  }
  Comm *c = comm();
  if (!c) return no_comm("multi-rank grid");
  CUP3D_NCCL(c->AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, is_max ? ncclMax : ncclSum, c->comm, st));
  return CUP3D_OK;
}


// Collective status agreement.  A collective entry point (mesh migration, the chi exchange of the tagging, ...) first does its
// rank-local part -- argument checks, plan building, allocations -- and may fail there on ONE rank only; if that rank simply returned,
// the others would block in the next send/recv for ever.  So every rank passes its local outcome here BEFORE the first exchange and
// all ranks get the same code back: 0, or the worst (most negative) code any rank arrived with.  The reference has no return codes
// and ends such runs with MPI_Abort on every rank (main.cpp:15265, 15289): all-or-none is the convention kept.  One MAX all-reduce of
// one double with a host wait: for entry points that run once per mesh adaptation, not for the per-iteration halos (their only
// rank-local failure, a missing communicator, is the same on every rank).
int agree(Sim *s, int rc, const char *where) {
  const Grid *g = s->grid;
  if (g->nranks == 1 || (g_virtual_ranks && !g_vcomm)) return rc;
  int worst = rc;
  if (g_vcomm) {
    VComm *vc = g_vcomm;
    vc->codes[g->rank] = rc;
    if (!vc->barrier()) { set_error("virtual communicator: a rank is missing at the status agreement of %s", where); return CUP3D_ECOMM; }
    for (int q = 0; q < vc->n; ++q) worst = std::min(worst, vc->codes[q]);
    if (!vc->barrier()) { set_error("virtual communicator: a rank is missing at the status agreement of %s", where); return CUP3D_ECOMM; }  // codes[] may be rewritten
  } else if (g_ht_on) {
    double v = (double)-rc;
    if (g_ht.allreduce(g_ht.ctx, &v, 1, 1)) { set_error("host transport: status agreement of %s failed", where); return CUP3D_ECOMM; }
    worst = -(int)v;
  } else {
    Comm *c = comm();
    if (!c) return rc ? rc : no_comm(where);
    hipStream_t st = s->comm_stream ? s->comm_stream : stream();
    double v = (double)-rc;
    if (!c->d_flag) CUP3D_HIP(hipMalloc((void **)&c->d_flag, sizeof(double)));
    CUP3D_HIP(hipStreamSynchronize(stream()));  // the collective proper starts on a drained compute stream, like the reference's blocking calls
    CUP3D_HIP(hipMemcpyAsync(c->d_flag, &v, sizeof v, hipMemcpyHostToDevice, st));
    CUP3D_NCCL(c->AllReduce(c->d_flag, c->d_flag, 1, ncclDouble, ncclMax, c->comm, st));
    CUP3D_HIP(hipMemcpyAsync(&v, c->d_flag, sizeof v, hipMemcpyDeviceToHost, st));
    CUP3D_HIP(hipStreamSynchronize(st));
    worst = -(int)v;
  }
  if (worst != 0 && rc == 0) set_error("%s: another rank could not take part (its status: %d); nothing was exchanged", where, worst);
  return worst;
}

}  // namespace cup3d

using namespace cup3d;

extern "C" {

int cup3d_comm_unique_id(void *id128) {
  if (!id128) return CUP3D_EINVAL;
  int rc = load_rccl();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  CUP3D_NCCL(g_comm.GetUniqueId(&id));
  memcpy(id128, &id, 128);
  return CUP3D_OK;
}

int cup3d_comm_init(int rank, int nranks, const void *id128) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return CUP3D_EINVAL;
  if (nranks == 1 && !debug_option("force_allreduce")) { g_comm.rank = 0; g_comm.nranks = 1; return CUP3D_OK; }
  if (!id128) return CUP3D_EINVAL;
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  CUP3D_NCCL(g_comm.CommInitRank(&g_comm.comm, nranks, id, rank));
  g_comm_broken = false;
  g_comm.rank = rank;
  g_comm.nranks = nranks;
  g_comm_ready = true;
  return CUP3D_OK;
}

int cup3d_comm_finalize(void) {
  if (g_comm_ready && g_comm.comm) {
    g_comm.CommDestroy(g_comm.comm);
    g_comm.comm = nullptr;
  }
  if (g_comm.d_flag) { hipFree(g_comm.d_flag); g_comm.d_flag = nullptr; }
  g_comm_ready = false;
  g_comm_broken = false;
  return CUP3D_OK;
}

// TEST SUPPORT: see cup3d_hip_testing.h
#ifdef CUP3D_TESTING  // test / tuning support: not in the release library at all
int cup3d_debug_host_transport(int rank, int nranks, const cup3d_host_transport *t) {
  if (!t) { g_ht_on = false; return CUP3D_OK; }
  if (nranks < 1 || rank < 0 || rank >= nranks || !t->exchange || !t->allreduce) return CUP3D_EINVAL;
  g_ht = *t;
  g_ht_on = nranks > 1;
  return CUP3D_OK;
}
#endif

// TEST SUPPORT: several ranks' sims in one process on one GPU; exchanges become no-ops
#ifdef CUP3D_TESTING  // test / tuning support: not in the release library at all
int cup3d_debug_virtual_ranks(int on) {
  g_virtual_ranks = on != 0;
  return CUP3D_OK;
}
#endif

// TEST SUPPORT: in-process communicator over `nranks` host threads (see VComm above); nranks = 0 tears it down.  Create it before
// the sims of the run (they register themselves by rank), then call the ordinary entry points from one thread per rank.
#ifdef CUP3D_TESTING  // test / tuning support: not in the release library at all
int cup3d_debug_virtual_comm(int nranks) {
  if (nranks < 0 || nranks > 16) return CUP3D_EINVAL;
  if (g_vcomm) {
    if (g_vcomm->d_tmp) hipFree(g_vcomm->d_tmp);
    delete g_vcomm;
    g_vcomm = nullptr;
    g_virtual_ranks = false;
  }
  if (nranks == 0) return CUP3D_OK;
  VComm *vc = new VComm();
  vc->n = nranks;
  vc->sims.assign(nranks, nullptr);
  vc->cur.assign(nranks, nullptr);
  vc->ptr.assign(nranks, nullptr);
  vc->counts.assign(nranks, nullptr);
  vc->codes.assign(nranks, 0);
  if (hipMalloc((void **)&vc->d_tmp, (size_t)nranks * 16 * sizeof(double)) != hipSuccess) { delete vc; set_error("virtual communicator: hipMalloc failed"); return CUP3D_EDEVICE; }
  g_vcomm = vc;
  g_virtual_ranks = true;
  return CUP3D_OK;
}
#endif

// TEST SUPPORT: fill `dst`'s halo slabs for (field, nc, w) by packing directly from peer
// sims living in the same process on the same GPU ("virtual ranks").  Exercises the plan
// ordering, the pack kernel and the kernels' halo-read path on one GPU; the RCCL call
// sequence itself is what halo_exchange() adds on top.
#ifdef CUP3D_TESTING  // test / tuning support: not in the release library at all
int cup3d_debug_halo_pull(cup3d_sim_t *dst_h, cup3d_sim_t *const *peers, int npeers, int field, int nc, int w) {
  if (!dst_h || !peers) return CUP3D_EINVAL;
  Sim *dst = reinterpret_cast<Sim *>(dst_h);
  const Grid *g = dst->grid;
  if (npeers != g->nranks) { set_error("need one sim per rank"); return CUP3D_EINVAL; }
  const size_t per_face = (size_t)nc * w * 64;
  size_t ro = 0;
  for (int p = 0; p < g->nranks; ++p) {
    const size_t nr = (size_t)g->recv_count[p] * per_face;
    if (!nr) continue;
    Sim *src = reinterpret_cast<Sim *>(peers[p]);
    const Grid *gp = src->grid;
    int ncs;
    const double *f = src->field(field, &ncs);
    if (!f || ncs != nc) { set_error("bad field for halo pull"); return CUP3D_EINVAL; }
    int rc = launch_pack(src, f, nc, w, stream());
    if (rc) return rc;
    size_t so = 0;
    for (int q = 0; q < g->rank; ++q) so += (size_t)gp->send_count[q] * per_face;
    if ((size_t)gp->send_count[g->rank] * per_face != nr) { set_error("plan mismatch: rank %d sends %ld faces, rank %d expects %ld", p, (long)gp->send_count[g->rank], g->rank, (long)g->recv_count[p]); return CUP3D_ESTATE; }
    CUP3D_HIP(hipMemcpyAsync(dst->halo_recv + ro, src->halo_send + so, nr * sizeof(double), hipMemcpyDeviceToDevice, stream()));
    ro += nr;
  }
  CUP3D_HIP(hipStreamSynchronize(stream()));
  return CUP3D_OK;
}
#endif

}  // extern "C"
