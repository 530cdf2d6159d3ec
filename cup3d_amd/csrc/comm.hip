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

#include <cstring>

#include "sim.hpp"

namespace cup3d {

int launch_pack(Sim *src, const double *field, int nc, int w, hipStream_t st);  // advdiff.hip
static bool g_virtual_ranks = false;  // test mode: halos are pre-filled by cup3d_debug_halo_pull
bool virtual_ranks() { return g_virtual_ranks; }

struct Comm {
  void *dl = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Comm g_comm;
static bool g_comm_ready = false;
Comm *comm() { return g_comm_ready ? &g_comm : nullptr; }

static int load_rccl() {
  if (g_comm.dl) return CUP3D_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names)
    if ((g_comm.dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!g_comm.dl) { set_error("cannot dlopen librccl: %s", dlerror()); return CUP3D_ECOMM; }
#define SYM(field, name)                                                      \
  *(void **)(&g_comm.field) = dlsym(g_comm.dl, name);                         \
  if (!g_comm.field) { set_error("librccl lacks %s", name); return CUP3D_ECOMM; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
  SYM(AllReduce, "ncclAllReduce") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  return CUP3D_OK;
}
#define CUP3D_NCCL(call)                                                      \
  do {                                                                        \
    ncclResult_t r_ = (call);                                                 \
    if (r_ != ncclSuccess) {                                                  \
      set_error("RCCL error %d (%s) in %s", (int)r_, g_comm.GetErrorString ? g_comm.GetErrorString(r_) : "?", #call); \
      return CUP3D_ECOMM;                                                     \
    }                                                                         \
  } while (0)

int halo_exchange(Sim *s, const double *field, int nc, int w) {
  const Grid *g = s->grid;
  if (g->multilevel) return amr_fill_ghosts(s, field, nc, w, s->halo_recv);  // coarse/fine ghost slabs take the halo slabs' place
  if (g->nranks == 1 || g_virtual_ranks) return CUP3D_OK;
  Comm *c = comm();
  if (!c) { set_error("multi-rank grid without cup3d_comm_init"); return CUP3D_ESTATE; }
  const size_t per_face = (size_t)nc * w * 64;
  ProfileScope ps("halo_exchange");
  {
    int rc = launch_pack(s, field, nc, w, stream());
    if (rc) return rc;
  }
  CUP3D_NCCL(c->GroupStart());
  size_t so = 0, ro = 0;
  for (int p = 0; p < g->nranks; ++p) {
    const size_t ns = (size_t)g->send_count[p] * per_face, nr = (size_t)g->recv_count[p] * per_face;
    if (ns) CUP3D_NCCL(c->Send(s->halo_send + so, ns, ncclDouble, p, c->comm, stream()));
    if (nr) CUP3D_NCCL(c->Recv(s->halo_recv + ro, nr, ncclDouble, p, c->comm, stream()));
    so += ns;
    ro += nr;
  }
  CUP3D_NCCL(c->GroupEnd());
  return CUP3D_OK;
}

// Overlapped form (the reference's inner/halo split, compute<>() main.cpp:5598-5618): the
// exchange runs on the communication stream while the caller launches the blocks that have
// no remote neighbour on the compute stream; halo_finish() then makes the compute stream wait
// for the slabs before the boundary blocks are launched.
int halo_begin(Sim *s, const double *field, int nc, int w) {
  const Grid *g = s->grid;
  if (g->multilevel) return amr_fill_ghosts(s, field, nc, w, s->halo_recv);
  if (g->nranks == 1 || g_virtual_ranks) return CUP3D_OK;
  Comm *c = comm();
  if (!c) { set_error("multi-rank grid without cup3d_comm_init"); return CUP3D_ESTATE; }
  const size_t per_face = (size_t)nc * w * 64;
  // the field (and the previous consumers of the slab buffers) live on the compute stream
  CUP3D_HIP(hipEventRecord(s->ev_h1, stream()));
  CUP3D_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_h1, 0));
  {
    int rc = launch_pack(s, field, nc, w, s->comm_stream);
    if (rc) return rc;
  }
  CUP3D_NCCL(c->GroupStart());
  size_t so = 0, ro = 0;
  for (int p = 0; p < g->nranks; ++p) {
    const size_t ns = (size_t)g->send_count[p] * per_face, nr = (size_t)g->recv_count[p] * per_face;
    if (ns) CUP3D_NCCL(c->Send(s->halo_send + so, ns, ncclDouble, p, c->comm, s->comm_stream));
    if (nr) CUP3D_NCCL(c->Recv(s->halo_recv + ro, nr, ncclDouble, p, c->comm, s->comm_stream));
    so += ns;
    ro += nr;
  }
  CUP3D_NCCL(c->GroupEnd());
  CUP3D_HIP(hipEventRecord(s->ev_h2, s->comm_stream));
  return CUP3D_OK;
}
int halo_finish(Sim *s) {
  if (s->grid->nranks == 1 || g_virtual_ranks) return CUP3D_OK;
  CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_h2, 0));
  return CUP3D_OK;
}

int allreduce(Sim *s, double *d_buf, int n, bool is_max, hipStream_t st) {
  // "force_allreduce": TEST SUPPORT, run the RCCL call on a 1-rank communicator to exercise the plumbing on one GPU
  if ((s->grid->nranks == 1 && !(debug_option("force_allreduce") && comm())) || g_virtual_ranks) return CUP3D_OK;
  Comm *c = comm();
  if (!c) { set_error("multi-rank grid without cup3d_comm_init"); return CUP3D_ESTATE; }
  CUP3D_NCCL(c->AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, is_max ? ncclMax : ncclSum, c->comm, st));
  return CUP3D_OK;
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
  g_comm_ready = false;
  return CUP3D_OK;
}

// TEST SUPPORT: several ranks' sims in one process on one GPU; exchanges become no-ops
int cup3d_debug_virtual_ranks(int on) { g_virtual_ranks = on != 0; return CUP3D_OK; }

// TEST SUPPORT: fill `dst`'s halo slabs for (field, nc, w) by packing directly from peer
// sims living in the same process on the same GPU ("virtual ranks").  Exercises the plan
// ordering, the pack kernel and the kernels' halo-read path on one GPU; the RCCL call
// sequence itself is what halo_exchange() adds on top.
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

}  // extern "C"
