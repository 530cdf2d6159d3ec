// TEST INFRASTRUCTURE -- a stand-in for librccl at RCCL's OWN API boundary, so that the production communication code of the library
// (cup3d_amd/csrc/comm.hip: dlopen, ncclGetUniqueId / ncclCommInitRank, grouped ncclSend / ncclRecv on the communication stream,
// ncclAllReduce, ncclCommAbort, cup3d_comm_finalize) and bench.py's --transport rccl path (unique id over gloo, one process per rank,
// release build of the library) EXECUTE on a box with ONE GPU.  RCCL itself refuses two ranks on one device; the builder's and the
// driver's test boxes have one.  Selected with CUP3D_RCCL_LIBRARY=<this .so> (comm.hip, load_rccl); never loaded otherwise.
//
// What it keeps of RCCL's contract (the part comm.hip relies on):
//   * every operation is ENQUEUED on the HIP stream it is given and completes in stream order; the host returns at once;
//   * sends and receives inside ncclGroupStart / ncclGroupEnd are issued together at the closing GroupEnd (all sends before all
//     receives: a rank never waits for a peer before its own sends are on their way);
//   * a send matches the receive the peer posts for it in the same order (one mailbox per ordered pair of ranks, sequence-numbered);
//   * ncclAllReduce of doubles with ncclSum / ncclMax: every rank reduces all contributions in rank order -- identical bits everywhere.
// How the bytes travel: device -> a POSIX shared-memory mailbox -> device, with stream-ordered copies and host functions
// (hipLaunchHostFunc) that raise / wait for the mailbox's sequence numbers.  The mailboxes are PINNED in both processes
// (hipHostRegister): an asynchronous copy from pageable memory would be staged when it is ENQUEUED -- before the host function in front of
// it has seen the message arrive.  Slow by design.  One node only.
// (The waits are BLOCKING host functions, and the HIP runtime may run every stream's host functions of a process on one thread.  All
//  RCCL calls of the library come from one stream per rank, so no wait can sit in front of the host function that would release it; a
//  host that spread them over streams could deadlock HERE where RCCL would not.  A form without host functions -- hipStreamWriteValue64 /
//  hipStreamWaitValue64 on the sequence numbers in the pinned shared segment -- would remove that difference; not built.)
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kMailboxBytes = 16u << 20;  // per ordered pair of ranks: one shared-memory segment each, created when the pair's ranks attach
constexpr int kReduceSlots = 4, kReduceMax = 64;

struct Mailbox {
  std::atomic<unsigned long long> sent, consumed;  // messages written / read so far
  size_t bytes;
  char pad[64 - 2 * sizeof(std::atomic<unsigned long long>) - sizeof(size_t)];
};
struct Shared {
  std::atomic<int> attached, detached;
  int nranks;
  std::atomic<unsigned long long> contributed[kReduceSlots];      // ranks that have written their operand of all-reduce number seq (slot seq % 4)
  std::atomic<unsigned long long> finished[kMaxRanks];            // all-reduces rank r has completed
  double operand[kReduceSlots][kMaxRanks][kReduceMax];
  Mailbox box[kMaxRanks][kMaxRanks];                              // [src][dst]: the sequence numbers; the payload lives in a segment of its own
};
inline size_t shared_bytes() { return sizeof(Shared); }

struct Comm {
  Shared *S = nullptr;
  char *out_box[kMaxRanks] = {nullptr}, *in_box[kMaxRanks] = {nullptr};  // payload of the mailboxes me -> p and p -> me, mapped and pinned here
  int rank = 0, nranks = 1;
  char name[64] = {0};
  unsigned long long reduce_seq = 0;
  unsigned long long sent_to[kMaxRanks] = {0}, recv_from[kMaxRanks] = {0};
  double *h_red = nullptr;  // pinned: the result of an all-reduce on its way back to the device
};

struct Pending { bool send; void *buf; size_t bytes; int peer; Comm *c; hipStream_t st; };
thread_local int g_depth = 0;
thread_local std::vector<Pending> g_pending;

[[noreturn]] void die(const char *what) {
  fprintf(stderr, "fake_rccl: %s\n", what);
  abort();
}
int wait_limit() {  // seconds a host function waits for the other rank before it ends the process (CUP3D_FAKE_RCCL_WAIT, default 120)
  static const int s = [] { const char *e = getenv("CUP3D_FAKE_RCCL_WAIT"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 120; }();
  return s;
}
template <class F>
void wait_until(F ok, const char *what, int a = -1, int b = -1, unsigned long long seq = 0) {
  const auto t0 = std::chrono::steady_clock::now();
  while (!ok()) {
    std::this_thread::sleep_for(std::chrono::microseconds(20));
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(wait_limit())) {
      fprintf(stderr, "fake_rccl: pid %d, ranks %d -> %d, operation number %llu:\n", (int)getpid(), a, b, seq);
      die(what);
    }
  }
}

// ---- host functions (run in stream order on the stream's callback thread; no HIP calls in here)
struct SendDone { Shared *S; int src, dst; size_t bytes; };
void mark_sent(void *p) {
  SendDone *d = static_cast<SendDone *>(p);
  d->S->box[d->src][d->dst].bytes = d->bytes;
  d->S->box[d->src][d->dst].sent.fetch_add(1, std::memory_order_release);
  delete d;
}
struct SlotFree { Shared *S; int src, dst; unsigned long long seq; };
void wait_slot_free(void *p) {  // the receiver has taken message seq - 1 out of the mailbox
  SlotFree *d = static_cast<SlotFree *>(p);
  wait_until([&] { return d->S->box[d->src][d->dst].consumed.load(std::memory_order_acquire) + 1 >= d->seq; }, "a send waited for the receiver to empty the mailbox", d->src, d->dst, d->seq);
  delete d;
}
struct RecvWait { Shared *S; int src, dst; unsigned long long seq; size_t bytes; };
void wait_arrival(void *p) {
  RecvWait *d = static_cast<RecvWait *>(p);
  wait_until([&] { return d->S->box[d->src][d->dst].sent.load(std::memory_order_acquire) >= d->seq; }, "a receive waited for its message", d->src, d->dst, d->seq);
  if (d->S->box[d->src][d->dst].bytes != d->bytes) die("a receive's size differs from the matching send's (plan mismatch between two ranks)");
  delete d;
}
void mark_consumed(void *p) {
  RecvWait *d = static_cast<RecvWait *>(p);
  d->S->box[d->src][d->dst].consumed.fetch_add(1, std::memory_order_release);
  delete d;
}
struct Reduce { Comm *c; unsigned long long seq; int n; bool is_max; };
void reduce_contribute(void *p) {
  Reduce *d = static_cast<Reduce *>(p);
  d->c->S->contributed[d->seq % kReduceSlots].fetch_add(1, std::memory_order_release);
  delete d;
}
void reduce_wait_slot(void *p) {  // slot seq % 4 was last used by all-reduce seq - 4: every rank must be done with it
  Reduce *d = static_cast<Reduce *>(p);
  Shared *S = d->c->S;
  if (d->seq > kReduceSlots)
    wait_until([&] { for (int r = 0; r < S->nranks; ++r) if (S->finished[r].load(std::memory_order_acquire) + kReduceSlots < d->seq) return false; return true; },
               "an all-reduce waited for the other ranks to finish an earlier one", d->c->rank, -1, d->seq);
  delete d;
}
void reduce_collect(void *p) {
  Reduce *d = static_cast<Reduce *>(p);
  Shared *S = d->c->S;
  const int slot = (int)(d->seq % kReduceSlots);
  wait_until([&] { return S->contributed[slot].load(std::memory_order_acquire) >= (unsigned long long)S->nranks * ((d->seq - 1) / kReduceSlots + 1); },
             "an all-reduce waited for the other ranks' operands", d->c->rank, -1, d->seq);
  for (int i = 0; i < d->n; ++i) {
    double a = S->operand[slot][0][i];
    for (int r = 1; r < S->nranks; ++r) a = d->is_max ? (a > S->operand[slot][r][i] ? a : S->operand[slot][r][i]) : a + S->operand[slot][r][i];  // rank order
    d->c->h_red[i] = a;
  }
  delete d;
}
void reduce_finished(void *p) {
  Reduce *d = static_cast<Reduce *>(p);
  d->c->S->finished[d->c->rank].store(d->seq, std::memory_order_release);
  delete d;
}

ncclResult_t issue(const Pending &q) {
  Shared *S = q.c->S;
  const int me = q.c->rank;
  if (q.bytes > kMailboxBytes) die("message larger than the fake mailbox");
  if (q.send) {
    const unsigned long long seq = ++q.c->sent_to[q.peer];
    if (hipLaunchHostFunc(q.st, wait_slot_free, new SlotFree{S, me, q.peer, seq}) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(q.c->out_box[q.peer], q.buf, q.bytes, hipMemcpyDeviceToHost, q.st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(q.st, mark_sent, new SendDone{S, me, q.peer, q.bytes}) != hipSuccess) return ncclUnhandledCudaError;
  } else {
    const unsigned long long seq = ++q.c->recv_from[q.peer];
    if (hipLaunchHostFunc(q.st, wait_arrival, new RecvWait{S, q.peer, me, seq, q.bytes}) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(q.buf, q.c->in_box[q.peer], q.bytes, hipMemcpyHostToDevice, q.st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(q.st, mark_consumed, new RecvWait{S, q.peer, me, seq, q.bytes}) != hipSuccess) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/cup3d_fake_rccl_%d_%lld", (int)getpid(),
           (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)shared_bytes()) != 0) return ncclSystemError;
  close(fd);  // zero-filled by the kernel: every counter starts at 0
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  int fd = -1;
  for (int tries = 0; tries < 3000 && fd < 0; ++tries) {
    fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  if (fd < 0) return ncclSystemError;
  void *p = mmap(nullptr, shared_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  Comm *c = new Comm();
  c->S = static_cast<Shared *>(p);
  c->rank = rank;
  c->nranks = nranks;
  snprintf(c->name, sizeof c->name, "%s", id.internal);
  if (hipHostMalloc((void **)&c->h_red, kReduceMax * sizeof(double), hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
  if (hipHostRegister(c->S, shared_bytes(), hipHostRegisterDefault) != hipSuccess) return ncclUnhandledCudaError;  // the all-reduce operands land in it
  for (int p2 = 0; p2 < nranks; ++p2) {  // the two mailboxes of every pair I am part of: whoever comes first creates the segment
    if (p2 == rank) continue;
    for (int dir = 0; dir < 2; ++dir) {
      char seg[96];
      snprintf(seg, sizeof seg, "%s_%d_%d", id.internal, dir ? p2 : rank, dir ? rank : p2);
      const int sfd = shm_open(seg, O_CREAT | O_RDWR, 0600);
      if (sfd < 0 || ftruncate(sfd, (off_t)kMailboxBytes) != 0) return ncclSystemError;
      void *m = mmap(nullptr, kMailboxBytes, PROT_READ | PROT_WRITE, MAP_SHARED, sfd, 0);
      close(sfd);
      if (m == MAP_FAILED) return ncclSystemError;
      if (hipHostRegister(m, kMailboxBytes, hipHostRegisterDefault) != hipSuccess) return ncclUnhandledCudaError;
      (dir ? c->in_box : c->out_box)[p2] = static_cast<char *>(m);
    }
  }
  if (rank == 0) c->S->nranks = nranks;
  c->S->attached.fetch_add(1);
  wait_until([&] { return c->S->attached.load() >= nranks; }, "ncclCommInitRank: not every rank arrived", rank);
  // every rank has opened everything it will ever open: the names can go now (the memory lives as long as it is mapped), so that a
  // rank that ends without ncclCommDestroy -- or is killed -- leaves nothing behind in /dev/shm
  for (int p2 = 0; p2 < nranks; ++p2) {
    if (p2 == rank) continue;
    for (int dir = 0; dir < 2; ++dir) {
      char seg[96];
      snprintf(seg, sizeof seg, "%s_%d_%d", id.internal, dir ? p2 : rank, dir ? rank : p2);
      shm_unlink(seg);
    }
  }
  shm_unlink(id.internal);
  *out = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

static ncclResult_t leave(ncclComm_t h) {
  Comm *c = reinterpret_cast<Comm *>(h);
  if (!c) return ncclInvalidArgument;
  for (int p2 = 0; p2 < c->nranks; ++p2)
    for (int dir = 0; dir < 2; ++dir) {
      char *m = (dir ? c->in_box : c->out_box)[p2];
      if (!m) continue;
      (void)hipHostUnregister(m);
      munmap(m, kMailboxBytes);
      char seg[96];
      snprintf(seg, sizeof seg, "%s_%d_%d", c->name, dir ? p2 : c->rank, dir ? c->rank : p2);
      shm_unlink(seg);  // (the second of the pair finds it gone already)
    }
  const bool last = c->S->detached.fetch_add(1) + 1 >= c->nranks;
  (void)hipHostUnregister(c->S);
  munmap(c->S, shared_bytes());
  if (last) shm_unlink(c->name);
  if (c->h_red) (void)hipHostFree(c->h_red);
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t h) { return leave(h); }
ncclResult_t ncclCommAbort(ncclComm_t h) { return leave(h); }

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  ncclResult_t rc = ncclSuccess;
  for (int pass = 0; pass < 2; ++pass)  // every send first, then the receives
    for (const Pending &q : g_pending)
      if (q.send == (pass == 0) && rc == ncclSuccess) rc = issue(q);
  g_pending.clear();
  return rc;
}
static ncclResult_t p2p(bool send, void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t h, hipStream_t st) {
  Comm *c = reinterpret_cast<Comm *>(h);
  if (!c || !buf || type != ncclDouble || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
  const Pending q{send, buf, count * sizeof(double), peer, c, st};
  if (g_depth > 0) { g_pending.push_back(q); return ncclSuccess; }
  return issue(q);
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t h, hipStream_t st) {
  return p2p(true, const_cast<void *>(buf), count, type, peer, h, st);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t h, hipStream_t st) { return p2p(false, buf, count, type, peer, h, st); }

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t h, hipStream_t st) {
  Comm *c = reinterpret_cast<Comm *>(h);
  if (!c || !sendbuf || !recvbuf || type != ncclDouble || count == 0 || count > (size_t)kReduceMax || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
  const unsigned long long seq = ++c->reduce_seq;
  const int slot = (int)(seq % kReduceSlots), n = (int)count;
  const bool is_max = op == ncclMax;
  if (hipLaunchHostFunc(st, reduce_wait_slot, new Reduce{c, seq, n, is_max}) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpyAsync(c->S->operand[slot][c->rank], sendbuf, count * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(st, reduce_contribute, new Reduce{c, seq, n, is_max}) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(st, reduce_collect, new Reduce{c, seq, n, is_max}) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpyAsync(recvbuf, c->h_red, count * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(st, reduce_finished, new Reduce{c, seq, n, is_max}) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument (fake rccl: doubles only, <= 64 values per all-reduce, <= 8 ranks)";
    case ncclInvalidUsage: return "invalid usage";
    case ncclSystemError: return "system error (shared memory)";
    case ncclUnhandledCudaError: return "HIP error";
    default: return "error";
  }
}

}  // extern "C"
