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
// How the bytes travel: device -> a POSIX shared-memory mailbox -> device, with stream-ordered copies.  The mailboxes are PINNED in
// both processes (hipHostRegister): an asynchronous copy from pageable memory would be staged when it is ENQUEUED, before the
// synchronisation in front of it has seen the message arrive.  Slow by design.  One node only.
//
// Two ways of ordering the copies of two processes (FAKE_RCCL_MODE):
//   values    (default where the device supports it) hipStreamWriteValue64 / hipStreamWaitValue64 on 64-bit sequence words in the pinned
//             shared segment: "message k is in the mailbox", "message k has been taken out", "my operand of all-reduce k is there".  The
//             waits are QUEUE operations: no host thread takes part, so operations on DIFFERENT streams of one process cannot block one
//             another -- as under RCCL, whose waits are device-side too.  The all-reduce itself is a one-wavefront kernel reading the ranks'
//             operands from the pinned segment in rank order.
//   hostfunc  round 4's form: hipLaunchHostFunc callbacks that raise / wait for the same words on the HOST.  The HIP runtime may run every
//             stream's host functions of a process on ONE thread, so a blocking wait on one stream can sit in front of the host function
//             of another stream that would release a peer: a host that spreads its RCCL calls over streams could deadlock here where
//             RCCL would not.  Kept as a variant (and for devices without stream memory operations).
// Injected latency (measurement support: what does an iteration cost when a collective takes as long as it does over xGMI?):
//   FAKE_RCCL_ALLREDUCE_US=<us>        every all-reduce completes <us> microseconds after the last operand arrived
//   FAKE_RCCL_SENDRECV_US_PER_MB=<us>  every send takes <us> microseconds per MB on top of its copies (1e6 / <us> = the link's MB/s)
// (values mode: a one-thread kernel spinning on the device's wall clock in stream order; hostfunc mode: a sleep in the callback).
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
constexpr int kReduceSlots = 4, kReduceMax = 64;
typedef unsigned long long u64;

size_t mailbox_bytes() {  // per ordered pair of ranks: one shared-memory segment each, created when the pair's ranks attach
  static const size_t n = [] { const char *e = getenv("FAKE_RCCL_MAILBOX_MB"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 16) << 20; }();
  return n;
}
double env_us(const char *name) {
  const char *e = getenv(name);
  const double v = e ? atof(e) : 0.0;
  return v > 0 ? v : 0.0;
}
double allreduce_us() { static const double v = env_us("FAKE_RCCL_ALLREDUCE_US"); return v; }
double sendrecv_us_per_mb() { static const double v = env_us("FAKE_RCCL_SENDRECV_US_PER_MB"); return v; }

// every word another process (or this process's device) waits on is a plain 64-bit word at its own cache line; the host side reads and
// writes them through std::atomic_ref-like casts (x86: naturally atomic), the device side through the stream memory operations
struct alignas(64) Word { volatile u64 v; char pad[56]; };
struct Mailbox {
  Word sent, consumed;  // messages written / read so far
  Word bytes;           // size of the message in the mailbox (checked by the receiver: plan mismatch between two ranks)
};
struct Shared {
  std::atomic<int> attached, detached;
  int nranks;
  Word error;                      // raised by a rank (host or device) that found a contract violation; every rank reports it
  Word contributed[kMaxRanks];     // number of the last all-reduce rank r's operand has been written for
  Word finished[kMaxRanks];        // all-reduces rank r has completed
  double operand[kReduceSlots][kMaxRanks][kReduceMax];
  Mailbox box[kMaxRanks][kMaxRanks];  // [src][dst]: the sequence words; the payload lives in a segment of its own
};
inline size_t shared_bytes() { return sizeof(Shared); }
inline u64 load(const Word &w) { return __atomic_load_n(&w.v, __ATOMIC_ACQUIRE); }
inline void store(Word &w, u64 v) { __atomic_store_n(&w.v, v, __ATOMIC_RELEASE); }

struct Comm {
  Shared *S = nullptr;   // host view of the shared segment
  char *Sd = nullptr;    // the same segment as the device addresses it (hipHostGetDevicePointer)
  char *out_box[kMaxRanks] = {nullptr}, *in_box[kMaxRanks] = {nullptr};  // payload of the mailboxes me -> p and p -> me, mapped and pinned here
  int rank = 0, nranks = 1;
  bool values = false;   // FAKE_RCCL_MODE
  long long ticks_per_us = 100;  // device wall clock
  char name[64] = {0};
  u64 reduce_seq = 0;
  u64 sent_to[kMaxRanks] = {0}, recv_from[kMaxRanks] = {0};
  double *h_red = nullptr;  // pinned: the result of an all-reduce on its way back to the device (hostfunc mode)
  std::thread watchdog;     // values mode: nobody on the host waits for a peer, so somebody has to notice one that never answers
  std::atomic<bool> leaving{false};
  template <class T> T *dev(T *host_ptr) const { return reinterpret_cast<T *>(Sd + (reinterpret_cast<char *>(host_ptr) - reinterpret_cast<char *>(S))); }
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
void wait_until(F ok, const char *what, int a = -1, int b = -1, u64 seq = 0) {
  const auto t0 = std::chrono::steady_clock::now();
  while (!ok()) {
    std::this_thread::sleep_for(std::chrono::microseconds(20));
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(wait_limit())) {
      fprintf(stderr, "fake_rccl: pid %d, ranks %d -> %d, operation number %llu:\n", (int)getpid(), a, b, seq);
      die(what);
    }
  }
}
void sleep_us(double us) {
  if (us > 0) std::this_thread::sleep_for(std::chrono::nanoseconds((long long)(us * 1e3)));
}

// ------------------------------------------------------------------ device side of the `values` mode
__global__ void k_delay(long long ticks) {  // injected latency: one thread watching the device's constant-rate wall clock
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
__device__ __forceinline__ u64 sys_load(const volatile u64 *p) { return __hip_atomic_load(const_cast<const u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// the receiver's check that both sides planned the same message
__global__ void k_check_bytes(const volatile u64 *have, u64 want, volatile u64 *error) {
  if (sys_load(have) != want) __hip_atomic_store(const_cast<u64 *>(error), (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// operands [rank][kReduceMax] in the pinned segment -> out[n], reduced in rank order (identical bits on every rank)
__global__ void k_reduce(const double *operands, int nranks, int n, int is_max, double *__restrict__ out) {
  const int i = threadIdx.x;
  if (i >= n) return;
  auto get = [&](int r) { return __builtin_bit_cast(double, sys_load(reinterpret_cast<const volatile u64 *>(operands + (size_t)r * kReduceMax + i))); };
  double a = get(0);
  for (int r = 1; r < nranks; ++r) { const double b = get(r); a = is_max ? (a > b ? a : b) : a + b; }
  out[i] = a;
}
#define FAKE_HIP(call) do { if ((call) != hipSuccess) return ncclUnhandledCudaError; } while (0)
ncclResult_t wait_geq(const Comm *c, hipStream_t st, Word *w, u64 v) {
  FAKE_HIP(hipStreamWaitValue64(st, (void *)&c->dev(w)->v, v, hipStreamWaitValueGte, ~0ull));
  return ncclSuccess;
}
ncclResult_t write_value(const Comm *c, hipStream_t st, Word *w, u64 v) {
  FAKE_HIP(hipStreamWriteValue64(st, (void *)&c->dev(w)->v, v, 0));
  return ncclSuccess;
}
ncclResult_t delay(const Comm *c, hipStream_t st, double us) {
  if (us <= 0) return ncclSuccess;
  hipLaunchKernelGGL(k_delay, dim3(1), dim3(1), 0, st, (long long)(us * (double)c->ticks_per_us));
  FAKE_HIP(hipGetLastError());
  return ncclSuccess;
}
#define FAKE_TRY(call) do { const ncclResult_t r__ = (call); if (r__ != ncclSuccess) return r__; } while (0)

ncclResult_t issue_values(const Pending &q) {
  Comm *c = q.c;
  Shared *S = c->S;
  const int me = c->rank;
  if (q.send) {
    Mailbox &m = S->box[me][q.peer];
    const u64 seq = ++c->sent_to[q.peer];
    if (seq > 1) FAKE_TRY(wait_geq(c, q.st, &m.consumed, seq - 1));  // the receiver has taken the previous message out
    FAKE_TRY(delay(c, q.st, sendrecv_us_per_mb() * (double)q.bytes / 1048576.0));
    FAKE_HIP(hipMemcpyAsync(c->out_box[q.peer], q.buf, q.bytes, hipMemcpyDeviceToHost, q.st));
    FAKE_TRY(write_value(c, q.st, &m.bytes, (u64)q.bytes));
    FAKE_TRY(write_value(c, q.st, &m.sent, seq));
  } else {
    Mailbox &m = S->box[q.peer][me];
    const u64 seq = ++c->recv_from[q.peer];
    FAKE_TRY(wait_geq(c, q.st, &m.sent, seq));
    hipLaunchKernelGGL(k_check_bytes, dim3(1), dim3(1), 0, q.st, &c->dev(&m.bytes)->v, (u64)q.bytes, &c->dev(&S->error)->v);
    FAKE_HIP(hipGetLastError());
    FAKE_HIP(hipMemcpyAsync(q.buf, c->in_box[q.peer], q.bytes, hipMemcpyHostToDevice, q.st));
    FAKE_TRY(write_value(c, q.st, &m.consumed, seq));
  }
  return ncclSuccess;
}
ncclResult_t allreduce_values(Comm *c, const void *sendbuf, void *recvbuf, int n, bool is_max, hipStream_t st) {
  Shared *S = c->S;
  const u64 seq = ++c->reduce_seq;
  const int slot = (int)(seq % kReduceSlots);
  if (seq > kReduceSlots)  // slot seq % 4 was last used by all-reduce seq - 4: every rank must be done with it
    for (int r = 0; r < c->nranks; ++r)
      if (r != c->rank) FAKE_TRY(wait_geq(c, st, &S->finished[r], seq - kReduceSlots));
  FAKE_HIP(hipMemcpyAsync(S->operand[slot][c->rank], sendbuf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  FAKE_TRY(write_value(c, st, &S->contributed[c->rank], seq));
  for (int r = 0; r < c->nranks; ++r)
    if (r != c->rank) FAKE_TRY(wait_geq(c, st, &S->contributed[r], seq));
  FAKE_TRY(delay(c, st, allreduce_us()));
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(64), 0, st, (const double *)c->dev(&S->operand[slot][0][0]), c->nranks, n, is_max ? 1 : 0, (double *)recvbuf);
  FAKE_HIP(hipGetLastError());
  FAKE_TRY(write_value(c, st, &S->finished[c->rank], seq));
  return ncclSuccess;
}

// ------------------------------------------------------------------ the `hostfunc` mode (round 4)
// host functions run in stream order on the stream's callback thread; no HIP calls in here
struct SendDone { Shared *S; int src, dst; size_t bytes; u64 seq; };
void mark_sent(void *p) {
  SendDone *d = static_cast<SendDone *>(p);
  store(d->S->box[d->src][d->dst].bytes, (u64)d->bytes);
  store(d->S->box[d->src][d->dst].sent, d->seq);
  delete d;
}
struct SlotFree { Shared *S; int src, dst; u64 seq; size_t bytes; };
void wait_slot_free(void *p) {  // the receiver has taken message seq - 1 out of the mailbox
  SlotFree *d = static_cast<SlotFree *>(p);
  wait_until([&] { return load(d->S->box[d->src][d->dst].consumed) + 1 >= d->seq; }, "a send waited for the receiver to empty the mailbox", d->src, d->dst, d->seq);
  sleep_us(sendrecv_us_per_mb() * (double)d->bytes / 1048576.0);
  delete d;
}
struct RecvWait { Shared *S; int src, dst; u64 seq; size_t bytes; };
void wait_arrival(void *p) {
  RecvWait *d = static_cast<RecvWait *>(p);
  wait_until([&] { return load(d->S->box[d->src][d->dst].sent) >= d->seq; }, "a receive waited for its message", d->src, d->dst, d->seq);
  if (load(d->S->box[d->src][d->dst].bytes) != (u64)d->bytes) die("a receive's size differs from the matching send's (plan mismatch between two ranks)");
  delete d;
}
void mark_consumed(void *p) {
  RecvWait *d = static_cast<RecvWait *>(p);
  store(d->S->box[d->src][d->dst].consumed, d->seq);
  delete d;
}
struct Reduce { Comm *c; u64 seq; int n; bool is_max; };
void reduce_contribute(void *p) {
  Reduce *d = static_cast<Reduce *>(p);
  store(d->c->S->contributed[d->c->rank], d->seq);
  delete d;
}
void reduce_wait_slot(void *p) {
  Reduce *d = static_cast<Reduce *>(p);
  Shared *S = d->c->S;
  if (d->seq > kReduceSlots)
    wait_until([&] { for (int r = 0; r < S->nranks; ++r) if (load(S->finished[r]) + kReduceSlots < d->seq) return false; return true; },
               "an all-reduce waited for the other ranks to finish an earlier one", d->c->rank, -1, d->seq);
  delete d;
}
void reduce_collect(void *p) {
  Reduce *d = static_cast<Reduce *>(p);
  Shared *S = d->c->S;
  const int slot = (int)(d->seq % kReduceSlots);
  wait_until([&] { for (int r = 0; r < S->nranks; ++r) if (load(S->contributed[r]) < d->seq) return false; return true; },
             "an all-reduce waited for the other ranks' operands", d->c->rank, -1, d->seq);
  sleep_us(allreduce_us());
  for (int i = 0; i < d->n; ++i) {
    double a = S->operand[slot][0][i];
    for (int r = 1; r < S->nranks; ++r) a = d->is_max ? (a > S->operand[slot][r][i] ? a : S->operand[slot][r][i]) : a + S->operand[slot][r][i];  // rank order
    d->c->h_red[i] = a;
  }
  delete d;
}
void reduce_finished(void *p) {
  Reduce *d = static_cast<Reduce *>(p);
  store(d->c->S->finished[d->c->rank], d->seq);
  delete d;
}
ncclResult_t issue_hostfunc(const Pending &q) {
  Shared *S = q.c->S;
  const int me = q.c->rank;
  if (q.send) {
    const u64 seq = ++q.c->sent_to[q.peer];
    FAKE_HIP(hipLaunchHostFunc(q.st, wait_slot_free, new SlotFree{S, me, q.peer, seq, q.bytes}));
    FAKE_HIP(hipMemcpyAsync(q.c->out_box[q.peer], q.buf, q.bytes, hipMemcpyDeviceToHost, q.st));
    FAKE_HIP(hipLaunchHostFunc(q.st, mark_sent, new SendDone{S, me, q.peer, q.bytes, seq}));
  } else {
    const u64 seq = ++q.c->recv_from[q.peer];
    FAKE_HIP(hipLaunchHostFunc(q.st, wait_arrival, new RecvWait{S, q.peer, me, seq, q.bytes}));
    FAKE_HIP(hipMemcpyAsync(q.buf, q.c->in_box[q.peer], q.bytes, hipMemcpyHostToDevice, q.st));
    FAKE_HIP(hipLaunchHostFunc(q.st, mark_consumed, new RecvWait{S, q.peer, me, seq, q.bytes}));
  }
  return ncclSuccess;
}
ncclResult_t allreduce_hostfunc(Comm *c, const void *sendbuf, void *recvbuf, int n, bool is_max, hipStream_t st) {
  const u64 seq = ++c->reduce_seq;
  const int slot = (int)(seq % kReduceSlots);
  FAKE_HIP(hipLaunchHostFunc(st, reduce_wait_slot, new Reduce{c, seq, n, is_max}));
  FAKE_HIP(hipMemcpyAsync(c->S->operand[slot][c->rank], sendbuf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  FAKE_HIP(hipLaunchHostFunc(st, reduce_contribute, new Reduce{c, seq, n, is_max}));
  FAKE_HIP(hipLaunchHostFunc(st, reduce_collect, new Reduce{c, seq, n, is_max}));
  FAKE_HIP(hipMemcpyAsync(recvbuf, c->h_red, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  FAKE_HIP(hipLaunchHostFunc(st, reduce_finished, new Reduce{c, seq, n, is_max}));
  return ncclSuccess;
}

// values mode: the waits sit in the device's queues, where a peer that never arrives would leave them for ever (and the host inside
// the next hipStreamSynchronize).  The hostfunc mode ends such a run after CUP3D_FAKE_RCCL_WAIT seconds from inside the waiting
// callback; here a thread per communicator does: operations of this rank outstanding and none of them completed for that long.
void watch(Comm *c) {
  auto done = [&] {
    u64 n = load(c->S->finished[c->rank]);
    for (int p = 0; p < c->nranks; ++p)
      if (p != c->rank) n += load(c->S->box[c->rank][p].sent) + load(c->S->box[p][c->rank].consumed);
    return n;
  };
  auto issued = [&] {
    u64 n = __atomic_load_n(&c->reduce_seq, __ATOMIC_RELAXED);
    for (int p = 0; p < c->nranks; ++p) n += __atomic_load_n(&c->sent_to[p], __ATOMIC_RELAXED) + __atomic_load_n(&c->recv_from[p], __ATOMIC_RELAXED);
    return n;
  };
  u64 last = done();
  auto t_last = std::chrono::steady_clock::now();
  while (!c->leaving.load()) {
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
    const u64 d = done();
    const auto now = std::chrono::steady_clock::now();
    if (d != last || d >= issued()) { last = d; t_last = now; continue; }
    if (now - t_last > std::chrono::seconds(wait_limit())) {
      fprintf(stderr, "fake_rccl: pid %d rank %d: %llu operations issued, %llu completed, none for %d s: all-reduces %llu of %llu", (int)getpid(), c->rank, issued(), d, wait_limit(),
              load(c->S->finished[c->rank]), c->reduce_seq);
      for (int p = 0; p < c->nranks; ++p)
        if (p != c->rank) fprintf(stderr, "; peer %d: sent %llu of %llu, received %llu of %llu", p, load(c->S->box[c->rank][p].sent), c->sent_to[p], load(c->S->box[p][c->rank].consumed), c->recv_from[p]);
      fprintf(stderr, "\n");
      die("a peer never answered (stream-memory-operation mode)");
    }
  }
}

ncclResult_t issue(const Pending &q) {
  if (q.bytes > mailbox_bytes()) die("message larger than the fake mailbox (FAKE_RCCL_MAILBOX_MB)");
  if (load(q.c->S->error)) die("a rank reported a contract violation (a receive whose size differs from the matching send's: plan mismatch between two ranks)");
  return q.c->values ? issue_values(q) : issue_hostfunc(q);
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
  // which of the two orderings: FAKE_RCCL_MODE, else stream memory operations where the device has them
  int dev = 0, can = 0, khz = 0;
  FAKE_HIP(hipGetDevice(&dev));
  (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev);
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) c->ticks_per_us = khz / 1000 > 0 ? khz / 1000 : 1;
  const char *mode = getenv("FAKE_RCCL_MODE");
  c->values = mode ? strcmp(mode, "hostfunc") != 0 : can != 0;
  if (c->values && !can) { fprintf(stderr, "fake_rccl: FAKE_RCCL_MODE=values, but this device has no stream memory operations\n"); return ncclInvalidUsage; }
  if (getenv("FAKE_RCCL_VERBOSE") && rank == 0)
    fprintf(stderr, "fake_rccl: %d ranks, mode %s, all-reduce +%g us, send +%g us/MB, wall clock %lld ticks/us\n", nranks, c->values ? "values" : "hostfunc", allreduce_us(),
            sendrecv_us_per_mb(), c->ticks_per_us);
  if (hipHostMalloc((void **)&c->h_red, kReduceMax * sizeof(double), hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
  if (hipHostRegister(c->S, shared_bytes(), hipHostRegisterDefault) != hipSuccess) return ncclUnhandledCudaError;  // operands and sequence words live in it
  if (hipHostGetDevicePointer((void **)&c->Sd, c->S, 0) != hipSuccess) return ncclUnhandledCudaError;
  for (int p2 = 0; p2 < nranks; ++p2) {  // the two mailboxes of every pair I am part of: whoever comes first creates the segment
    if (p2 == rank) continue;
    for (int dir = 0; dir < 2; ++dir) {
      char seg[96];
      snprintf(seg, sizeof seg, "%s_%d_%d", id.internal, dir ? p2 : rank, dir ? rank : p2);
      const int sfd = shm_open(seg, O_CREAT | O_RDWR, 0600);
      if (sfd < 0 || ftruncate(sfd, (off_t)mailbox_bytes()) != 0) return ncclSystemError;
      void *m = mmap(nullptr, mailbox_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, sfd, 0);
      close(sfd);
      if (m == MAP_FAILED) return ncclSystemError;
      if (hipHostRegister(m, mailbox_bytes(), hipHostRegisterDefault) != hipSuccess) return ncclUnhandledCudaError;
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
  if (c->values) c->watchdog = std::thread(watch, c);
  *out = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

static ncclResult_t leave(ncclComm_t h, bool drain) {
  Comm *c = reinterpret_cast<Comm *>(h);
  if (!c) return ncclInvalidArgument;
  if (drain) (void)hipDeviceSynchronize();  // nothing of this rank is still on its way through the segments that go away now (not on abort: a peer may never answer)
  c->leaving.store(true);
  if (c->watchdog.joinable()) c->watchdog.join();
  for (int p2 = 0; p2 < c->nranks; ++p2)
    for (int dir = 0; dir < 2; ++dir) {
      char *m = (dir ? c->in_box : c->out_box)[p2];
      if (!m) continue;
      (void)hipHostUnregister(m);
      munmap(m, mailbox_bytes());
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
ncclResult_t ncclCommDestroy(ncclComm_t h) { return leave(h, true); }
ncclResult_t ncclCommAbort(ncclComm_t h) { return leave(h, false); }

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
  if (load(c->S->error)) die("a rank reported a contract violation (plan mismatch between two ranks)");
  return c->values ? allreduce_values(c, sendbuf, recvbuf, (int)count, op == ncclMax, st) : allreduce_hostfunc(c, sendbuf, recvbuf, (int)count, op == ncclMax, st);
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
