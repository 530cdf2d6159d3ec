// Device block store, host<->device transfers, pointwise operators.  See sim.hpp.
#include <atomic>
#include <cstdarg>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

#include "sim.hpp"
#include "tile.hpp"

namespace cup3d {

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return CUP3D_EDEVICE;
}

// ------------------------------------------------------------------ stream / profile
static hipStream_t g_stream = nullptr;
hipStream_t stream() { return g_stream; }

struct ProfileRec { int idx; hipEvent_t a, b; };
static bool g_prof_on = false;
bool profile_on() { return g_prof_on; }
static std::vector<std::string> g_prof_names;
static std::vector<long> g_prof_launches;
static std::vector<double> g_prof_ms;
static std::vector<ProfileRec> g_prof_open;
static std::vector<hipEvent_t> g_event_pool;  // events are reused: creating two per launch costs more host time than the launch
// one lock around the profiler's tables and the event pool: the ranks of the in-process test communicator are host threads that all
// open ProfileScopes (uncontended in a one-thread process: ~20 ns per scope, and only while profiling is on)
static std::mutex g_prof_mutex;
static hipEvent_t pooled_event() {
  if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}

static void profile_drain() {
  for (auto &r : g_prof_open) {
    float ms = 0;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      g_prof_ms[r.idx] += ms;
      g_prof_launches[r.idx] += 1;
    }
    g_event_pool.push_back(r.a);
    g_event_pool.push_back(r.b);
  }
  g_prof_open.clear();
}
ProfileScope::ProfileScope(const char *name) : ProfileScope(name, g_stream) {}
ProfileScope::ProfileScope(const char *name, hipStream_t on) : idx(-1), start(nullptr), st(on) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mutex);
  for (size_t i = 0; i < g_prof_names.size(); ++i)
    if (g_prof_names[i] == name) idx = (int)i;
  if (idx < 0) {
    idx = (int)g_prof_names.size();
    g_prof_names.push_back(name);
    g_prof_launches.push_back(0);
    g_prof_ms.push_back(0);
  }
  if (!(start = pooled_event())) { idx = -1; return; }
  hipEventRecord(start, st);
}
ProfileScope::~ProfileScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mutex);
  hipEvent_t stop = pooled_event();
  if (!stop) { g_event_pool.push_back(start); return; }
  hipEventRecord(stop, st);
  g_prof_open.push_back({idx, start, stop});
  if (g_prof_open.size() > 4096) profile_drain();
}

#ifdef CUP3D_TESTING
static std::map<std::string, int> g_debug_opts;
#endif
int not_in_release(const char *what) {
  set_error("%s is test / tuning support: load libcup3d_hip_testing.so (built with -DCUP3D_TESTING)", what);
  return CUP3D_ESTATE;
}
// ---- cup3d_stats_*: process-wide counters (the ranks of the in-process test communicator are threads, hence atomics)
static std::atomic<long> g_st_halo{0}, g_st_halo_bytes{0}, g_st_allreduce{0}, g_st_waits{0}, g_st_wait_ns{0}, g_st_iters{0}, g_st_h2d{0}, g_st_d2h{0};
static void stats_field_transfer(bool up, size_t bytes) { (up ? g_st_h2d : g_st_d2h) += (long)bytes; }
void stats_host_wait(double seconds) { g_st_waits++; g_st_wait_ns += (long)(seconds * 1e9); }
void stats_solver_iterations(long n) { g_st_iters += n; }
void stats_halo(size_t bytes_sent) { g_st_halo++; g_st_halo_bytes += (long)bytes_sent; }
void stats_allreduce() { g_st_allreduce++; }

#ifdef CUP3D_TESTING
int debug_option(const char *name) {
  auto it = g_debug_opts.find(name);
  return it == g_debug_opts.end() ? 0 : it->second;
}
#endif

// ------------------------------------------------------------------ Sim
GridDev Sim::gdev(bool boundary_only, bool inner_only) const {
  GridDev g;
  g.nbr = d_nbr;
  g.h = grid->h;
  g.hb = d_hb;
  g.flux = d_flux;
  g.raw = d_raw_mask;
  if (boundary_only) {
    g.list = d_boundary;
    g.nblocks = (int)grid->boundary.size();
  } else if (inner_only) {
    g.list = d_inner;
    g.nblocks = (int)grid->inner.size();
  } else {
    g.list = nullptr;
    g.nblocks = (int)nb;
  }
  g.chunk = (g.nblocks + 7) / 8;
  return g;
}
GridDev Sim::gdev_list(const int32_t *l, unsigned n) const {
  GridDev g = gdev();
  g.list = l;
  g.nblocks = (int)n;
  g.chunk = (g.nblocks + 7) / 8;
  return g;
}
double *Sim::field(int id, int *ncomp) const {
  switch (id) {
    case CUP3D_FIELD_CHI: *ncomp = 1; return chi;
    case CUP3D_FIELD_PRES: *ncomp = 1; return pres;
    case CUP3D_FIELD_LHS: *ncomp = 1; return lhs;
    case CUP3D_FIELD_VEL: *ncomp = 3; return vel;
    case CUP3D_FIELD_TMPV: *ncomp = 3; return tmpV;
  }
  *ncomp = 0;
  return nullptr;
}
int sim_alloc(double **p, size_t n, Sim *s) {
  CUP3D_HIP(hipMalloc((void **)p, n * sizeof(double)));
  CUP3D_HIP(hipMemsetAsync(*p, 0, n * sizeof(double), g_stream));
  s->bytes += n * sizeof(double);
  return CUP3D_OK;
}

// ------------------------------------------------------------------ layout conversion
// reference block memory [blk][cell][c]  <->  device slab [blk][c][cell]
__global__ void __launch_bounds__(256) k_aos_to_soa(const double *__restrict__ aos, double *__restrict__ soa, long ncell_total, int nc) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ncell_total * nc) return;
  const long blk = i / (512L * nc);
  const int r = (int)(i - blk * 512L * nc);
  const int c = r / 512, cell = r - c * 512;
  soa[i] = aos[(blk * 512 + cell) * nc + c];
}
__global__ void __launch_bounds__(256) k_soa_to_aos(const double *__restrict__ soa, double *__restrict__ aos, long ncell_total, int nc) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ncell_total * nc) return;
  const long blk = i / (512L * nc);
  const int r = (int)(i - blk * 512L * nc);
  const int cell = r / nc, c = r - cell * nc;
  aos[i] = soa[(blk * nc + c) * 512 + cell];
}
// the same for a list of block slots (partial transfers): staged block b <-> field slot slots[b]
__global__ void __launch_bounds__(256) k_aos_to_soa_list(const double *__restrict__ aos, double *__restrict__ field, const int32_t *__restrict__ slots,
                                                         long ncell_total, int nc) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ncell_total * nc) return;
  const long blk = i / (512L * nc);
  const int r = (int)(i - blk * 512L * nc);
  const int c = r / 512, cell = r - c * 512;
  field[((size_t)slots[blk] * nc + c) * 512 + cell] = aos[(blk * 512 + cell) * nc + c];
}
__global__ void __launch_bounds__(256) k_soa_to_aos_list(const double *__restrict__ field, double *__restrict__ aos, const int32_t *__restrict__ slots,
                                                         long ncell_total, int nc) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ncell_total * nc) return;
  const long blk = i / (512L * nc);
  const int r = (int)(i - blk * 512L * nc);
  const int cell = r / nc, c = r - cell * nc;
  aos[i] = field[((size_t)slots[blk] * nc + c) * 512 + cell];
}
// wrapping sum of the 64-bit patterns of n doubles (cup3d_sim_checksum): one atomic per workgroup; integer addition commutes, so the
// result does not depend on the launch geometry or on how the blocks are spread over ranks
__global__ void __launch_bounds__(256) k_checksum(const double *__restrict__ p, long n, unsigned long long *__restrict__ out) {
  __shared__ unsigned long long part[256];
  unsigned long long a = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a += __builtin_bit_cast(unsigned long long, p[i]);
  part[threadIdx.x] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, part[0]);
}
static unsigned stride_groups(long n);
int checksum_array(Sim *s, const double *p, long n, unsigned long long *sum) {
  unsigned long long *d = reinterpret_cast<unsigned long long *>(s->d_red + kRedChecksum);  // its own slot of d_red (RedSlot)
  CUP3D_HIP(hipMemsetAsync(d, 0, sizeof *d, g_stream));
  hipLaunchKernelGGL(k_checksum, dim3(stride_groups(n)), dim3(256), 0, g_stream, p, n, d);
  CUP3D_HIP(hipGetLastError());
  CUP3D_HIP(hipMemcpyAsync(sum, d, sizeof *d, hipMemcpyDeviceToHost, g_stream));
  CUP3D_HIP(hipStreamSynchronize(g_stream));
  return CUP3D_OK;
}
__global__ void __launch_bounds__(256) k_fill(double *__restrict__ p, long n, double v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}
// findMaxU, main.cpp:8603-8623
__global__ void __launch_bounds__(256) k_max_u(const double *__restrict__ vel, long nblk, double u0, double u1, double u2, double *__restrict__ partial) {
  __shared__ double sh[4];
  double m = 0;
  const long n = nblk * 1536;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)((i / 512) % 3);
    const double uinf = c == 0 ? u0 : (c == 1 ? u1 : u2);
    m = fmax(m, fabs(vel[i] + uinf));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
__global__ void __launch_bounds__(256) k_max_final(const double *__restrict__ partial, int n, double *__restrict__ out) {
  __shared__ double sh[4];
  double m = 0;
  for (int i = threadIdx.x; i < n; i += 256) m = fmax(m, partial[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
// ExternalForcing::operator(), main.cpp:10581-10596 : component 0 += gradPdt
__global__ void __launch_bounds__(256) k_forcing(double *__restrict__ vel, long nblk, double g) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nblk * 512; i += (long)gridDim.x * 256) {
    const long blk = i >> 9;
    vel[blk * 1536 + (i & 511)] += g;
  }
}

// fn(i) for i in [0, n) on the calling thread plus up to 15 helpers (contiguous ranges)
template <class F>
static void host_parallel(size_t n, F fn) {
  unsigned hw = std::thread::hardware_concurrency();
  size_t nt = std::min<size_t>(std::min<size_t>(16, hw ? hw : 1), (n + 255) / 256);
  if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  auto range = [&](size_t t) { const size_t b = n * t / nt, e = n * (t + 1) / nt; for (size_t i = b; i < e; ++i) fn(i); };
  for (size_t t = 1; t < nt; ++t) th.emplace_back(range, t);
  range(0);
  for (auto &t : th) t.join();
}
static unsigned stride_groups(long n) {
  long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace cup3d

using namespace cup3d;

extern "C" {

const char *cup3d_last_error(void) { return g_err; }
const char *cup3d_version(void) { return "cup3d-hip 0.1 (gfx950)"; }

int cup3d_device_count(int *n) {
  if (!n) return CUP3D_EINVAL;
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess) { *n = 0; return hip_fail(e, "hipGetDeviceCount", __FILE__, __LINE__); }
  return CUP3D_OK;
}
int cup3d_device_init(int device) {
  int n = 0;
  if (cup3d_device_count(&n) != CUP3D_OK || n == 0) {
    set_error("no HIP device visible: the cup3d hot path runs on MI355X (gfx950) only and has no CPU fallback");
    return CUP3D_EDEVICE;
  }
  if (device < 0 || device >= n) { set_error("device %d out of range (%d visible)", device, n); return CUP3D_EINVAL; }
  CUP3D_HIP(hipSetDevice(device));
  hipDeviceProp_t p;
  CUP3D_HIP(hipGetDeviceProperties(&p, device));
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, p.gcnArchName);
    return CUP3D_EDEVICE;
  }
  return CUP3D_OK;
}
int cup3d_set_stream(void *s) { g_stream = (hipStream_t)s; return CUP3D_OK; }
int cup3d_device_synchronize(void) {
  CUP3D_HIP(hipStreamSynchronize(g_stream));
  CUP3D_HIP(hipDeviceSynchronize());
  return CUP3D_OK;
}

// TEST / TUNING SUPPORT: select kernel variants (A/B timing, ablations); 0 = production
#ifdef CUP3D_TESTING  // test / tuning support: not in the release library at all
int cup3d_debug_set_option(const char *name, int value) {
  if (!name) return CUP3D_EINVAL;
  g_debug_opts[name] = value;
  return CUP3D_OK;
}
#endif

int cup3d_profile_enable(int on) { g_prof_on = on != 0; return CUP3D_OK; }
int cup3d_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mutex);
  profile_drain();
  std::fill(g_prof_ms.begin(), g_prof_ms.end(), 0.0);
  std::fill(g_prof_launches.begin(), g_prof_launches.end(), 0L);
  return CUP3D_OK;
}
int cup3d_profile_read(cup3d_profile_entry *e, int max, int *n) {
  std::lock_guard<std::mutex> lk(g_prof_mutex);
  profile_drain();
  const int m = (int)g_prof_names.size();
  if (n) *n = m;
  for (int i = 0; i < m && i < max; ++i) {
    snprintf(e[i].name, sizeof e[i].name, "%s", g_prof_names[i].c_str());
    e[i].launches = g_prof_launches[i];
    e[i].total_ms = g_prof_ms[i];
  }
  return CUP3D_OK;
}

static int sim_build(Sim *s, const Grid *g) {
  s->grid = g;
  s->nb = g->nblocks();
  s->nvis = (int64_t)g->Z.size();  // rank views: local + ghost blocks
  const size_t nb = (size_t)s->nb, nv = (size_t)s->nvis;
  const bool view = g->n_local >= 0;
  int rc;
  auto up = [&](int32_t **d, const std::vector<int32_t> &v) -> int {
    if (v.empty()) { *d = nullptr; return CUP3D_OK; }
    CUP3D_HIP(hipMalloc((void **)d, v.size() * sizeof(int32_t)));
    CUP3D_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return CUP3D_OK;
  };
#define A(ptr, n) if ((rc = sim_alloc(&(ptr), (n), s)) != CUP3D_OK) return rc;
  A(s->vel, nv * 1536) A(s->vel2, nv * 1536) A(s->tmpV, nv * 1536)
  A(s->pres, nv * 512) A(s->lhs, nv * 512) A(s->chi, nv * 512) A(s->pold, nv * 512)
  s->max_groups = 4096;
  A(s->d_partials, (size_t)s->max_groups * 8 + nb) A(s->d_red, kRedSize)
  CUP3D_HIP(hipHostMalloc((void **)&s->h_red, 17 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));  // 16 totals + the sequence word of Reducer
  memset(s->h_red, 0, 17 * sizeof(double));
  CUP3D_HIP(hipHostGetDevicePointer((void **)&s->h_red_dev, s->h_red, 0));
  CUP3D_HIP(hipMalloc((void **)&s->d_counters, 4 * sizeof(unsigned)));
  CUP3D_HIP(hipMemsetAsync(s->d_counters, 0, 4 * sizeof(unsigned), g_stream));
  if ((rc = up(&s->d_nbr, g->nbr))) return rc;
  if (g->nranks > 1 && ((rc = up(&s->d_inner, g->inner)) || (rc = up(&s->d_boundary, g->boundary)) || (rc = up(&s->d_send_faces, g->send_faces)))) return rc;
  if (g->nranks > 1 || debug_option("force_allreduce")) {  // ("force_allreduce", testing build: ONE rank whose scalars go through the communicator -- latency measurements)
    // the communication stream outranks the compute stream: its pack kernels, RCCL's send / receive kernels and the one-thread
    // recurrence steps are dispatched ahead of the tens of thousands of loop-kernel workgroups queued on the compute stream, so an
    // exchange started with the inner blocks' pass really runs beside it instead of behind it
    int prio_least = 0, prio_greatest = 0;
    CUP3D_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    CUP3D_HIP(hipStreamCreateWithPriority(&s->comm_stream, hipStreamNonBlocking, prio_greatest));
  }
  if (g->nranks > 1 && !g->multilevel) {
    const size_t slab = 3 * 3 * 64;  // widest exchange: 3 components x 3 layers
    if (g->n_recv_faces) A(s->halo_recv, (size_t)g->n_recv_faces * slab)
    if (!g->send_faces.empty()) A(s->halo_send, g->send_faces.size() * slab)
  }
  if (g->multilevel) {
    // ghost slabs are produced for the LOCAL interface faces only (a view also lists the fine faces of ghost blocks, whose
    // flux arrays arrive from their owners)
    const int64_t nlf = view ? g->n_local_faces : g->n_amr_faces();
    std::vector<int32_t> lr, lp;
    // the faces of inner blocks first: their slabs are produced while the ghost blocks of a view are still travelling (halo_begin)
    std::vector<char> waits(std::max<size_t>(nb, 1), 0);
    if (view) for (int32_t b : g->boundary) waits[(size_t)b] = 1;
    for (int pass = 0; pass < 2; ++pass) {
      for (int64_t e = 0; e < nlf; ++e)
        if (waits[(size_t)(g->amr_faces[2 * e] / 6)] == pass) (g->amr_faces[2 * e + 1] ? lr : lp).push_back((int32_t)e);
      if (pass == 0) { s->n_restrict_inner = (unsigned)lr.size(); s->n_prolong_inner = (unsigned)lp.size(); }
    }
    s->n_restrict = (unsigned)lr.size();
    s->n_prolong = (unsigned)lp.size();
    if ((rc = up(&s->d_amr_faces, g->amr_faces)) || (rc = up(&s->d_amr_fine, g->amr_fine)) || (rc = up(&s->d_nbr27, g->nbr27)) ||
        (rc = up(&s->d_index, g->index)) || (rc = up(&s->d_restrict_list, lr)) || (rc = up(&s->d_prolong_list, lp)) ||
        (rc = up(&s->d_fix_list[0], g->fix_faces[0])) || (rc = up(&s->d_fix_list[1], g->fix_faces[1])) || (rc = up(&s->d_fix_list[2], g->fix_faces[2])))
      return rc;
    {  // the corrected faces grouped by block (k_flux_fix_blocks): blocks in slot order, faces x-, x+, y-, y+, z-, z+
      std::vector<int32_t> of_slot((size_t)std::max<int64_t>(nv, 1), -1), tab;
      for (int d = 0; d < 3; ++d)
        for (int32_t e : g->fix_faces[d]) {
          const int32_t sf = g->amr_faces[2 * (size_t)e], slot = sf / 6, f = sf % 6;
          if (of_slot[(size_t)slot] < 0) { of_slot[(size_t)slot] = (int32_t)(tab.size() / 6); tab.insert(tab.end(), 6, -1); }
          tab[(size_t)of_slot[(size_t)slot] * 6 + f] = e;
        }
      s->n_fix_blocks = (unsigned)(tab.size() / 6);
      if ((rc = up(&s->d_fix_blocks, tab))) return rc;
    }
    A(s->d_hb, nv)
    CUP3D_HIP(hipMemcpy(s->d_hb, g->hb.data(), nv * sizeof(double), hipMemcpyHostToDevice));
    {
      std::vector<unsigned char> mask(std::max<size_t>(nb, 1), 0);
      std::vector<int32_t> raw;
      for (int64_t e = 0; e < nlf; ++e)
        if (g->amr_faces[2 * e + 1] == 1) mask[(size_t)(g->amr_faces[2 * e] / 6)] = 1;
      for (size_t b = 0; b < nb; ++b) if (mask[b]) raw.push_back((int32_t)b);
      s->n_raw = (unsigned)raw.size();
      CUP3D_HIP(hipMalloc((void **)&s->d_raw_mask, mask.size()));
      CUP3D_HIP(hipMemcpy(s->d_raw_mask, mask.data(), mask.size(), hipMemcpyHostToDevice));
      if ((rc = up(&s->d_raw_list, raw))) return rc;
    }
    if (!view) {  // one rank: which blocks have an interface face, which have none
      std::vector<int32_t> iface, plain;
      for (size_t b = 0; b < nb; ++b) {
        bool any = false;
        for (int f = 0; f < 6; ++f) any = any || g->nbr[6 * b + f] >= kNbrHalo;
        (any ? iface : plain).push_back((int32_t)b);
        if ((int32_t)b == g->corner_slot) s->corner_is_plain = !any;
      }
      s->n_iface = (unsigned)iface.size();
      s->n_plain = (unsigned)plain.size();
      if ((rc = up(&s->d_iface_list, iface)) || (rc = up(&s->d_plain_list, plain))) return rc;
    }
    const size_t ne = (size_t)std::max<int64_t>(g->n_amr_faces(), 1);
    // ghost slabs: widest use = 3 components x 3 layers; the pressure RHS keeps a second set (udef) behind the first
    A(s->halo_recv, ne * 9 * 64)
    A(s->d_flux, ne * 3 * 64)
    if (view) {
      if ((rc = up(&s->d_send_blocks, g->send_blocks)) || (rc = up(&s->d_send_flux, g->send_flux_faces))) return rc;
      const size_t need = std::max(g->send_blocks.size() * 1536, g->send_flux_faces.size() * 3 * 64);
      if (need) A(s->halo_send, need)
      // sub-box form of the ghost-block exchange: boxes + packed-message offsets of both width classes, one staging buffer
      size_t most = 0;
      for (int k = 0; k < 2 && !g->send_cells[k].empty(); ++k) {
        auto upload = [&](const std::vector<uint8_t> &box, unsigned char **d_box, long long **d_off, size_t *total) -> int {
          std::vector<long long> off(box.size() / 6 + 1, 0);
          for (size_t i = 0; i < box.size() / 6; ++i)
            off[i + 1] = off[i] + (long long)(box[6 * i + 3] - box[6 * i]) * (box[6 * i + 4] - box[6 * i + 1]) * (box[6 * i + 5] - box[6 * i + 2]);
          *total = (size_t)off.back();
          CUP3D_HIP(hipMalloc((void **)d_box, std::max<size_t>(box.size(), 1)));
          CUP3D_HIP(hipMalloc((void **)d_off, off.size() * sizeof(long long)));
          if (!box.empty()) CUP3D_HIP(hipMemcpy(*d_box, box.data(), box.size(), hipMemcpyHostToDevice));
          CUP3D_HIP(hipMemcpy(*d_off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice));
          return CUP3D_OK;
        };
        size_t ts = 0, tr = 0;
        if ((rc = upload(g->send_box[k], &s->d_send_box[k], &s->d_send_off[k], &ts)) || (rc = upload(g->ghost_box[k], &s->d_ghost_box[k], &s->d_ghost_off[k], &tr))) return rc;
        most = std::max(most, tr);
      }
      if (most) A(s->box_recv, most * 3)
    }
  }
#undef A
  CUP3D_HIP(hipEventCreateWithFlags(&s->ev_a, hipEventDisableTiming));
  CUP3D_HIP(hipEventCreateWithFlags(&s->ev_b, hipEventDisableTiming));
  CUP3D_HIP(hipEventCreateWithFlags(&s->ev_h1, hipEventDisableTiming));
  CUP3D_HIP(hipEventCreateWithFlags(&s->ev_h2, hipEventDisableTiming));
  CUP3D_HIP(hipEventCreateWithFlags(&s->ev_m, hipEventDisableTiming));
  if (g->nranks > 1) {
    CUP3D_HIP(hipEventCreateWithFlags(&s->ev_vc_pack, hipEventDisableTiming));
    CUP3D_HIP(hipEventCreateWithFlags(&s->ev_vc_done, hipEventDisableTiming));
  }
  CUP3D_HIP(hipStreamSynchronize(g_stream));
  return CUP3D_OK;
}

}  // extern "C"

namespace cup3d {
// The part of a Sim the face-slab exchange needs, for a grid that holds no fields of its own: the coarse levels of the multigrid
// preconditioner exchange their iterates through halo_exchange() like every other kernel's input (multigrid.hip).  The communication
// stream is the finest level's (every RCCL call of a rank goes through one stream, in one order).
Sim *sim_comm_only(const Grid *g, hipStream_t comm_stream) {
  Sim *s = new Sim();
  s->grid = g;
  s->nb = s->nvis = g->nblocks();
  s->comm_stream = comm_stream;
  bool ok = true;
  auto up = [&](int32_t **d, const std::vector<int32_t> &v) {
    if (v.empty()) return;
    ok = ok && hipMalloc((void **)d, v.size() * sizeof(int32_t)) == hipSuccess && hipMemcpy(*d, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
  };
  up(&s->d_send_faces, g->send_faces);
  if (g->n_recv_faces) ok = ok && hipMalloc((void **)&s->halo_recv, (size_t)g->n_recv_faces * 64 * sizeof(double)) == hipSuccess;
  if (!g->send_faces.empty()) ok = ok && hipMalloc((void **)&s->halo_send, g->send_faces.size() * 64 * sizeof(double)) == hipSuccess;
  hipEvent_t *ev[] = {&s->ev_h1, &s->ev_h2, &s->ev_vc_pack, &s->ev_vc_done};
  for (hipEvent_t *e : ev) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  if (!ok) { sim_comm_only_destroy(s); return nullptr; }
  return s;
}
void sim_comm_only_destroy(Sim *s) {
  if (!s) return;
  vcomm_unregister(s);
  if (s->d_send_faces) hipFree(s->d_send_faces);
  if (s->halo_recv) hipFree(s->halo_recv);
  if (s->halo_send) hipFree(s->halo_send);
  hipEvent_t ev[] = {s->ev_h1, s->ev_h2, s->ev_vc_pack, s->ev_vc_done};
  for (hipEvent_t e : ev) if (e) hipEventDestroy(e);
  delete s;  // the communication stream belongs to the finest level's Sim
}
}  // namespace cup3d

extern "C" {

int cup3d_sim_create(const cup3d_grid_t *gh, cup3d_sim_t **out) {
  if (!gh || !out) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("cup3d_sim_create: no HIP device (call cup3d_device_init first)"); return CUP3D_EDEVICE; }
  if (g->nranks > 1 && !comm() && !virtual_ranks() && !host_transport()) { set_error("grid spans %d ranks but cup3d_comm_init was not called", g->nranks); return CUP3D_ESTATE; }
  Sim *s = new Sim();
  const int rc = sim_build(s, g);
  if (rc != CUP3D_OK) {  // whatever was allocated so far goes back (an out-of-memory at 512^3 must not strand the slabs already made)
    cup3d_sim_destroy(reinterpret_cast<cup3d_sim_t *>(s));
    return rc;
  }
  vcomm_register(s);
  *out = reinterpret_cast<cup3d_sim_t *>(s);
  return CUP3D_OK;
}

static void release_stage(Sim *s);
void cup3d_sim_destroy(cup3d_sim_t *h) {
  if (!h) return;
  Sim *s = reinterpret_cast<Sim *>(h);
  vcomm_unregister(s);
  hipStreamSynchronize(g_stream);
  mg_destroy(s);
  double *ptrs[] = {s->vel, s->vel2, s->tmpV, s->pres, s->lhs, s->chi, s->pold, s->d_partials, s->d_red, s->halo_recv, s->halo_send, s->d_block_dots,
                    s->d_hb, s->d_flux};
  for (double *p : ptrs) if (p) hipFree(p);
  for (double *p : s->sv) if (p) hipFree(p);
  if (s->h_red) hipHostFree(s->h_red);
  if (s->d_ctl) hipFree(s->d_ctl);
  if (s->h_ctl) hipHostFree(s->h_ctl);
  if (s->d_counters) hipFree(s->d_counters);
  if (s->d_cg_iters) hipFree(s->d_cg_iters);
  if (s->d_arrive_sums) hipFree(s->d_arrive_sums);
  if (s->d_arrive) hipFree(s->d_arrive);
  if (s->d_loop_sums) hipFree(s->d_loop_sums);
  if (s->h_early_fail) hipHostFree(s->h_early_fail);
  release_stage(s);
  int32_t *ip[] = {s->d_nbr, s->d_inner, s->d_boundary, s->d_send_faces, s->d_amr_faces, s->d_amr_fine, s->d_nbr27, s->d_index,
                   s->d_restrict_list, s->d_prolong_list, s->d_fix_list[0], s->d_fix_list[1], s->d_fix_list[2], s->d_fix_blocks, s->d_send_blocks, s->d_send_flux, s->d_raw_list, s->d_iface_list, s->d_plain_list};
  if (s->d_raw_mask) hipFree(s->d_raw_mask);
  for (int k = 0; k < 2; ++k) {
    void *bp[] = {s->d_send_box[k], s->d_ghost_box[k], s->d_send_off[k], s->d_ghost_off[k]};
    for (void *p : bp) if (p) hipFree(p);
  }
  if (s->box_recv) hipFree(s->box_recv);
  for (int32_t *p : ip) if (p) hipFree(p);
  if (s->comm_stream) hipStreamDestroy(s->comm_stream);
  if (s->ev_a) hipEventDestroy(s->ev_a);
  if (s->ev_b) hipEventDestroy(s->ev_b);
  if (s->ev_h1) hipEventDestroy(s->ev_h1);
  if (s->ev_h2) hipEventDestroy(s->ev_h2);
  if (s->ev_m) hipEventDestroy(s->ev_m);
  if (s->ev_vc_pack) hipEventDestroy(s->ev_vc_pack);
  if (s->ev_vc_done) hipEventDestroy(s->ev_vc_done);
  delete s;
}
size_t cup3d_sim_device_bytes(const cup3d_sim_t *h) { return h ? reinterpret_cast<const Sim *>(h)->bytes : 0; }

int cup3d_sim_device_ptr(cup3d_sim_t *h, int field, void **ptr) {
  if (!h || !ptr) return CUP3D_EINVAL;
  int nc;
  double *p = reinterpret_cast<Sim *>(h)->field(field, &nc);
  if (!p) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  *ptr = p;
  return CUP3D_OK;
}

static int mark_written(Sim *s, int field) {
  if (field == CUP3D_FIELD_CHI) s->chi_nonzero = true;   // obstacles present: KernelPressureRHS must read chi/udef
  if (field == CUP3D_FIELD_TMPV) s->udef_nonzero = true;  // the caller placed udef in tmpV for the next projection
  return CUP3D_OK;
}
int cup3d_sim_set_obstacles(cup3d_sim_t *h, int any_rank_has_obstacles) {
  if (!h) return CUP3D_EINVAL;
  reinterpret_cast<Sim *>(h)->obstacles_global = any_rank_has_obstacles < 0 ? -1 : (any_rank_has_obstacles != 0);
  return CUP3D_OK;
}
// zero-copy hosts write through cup3d_sim_device_ptr; the library cannot see those stores, so they say so here
int cup3d_sim_mark_written(cup3d_sim_t *h, int field) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int nc;
  if (!s->field(field, &nc)) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  return mark_written(s, field);
}

int cup3d_sim_upload_blocks(cup3d_sim_t *h, int field, const void *const *ptrs);
int cup3d_sim_download_blocks(cup3d_sim_t *h, int field, void *const *ptrs);

// one contiguous host array [nb][8][8][8][nc]: the same pipelined, multi-threaded staging as the per-block form below (the host
// memory is pageable either way)
int cup3d_sim_upload(cup3d_sim_t *h, int field, const double *blocks) {
  if (!h || !blocks) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int nc;
  if (!s->field(field, &nc)) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  std::vector<const void *> ptrs((size_t)s->nb);
  for (size_t i = 0; i < ptrs.size(); ++i) ptrs[i] = blocks + i * 512 * (size_t)nc;
  return cup3d_sim_upload_blocks(h, field, ptrs.data());
}

int cup3d_sim_download(cup3d_sim_t *h, int field, double *blocks) {
  if (!h || !blocks) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int nc;
  if (!s->field(field, &nc)) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  std::vector<void *> ptrs((size_t)s->nb);
  for (size_t i = 0; i < ptrs.size(); ++i) ptrs[i] = blocks + i * 512 * (size_t)nc;
  return cup3d_sim_download_blocks(h, field, ptrs.data());
}

// One pointer per block (Info::block of the reference's per-block allocations, main.cpp:877-884).  The host side of this boundary
// is 262 144 separate 4-12 KiB allocations at 512^3, so the transfer is a gather (or scatter) on the host plus a PCIe copy plus the
// AoS <-> SoA kernel.  Round 1 did the three one after the other, with one host thread: 11 GB/s down, 55 GB/s up.  Now:
//   * the gather / scatter runs on up to 16 host threads (more were slower on the two-socket GPU host) (a single core copies ~10 GB/s, PCIe Gen5 x16 carries ~55);
//   * two pinned staging buffers and two device staging buffers alternate, so that the host works on chunk k+1 while chunk k is on
//     the bus and in the layout kernel (events, no stream synchronisation inside the loop).
static void release_stage(Sim *s) {
  if (s->h_stage) hipHostFree(s->h_stage);
  if (s->d_stage) hipFree(s->d_stage);
  if (s->h_stage_slots) hipHostFree(s->h_stage_slots);
  if (s->d_stage_slots) hipFree(s->d_stage_slots);
  for (hipEvent_t &e : s->ev_stage) { if (e) hipEventDestroy(e); e = nullptr; }
  s->h_stage = s->d_stage = nullptr;
  s->h_stage_slots = s->d_stage_slots = nullptr;
}
static int ensure_stage(Sim *s) {
  if (s->stage_ready) return CUP3D_OK;
  // per buffer: at most 16384 vector blocks (192 MiB), never more than the sim holds -- a 64-block test sim pins 1.5 MiB, not 400
  s->stage_blocks = (size_t)std::min<int64_t>(16384, std::max<int64_t>(s->nb, 1));
  const size_t nd = 2 * s->stage_blocks * 1536;
  hipError_t e = hipHostMalloc((void **)&s->h_stage, nd * sizeof(double), hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc((void **)&s->d_stage, nd * sizeof(double));
  if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_stage_slots, s->stage_blocks * sizeof(int32_t), hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc((void **)&s->d_stage_slots, s->stage_blocks * sizeof(int32_t));
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&s->ev_stage[i], hipEventDisableTiming);
  if (e != hipSuccess) {  // all or nothing: a later call must not find half a staging area
    release_stage(s);
    return hip_fail(e, "ensure_stage", __FILE__, __LINE__);
  }
  s->bytes += nd * sizeof(double) + s->stage_blocks * sizeof(int32_t);  // the device half (the pinned host half is as large)
  s->stage_ready = true;
  return CUP3D_OK;
}
int cup3d_sim_upload_blocks(cup3d_sim_t *h, int field, const void *const *ptrs) {
  if (!h || !ptrs) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int nc, rc;
  double *dst = s->field(field, &nc);
  if (!dst) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  if ((rc = ensure_stage(s))) return rc;
  const size_t per = 512 * (size_t)nc, cap = s->stage_blocks * 1536;
  stats_field_transfer(true, (size_t)s->nb * per * sizeof(double));
  int k = 0;
  for (size_t b0 = 0; b0 < (size_t)s->nb; b0 += s->stage_blocks, ++k) {
    const size_t n = std::min(s->stage_blocks, (size_t)s->nb - b0);
    const int buf = k & 1;
    double *hs = s->h_stage + buf * cap, *ds = s->d_stage + buf * cap;
    if (k >= 2) CUP3D_HIP(hipEventSynchronize(s->ev_stage[buf]));  // the copy that last read this pinned buffer is done
    host_parallel(n, [&](size_t i) { memcpy(hs + i * per, ptrs[b0 + i], per * sizeof(double)); });
    CUP3D_HIP(hipMemcpyAsync(ds, hs, n * per * sizeof(double), hipMemcpyHostToDevice, g_stream));
    CUP3D_HIP(hipEventRecord(s->ev_stage[buf], g_stream));
    if (nc == 1) CUP3D_HIP(hipMemcpyAsync(dst + b0 * per, ds, n * per * sizeof(double), hipMemcpyDeviceToDevice, g_stream));
    else hipLaunchKernelGGL(k_aos_to_soa, dim3((unsigned)((n * per + 255) / 256)), dim3(256), 0, g_stream, ds, dst + b0 * per, (long)n * 512, nc);
    CUP3D_HIP(hipGetLastError());
  }
  CUP3D_HIP(hipStreamSynchronize(g_stream));
  return mark_written(s, field);
}
int cup3d_sim_download_blocks(cup3d_sim_t *h, int field, void *const *ptrs) {
  if (!h || !ptrs) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int nc, rc;
  double *src = s->field(field, &nc);
  if (!src) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  if ((rc = ensure_stage(s))) return rc;
  const size_t per = 512 * (size_t)nc, cap = s->stage_blocks * 1536;
  stats_field_transfer(false, (size_t)s->nb * per * sizeof(double));
  const size_t nchunks = ((size_t)s->nb + s->stage_blocks - 1) / s->stage_blocks;
  auto issue = [&](size_t c) -> int {  // layout kernel + copy of chunk c into pinned buffer c & 1
    const size_t b0 = c * s->stage_blocks, n = std::min(s->stage_blocks, (size_t)s->nb - b0);
    const int buf = (int)(c & 1);
    double *hs = s->h_stage + buf * cap, *ds = s->d_stage + buf * cap;
    if (nc == 1) CUP3D_HIP(hipMemcpyAsync(hs, src + b0 * per, n * per * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    else {
      hipLaunchKernelGGL(k_soa_to_aos, dim3((unsigned)((n * per + 255) / 256)), dim3(256), 0, g_stream, src + b0 * per, ds, (long)n * 512, nc);
      CUP3D_HIP(hipGetLastError());
      CUP3D_HIP(hipMemcpyAsync(hs, ds, n * per * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    }
    CUP3D_HIP(hipEventRecord(s->ev_stage[buf], g_stream));
    return CUP3D_OK;
  };
  if (nchunks && (rc = issue(0))) return rc;
  for (size_t c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks && (rc = issue(c + 1))) return rc;  // the next chunk travels while this one is scattered on the host
    const size_t b0 = c * s->stage_blocks, n = std::min(s->stage_blocks, (size_t)s->nb - b0);
    const double *hs = s->h_stage + (c & 1) * cap;
    CUP3D_HIP(hipEventSynchronize(s->ev_stage[c & 1]));
    host_parallel(n, [&](size_t i) { memcpy(ptrs[b0 + i], hs + i * per, per * sizeof(double)); });
  }
  return CUP3D_OK;
}

// Partial transfers: only the listed block slots move (e.g. the blocks an obstacle touches, so that the host-side obstacle
// operators can run between two device operators without a full-field round trip).
static int check_list(Sim *s, long n, const int32_t *slots) {
  for (long i = 0; i < n; ++i)
    if (slots[i] < 0 || slots[i] >= s->nb) { set_error("block slot %d out of range", (int)slots[i]); return CUP3D_EINVAL; }
  return CUP3D_OK;
}
int cup3d_sim_upload_block_list(cup3d_sim_t *h, int field, long nlist, const int32_t *slots, const void *const *ptrs) {
  if (!h || nlist < 0 || (nlist > 0 && (!slots || !ptrs))) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int nc, rc;
  double *dst = s->field(field, &nc);
  if (!dst) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  if ((rc = check_list(s, nlist, slots)) || (rc = ensure_stage(s))) return rc;
  const size_t per = 512 * (size_t)nc;
  stats_field_transfer(true, (size_t)nlist * per * sizeof(double));
  for (size_t b0 = 0; b0 < (size_t)nlist; b0 += s->stage_blocks) {
    const size_t n = std::min(s->stage_blocks, (size_t)nlist - b0);
    for (size_t i = 0; i < n; ++i) {
      memcpy(s->h_stage + i * per, ptrs[b0 + i], per * sizeof(double));
      s->h_stage_slots[i] = slots[b0 + i];
    }
    CUP3D_HIP(hipMemcpyAsync(s->d_stage, s->h_stage, n * per * sizeof(double), hipMemcpyHostToDevice, g_stream));
    CUP3D_HIP(hipMemcpyAsync(s->d_stage_slots, s->h_stage_slots, n * sizeof(int32_t), hipMemcpyHostToDevice, g_stream));
    hipLaunchKernelGGL(k_aos_to_soa_list, dim3((unsigned)((n * per + 255) / 256)), dim3(256), 0, g_stream, s->d_stage, dst, s->d_stage_slots, (long)n * 512, nc);
    CUP3D_HIP(hipGetLastError());
    CUP3D_HIP(hipStreamSynchronize(g_stream));
  }
  return nlist > 0 ? mark_written(s, field) : CUP3D_OK;
}
int cup3d_sim_download_block_list(cup3d_sim_t *h, int field, long nlist, const int32_t *slots, void *const *ptrs) {
  if (!h || nlist < 0 || (nlist > 0 && (!slots || !ptrs))) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int nc, rc;
  double *src = s->field(field, &nc);
  if (!src) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  if ((rc = check_list(s, nlist, slots)) || (rc = ensure_stage(s))) return rc;
  const size_t per = 512 * (size_t)nc;
  stats_field_transfer(false, (size_t)nlist * per * sizeof(double));
  for (size_t b0 = 0; b0 < (size_t)nlist; b0 += s->stage_blocks) {
    const size_t n = std::min(s->stage_blocks, (size_t)nlist - b0);
    for (size_t i = 0; i < n; ++i) s->h_stage_slots[i] = slots[b0 + i];
    CUP3D_HIP(hipMemcpyAsync(s->d_stage_slots, s->h_stage_slots, n * sizeof(int32_t), hipMemcpyHostToDevice, g_stream));
    hipLaunchKernelGGL(k_soa_to_aos_list, dim3((unsigned)((n * per + 255) / 256)), dim3(256), 0, g_stream, src, s->d_stage, s->d_stage_slots, (long)n * 512, nc);
    CUP3D_HIP(hipGetLastError());
    CUP3D_HIP(hipMemcpyAsync(s->h_stage, s->d_stage, n * per * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    CUP3D_HIP(hipStreamSynchronize(g_stream));
    for (size_t i = 0; i < n; ++i) memcpy(ptrs[b0 + i], s->h_stage + i * per, per * sizeof(double));
  }
  return CUP3D_OK;
}

int cup3d_stats_reset(void) {
  g_st_halo = 0; g_st_halo_bytes = 0; g_st_allreduce = 0; g_st_waits = 0; g_st_wait_ns = 0; g_st_iters = 0; g_st_h2d = 0; g_st_d2h = 0;
  return CUP3D_OK;
}
int cup3d_stats_read(cup3d_run_stats *o) {
  if (!o) return CUP3D_EINVAL;
  o->halo_exchanges = g_st_halo; o->halo_bytes_sent = (double)g_st_halo_bytes; o->allreduces = g_st_allreduce;
  o->host_waits = g_st_waits; o->host_wait_seconds = g_st_wait_ns * 1e-9; o->solver_iterations = g_st_iters;
  o->field_bytes_uploaded = (double)g_st_h2d; o->field_bytes_downloaded = (double)g_st_d2h;
  return CUP3D_OK;
}

int cup3d_sim_checksum(cup3d_sim_t *h, int field, unsigned long long *sum) {
  if (!h || !sum) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int nc;
  const double *p = s->field(field, &nc);
  if (!p) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  return checksum_array(s, p, (long)s->nb * 512 * nc, sum);  // the rank's own blocks; ghost blocks of a rank view sit behind them
}

int cup3d_sim_fill(cup3d_sim_t *h, int field, double value) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int nc;
  double *p = s->field(field, &nc);
  if (!p) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  const long n = s->nb * 512L * nc;
  hipLaunchKernelGGL(k_fill, dim3(stride_groups(n)), dim3(256), 0, g_stream, p, n, value);
  CUP3D_HIP(hipGetLastError());
  if (field == CUP3D_FIELD_CHI) s->chi_nonzero = value != 0.0;
  if (field == CUP3D_FIELD_TMPV) s->udef_nonzero = true;
  return CUP3D_OK;
}

int cup3d_max_u(cup3d_sim_t *h, const double uinf[3], double *umax) {
  if (!h || !uinf || !umax) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  const unsigned groups = std::min<unsigned>(stride_groups(s->nb * 1536), (unsigned)s->max_groups);
  {
    ProfileScope ps("max_u");
    hipLaunchKernelGGL(k_max_u, dim3(groups), dim3(256), 0, g_stream, s->vel, (long)s->nb, uinf[0], uinf[1], uinf[2], s->d_partials);
    hipLaunchKernelGGL(k_max_final, dim3(1), dim3(256), 0, g_stream, s->d_partials, (int)groups, s->d_red);
  }
  CUP3D_HIP(hipGetLastError());
  hipStream_t cs = scalar_stream(s);  // MPI_Allreduce MAX, main.cpp:8620 -- on the stream every RCCL call of the library uses
  if (cs != g_stream) {
    CUP3D_HIP(hipEventRecord(s->ev_b, g_stream));
    CUP3D_HIP(hipStreamWaitEvent(cs, s->ev_b, 0));
  }
  int rc = allreduce(s, s->d_red, 1, /*is_max=*/true, cs);
  if (rc) return rc;
  CUP3D_HIP(hipMemcpyAsync(s->h_red, s->d_red, sizeof(double), hipMemcpyDeviceToHost, cs));
  CUP3D_HIP(hipStreamSynchronize(cs));
  *umax = s->h_red[0];
  return CUP3D_OK;
}

int cup3d_external_forcing(cup3d_sim_t *h, double umax_forced, double nu, double H, double dt) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  const double gradPdt = 8 * umax_forced * nu / H / H * dt;  // main.cpp:10584
  ProfileScope ps("external_forcing");
  hipLaunchKernelGGL(k_forcing, dim3(stride_groups(s->nb * 512)), dim3(256), 0, g_stream, s->vel, (long)s->nb, gradPdt);
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

}  // extern "C"
