// See grid.hpp.
#include <map>
#include <memory>
#include "grid.hpp"

#include <algorithm>
#include <stdexcept>
#include <tuple>

namespace cup3d {

void Grid::partition(int64_t total, int rank, int nranks, int64_t *begin, int64_t *count) {
  // GridMPI ctor, main.cpp:2970-2980: the first (total % nranks) ranks own one block more
  const int64_t q = total / nranks, rem = total % nranks;
  *count = q + (rank < rem ? 1 : 0);
  *begin = (int64_t)rank * q + std::min<int64_t>(rank, rem);
}

int Grid::owner_of(int64_t z) const {
  const int64_t q = total_blocks / nranks, rem = total_blocks % nranks;
  if (z < rem * (q + 1)) return (int)(z / (q + 1));
  return (int)(rem + (z - rem * (q + 1)) / q);
}

Grid::Grid(const int bpd_[3], int level_max_, int level_, double maxextent_, const int bc_[3], int rank_, int nranks_)
    : level_max(level_max_), level(level_), maxextent(maxextent_), rank(rank_), nranks(nranks_) {
  if (level < 0 || level >= level_max) throw std::invalid_argument("level must be in [0, level_max)");
  if (nranks < 1 || rank < 0 || rank >= nranks) throw std::invalid_argument("bad rank / nranks");
  for (int d = 0; d < 3; ++d) {
    if (bpd_[d] < 1) throw std::invalid_argument("bpd must be >= 1");
    if (bc_[d] < 0 || bc_[d] > 2) throw std::invalid_argument("bc must be freespace(0), periodic(1) or wall(2)");
    bpd[d] = bpd_[d];
    bc[d] = bc_[d];
    nbd[d] = bpd[d] << level;
  }
  sfc.reset(new HilbertCurve(bpd[0], bpd[1], bpd[2], level_max));
  // h0 = maxextent / (max(bpd) * 8), h = h0 / 2^level  (Grid::getInfoAll, main.cpp:1059-1062)
  const int widest = std::max(bpd[0], std::max(bpd[1], bpd[2]));
  const double h0 = maxextent / (double)(widest * kBS);
  h = h0 / (double)(1 << level);
  total_blocks = (int64_t)nbd[0] * nbd[1] * nbd[2];
  if (total_blocks < nranks) throw std::invalid_argument("fewer blocks than ranks");
  partition(total_blocks, rank, nranks, &z_begin, &z_count);

  // local blocks ordered by blockID_2
  std::vector<std::pair<int64_t, int64_t>> order(z_count);
  for (int64_t n = 0; n < z_count; ++n) {
    int c[3];
    sfc->inverse(z_begin + n, level, c);
    order[n] = {sfc->encode(level, c), z_begin + n};
  }
  std::sort(order.begin(), order.end());
  Z.resize(z_count);
  id2.resize(z_count);
  index.resize(3 * z_count);
  slot_of_z.assign(z_count, -1);
  for (int64_t s = 0; s < z_count; ++s) {
    id2[s] = order[s].first;
    Z[s] = order[s].second;
    int c[3];
    sfc->inverse(Z[s], level, c);
    index[3 * s + 0] = c[0];
    index[3 * s + 1] = c[1];
    index[3 * s + 2] = c[2];
    slot_of_z[Z[s] - z_begin] = (int32_t)s;
    if (c[0] == 0 && c[1] == 0 && c[2] == 0) corner_slot = (int32_t)s;
  }

  // face neighbours + halo plan
  nbr.assign(6 * z_count, 0);
  struct Need { int peer; int64_t sender_z; int sender_face; int64_t slot; int face; };
  std::vector<Need> recv_needs;                                   // what this rank receives
  std::vector<std::tuple<int, int64_t, int, int32_t>> send_list;  // (peer, my Z, my face, slot*6+face)
  for (int64_t s = 0; s < z_count; ++s) {
    bool has_remote = false;
    for (int f = 0; f < 6; ++f) {
      const int d = f >> 1, side = f & 1;
      int c[3] = {index[3 * s], index[3 * s + 1], index[3 * s + 2]};
      const bool at_face = side ? (c[d] == nbd[d] - 1) : (c[d] == 0);
      if (at_face && bc[d] != 1) {  // domain face with a boundary condition
        nbr[6 * s + f] = -1 - bc[d];
        continue;
      }
      c[d] = (c[d] + (side ? 1 : -1) + nbd[d]) % nbd[d];
      const int64_t zn = sfc->forward(level, c[0], c[1], c[2]);
      const int p = owner_of(zn);
      if (p == rank) {
        nbr[6 * s + f] = slot_of_z[zn - z_begin];
      } else {
        has_remote = true;
        recv_needs.push_back({p, zn, f ^ 1, s, f});
        // symmetric relation: my slab behind face f is what that neighbour needs
        send_list.emplace_back(p, Z[s], f, (int32_t)(6 * s + f));
      }
    }
    (has_remote ? boundary : inner).push_back((int32_t)s);
  }
  // both sides order slabs by (peer, sender Z, sender face)
  std::sort(recv_needs.begin(), recv_needs.end(), [](const Need &a, const Need &b) {
    return std::tie(a.peer, a.sender_z, a.sender_face) < std::tie(b.peer, b.sender_z, b.sender_face);
  });
  std::sort(send_list.begin(), send_list.end());
  send_count.assign(nranks, 0);
  recv_count.assign(nranks, 0);
  n_recv_faces = (int64_t)recv_needs.size();
  for (int64_t e = 0; e < n_recv_faces; ++e) {
    const Need &n = recv_needs[e];
    recv_count[n.peer]++;
    nbr[6 * n.slot + n.face] = kNbrHalo + (int32_t)e;
  }
  send_faces.reserve(send_list.size());
  for (auto &t : send_list) {
    send_count[std::get<0>(t)]++;
    send_faces.push_back(std::get<3>(t));
  }
}

// ---------------------------------------------------------------- multi-level mesh
Grid::Grid(const int bpd_[3], int level_max_, double maxextent_, const int bc_[3], int64_t nleaves, const int32_t *levels, const int64_t *Zs)
    : level_max(level_max_), level(0), maxextent(maxextent_), rank(0), nranks(1) {
  if (nleaves < 1 || !levels || !Zs) throw std::invalid_argument("empty leaf list");
  for (int d = 0; d < 3; ++d) {
    if (bpd_[d] < 1) throw std::invalid_argument("bpd must be >= 1");
    if (bc_[d] < 0 || bc_[d] > 2) throw std::invalid_argument("bc must be freespace(0), periodic(1) or wall(2)");
    bpd[d] = bpd_[d];
    bc[d] = bc_[d];
  }
  multilevel = true;
  sfc.reset(new HilbertCurve(bpd[0], bpd[1], bpd[2], level_max));
  const int widest = std::max(bpd[0], std::max(bpd[1], bpd[2]));
  const double h0 = maxextent / (double)(widest * kBS);
  total_blocks = z_count = nleaves;
  z_begin = 0;
  // m_vInfo order: sorted by blockID_2 (FillPos, main.cpp:943-964)
  std::vector<std::tuple<int64_t, int32_t, int64_t>> order(nleaves);
  int lmin = level_max;
  for (int64_t n = 0; n < nleaves; ++n) {
    if (levels[n] < 0 || levels[n] >= level_max) throw std::invalid_argument("leaf level out of range");
    int c[3];
    sfc->inverse(Zs[n], levels[n], c);
    order[n] = std::make_tuple(sfc->encode(levels[n], c), levels[n], Zs[n]);
    lmin = std::min(lmin, (int)levels[n]);
  }
  std::sort(order.begin(), order.end());
  level = lmin;
  h = h0 / (double)(1 << level);
  for (int d = 0; d < 3; ++d) nbd[d] = bpd[d] << level;
  Z.resize(nleaves); id2.resize(nleaves); index.resize(3 * nleaves); blevel.resize(nleaves); hb.resize(nleaves);
  // dense (level, i, j, k) -> slot maps
  std::vector<std::vector<int32_t>> &at = at_;
  at.assign(level_max, std::vector<int32_t>());
  auto dims = [&](int l, int d) { return bpd[d] << l; };
  for (int l = 0; l < level_max; ++l) at[l].assign((size_t)dims(l, 0) * dims(l, 1) * dims(l, 2), -1);
  auto key = [&](int l, const int c[3]) { return ((size_t)c[2] * dims(l, 1) + c[1]) * dims(l, 0) + c[0]; };
  for (int64_t s = 0; s < nleaves; ++s) {
    id2[s] = std::get<0>(order[s]);
    blevel[s] = std::get<1>(order[s]);
    Z[s] = std::get<2>(order[s]);
    hb[s] = h0 / (double)(1 << blevel[s]);
    int c[3];
    sfc->inverse(Z[s], blevel[s], c);
    for (int d = 0; d < 3; ++d) index[3 * s + d] = c[d];
    if (at[blevel[s]][key(blevel[s], c)] != -1) throw std::invalid_argument("duplicate leaf");
    at[blevel[s]][key(blevel[s], c)] = (int32_t)s;
    if (c[0] == 0 && c[1] == 0 && c[2] == 0) corner_slot = (int32_t)s;  // the last one in m_vInfo order, main.cpp:9287-9289
  }
  // neighbour states of all 26 codes (BlockLab::load 3690-3712)
  nbr27.assign(27 * (size_t)nleaves, kNbrSkipped);
  for (int64_t s = 0; s < nleaves; ++s) {
    const int l = blevel[s];
    const int32_t *idx = &index[3 * s];
    for (int icode = 0; icode < 27; ++icode) {
      const int code[3] = {icode % 3 - 1, (icode / 3) % 3 - 1, icode / 9 - 1};
      if (icode == 13) { nbr27[27 * s + icode] = (int32_t)s; continue; }
      bool skipped = false;
      int c[3];
      for (int d = 0; d < 3; ++d) {
        const int n = dims(l, d);
        const bool skin = idx[d] == 0 || idx[d] == n - 1;
        const int skip = idx[d] == 0 ? -1 : 1;
        if (bc[d] != 1 && code[d] == skip && skin) skipped = true;
        c[d] = (idx[d] + code[d] + n) % n;
      }
      if (skipped) continue;
      int32_t v = at[l][key(l, c)];
      if (v < 0) {
        v = kNbrFiner;
        for (int k = 1; k <= l; ++k) {
          const int pc[3] = {c[0] >> k, c[1] >> k, c[2] >> k};
          const int32_t a = at[l - k][key(l - k, pc)];
          if (a >= 0) {
            if (k > 1) throw std::invalid_argument("mesh is not 2:1 balanced");
            v = kNbrCoarser + a;
            break;
          }
        }
      }
      nbr27[27 * s + icode] = v;
    }
  }
  // face neighbours; interface faces in (slot, face) order
  nbr.assign(6 * (size_t)nleaves, 0);
  std::vector<int32_t> face_e(6 * (size_t)nleaves, -1);
  for (int64_t s = 0; s < nleaves; ++s)
    for (int f = 0; f < 6; ++f) {
      const int d = f >> 1, side = f & 1;
      int code[3] = {0, 0, 0};
      code[d] = side ? 1 : -1;
      const int32_t v = nbr27[27 * s + (code[0] + 1) + 3 * (code[1] + 1) + 9 * (code[2] + 1)];
      if (v == kNbrSkipped) nbr[6 * s + f] = -1 - bc[d];
      else if (v >= 0 && v < kNbrCoarser) nbr[6 * s + f] = v;
      else {
        const int32_t e = (int32_t)n_amr_faces();
        face_e[6 * s + f] = e;
        nbr[6 * s + f] = kNbrHalo + e;
        amr_faces.push_back((int32_t)(6 * s + f));
        amr_faces.push_back(v == kNbrFiner ? 1 : 0);
      }
    }
  amr_fine.assign(4 * (size_t)n_amr_faces(), -1);
  for (int64_t e = 0; e < n_amr_faces(); ++e) {
    if (amr_faces[2 * e + 1] != 1) continue;
    const int64_t s = amr_faces[2 * e] / 6;
    const int f = amr_faces[2 * e] % 6, d = f >> 1, side = f & 1, l = blevel[s];
    const int dfast = d == 0 ? 1 : 0, dslow = d == 2 ? 1 : 2;
    fix_faces[d].push_back((int32_t)e);
    for (int B = 0; B < 4; ++B) {
      int c[3];
      for (int k = 0; k < 3; ++k) c[k] = 2 * index[3 * s + k];
      c[d] = 2 * index[3 * s + d] + (side ? 2 : -1);
      c[dfast] += B % 2;
      c[dslow] += B / 2;
      for (int k = 0; k < 3; ++k) { const int n = dims(l + 1, k); c[k] = (c[k] + n) % n; }
      const int32_t fs = at[l + 1][key(l + 1, c)];
      if (fs < 0) throw std::invalid_argument("mesh is not 2:1 balanced");
      amr_fine[4 * e + B] = face_e[6 * (size_t)fs + (f ^ 1)];
      if (amr_fine[4 * e + B] < 0) throw std::logic_error("interface face bookkeeping");
    }
  }
  inner.resize(nleaves);
  for (int64_t s = 0; s < nleaves; ++s) inner[s] = (int32_t)s;
  send_count.assign(1, 0);
  recv_count.assign(1, 0);
  slot_of_z.clear();
}

Grid::Grid(const Grid &p, int)
    : level_max(p.level_max), level(p.level), maxextent(p.maxextent), h(p.h), rank(0), nranks(1), total_blocks(p.total_blocks), z_begin(0),
      z_count(p.z_count) {
  for (int d = 0; d < 3; ++d) { bpd[d] = p.bpd[d]; bc[d] = p.bc[d]; nbd[d] = p.nbd[d]; }
  sfc.reset(new HilbertCurve(bpd[0], bpd[1], bpd[2], level_max));
  multilevel = true;
  send_count.assign(1, 0);
  recv_count.assign(1, 0);
}

// One rank's view of this (global, multi-level) mesh: see grid.hpp.  The tables are the global ones renumbered, so what a kernel
// reads through them on rank r is what it reads on one rank -- provided the ghost blocks / ghost face fluxes hold the owners' data,
// which is what the two exchange plans are for.  (In the reference every rank keeps the whole octree too: Grid::Octree with the
// owner in TreePosition, main.cpp:815-855; SynchronizerMPI_AMR::_Setup 1979-2286 derives its messages from it.)
// The cells of REMOTE blocks that the star-stencil consumers of rank r's blocks read, for a stencil of width W (1 or 3), as one bounding
// box per remote block (global slot -> lo[3], hi[3]).  Each consumer's index arithmetic is replayed here exactly:
//   * a stencil kernel's tile: the W layers of a same-level face neighbour behind the shared face;
//   * k_ghost_restrict (amr.hip): the 2W layers of the four finer leaves behind a face;
//   * k_ghost_prolong (amr.hip): the coarse shadow patch of W layers x 6 x 6 behind a face whose neighbour is coarser -- cells of coarser
//     leaves, and 2x2x2 averages over cells of same-level edge / corner neighbours; domain faces clamp the patch as the kernel does.
namespace {
struct CellBox { uint8_t lo[3] = {8, 8, 8}, hi[3] = {0, 0, 0}; };
}
static void star_boxes(const Grid &G, const int32_t *owner, int r, int W, std::map<int32_t, CellBox> &boxes) {
  auto mark = [&](int32_t slot, const int lo[3], const int hi[3]) {
    if (owner[slot] == r) return;
    CellBox &b = boxes[slot];
    for (int d = 0; d < 3; ++d) {
      if (lo[d] < 0 || hi[d] > 8 || lo[d] >= hi[d]) throw std::logic_error("star_boxes: a consumer's cell range leaves the block");
      b.lo[d] = (uint8_t)std::min<int>(b.lo[d], lo[d]);
      b.hi[d] = (uint8_t)std::max<int>(b.hi[d], hi[d]);
    }
  };
  const int64_t nb = G.nblocks();
  for (int64_t s = 0; s < nb; ++s) {
    if (owner[s] != r) continue;
    for (int f = 0; f < 6; ++f) {
      const int32_t n = G.nbr[6 * s + f];
      const int ax = f >> 1, side = f & 1;
      if (n >= 0 && n < kNbrHalo) {  // same-level face neighbour: the W layers behind the shared face
        int lo[3] = {0, 0, 0}, hi[3] = {8, 8, 8};
        if (side) hi[ax] = W; else lo[ax] = 8 - W;
        mark(n, lo, hi);
      } else if (n >= kNbrHalo) {
        const int32_t e = n - kNbrHalo;
        if (G.amr_faces[2 * e + 1] == 1) {  // neighbour finer: k_ghost_restrict, layers n0 = side ? 2 gl : 6 - 2 gl (+1), gl < W
          for (int B = 0; B < 4; ++B) {
            const int32_t fe = G.amr_fine[4 * e + B];
            if (fe < 0) continue;
            int lo[3] = {0, 0, 0}, hi[3] = {8, 8, 8};
            if (side) hi[ax] = 2 * W; else lo[ax] = 8 - 2 * W;
            mark(G.amr_faces[2 * fe] / 6, lo, hi);
          }
        } else {  // neighbour coarser: the patch of k_ghost_prolong
          const int ax1 = ax == 0 ? 1 : 0, ax2 = ax == 2 ? 1 : 2;
          const int par[3] = {G.index[3 * s] & 1, G.index[3 * s + 1] & 1, G.index[3 * s + 2] & 1};
          for (int L = 0; L < W; ++L)
            for (int q = 0; q < 36; ++q) {
              int P[3], code[3];
              P[ax] = side ? 4 + L : -1 - L;
              P[ax1] = q % 6 - 1;
              P[ax2] = q / 6 - 1;
              code[ax] = side ? 1 : -1;
              for (int k = 0; k < 2; ++k) {
                const int t = k ? ax2 : ax1;
                int rg = P[t] < 0 ? -1 : (P[t] > 3 ? 1 : 0);
                if (rg != 0 && G.nbr[6 * s + 2 * t + (rg > 0)] < 0) { P[t] = rg < 0 ? 0 : 3; rg = 0; }  // domain face: the face cell
                code[t] = rg;
              }
              const int32_t v = G.nbr27[27 * s + (code[0] + 1) + 3 * (code[1] + 1) + 9 * (code[2] + 1)];
              if (v >= kNbrCoarser) {
                int lo[3], hi[3];
                for (int d = 0; d < 3; ++d) { lo[d] = (par[d] * 4 + P[d] + 8) & 7; hi[d] = lo[d] + 1; }
                mark(v - kNbrCoarser, lo, hi);
              } else if (v >= 0) {  // avg_block(blk, 2 P - 8 code): a 2 x 2 x 2 cube
                int lo[3], hi[3];
                for (int d = 0; d < 3; ++d) { lo[d] = 2 * P[d] - 8 * code[d]; hi[d] = lo[d] + 2; }
                mark(v, lo, hi);
              }
            }
        }
      }
    }
  }
}

std::unique_ptr<Grid> Grid::rank_view(const int32_t *owner, int rank_, int nranks_, bool tensorial) const {
  if (!multilevel || n_local >= 0) throw std::invalid_argument("rank_view needs a global multi-level mesh");
  if (!owner || nranks_ < 1 || rank_ < 0 || rank_ >= nranks_) throw std::invalid_argument("bad rank / nranks");
  const int64_t nb = nblocks(), ne = n_amr_faces();
  for (int64_t s = 0; s < nb; ++s)
    if (owner[s] < 0 || owner[s] >= nranks_) throw std::invalid_argument("owner out of range");
  // which leaves / fine faces does a rank need from others?  needed_by[s] = bit set of ranks (<= 64 ranks per node group is
  // plenty here; use a vector<bool> matrix for generality)
  auto view_of = [&](int r, std::vector<int32_t> &ghost, std::vector<int32_t> &gfaces) {
    std::vector<char> want(nb, 0), wantf(ne, 0);
    for (int64_t s = 0; s < nb; ++s) {
      if (owner[s] != r) continue;
      for (int c = 0; c < 27; ++c) {
        const int32_t v = nbr27[27 * s + c];
        if (v >= kNbrCoarser) { if (owner[v - kNbrCoarser] != r) want[v - kNbrCoarser] = 1; }
        else if (v >= 0 && owner[v] != r) want[v] = 1;
        else if (v == kNbrFiner && tensorial) {  // the finer leaves behind this position, face / edge / corner alike
          const int code[3] = {c % 3 - 1, (c / 3) % 3 - 1, c / 9 - 1};
          for (int q = 0; q < 8; ++q) {
            int fi[3];
            bool used = true;
            for (int d = 0; d < 3; ++d) {
              const int bit = (q >> d) & 1;
              if (code[d] != 0 && bit) used = false;
              fi[d] = 2 * index[3 * s + d] + (code[d] < 0 ? -1 : (code[d] > 0 ? 2 : bit));
            }
            if (!used) continue;
            const int32_t fs = leaf(blevel[s] + 1, fi);
            if (fs >= 0 && owner[fs] != r) want[fs] = 1;
          }
        }
      }
    }
    for (int64_t e = 0; e < ne; ++e) {
      const int64_t s = amr_faces[2 * e] / 6;
      if (owner[s] != r || amr_faces[2 * e + 1] != 1) continue;
      for (int B = 0; B < 4; ++B) {
        const int32_t fe = amr_fine[4 * e + B];
        const int32_t fs = amr_faces[2 * fe] / 6;
        if (owner[fs] != r) { want[fs] = 1; wantf[fe] = 1; }
      }
    }
    ghost.clear();
    gfaces.clear();
    for (int p = 0; p < nranks_; ++p) {  // (owner, global order)
      if (p == r) continue;
      for (int64_t s = 0; s < nb; ++s) if (want[s] && owner[s] == p) ghost.push_back((int32_t)s);
      for (int64_t e = 0; e < ne; ++e) if (wantf[e] && owner[amr_faces[2 * e] / 6] == p) gfaces.push_back((int32_t)e);
    }
  };
  std::vector<int32_t> ghost, gfaces;
  view_of(rank_, ghost, gfaces);

  std::unique_ptr<Grid> v(new Grid(*this, 0));
  Grid &V = *v;
  V.rank = rank_;
  V.nranks = nranks_;
  V.send_count.assign(nranks_, 0);  // no face-slab plan: everything remote arrives as ghost blocks
  V.recv_count.assign(nranks_, 0);
  // slot maps
  std::vector<int32_t> local;
  for (int64_t s = 0; s < nb; ++s) if (owner[s] == rank_) local.push_back((int32_t)s);
  if (local.empty()) throw std::invalid_argument("a rank without blocks");
  V.n_local = (int64_t)local.size();
  V.global_slot = local;
  V.global_slot.insert(V.global_slot.end(), ghost.begin(), ghost.end());
  std::vector<int32_t> to_view(nb, -1);
  for (size_t i = 0; i < V.global_slot.size(); ++i) to_view[V.global_slot[i]] = (int32_t)i;
  const size_t nv = V.global_slot.size();
  V.Z.resize(nv); V.id2.resize(nv); V.index.resize(3 * nv); V.blevel.resize(nv); V.hb.resize(nv);
  V.at_.assign(level_max, std::vector<int32_t>());
  for (int l = 0; l < level_max; ++l) V.at_[l].assign(at_[l].size(), -1);
  for (size_t i = 0; i < nv; ++i) {
    const int32_t g = V.global_slot[i];
    V.Z[i] = Z[g]; V.id2[i] = id2[g]; V.blevel[i] = blevel[g]; V.hb[i] = hb[g];
    for (int d = 0; d < 3; ++d) V.index[3 * i + d] = index[3 * g + d];
  }
  for (int l = 0; l < level_max; ++l)
    for (size_t k = 0; k < at_[l].size(); ++k)
      if (at_[l][k] >= 0 && to_view[at_[l][k]] >= 0) V.at_[l][k] = to_view[at_[l][k]];
  V.ghost_owner.resize(ghost.size());
  for (size_t i = 0; i < ghost.size(); ++i) V.ghost_owner[i] = owner[ghost[i]];
  // interface faces: local ones in global order, then the ghost fine faces
  std::vector<int32_t> face_to_view(ne, -1);
  V.global_face.clear();
  for (int64_t e = 0; e < ne; ++e)
    if (owner[amr_faces[2 * e] / 6] == rank_) { face_to_view[e] = (int32_t)V.global_face.size(); V.global_face.push_back((int32_t)e); }
  V.n_local_faces = (int64_t)V.global_face.size();
  for (int32_t e : gfaces) { face_to_view[e] = (int32_t)V.global_face.size(); V.global_face.push_back(e); }
  const size_t nfv = V.global_face.size();
  V.amr_faces.resize(2 * nfv);
  V.amr_fine.assign(4 * nfv, -1);
  for (int d = 0; d < 3; ++d) V.fix_faces[d].clear();
  for (size_t i = 0; i < nfv; ++i) {
    const int32_t e = V.global_face[i];
    const int32_t gs = amr_faces[2 * e] / 6, f = amr_faces[2 * e] % 6;
    V.amr_faces[2 * i] = 6 * to_view[gs] + f;
    V.amr_faces[2 * i + 1] = amr_faces[2 * e + 1];
    if ((int64_t)i < V.n_local_faces && amr_faces[2 * e + 1] == 1) {
      V.fix_faces[f >> 1].push_back((int32_t)i);
      for (int B = 0; B < 4; ++B) V.amr_fine[4 * i + B] = face_to_view[amr_fine[4 * e + B]];
    }
  }
  // neighbour tables of the local blocks
  V.nbr.assign(6 * (size_t)V.n_local, 0);
  V.nbr27.assign(27 * (size_t)V.n_local, kNbrSkipped);
  V.corner_slot = -1;
  for (int64_t i = 0; i < V.n_local; ++i) {
    const int32_t g = local[i];
    if (g == corner_slot) V.corner_slot = (int32_t)i;
    for (int c = 0; c < 27; ++c) {
      const int32_t x = nbr27[27 * (size_t)g + c];
      V.nbr27[27 * i + c] = x >= kNbrCoarser ? kNbrCoarser + to_view[x - kNbrCoarser] : (x >= 0 ? to_view[x] : x);
    }
    for (int f = 0; f < 6; ++f) {
      const int32_t x = nbr[6 * (size_t)g + f];
      V.nbr[6 * i + f] = x >= kNbrHalo ? kNbrHalo + face_to_view[x - kNbrHalo] : (x >= 0 ? to_view[x] : x);
    }
  }
  // inner / boundary split (the reference's inner_blocks / halo_blocks, main.cpp:2196-2199): a local block is a boundary block when
  // anything its tables lead to lives in a ghost slot -- a same-level or coarser neighbour in one of the 27 directions, or a finer
  // leaf behind one of its faces (its ghost slab is restricted from that leaf) -- and only those wait for the exchange
  {
    std::vector<char> remote((size_t)V.n_local, 0);
    for (int64_t i = 0; i < V.n_local; ++i)
      for (int c = 0; c < 27 && !remote[i]; ++c) {
        const int32_t x = V.nbr27[27 * i + c];
        const int32_t sl = x >= kNbrCoarser ? x - kNbrCoarser : x;
        if (sl >= V.n_local) remote[i] = 1;
      }
    for (int64_t e = 0; e < V.n_local_faces; ++e) {
      if (V.amr_faces[2 * e + 1] != 1) continue;
      for (int B = 0; B < 4; ++B) {
        const int32_t fe = V.amr_fine[4 * e + B];
        if (fe >= V.n_local_faces || (fe >= 0 && V.amr_faces[2 * fe] / 6 >= V.n_local)) remote[V.amr_faces[2 * e] / 6] = 1;
      }
    }
    V.inner.clear();
    V.boundary.clear();
    for (int64_t i = 0; i < V.n_local; ++i) (remote[i] ? V.boundary : V.inner).push_back((int32_t)i);
  }
  V.total_blocks = nb;
  // exchange plans: what I receive from p is the (p-owned) part of my ghost list; what I send to p is the (me-owned) part of p's
  V.recv_block_count.assign(nranks_, 0);
  V.recv_flux_count.assign(nranks_, 0);
  for (int32_t g : ghost) V.recv_block_count[owner[g]]++;
  for (int32_t e : gfaces) V.recv_flux_count[owner[amr_faces[2 * e] / 6]]++;
  V.send_block_count.assign(nranks_, 0);
  V.send_flux_count.assign(nranks_, 0);
  V.send_blocks.clear();
  V.send_flux_faces.clear();
  std::vector<int32_t> pg, pf;
  for (int p = 0; p < nranks_; ++p) {
    if (p == rank_) continue;
    view_of(p, pg, pf);
    for (int32_t g : pg) if (owner[g] == rank_) { V.send_blocks.push_back(to_view[g]); V.send_block_count[p]++; }
    for (int32_t e : pf) if (owner[amr_faces[2 * e] / 6] == rank_) { V.send_flux_faces.push_back(face_to_view[e]); V.send_flux_count[p]++; }
  }
  // sub-box form of the ghost-block exchange: the boxes of my ghosts, and -- replaying every peer's consumers -- of what I send
  if (!tensorial) {
    for (int k = 0; k < 2; ++k) {
      const int W = k ? 3 : 1;
      std::map<int32_t, CellBox> mine;
      star_boxes(*this, owner, rank_, W, mine);
      V.ghost_box[k].assign(6 * ghost.size(), 0);
      V.recv_cells[k].assign(nranks_, 0);
      V.send_cells[k].assign(nranks_, 0);
      for (size_t i = 0; i < ghost.size(); ++i) {
        auto it = mine.find(ghost[i]);
        if (it == mine.end()) continue;  // a ghost only the tensorial consumers or the tables name: nothing of it is read here
        int64_t vol = 1;
        for (int d = 0; d < 3; ++d) { V.ghost_box[k][6 * i + d] = it->second.lo[d]; V.ghost_box[k][6 * i + 3 + d] = it->second.hi[d]; vol *= it->second.hi[d] - it->second.lo[d]; }
        V.recv_cells[k][owner[ghost[i]]] += vol;
      }
      for (auto &kv : mine)  // every block a consumer reads must be in the ghost list
        if (to_view[kv.first] < 0) throw std::logic_error("rank_view: a star-stencil consumer reads a block that is not a ghost");
      V.send_box[k].clear();
      for (int p = 0; p < nranks_; ++p) {
        if (p == rank_) continue;
        std::map<int32_t, CellBox> theirs;
        star_boxes(*this, owner, p, W, theirs);
        view_of(p, pg, pf);
        for (int32_t g : pg) {
          if (owner[g] != rank_) continue;
          uint8_t bx[6] = {0, 0, 0, 0, 0, 0};
          auto it = theirs.find(g);
          int64_t vol = 0;
          if (it != theirs.end()) {
            vol = 1;
            for (int d = 0; d < 3; ++d) { bx[d] = it->second.lo[d]; bx[3 + d] = it->second.hi[d]; vol *= it->second.hi[d] - it->second.lo[d]; }
          }
          V.send_box[k].insert(V.send_box[k].end(), bx, bx + 6);
          V.send_cells[k][p] += vol;
        }
      }
    }
  }
  // the multigrid option's hierarchy of this rank needs the global mesh, which the view does not keep: leave what it takes to build one
  // on first use (mg_plan_get).  Multi-level meshes only: the multigrid path of a uniform mesh never reads it.
  if (!tensorial && multilevel) {
    auto src = std::make_shared<MGSource>();
    for (int d = 0; d < 3; ++d) { src->bpd[d] = bpd[d]; src->bc[d] = bc[d]; }
    src->maxextent = maxextent;
    src->blevel.assign(blevel.begin(), blevel.end());
    src->index.assign(index.begin(), index.end());
    src->owner.assign(owner, owner + nblocks());
    src->leaf_slot = to_view;
    src->rank = rank_;
    src->nranks = nranks_;
    V.mg_source = src;
  }
  return v;
}

std::shared_ptr<MGHierarchy> Grid::mg_hierarchy(const int32_t *owner, int rank_, int nranks_, const std::vector<int32_t> *leaf_slot) const {
  if (!multilevel || n_local >= 0) throw std::invalid_argument("mg_hierarchy needs a global multi-level mesh");
  MGSource src;
  for (int d = 0; d < 3; ++d) { src.bpd[d] = bpd[d]; src.bc[d] = bc[d]; }
  src.maxextent = maxextent;
  src.blevel.assign(blevel.begin(), blevel.end());
  src.index.assign(index.begin(), index.end());
  if (owner) src.owner.assign(owner, owner + nblocks());
  if (leaf_slot) src.leaf_slot = *leaf_slot;
  src.rank = rank_;
  src.nranks = nranks_;
  return build_mg_hierarchy(src);
}

std::shared_ptr<const MGHierarchy> Grid::mg_plan_get() const {
  if (!multilevel) throw std::invalid_argument("mg_plan_get needs a multi-level mesh");  // the precondition Grid::mg_hierarchy enforces
  if (!mg_plan && mg_source) mg_plan = build_mg_hierarchy(*mg_source);
  return mg_plan;
}

std::shared_ptr<MGHierarchy> build_mg_hierarchy(const MGSource &src) {
  const int *bpd = src.bpd, *bc = src.bc;
  const double maxextent = src.maxextent;
  const std::vector<int32_t> &blevel = src.blevel, &index = src.index;
  const int32_t *owner = src.owner.empty() ? nullptr : src.owner.data();
  const std::vector<int32_t> *leaf_slot = src.leaf_slot.empty() ? nullptr : &src.leaf_slot;
  const int rank_ = src.rank, nranks_ = src.nranks;
  const int64_t nb = (int64_t)blevel.size();
  int lmax = 0;
  for (int64_t b = 0; b < nb; ++b) lmax = std::max(lmax, (int)blevel[(size_t)b]);
  const int nlev = lmax + 1;
  struct Node { int c[3]; int32_t leaf; int32_t owner; };
  std::vector<std::vector<Node>> nodes(nlev);
  std::vector<std::vector<int32_t>> map(nlev);
  auto dim = [&](int l, int d) { return bpd[d] << l; };
  auto at = [&](int l, const int c[3]) -> int32_t & { return map[l][((size_t)c[2] * dim(l, 1) + c[1]) * dim(l, 0) + c[0]]; };
  for (int l = 0; l < nlev; ++l) map[l].assign((size_t)dim(l, 0) * dim(l, 1) * dim(l, 2), -1);
  for (int64_t b = 0; b < nb; ++b) {
    const int l = blevel[(size_t)b];
    Node n{{index[3 * b], index[3 * b + 1], index[3 * b + 2]}, (int32_t)b, owner ? owner[b] : 0};
    if (n.owner < 0 || n.owner >= nranks_) throw std::invalid_argument("owner out of range");
    at(l, n.c) = (int32_t)nodes[l].size();
    nodes[l].push_back(n);
  }
  for (int l = lmax; l >= 1; --l)  // ancestors: nodes[l] is complete when level l is visited
    for (size_t i = 0; i < nodes[l].size(); ++i) {
      const int pc[3] = {nodes[l][i].c[0] >> 1, nodes[l][i].c[1] >> 1, nodes[l][i].c[2] >> 1};
      if (at(l - 1, pc) < 0) {
        at(l - 1, pc) = (int32_t)nodes[l - 1].size();
        nodes[l - 1].push_back(Node{{pc[0], pc[1], pc[2]}, -1, -1});
      }
    }
  for (int l = lmax - 1; l >= 0; --l)  // an ancestor lives where its first child lives (the children of level l + 1 have their owners by now)
    for (Node &n : nodes[l]) {
      if (n.leaf >= 0) continue;
      const int cc[3] = {2 * n.c[0], 2 * n.c[1], 2 * n.c[2]};
      const int32_t ch = at(l + 1, cc);
      if (ch < 0) throw std::logic_error("multigrid: an ancestor without its first child");
      n.owner = nodes[l + 1][ch].owner;
    }
  // face neighbour of node (l, i) across face f: >= 0 same-level node; -1 domain face; -2 - cs: only the coarse node cs of level l - 1 exists
  auto neighbour = [&](int l, const Node &nd, int f) -> int32_t {
    const int d = f >> 1, side = f & 1;
    int c[3] = {nd.c[0], nd.c[1], nd.c[2]};
    c[d] += side ? 1 : -1;
    if (c[d] < 0 || c[d] >= dim(l, d)) {
      if (bc[d] != 1) return -1;  // zero-gradient pressure tile behind every non-periodic domain face
      c[d] = (c[d] + dim(l, d)) % dim(l, d);
    }
    const int32_t m = at(l, c);
    if (m >= 0) return m;
    const int cc[3] = {c[0] >> 1, c[1] >> 1, c[2] >> 1};
    const int32_t cs = l > 0 ? at(l - 1, cc) : -1;
    if (cs < 0) throw std::invalid_argument("multigrid: the mesh is not 2:1 balanced");
    return -2 - cs;
  };
  // ghost lists of EVERY rank (a rank's send list is the part of the others' ghost lists it owns): want[l][r] = global node numbers
  std::vector<std::vector<std::vector<int32_t>>> want(nlev, std::vector<std::vector<int32_t>>(nranks_));
  if (nranks_ > 1)
    for (int l = 0; l < nlev; ++l)
      for (size_t i = 0; i < nodes[l].size(); ++i) {
        const Node &nd = nodes[l][i];
        const int r = nd.owner;
        for (int f = 0; f < 6; ++f) {
          const int32_t m = neighbour(l, nd, f);
          if (m >= 0) { if (nodes[l][m].owner != r) want[l][r].push_back(m); }
          else if (m <= -2) { const int32_t cs = -2 - m; if (nodes[l - 1][cs].owner != r) want[l - 1][r].push_back(cs); }
        }
        if (l > 0) {
          const int pc[3] = {nd.c[0] >> 1, nd.c[1] >> 1, nd.c[2] >> 1};
          const int32_t pn = at(l - 1, pc);
          if (nodes[l - 1][pn].owner != r) want[l - 1][r].push_back(pn);
        }
      }
  for (int l = 0; l < nlev; ++l)
    for (int r = 0; r < nranks_; ++r) {
      auto &w = want[l][r];
      std::sort(w.begin(), w.end(), [&](int32_t a, int32_t b) { return nodes[l][a].owner != nodes[l][b].owner ? nodes[l][a].owner < nodes[l][b].owner : a < b; });
      w.erase(std::unique(w.begin(), w.end()), w.end());
    }
  auto H = std::make_shared<MGHierarchy>();
  H->rank = rank_;
  H->nranks = nranks_;
  H->lev.resize(nlev);
  std::vector<std::vector<int32_t>> loc(nlev);  // global node -> slot in rank_'s arrays, -1 invisible
  for (int l = 0; l < nlev; ++l) {
    MGLevelPlan &P = H->lev[l];
    loc[l].assign(nodes[l].size(), -1);
    for (size_t i = 0; i < nodes[l].size(); ++i)
      if (nodes[l][i].owner == rank_) { loc[l][i] = (int32_t)P.gid.size(); P.gid.push_back((int32_t)i); }
    P.n_owned = (int64_t)P.gid.size();
    for (int32_t gnode : want[l][rank_]) { loc[l][gnode] = (int32_t)P.gid.size(); P.gid.push_back(gnode); }
    P.n_ghost = (int64_t)P.gid.size() - P.n_owned;
    P.n_global = (int64_t)nodes[l].size();
    P.h = maxextent / (8.0 * std::max(std::max(dim(l, 0), dim(l, 1)), dim(l, 2)));  // Info::h of level l (h_gridpoint, main.cpp:15405-15415)
  }
  for (int l = 0; l < nlev; ++l) {
    MGLevelPlan &P = H->lev[l];
    const size_t n = (size_t)P.n_owned;
    P.nbr.assign(6 * n, -1);
    P.parent.assign(2 * n, 0);
    P.leaf.assign(n, -1);
    for (size_t i = 0; i < n; ++i) {
      const Node &nd = nodes[l][P.gid[i]];
      P.leaf[i] = nd.leaf < 0 ? -1 : (leaf_slot ? (*leaf_slot)[nd.leaf] : nd.leaf);
      if (nd.leaf >= 0 && P.leaf[i] < 0) throw std::logic_error("multigrid: an owned leaf without a local slot");
      if (l > 0) {
        const int pc[3] = {nd.c[0] >> 1, nd.c[1] >> 1, nd.c[2] >> 1};
        P.parent[2 * i] = loc[l - 1][at(l - 1, pc)];
        P.parent[2 * i + 1] = (nd.c[0] & 1) + 2 * (nd.c[1] & 1) + 4 * (nd.c[2] & 1);
        if (P.parent[2 * i] < 0) throw std::logic_error("multigrid: an owned node's parent is not visible");
      }
      for (int f = 0; f < 6; ++f) {
        const int32_t m = neighbour(l, nd, f);
        if (m >= 0) {
          if (loc[l][m] < 0) throw std::logic_error("multigrid: an owned node's neighbour is not visible");
          P.nbr[6 * i + f] = loc[l][m];
        } else if (m <= -2) {
          const int32_t cs = loc[l - 1][-2 - m];
          if (cs < 0) throw std::logic_error("multigrid: an owned node's coarse neighbour is not visible");
          const int d = f >> 1, t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;  // the slab's (a1, a2) directions, face1()
          P.nbr[6 * i + f] = kNbrHalo + (int32_t)(P.cf.size() / 4);
          P.cf.push_back(cs); P.cf.push_back(d); P.cf.push_back(f & 1); P.cf.push_back((nd.c[t1] & 1) | ((nd.c[t2] & 1) << 1));
        }
      }
    }
    // exchange plans
    P.send_count.assign(nranks_, 0); P.recv_count.assign(nranks_, 0);
    P.rsend_count.assign(nranks_, 0); P.rrecv_count.assign(nranks_, 0);
    for (int32_t gnode : want[l][rank_]) P.recv_count[nodes[l][gnode].owner]++;
    for (int p = 0; p < nranks_; ++p) {
      if (p == rank_) continue;
      for (int32_t gnode : want[l][p])
        if (nodes[l][gnode].owner == rank_) { P.send_slots.push_back(loc[l][gnode]); P.send_count[p]++; }
    }
    if (l > 0 && nranks_ > 1) {
      for (int p = 0; p < nranks_; ++p) {  // peer-major, children in global order -- on both sides
        if (p == rank_) continue;
        for (size_t i = 0; i < nodes[l].size(); ++i) {
          const Node &nd = nodes[l][i];
          const int pc[3] = {nd.c[0] >> 1, nd.c[1] >> 1, nd.c[2] >> 1};
          const int32_t pn = at(l - 1, pc);
          const int po = nodes[l - 1][pn].owner, oct = (nd.c[0] & 1) + 2 * (nd.c[1] & 1) + 4 * (nd.c[2] & 1);
          if (nd.owner == rank_ && po == p) { P.rsend.push_back(loc[l - 1][pn]); P.rsend.push_back(oct); P.rsend_count[p]++; }
          if (nd.owner == p && po == rank_) { P.rrecv.push_back(loc[l - 1][pn]); P.rrecv.push_back(oct); P.rrecv_count[p]++; }
        }
      }
    }
  }
  return H;
}

int32_t Grid::leaf(int l, const int c[3]) const {
  if (l < 0 || l >= level_max || at_.empty()) return -1;
  int w[3];
  for (int d = 0; d < 3; ++d) { const int n = bpd[d] << l; w[d] = ((c[d] % n) + n) % n; }
  return at_[l][((size_t)w[2] * (bpd[1] << l) + w[1]) * (bpd[0] << l) + w[0]];
}

std::unique_ptr<Grid> Grid::as_mesh() const {
  if (nranks != 1) throw std::invalid_argument("mesh adaptation is supported on one rank");
  std::vector<int32_t> lv(nblocks());
  for (int64_t s = 0; s < nblocks(); ++s) lv[s] = multilevel ? blevel[s] : level;
  return std::unique_ptr<Grid>(new Grid(bpd, level_max, maxextent, bc, nblocks(), lv.data(), Z.data()));
}

void Grid::valid_states(int8_t *st) const {
  if (!multilevel) throw std::invalid_argument("valid_states needs a multi-level mesh object (Grid::as_mesh)");
  if (n_local >= 0) throw std::invalid_argument("valid_states needs the global mesh, not one rank's view");
  const int64_t nb = nblocks();
  for (int64_t b = 0; b < nb; ++b)
    if ((st[b] == 1 && blevel[b] == level_max - 1) || (st[b] == -1 && blevel[b] == 0)) st[b] = 0;
  for (int lv = level_max - 1; lv >= 0; --lv) {
    // refinement propagates from finer neighbours; a block next to finer blocks may not compress (5352-5409)
    for (int64_t b = 0; b < nb; ++b) {
      if (!(blevel[b] == lv && st[b] != 1 && blevel[b] != level_max - 1)) continue;
      const int32_t *idx = &index[3 * b];
      for (int icode = 0; icode < 27; ++icode) {
        if (st[b] == 1) break;
        if (icode == 13 || nbr27[27 * b + icode] != kNbrFiner) continue;
        if (st[b] == -1) st[b] = 0;
        const int code[3] = {icode % 3 - 1, (icode / 3) % 3 - 1, icode / 9 - 1};
        const int tmp = std::abs(code[0]) + std::abs(code[1]) + std::abs(code[2]);
        const int Bstep = tmp == 2 ? 3 : (tmp == 3 ? 4 : 1);
        for (int B = 0; B <= 3; B += Bstep) {
          const int aux = std::abs(code[0]) == 1 ? B % 2 : B / 2;
          const int fi[3] = {2 * idx[0] + std::max(code[0], 0) + code[0] + (B % 2) * std::max(0, 1 - std::abs(code[0])),
                             2 * idx[1] + std::max(code[1], 0) + code[1] + aux * std::max(0, 1 - std::abs(code[1])),
                             2 * idx[2] + std::max(code[2], 0) + code[2] + (B / 2) * std::max(0, 1 - std::abs(code[2]))};
          const int32_t fb = leaf(lv + 1, fi);
          if (fb >= 0 && st[fb] == 1) { st[b] = 1; break; }
        }
      }
    }
    if (lv == 0) break;
    // no compression next to a refining block of the same level (5413-5448)
    for (int64_t b = 0; b < nb; ++b) {
      if (!(blevel[b] == lv && st[b] == -1)) continue;
      for (int icode = 0; icode < 27; ++icode) {
        const int32_t v = nbr27[27 * b + icode];
        if (icode != 13 && v >= 0 && v < kNbrCoarser && st[v] == 1) { st[b] = 0; break; }
      }
    }
  }
  // an octet compresses only if all eight siblings exist and agree (5451-5491)
  std::vector<int8_t> veto(nb, 0);
  for (int64_t b = 0; b < nb; ++b) {
    if (st[b] != -1) continue;
    const int32_t *idx = &index[3 * b];
    for (int q = 0; q < 8; ++q) {
      const int s[3] = {2 * (idx[0] / 2) + (q & 1), 2 * (idx[1] / 2) + ((q >> 1) & 1), 2 * (idx[2] / 2) + (q >> 2)};
      const int32_t sb = leaf(blevel[b], s);
      if (sb < 0 || st[sb] != -1) { veto[b] = 1; break; }
    }
  }
  for (int64_t b = 0; b < nb; ++b)
    if (veto[b]) st[b] = 0;
}

void Grid::adapted_leaves(const int8_t *st, std::vector<int32_t> &levels, std::vector<int64_t> &Zs) const {
  if (!multilevel) throw std::invalid_argument("adapted_leaves needs a multi-level mesh object (Grid::as_mesh)");
  if (n_local >= 0) throw std::invalid_argument("adapted_leaves needs the global mesh, not one rank's view");
  levels.clear();
  Zs.clear();
  for (int64_t b = 0; b < nblocks(); ++b) {
    const int l = blevel[b];
    const int32_t *idx = &index[3 * b];
    if (st[b] == 1) {
      for (int q = 0; q < 8; ++q) {
        levels.push_back(l + 1);
        Zs.push_back(sfc->forward(l + 1, 2 * idx[0] + (q & 1), 2 * idx[1] + ((q >> 1) & 1), 2 * idx[2] + (q >> 2)));
      }
    } else if (st[b] == -1) {
      if (idx[0] % 2 == 0 && idx[1] % 2 == 0 && idx[2] % 2 == 0) {
        levels.push_back(l - 1);
        Zs.push_back(sfc->forward(l - 1, idx[0] / 2, idx[1] / 2, idx[2] / 2));
      }
    } else {
      levels.push_back(l);
      Zs.push_back(Z[b]);
    }
  }
}

void Grid::adapted_owners(const int32_t *owner, const int8_t *st, int nranks_, const Grid &adapted, int32_t *new_owner) const {
  if (!multilevel || !adapted.multilevel) throw std::invalid_argument("adapted_owners needs multi-level mesh objects");
  if (nranks_ < 1) throw std::invalid_argument("bad number of ranks");
  std::vector<std::vector<int32_t>> list(nranks_);
  auto put = [&](int r, int l, const int c[3]) {
    const int32_t s = adapted.leaf(l, c);
    if (s < 0) throw std::invalid_argument("the adapted mesh does not match the states");
    list[r].push_back(s);
  };
  for (int64_t b = 0; b < nblocks(); ++b) {
    const int l = blevel[b], r = owner[b];
    const int32_t *idx = &index[3 * b];
    if (r < 0 || r >= nranks_) throw std::invalid_argument("owner out of range");
    if (st[b] == 1) {
      for (int q = 0; q < 8; ++q) {
        const int c[3] = {2 * idx[0] + (q & 1), 2 * idx[1] + ((q >> 1) & 1), 2 * idx[2] + (q >> 2)};
        put(r, l + 1, c);  // refine_1 / refine_2 allocate the children where the parent is (5227-5271)
      }
    } else if (st[b] == -1) {
      if (idx[0] % 2 == 0 && idx[1] % 2 == 0 && idx[2] % 2 == 0) {  // the base block's rank receives the octet (4729-4804)
        const int c[3] = {idx[0] / 2, idx[1] / 2, idx[2] / 2};
        put(r, l - 1, c);
      }
    } else {
      const int c[3] = {idx[0], idx[1], idx[2]};
      put(r, l, c);
    }
  }
  int64_t mx = 0, mn = INT64_MAX, total = 0;
  for (auto &v : list) {
    std::sort(v.begin(), v.end());  // slots of `adapted` are in blockID_2 order (Info::operator<)
    mx = std::max<int64_t>(mx, (int64_t)v.size());
    mn = std::min<int64_t>(mn, (int64_t)v.size());
    total += (int64_t)v.size();
  }
  if (mn == 0 || (double)mx / (double)mn > 1.01) {  // Balance_Global (4906-5021): even cut of the rank-major concatenation
    int r = 0;
    int64_t left = total / nranks_ + (0 < total % nranks_ ? 1 : 0);
    for (auto &v : list)
      for (int32_t s : v) {
        while (left == 0) { ++r; left = total / nranks_ + (r < total % nranks_ ? 1 : 0); }
        new_owner[s] = r;
        --left;
      }
    return;
  }
  for (int r = 0; r < nranks_; ++r)
    for (int32_t s : list[r]) new_owner[s] = r;
  for (int r = 0; r < nranks_; ++r) {  // Balance_Diffusion (4821-4905)
    const int64_t my = (int64_t)list[r].size();
    const int64_t fl = r == 0 ? 0 : (my - (int64_t)list[r - 1].size()) / 4;
    const int64_t fr = r == nranks_ - 1 ? 0 : (my - (int64_t)list[r + 1].size()) / 4;
    for (int64_t i = 0; i < fl; ++i) new_owner[list[r][i]] = r - 1;
    for (int64_t i = 0; i < fr; ++i) new_owner[list[r][my - 1 - i]] = r + 1;
  }
}

int32_t Grid::slot_of_index(int i, int j, int k) const {
  const int64_t z = sfc->forward(level, i, j, k);
  if (z < z_begin || z >= z_begin + z_count) return -1;
  return slot_of_z[z - z_begin];
}

std::vector<int32_t> Grid::neighbours27() const {
  std::vector<int32_t> out(27 * (size_t)nblocks());
  for (int64_t s = 0; s < nblocks(); ++s) {
    const int32_t *idx = &index[3 * s];
    for (int cz = -1; cz <= 1; ++cz)
      for (int cy = -1; cy <= 1; ++cy)
        for (int cx = -1; cx <= 1; ++cx) {
          const int code[3] = {cx, cy, cz};
          bool skipped = false;
          int c[3];
          for (int d = 0; d < 3; ++d) {
            const bool skin = idx[d] == 0 || idx[d] == nbd[d] - 1;
            const int skip = idx[d] == 0 ? -1 : 1;  // main.cpp:3681-3686, 3696-3701
            if (bc[d] != 1 && code[d] == skip && skin) skipped = true;
            c[d] = (idx[d] + code[d] + nbd[d]) % nbd[d];
          }
          int32_t v = -1;
          if (!skipped) {
            v = slot_of_index(c[0], c[1], c[2]);
            if (v < 0) v = -2;
          }
          out[27 * s + (cx + 1) + 3 * (cy + 1) + 9 * (cz + 1)] = v;
        }
  }
  return out;
}

}  // namespace cup3d
