// See grid.hpp.
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
