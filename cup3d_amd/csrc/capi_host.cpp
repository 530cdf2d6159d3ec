// Host-only part of the C ABI (include/cup3d_hip.h): indexing, topology, time step.
// No HIP calls here: these entry points work on a machine without a GPU.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../../include/cup3d_hip.h"
#ifdef CUP3D_TESTING
#include "../../include/cup3d_hip_testing.h"
#endif
#include "grid.hpp"

namespace cup3d {
void set_error(const char *fmt, ...);
}
using namespace cup3d;

extern "C" {

int cup3d_sfc_create(int bx, int by, int bz, int level_max, cup3d_sfc_t **out) {
  if (!out || bx < 1 || by < 1 || bz < 1 || level_max < 1) { set_error("cup3d_sfc_create: bad arguments"); return CUP3D_EINVAL; }
  try {
    *out = reinterpret_cast<cup3d_sfc_t *>(new HilbertCurve(bx, by, bz, level_max));
  } catch (const std::exception &e) {
    set_error("cup3d_sfc_create: %s", e.what());
    return CUP3D_ENOMEM;
  }
  return CUP3D_OK;
}
void cup3d_sfc_destroy(cup3d_sfc_t *s) { delete reinterpret_cast<HilbertCurve *>(s); }
long long cup3d_sfc_forward(const cup3d_sfc_t *s, int level, int i, int j, int k) {
  return reinterpret_cast<const HilbertCurve *>(s)->forward(level, i, j, k);
}
void cup3d_sfc_inverse(const cup3d_sfc_t *s, long long Z, int level, int ijk[3]) {
  reinterpret_cast<const HilbertCurve *>(s)->inverse(Z, level, ijk);
}
long long cup3d_sfc_encode(const cup3d_sfc_t *s, int level, const int index[3]) {
  return reinterpret_cast<const HilbertCurve *>(s)->encode(level, index);
}
void cup3d_sfc_info(const cup3d_sfc_t *s, int level, const int index[3], long long nei[27], long long child[8], long long *parent) {
  int64_t n[27], c[8], p;
  reinterpret_cast<const HilbertCurve *>(s)->info(level, index, n, c, &p);
  for (int i = 0; i < 27; ++i) nei[i] = n[i];
  for (int i = 0; i < 8; ++i) child[i] = c[i];
  *parent = p;
}

int cup3d_grid_create_uniform(const int bpd[3], int level_max, int level, double maxextent, const int bc[3], int rank, int nranks,
                              cup3d_grid_t **out) {
  if (!bpd || !bc || !out) { set_error("cup3d_grid_create_uniform: null argument"); return CUP3D_EINVAL; }
  try {
    *out = reinterpret_cast<cup3d_grid_t *>(new Grid(bpd, level_max, level, maxextent, bc, rank, nranks));
  } catch (const std::invalid_argument &e) {
    set_error("cup3d_grid_create_uniform: %s", e.what());
    return CUP3D_EINVAL;
  } catch (const std::exception &e) {
    set_error("cup3d_grid_create_uniform: %s", e.what());
    return CUP3D_ENOMEM;
  }
  return CUP3D_OK;
}
int cup3d_grid_create_mesh(const int bpd[3], int level_max, double maxextent, const int bc[3], long nleaves, const int32_t *levels,
                           const int64_t *Zs, cup3d_grid_t **out) {
  if (!bpd || !bc || !out || !levels || !Zs || level_max < 1) { set_error("cup3d_grid_create_mesh: bad argument"); return CUP3D_EINVAL; }
  try {
    *out = reinterpret_cast<cup3d_grid_t *>(new Grid(bpd, level_max, maxextent, bc, nleaves, levels, Zs));
  } catch (const std::invalid_argument &e) {
    set_error("cup3d_grid_create_mesh: %s", e.what());
    return CUP3D_EINVAL;
  } catch (const std::exception &e) {
    set_error("cup3d_grid_create_mesh: %s", e.what());
    return CUP3D_ENOMEM;
  }
  return CUP3D_OK;
}
long cup3d_grid_ninterface_faces(const cup3d_grid_t *g) { return (long)reinterpret_cast<const Grid *>(g)->n_amr_faces(); }
int cup3d_grid_interface(const cup3d_grid_t *gh, int32_t *faces2, int32_t *fine4, int32_t *nbr27) {
  if (!gh) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  if (!g->multilevel) { set_error("cup3d_grid_interface: not a multi-level mesh"); return CUP3D_EINVAL; }
  if (faces2) memcpy(faces2, g->amr_faces.data(), g->amr_faces.size() * sizeof(int32_t));
  if (fine4) memcpy(fine4, g->amr_fine.data(), g->amr_fine.size() * sizeof(int32_t));
  if (nbr27) memcpy(nbr27, g->nbr27.data(), g->nbr27.size() * sizeof(int32_t));
  return CUP3D_OK;
}
int cup3d_grid_rank_view(const cup3d_grid_t *gh, const int32_t *owner, int rank, int nranks, cup3d_grid_t **out) {
  if (!gh || !owner || !out) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  try {
    std::unique_ptr<Grid> tmp;
    const Grid *m = g;
    if (!g->multilevel) { tmp = g->as_mesh(); m = tmp.get(); }
    *out = reinterpret_cast<cup3d_grid_t *>(m->rank_view(owner, rank, nranks).release());
  } catch (const std::exception &e) {
    set_error("cup3d_grid_rank_view: %s", e.what());
    return CUP3D_EINVAL;
  }
  return CUP3D_OK;
}
int cup3d_grid_view_sizes(const cup3d_grid_t *gh, long out[6]) {
  if (!gh || !out) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  if (g->n_local < 0) { set_error("cup3d_grid_view_sizes: not a rank view"); return CUP3D_EINVAL; }
  out[0] = (long)g->n_local;
  out[1] = (long)g->nghost();
  out[2] = (long)g->n_local_faces;
  out[3] = (long)(g->n_amr_faces() - g->n_local_faces);
  out[4] = (long)g->send_blocks.size();
  out[5] = (long)g->send_flux_faces.size();
  return CUP3D_OK;
}
int cup3d_grid_view_plan(const cup3d_grid_t *gh, int32_t *global_slot, int32_t *global_face, int32_t *send_blocks, long *send_block_count,
                         long *recv_block_count, int32_t *send_flux_faces, long *send_flux_count, long *recv_flux_count) {
  if (!gh) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  if (g->n_local < 0) { set_error("cup3d_grid_view_plan: not a rank view"); return CUP3D_EINVAL; }
  auto cp = [](int32_t *dst, const std::vector<int32_t> &v) { if (dst && !v.empty()) memcpy(dst, v.data(), v.size() * sizeof(int32_t)); };
  cp(global_slot, g->global_slot);
  cp(global_face, g->global_face);
  cp(send_blocks, g->send_blocks);
  cp(send_flux_faces, g->send_flux_faces);
  for (int p = 0; p < g->nranks; ++p) {
    if (send_block_count) send_block_count[p] = (long)g->send_block_count[p];
    if (recv_block_count) recv_block_count[p] = (long)g->recv_block_count[p];
    if (send_flux_count) send_flux_count[p] = (long)g->send_flux_count[p];
    if (recv_flux_count) recv_flux_count[p] = (long)g->recv_flux_count[p];
  }
  return CUP3D_OK;
}
int cup3d_grid_view_boxes(const cup3d_grid_t *gh, int k, unsigned char *ghost_box, unsigned char *send_box, long *send_cells, long *recv_cells) {
  if (!gh || k < 0 || k > 1) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  if (g->n_local < 0 || g->send_cells[k].empty()) { set_error("cup3d_grid_view_boxes: not a rank view with a sub-box plan"); return CUP3D_EINVAL; }
  if (ghost_box && !g->ghost_box[k].empty()) memcpy(ghost_box, g->ghost_box[k].data(), g->ghost_box[k].size());
  if (send_box && !g->send_box[k].empty()) memcpy(send_box, g->send_box[k].data(), g->send_box[k].size());
  for (int p = 0; p < g->nranks; ++p) {
    if (send_cells) send_cells[p] = (long)g->send_cells[k][p];
    if (recv_cells) recv_cells[p] = (long)g->recv_cells[k][p];
  }
  return CUP3D_OK;
}
int cup3d_grid_valid_states(const cup3d_grid_t *gh, signed char *states) {
  if (!gh || !states) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  try {
    if (g->multilevel) g->valid_states(reinterpret_cast<int8_t *>(states));
    else g->as_mesh()->valid_states(reinterpret_cast<int8_t *>(states));
  } catch (const std::exception &e) {
    set_error("cup3d_grid_valid_states: %s", e.what());
    return CUP3D_EINVAL;
  }
  return CUP3D_OK;
}
int cup3d_grid_adapted(const cup3d_grid_t *gh, const signed char *states, cup3d_grid_t **out) {
  if (!gh || !states || !out) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  try {
    std::unique_ptr<Grid> tmp;
    const Grid *m = g;
    if (!g->multilevel) { tmp = g->as_mesh(); m = tmp.get(); }
    std::vector<int32_t> lv;
    std::vector<int64_t> zs;
    m->adapted_leaves(reinterpret_cast<const int8_t *>(states), lv, zs);
    *out = reinterpret_cast<cup3d_grid_t *>(new Grid(g->bpd, g->level_max, g->maxextent, g->bc, (int64_t)lv.size(), lv.data(), zs.data()));
  } catch (const std::exception &e) {
    set_error("cup3d_grid_adapted: %s", e.what());
    return CUP3D_EINVAL;
  }
  return CUP3D_OK;
}
int cup3d_grid_adapted_owners(const cup3d_grid_t *gh, const int32_t *owner, const signed char *states, int nranks, const cup3d_grid_t *ah,
                              int32_t *new_owner) {
  if (!gh || !owner || !states || !ah || !new_owner) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh), *a = reinterpret_cast<const Grid *>(ah);
  try {
    std::unique_ptr<Grid> tmp;
    const Grid *m = g;
    if (!g->multilevel) { tmp = g->as_mesh(); m = tmp.get(); }
    m->adapted_owners(owner, reinterpret_cast<const int8_t *>(states), nranks, *a, new_owner);
  } catch (const std::exception &e) {
    set_error("cup3d_grid_adapted_owners: %s", e.what());
    return CUP3D_EINVAL;
  }
  return CUP3D_OK;
}
void cup3d_grid_destroy(cup3d_grid_t *g) { delete reinterpret_cast<Grid *>(g); }
long cup3d_grid_nblocks(const cup3d_grid_t *g) { return (long)reinterpret_cast<const Grid *>(g)->nblocks(); }
long cup3d_grid_nblocks_global(const cup3d_grid_t *g) { return (long)reinterpret_cast<const Grid *>(g)->total_blocks; }
long cup3d_grid_nhalo_faces(const cup3d_grid_t *g) { return (long)reinterpret_cast<const Grid *>(g)->n_recv_faces; }
long cup3d_grid_nsend_faces(const cup3d_grid_t *g) { return (long)reinterpret_cast<const Grid *>(g)->send_faces.size(); }
long cup3d_grid_ninner(const cup3d_grid_t *g) { return (long)reinterpret_cast<const Grid *>(g)->inner.size(); }

int cup3d_grid_tables(const cup3d_grid_t *gh, long long *t, double *geom) {
  if (!gh || !t || !geom) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  for (int64_t s = 0; s < g->nblocks(); ++s) {
    const double h = g->multilevel ? g->hb[s] : g->h;
    t[6 * s + 0] = g->multilevel ? g->blevel[s] : g->level;
    t[6 * s + 1] = g->Z[s];
    for (int d = 0; d < 3; ++d) t[6 * s + 2 + d] = g->index[3 * s + d];
    t[6 * s + 5] = g->id2[s];
    geom[4 * s] = h;
    for (int d = 0; d < 3; ++d) geom[4 * s + 1 + d] = g->index[3 * s + d] * kBS * h;  // origin, main.cpp:1066-1068
  }
  return CUP3D_OK;
}
int cup3d_grid_neighbours(const cup3d_grid_t *gh, int32_t *nbr) {
  if (!gh || !nbr) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  memcpy(nbr, g->nbr.data(), g->nbr.size() * sizeof(int32_t));
  return CUP3D_OK;
}
int cup3d_grid_halo_plan(const cup3d_grid_t *gh, long *send_count, long *recv_count, int32_t *send_faces) {
  if (!gh || !send_count || !recv_count) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  for (int p = 0; p < g->nranks; ++p) {
    send_count[p] = (long)g->send_count[p];
    recv_count[p] = (long)g->recv_count[p];
  }
  if (send_faces) memcpy(send_faces, g->send_faces.data(), g->send_faces.size() * sizeof(int32_t));
  return CUP3D_OK;
}

#ifdef CUP3D_TESTING
// TEST SUPPORT (no GPU): the multigrid hierarchies of ALL ranks for the given ownership, checked against each other -- every table
// entry resolves to an owned or a ghost slot; what rank r sends to rank p is, node for node, what p expects from r (ghost exchange and
// restriction octants); every owned ancestor receives each of its eight octants exactly once (from a local child or from a message)
int cup3d_debug_mg_plan_check(const cup3d_grid_t *gh, const int32_t *owner, int nranks) {
  if (!gh || nranks < 1) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  try {
    std::vector<std::shared_ptr<MGHierarchy>> H(nranks);
    for (int r = 0; r < nranks; ++r) H[r] = g->mg_hierarchy(nranks > 1 ? owner : nullptr, r, nranks, nullptr);
    const size_t nlev = H[0]->lev.size();
    for (size_t l = 0; l < nlev; ++l) {
      int64_t owned_total = 0;
      for (int r = 0; r < nranks; ++r) {
        const MGLevelPlan &P = H[r]->lev[l];
        owned_total += P.n_owned;
        const int64_t nvis = P.n_owned + P.n_ghost, nvis_c = l > 0 ? H[r]->lev[l - 1].n_owned + H[r]->lev[l - 1].n_ghost : 0;
        if ((int64_t)P.gid.size() != nvis) throw std::logic_error("gid size");
        for (int64_t i = 0; i < P.n_owned; ++i) {
          for (int f = 0; f < 6; ++f) {
            const int32_t v = P.nbr[6 * i + f];
            if (v >= kNbrHalo) { if ((size_t)(v - kNbrHalo) >= P.cf.size() / 4) throw std::logic_error("cf index out of range"); }
            else if (v >= 0) { if (v >= nvis) throw std::logic_error("neighbour slot out of range"); }
            else if (v != -1) throw std::logic_error("bad neighbour code");
          }
          if (l > 0 && (P.parent[2 * i] < 0 || P.parent[2 * i] >= nvis_c || P.parent[2 * i + 1] < 0 || P.parent[2 * i + 1] > 7)) throw std::logic_error("parent out of range");
        }
        for (size_t e = 0; e < P.cf.size() / 4; ++e)
          if (P.cf[4 * e] < 0 || P.cf[4 * e] >= nvis_c) throw std::logic_error("coarse neighbour slot out of range");
        // ghost exchange: my receive from p == p's send to me, node for node
        int64_t ghost_at = P.n_owned;
        for (int p = 0; p < nranks; ++p) {
          const MGLevelPlan &Q = H[p]->lev[l];
          if (P.recv_count[p] != Q.send_count[r]) throw std::logic_error("ghost exchange counts differ");
          int64_t off = 0;
          for (int q = 0; q < r; ++q) off += Q.send_count[q];
          for (int64_t k = 0; k < P.recv_count[p]; ++k)
            if (Q.gid[Q.send_slots[off + k]] != P.gid[ghost_at + k]) throw std::logic_error("ghost exchange order differs");
          ghost_at += P.recv_count[p];
        }
        if (ghost_at != nvis) throw std::logic_error("ghost counts do not add up");
        // restriction octants
        if (l > 0) {
          const MGLevelPlan &C = H[r]->lev[l - 1];
          int64_t at = 0;
          for (int p = 0; p < nranks; ++p) {
            const MGLevelPlan &Q = H[p]->lev[l];
            if (P.rrecv_count[p] != Q.rsend_count[r]) throw std::logic_error("restriction counts differ");
            int64_t off = 0;
            for (int q = 0; q < r; ++q) off += Q.rsend_count[q];
            const MGLevelPlan &QC = H[p]->lev[l - 1];
            for (int64_t k = 0; k < P.rrecv_count[p]; ++k) {
              const int32_t mine = P.rrecv[2 * (at + k)], theirs = Q.rsend[2 * (off + k)];
              if (mine >= C.n_owned) throw std::logic_error("a received octant goes into a ghost parent");
              if (theirs < QC.n_owned) throw std::logic_error("a sent octant comes from an owned parent");
              if (C.gid[mine] != QC.gid[theirs] || P.rrecv[2 * (at + k) + 1] != Q.rsend[2 * (off + k) + 1]) throw std::logic_error("restriction order differs");
            }
            at += P.rrecv_count[p];
          }
          // every owned ancestor of level l - 1 gets each octant exactly once
          std::vector<unsigned char> seen((size_t)C.n_owned, 0);
          for (int64_t i = 0; i < P.n_owned; ++i)
            if (P.parent[2 * i] < C.n_owned) seen[P.parent[2 * i]] |= (unsigned char)(1u << P.parent[2 * i + 1]);
          for (size_t k = 0; k < P.rrecv.size() / 2; ++k) {
            unsigned char &m = seen[P.rrecv[2 * k]];
            if (m & (1u << P.rrecv[2 * k + 1])) throw std::logic_error("an octant arrives twice");
            m |= (unsigned char)(1u << P.rrecv[2 * k + 1]);
          }
          for (int64_t i = 0; i < C.n_owned; ++i)
            if (C.leaf[i] < 0 ? seen[i] != 0xff : seen[i] != 0) throw std::logic_error("an ancestor misses an octant (or a leaf receives one)");
        }
      }
      if (owned_total != H[0]->lev[l].n_global) throw std::logic_error("the ranks' owned nodes do not add up to the level");
    }
  } catch (const std::exception &e) {
    set_error("cup3d_debug_mg_plan_check: %s", e.what());
    return CUP3D_ESTATE;
  }
  return CUP3D_OK;
}
// TEST SUPPORT (no GPU): the local slots of a rank's grid / view whose kernels are launched BEFORE the halo exchange has completed
// (Grid::inner; cup3d_grid_ninner gives their number)
int cup3d_debug_grid_inner_blocks(const cup3d_grid_t *gh, int32_t *slots) {
  if (!gh || !slots) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  for (size_t i = 0; i < g->inner.size(); ++i) slots[i] = g->inner[i];
  return CUP3D_OK;
}
// TEST SUPPORT (no GPU): a rank's TENSORIAL view of a mesh -- the one cup3d_adapt_migrate and cup3d_grad_chi_on_tmp_over_ranks build
// internally (edge / corner neighbours and the finer leaves behind them are ghosts too; whole blocks travel) -- so that its ghost list
// can be checked against an independent consumer (tests/test_host_indexing.py)
int cup3d_debug_grid_rank_view_tensorial(const cup3d_grid_t *gh, const int32_t *owner, int rank, int nranks, cup3d_grid_t **out) {
  if (!gh || !owner || !out) return CUP3D_EINVAL;
  const Grid *g = reinterpret_cast<const Grid *>(gh);
  try {
    std::unique_ptr<Grid> tmp;
    const Grid *m = g;
    if (!g->multilevel) { tmp = g->as_mesh(); m = tmp.get(); }
    *out = reinterpret_cast<cup3d_grid_t *>(m->rank_view(owner, rank, nranks, /*tensorial=*/true).release());
  } catch (const std::exception &e) {
    set_error("cup3d_debug_grid_rank_view_tensorial: %s", e.what());
    return CUP3D_EINVAL;
  }
  return CUP3D_OK;
}
#endif

double cup3d_calc_max_timestep(double hmin, double umax, double nu, double cfl, int step, int rampup, double dt_old, double coefU[3]) {
  return cup3d_calc_max_timestep2(hmin, umax, nu, cfl, step, rampup, dt_old, coefU, 0);
}

double cup3d_calc_max_timestep2(double hmin, double umax, double nu, double cfl, int step, int rampup, double dt_old, double coefU[3],
                                int implicit_diffusion) {
  // Simulation::calcMaxTimestep, main.cpp:15268-15303 (CFL > 0); with -implicitDiffusion the diffusive limit is 0.1 after step 10
  const double dt_diffusion = (implicit_diffusion && step > 10) ? 0.1 : (1.0 / 6.0) * hmin * hmin / (nu + (1.0 / 6.0) * hmin * umax);
  const double dt_advection = hmin / (umax + 1e-8);
  double dt;
  if (step < rampup) {
    const double x = step / (double)rampup;
    const double ramp_cfl = std::exp(std::log(1e-3) * (1 - x) + std::log(cfl) * x);
    dt = std::min(dt_diffusion, ramp_cfl * dt_advection);
  } else {
    dt = std::min(dt_diffusion, cfl * dt_advection);
  }
  if (step > 2 && coefU) {  // step_2nd_start = 2
    const double a = dt_old, b = dt;
    const double c1 = -(a + b) / (a * b);
    const double c2 = b / (a + b) / a;
    coefU[0] = -b * (c1 + c2);
    coefU[1] = b * c1;
    coefU[2] = b * c2;
  }
  return dt;
}

}  // extern "C"
