// Block topology of one rank: ownership, ordering, face neighbours, halo plan (host).
//
// Mirrors what the hot path needs from the reference's Grid/GridMPI/Synchronizer:
//   * ownership = contiguous Hilbert (Z) ranges, GridMPI ctor, main.cpp:2959-2986
//   * local order = sorted by blockID_2 (FillPos, main.cpp:943-964)
//   * neighbours = Info::Znei with periodic wrap (384-420) + owner lookup (Tree().rank())
//   * halo plan = SynchronizerMPI_AMR::_Setup (1979-2286) reduced to face slabs, which is
//     all the star-shaped hot-path stencils read; inner/halo block split (2196-2199).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <vector>

#include "hilbert.hpp"

namespace cup3d {

constexpr int kBS = 8;             // _BS_ (Makefile:11)
constexpr int kCells = 512;        // cells per block
constexpr int32_t kNbrHalo = 0x40000000;
// nbr27 entries of a multi-level mesh: [0, kNbrCoarser) same-level slot; kNbrCoarser + slot = the coarser leaf that
// covers the neighbour position (TreePosition CheckCoarser); kNbrSkipped = not loaded (domain face with a boundary
// condition, BlockLab::load 3696-3701); kNbrFiner = finer leaves there (CheckFiner)
constexpr int32_t kNbrCoarser = 0x20000000;
constexpr int32_t kNbrSkipped = -1, kNbrFiner = -3;

// ---- the level hierarchy of the multigrid option on a multi-level mesh (multigrid.hip), for one rank.  Level l holds, as NODES, the leaves of
// level l and the ancestors (at level l) of every finer leaf.  A leaf node belongs to the leaf's owner, an ancestor to the owner of its
// first child (octant 0), recursively.  A rank's arrays of level l hold its OWNED nodes (global order) followed by GHOST nodes, ordered by
// (owner, global order): every remote node an owned node's tables name -- same-level face neighbours, the parent and the coarse neighbours
// behind coarse/fine faces of the owned nodes one level up.  All slots below are slots of those arrays.
struct MGLevelPlan {
  int64_t n_owned = 0, n_ghost = 0, n_global = 0;
  double h = 0;
  std::vector<int32_t> nbr;     // [n_owned][6]: slot | -1 zero-gradient domain face | kNbrHalo + index into cf
  std::vector<int32_t> parent;  // [n_owned][2]: slot on level l - 1, octant (x + 2y + 4z)
  std::vector<int32_t> leaf;    // [n_owned]: slot of the leaf in the rank's field arrays, -1 for an ancestor
  std::vector<int32_t> cf;      // [ncf][4]: coarse node's slot on level l - 1, direction, side, tangential parities
  std::vector<int32_t> gid;     // [n_owned + n_ghost]: global node number (checks only)
  // ghost exchange of an array of this level: what I send to p = the nodes of p's ghost list that I own, in p's ghost order
  std::vector<int64_t> send_count, recv_count;  // [nranks]
  std::vector<int32_t> send_slots;
  // restriction into level l - 1: octants of REMOTE parents that my owned nodes of this level produce (sent to the parent's owner,
  // peer-major, children in global order), and the octants of my owned parents that arrive (sender-major, the same order)
  std::vector<int64_t> rsend_count, rrecv_count;  // [nranks]
  std::vector<int32_t> rsend, rrecv;              // pairs: slot on level l - 1 (a ghost slot when sending, an owned one when receiving), octant
};
// what Grid::mg_hierarchy reads of a global multi-level mesh (kept by rank views so that the hierarchy can be built on first use)
struct MGSource {
  int bpd[3], bc[3];
  double maxextent;
  std::vector<int32_t> blevel, index;  // [nb], [nb][3] of ALL leaves, global order
  std::vector<int32_t> owner;          // [nb]; empty: one rank owns everything
  std::vector<int32_t> leaf_slot;      // [nb] slot of a leaf in the rank's field arrays (-1: not visible); empty: the global slot
  int rank, nranks;
};
struct MGHierarchy;
std::shared_ptr<MGHierarchy> build_mg_hierarchy(const MGSource &src);
struct MGHierarchy {
  int rank = 0, nranks = 1;
  std::vector<MGLevelPlan> lev;  // [0] coarsest ... finest level present
};

struct Grid {
  Grid(const int bpd[3], int level_max, int level, double maxextent, const int bc[3], int rank, int nranks);
  // multi-level (AMR) mesh on one rank: the leaves (level, Z) in any order -- what Grid::m_vInfo holds after
  // MeshAdaptation::Adapt (main.cpp:5086-5159); the octree states follow from the leaf set (Tree(), 840-855)
  Grid(const int bpd[3], int level_max, double maxextent, const int bc[3], int64_t nleaves, const int32_t *levels, const int64_t *Zs);

  std::unique_ptr<HilbertCurve> sfc;
  int bpd[3], level_max, level, bc[3];
  int nbd[3];          // blocks per dimension at `level`
  double maxextent, h;
  int rank, nranks;
  int64_t total_blocks, z_begin, z_count;  // this rank owns Z in [z_begin, z_begin+z_count)

  // local blocks in m_vInfo order
  std::vector<int64_t> Z, id2;
  std::vector<int32_t> index;  // [nb][3]
  std::vector<int32_t> nbr;    // [nb][6]: x-,x+,y-,y+,z-,z+
  std::vector<int32_t> inner, boundary;  // slots without / with a remote neighbour

  // halo plan, peer-major; a "face slab" is the w-deep layer of a block behind one face
  std::vector<int64_t> send_count, recv_count;  // [nranks]
  std::vector<int32_t> send_faces;              // slot*6 + face of each slab sent, in send order
  int64_t n_recv_faces = 0;
  int32_t corner_slot = -1;  // local slot of the block with index (0,0,0), or -1 (mean constraint, main.cpp:9287-9289)

  // ---- multi-level meshes only (multilevel == true; `level` is then the coarsest level present, `h` its spacing)
  bool multilevel = false;
  std::vector<int32_t> blevel;   // [nb] level of each block
  std::vector<double> hb;        // [nb] grid spacing of each block (Info::h)
  std::vector<int32_t> nbr27;    // [nb][27], code = (cx+1) + 3*(cy+1) + 9*(cz+1); see kNbrCoarser above
  // interface faces = faces whose same-level neighbour does not exist (FluxCorrection::prepare 676-711): each one has a
  // ghost slab produced on the device before a stencil kernel runs and a stored face-flux array.  nbr[6*slot+f] = kNbrHalo + e.
  std::vector<int32_t> amr_faces;   // [ne][2]: 6*slot + face, kind (0: neighbour is coarser -> prolong; 1: finer -> restrict)
  std::vector<int32_t> amr_fine;    // [ne][4]: interface-face index of the four finer blocks' opposite faces, quadrant
                                    // B = (fast half) + 2*(slow half) of the face (FillCase 601-661); -1 for kind 0
  std::vector<int32_t> fix_faces[3];  // interface faces of kind 1 with normal direction d (flux correction order x, y, z)
  int64_t n_amr_faces() const { return (int64_t)amr_faces.size() / 2; }
  std::vector<std::vector<int32_t>> at_;  // [level][(k*ny + j)*nx + i] -> slot of the leaf there, or -1
  // slot of the leaf (l, c) with periodic wrap, -1 if there is none at that level (multi-level meshes)
  int32_t leaf(int l, const int c[3]) const;
  // MeshAdaptation::ValidStates (main.cpp:5330-5492) on one rank: in/out states[nb] in {-1 Compress, 0 Leave, 1 Refine}
  void valid_states(int8_t *states) const;
  // leaves after MeshAdaptation::Adapt (5086-5159) applied the valid states
  void adapted_leaves(const int8_t *states, std::vector<int32_t> &levels, std::vector<int64_t> &Zs) const;
  // owner rank of every leaf of `adapted` (= this mesh after Adapt with `states`) when the leaves of this mesh are spread over
  // `nranks` ranks as owner[nb] says: children stay with the refined parent, a compressed octet's parent appears on the rank of
  // its base block (LoadBalancer::PrepareCompression 4729-4804), then Balance_Diffusion / Balance_Global (4805-5021)
  void adapted_owners(const int32_t *owner, const int8_t *states, int nranks, const Grid &adapted, int32_t *new_owner) const;
  // the same blocks as a multi-level mesh object (for uniform one-rank grids; a multi-level grid returns a copy of itself)
  std::unique_ptr<Grid> as_mesh() const;

  std::vector<int32_t> slot_of_z;  // Z - z_begin -> local slot
  // local slot of the block at (i,j,k) of this level (periodic wrap NOT applied); -1 if owned by another rank
  int32_t slot_of_index(int i, int j, int k) const;
  // all 26 neighbours + self of every local block, code = (cx+1) + 3*(cy+1) + 9*(cz+1); -1 = not loaded
  // (domain face with a boundary condition, BlockLab::load 3696-3701), -2 = owned by another rank
  std::vector<int32_t> neighbours27() const;

  int64_t nblocks() const { return n_local >= 0 ? n_local : (int64_t)Z.size(); }

  // ---- one rank's view of a multi-level mesh that is spread over several ranks (rank_view).  Block slots [0, n_local) are the
  // rank's own leaves (global order), [n_local, n_local + nghost()) are GHOST leaves: every remote leaf a local block's tables
  // refer to (same-level or coarser neighbours of the 27-point neighbourhood, finer leaves behind coarse/fine faces), ordered by
  // (owner, global order) so that each peer's blocks arrive as one contiguous run.  Z/id2/index/blevel/hb cover all visible
  // slots; nbr/nbr27 only the local ones.  Interface faces: the local blocks' faces first (n_local_faces, global order), then
  // the fine faces of ghost blocks that local coarse-side faces need for the restriction and the flux correction, ordered by
  // (owner, owner's face order).  -1 / empty on ordinary grids.
  int64_t n_local = -1, n_local_faces = -1;
  int64_t nghost() const { return n_local >= 0 ? (int64_t)Z.size() - n_local : 0; }
  std::vector<int32_t> global_slot;   // [visible slots] slot in the global mesh
  std::vector<int32_t> global_face;   // [interface faces of the view] interface-face index in the global mesh
  std::vector<int32_t> ghost_owner;   // [nghost]
  // exchange plans, peer-major: whole blocks before every stencil kernel, face-flux arrays after every flux-corrected one
  std::vector<int32_t> send_blocks;       // local slots, in the receiver's ghost order
  std::vector<int64_t> send_block_count, recv_block_count;   // [nranks]
  std::vector<int32_t> send_flux_faces;   // local interface faces (fine side), in the receiver's ghost-face order
  std::vector<int64_t> send_flux_count, recv_flux_count;     // [nranks]
  // Sub-box form of the ghost-block exchange (comm.hip): of a ghost block only the cells the rank's STAR-stencil consumers read travel
  // -- the w layers behind a shared face (same-level neighbour), the 2w layers a restriction averages (finer leaf), the cells the coarse
  // shadow patch of an interpolation takes (coarser leaves; same-level edge / corner neighbours averaged down).  Per stencil-width class
  // k (0: w = 1, 1: w = 3) the bounding box of those cells, found by replaying the consumers' index arithmetic (grid.cpp, star_boxes):
  // lo x, y, z, hi x, y, z (hi exclusive; lo = hi = 0: nothing is read).  Empty for tensorial views (mesh adaptation ships whole blocks).
  std::vector<uint8_t> ghost_box[2];                  // [nghost][6], ghost order
  std::vector<uint8_t> send_box[2];                   // [send_blocks.size()][6], send order
  std::vector<int64_t> send_cells[2], recv_cells[2];  // [nranks]: cells (per component) to / from each peer
  // tensorial: also the finer leaves behind EDGE and CORNER positions become ghosts -- what the tensorial [-1,2) tile of mesh adaptation
  // (refine_1 / RefineBlocks) averages down; the star-shaped stencils of the time step never read them
  std::unique_ptr<Grid> rank_view(const int32_t *owner, int rank, int nranks, bool tensorial = false) const;
  // the multigrid hierarchy of rank `rank` when the leaves of this (global, multi-level) mesh are owned as `owner` says (nullptr: one
  // rank owns everything); leaf_slot[global leaf] = slot of the leaf in that rank's field arrays (nullptr: the global slot itself)
  std::shared_ptr<MGHierarchy> mg_hierarchy(const int32_t *owner, int rank, int nranks, const std::vector<int32_t> *leaf_slot) const;
  // rank views: the hierarchy needs the GLOBAL leaf table, which a view does not keep -- so rank_view() leaves the little it takes to
  // build one (levels, indices and owners of all leaves: 20 bytes per leaf) in mg_source, and the hierarchy itself, with its dense maps
  // per level and the want-lists of every rank, is built by mg_plan_get() when block_solver 5 first asks for it (never, in most runs).
  // Throws std::invalid_argument for a mesh that is not 2:1 balanced; std::logic_error (a plan bug) and std::bad_alloc are not swallowed.
  std::shared_ptr<const MGSource> mg_source;
  std::shared_ptr<const MGHierarchy> mg_plan_get() const;
  mutable std::shared_ptr<const MGHierarchy> mg_plan;  // filled by mg_plan_get (one thread drives a Grid)
  Grid(const Grid &proto, int basics_only);  // box, curve and spacing of `proto`, no blocks (used by rank_view)
  int owner_of(int64_t z) const;
  static void partition(int64_t total, int rank, int nranks, int64_t *begin, int64_t *count);
};

}  // namespace cup3d
