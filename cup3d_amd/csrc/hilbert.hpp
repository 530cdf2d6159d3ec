// Hilbert-curve block indexing for the block-structured grid (host, integers only).
//
// Contract (bit-exact with the reference): class SpaceFillingCurve,
// slitvinov/CUP3D main.cpp:95-319 — forward() 237-255, inverse() 256-276,
// Encode() 287-318 — and the neighbour/child/parent ids of Info::setup, 384-420.
// The reference builds the level-0 compaction table in O(N^2) (215-235); here it is
// one O(8^base) sweep over the enclosing cube, which yields the same table.
#pragma once
#include <cstdint>
#include <vector>

namespace cup3d {

class HilbertCurve {
public:
  HilbertCurve(int bx, int by, int bz, int level_max);

  int64_t forward(int level, int i, int j, int k) const;
  void inverse(int64_t Z, int level, int ijk[3]) const;
  // cross-level sort key "blockID_2"
  int64_t encode(int level, const int index[3]) const;
  // Znei[3][3][3] (x slowest), Zchild[2][2][2], Zparent of the block at (level,index)
  void info(int level, const int index[3], int64_t nei[27], int64_t child[8], int64_t *parent) const;

  int level_max() const { return level_max_; }
  int bpd(int d) const { return b_[d]; }
  bool full_cube() const { return full_cube_; }

private:
  // Skilling's "transpose" form of the 3-D Hilbert curve on a 2^bits cube
  static int64_t cube_index(int x, int y, int z, int bits);
  static void cube_coords(int64_t h, int bits, int64_t xyz[3]);

  int b_[3];
  int level_max_;
  int base_bits_;   // smallest b with 2^b >= max(bpd)
  bool full_cube_;  // the level-0 box fills its enclosing 2^b cube ("isRegular")
  std::vector<int64_t> rank_of_cell_;  // level-0 (k*by+j)*bx+i -> position along the curve, box cells only
  std::vector<int32_t> cell_of_rank_;  // inverse: 3 ints per position
};

}  // namespace cup3d
