// See hilbert.hpp.  Integer-only; must agree bit for bit with the reference's
// SpaceFillingCurve (main.cpp:95-319); checked against tests/golden/sfc_tables.npz.
#include "hilbert.hpp"

#include <algorithm>

namespace cup3d {

// Coordinates -> Hilbert index on a cube of side 2^bits.  Skilling, "Programming the
// Hilbert curve" (2004): undo the excess work of the Gray-coded transpose, Gray
// encode, then interleave the bit planes with x as the most significant of each triple
// (the reference's AxestoTranspose, main.cpp:107-152).
int64_t HilbertCurve::cube_index(int x, int y, int z, int bits) {
  if (bits == 0) return 0;
  uint32_t a[3] = {(uint32_t)x, (uint32_t)y, (uint32_t)z};
  const uint32_t top = 1u << (bits - 1);
  for (uint32_t q = top; q > 1; q >>= 1) {
    const uint32_t low = q - 1;
    for (int d = 0; d < 3; ++d) {
      if (a[d] & q) {
        a[0] ^= low;  // invert the low bits of the first axis
      } else {        // exchange the low bits of axis 0 and axis d
        const uint32_t swap = (a[0] ^ a[d]) & low;
        a[0] ^= swap;
        a[d] ^= swap;
      }
    }
  }
  a[1] ^= a[0];
  a[2] ^= a[1];
  uint32_t fold = 0;
  for (uint32_t q = top; q > 1; q >>= 1)
    if (a[2] & q) fold ^= q - 1;
  a[0] ^= fold;
  a[1] ^= fold;
  a[2] ^= fold;
  int64_t h = 0;
  for (int bit = 0; bit < bits; ++bit) {
    const int64_t triple = (int64_t)((a[2] >> bit) & 1u) | ((int64_t)((a[1] >> bit) & 1u) << 1) |
                           ((int64_t)((a[0] >> bit) & 1u) << 2);
    h |= triple << (3 * bit);
  }
  return h;
}

// Hilbert index -> coordinates (the reference's TransposetoAxes, main.cpp:153-192).
void HilbertCurve::cube_coords(int64_t h, int bits, int64_t xyz[3]) {
  int64_t a[3] = {0, 0, 0};
  for (int bit = 0; h > 0; ++bit, h >>= 3) {
    a[2] |= (h & 1) << bit;
    a[1] |= ((h >> 1) & 1) << bit;
    a[0] |= ((h >> 2) & 1) << bit;
  }
  if (bits > 0) {
    const int64_t side = (int64_t)2 << (bits - 1);
    const int64_t carry = a[2] >> 1;  // Gray decode
    a[2] ^= a[1];
    a[1] ^= a[0];
    a[0] ^= carry;
    for (int64_t q = 2; q != side; q <<= 1) {
      const int64_t low = q - 1;
      for (int d = 2; d >= 0; --d) {
        if (a[d] & q) {
          a[0] ^= low;
        } else {
          const int64_t swap = (a[0] ^ a[d]) & low;
          a[0] ^= swap;
          a[d] ^= swap;
        }
      }
    }
  }
  xyz[0] = a[0];
  xyz[1] = a[1];
  xyz[2] = a[2];
}

HilbertCurve::HilbertCurve(int bx, int by, int bz, int level_max) : level_max_(level_max) {
  b_[0] = bx;
  b_[1] = by;
  b_[2] = bz;
  const int widest = std::max(bx, std::max(by, bz));
  base_bits_ = 0;
  while ((1 << base_bits_) < widest) ++base_bits_;
  const int64_t ncell = (int64_t)bx * by * bz;
  const int64_t cube = (int64_t)1 << (3 * base_bits_);
  // The reference indexes with the enclosing cube's curve whenever no cube cell outside the box precedes a box cell on the
  // curve (`isRegular`, main.cpp:216-234) -- true for full cubes, but also for boxes such as 1x1x2 or 1x2x2 that happen to be
  // a prefix of the curve -- and compacts the curve otherwise.
  full_cube_ = true;
  bool seen_outside = false;
  rank_of_cell_.assign(ncell, -1);
  cell_of_rank_.assign(3 * ncell, -1);
  // Walk the enclosing cube along the curve; box cells keep their relative order.
  int64_t next = 0;
  for (int64_t h = 0; h < cube; ++h) {
    int64_t c[3];
    cube_coords(h, base_bits_, c);
    if (c[0] >= bx || c[1] >= by || c[2] >= bz) { seen_outside = true; continue; }
    if (seen_outside) full_cube_ = false;
    rank_of_cell_[(c[2] * by + c[1]) * bx + c[0]] = next;
    cell_of_rank_[3 * next + 0] = (int32_t)c[0];
    cell_of_rank_[3 * next + 1] = (int32_t)c[1];
    cell_of_rank_[3 * next + 2] = (int32_t)c[2];
    ++next;
  }
}

int64_t HilbertCurve::forward(int level, int i, int j, int k) const {
  if (level >= level_max_) return 0;  // main.cpp:239-240
  if (full_cube_) return cube_index(i, j, k, level + base_bits_);
  // per level-0 block: local curve of 8^level cells, blocks ordered by the compacted curve
  const int side = 1 << level;
  const int I = i / side, J = j / side, K = k / side;
  const int64_t local = cube_index(i - I * side, j - J * side, k - K * side, level);
  return local + rank_of_cell_[((int64_t)K * b_[1] + J) * b_[0] + I] * side * side * side;
}

void HilbertCurve::inverse(int64_t Z, int level, int ijk[3]) const {
  int64_t c[3];
  if (full_cube_) {
    cube_coords(Z, level + base_bits_, c);
    ijk[0] = (int)c[0];
    ijk[1] = (int)c[1];
    ijk[2] = (int)c[2];
    return;
  }
  const int64_t side = (int64_t)1 << level;
  const int64_t per_block = side * side * side;
  cube_coords(Z % per_block, level, c);
  const int64_t r = Z / per_block;
  ijk[0] = (int)(c[0] + cell_of_rank_[3 * r + 0] * side);
  ijk[1] = (int)(c[1] + cell_of_rank_[3 * r + 1] * side);
  ijk[2] = (int)(c[2] + cell_of_rank_[3 * r + 2] * side);
}

int64_t HilbertCurve::encode(int level, const int index[3]) const {
  // sum of the ancestors' curve positions ...
  int64_t key = 0;
  int c[3] = {index[0], index[1], index[2]};
  for (int l = level; l >= 0; --l) {
    key += forward(l, c[0], c[1], c[2]);
    c[0] /= 2;
    c[1] /= 2;
    c[2] /= 2;
  }
  // ... plus, for every finer level, the first child octet on the curve (main.cpp:300-315)
  c[0] = 2 * index[0];
  c[1] = 2 * index[1];
  c[2] = 2 * index[2];
  for (int l = level + 1; l < level_max_; ++l) {
    int64_t zc = forward(l, c[0], c[1], c[2]);
    zc -= zc % 8;
    key += zc;
    int first[3];
    inverse(zc, l, first);
    c[0] = 2 * first[0];
    c[1] = 2 * first[1];
    c[2] = 2 * first[2];
  }
  return key + level;
}

void HilbertCurve::info(int level, const int index[3], int64_t nei[27], int64_t child[8], int64_t *parent) const {
  const int n[3] = {b_[0] << level, b_[1] << level, b_[2] << level};
  int slot = 0;
  for (int di = -1; di <= 1; ++di)
    for (int dj = -1; dj <= 1; ++dj)
      for (int dk = -1; dk <= 1; ++dk)
        nei[slot++] = forward(level, (index[0] + di + n[0]) % n[0], (index[1] + dj + n[1]) % n[1],
                              (index[2] + dk + n[2]) % n[2]);
  slot = 0;
  for (int di = 0; di < 2; ++di)
    for (int dj = 0; dj < 2; ++dj)
      for (int dk = 0; dk < 2; ++dk)
        child[slot++] = forward(level + 1, 2 * index[0] + di, 2 * index[1] + dj, 2 * index[2] + dk);
  *parent = level == 0 ? 0
                       : forward(level - 1, (index[0] / 2 + n[0]) % n[0], (index[1] / 2 + n[1]) % n[1],
                                 (index[2] / 2 + n[2]) % n[2]);
}

}  // namespace cup3d
