// Zone-colour logic of the renderer, shared by the host (static table build at engine
// creation) and the gfx950 kernels (per-environment patches).
//
// A cell is drawn as 3 x 3 zones: [0, bw) | [bw, ppc - bw) | [ppc - bw, ppc) in x and y.  For
// one cell layer with absent-neighbour mask m (pw_format.h) zone (zx, zy) is a border zone iff
// a strip of puzzle.py:631-638 covers it.
#ifndef PW_ZONE_H_
#define PW_ZONE_H_

#include <stdint.h>

#include "pw_format.h"

#if defined(__HIPCC__)
#define PW_HD __host__ __device__ __forceinline__
#else
#define PW_HD inline
#endif

// bit zx of the result: zone (zx, zy) is covered by a border strip
PW_HD uint32_t pw_zone_border_bits(uint32_t m, int zy) {
  uint32_t u = 0, cl = 0, cr = 0;
  if (zy == 0) {
    u = (m >> 2) & 1u;   // U strip
    cl = (m >> 4) & 1u;  // UL corner
    cr = (m >> 5) & 1u;  // UR corner
  } else if (zy == 2) {
    u = (m >> 3) & 1u;   // D strip
    cl = (m >> 6) & 1u;  // DL corner
    cr = (m >> 7) & 1u;  // DR corner
  }
  const uint32_t b0 = (m & 1u) | u | cl;
  const uint32_t b2 = ((m >> 1) & 1u) | u | cr;
  return b0 | (u << 1) | (b2 << 2);
}

// Zone-table entry of one cell sub-row: colour(zx=0) | colour(zx=1) << 4 | colour(zx=2) << 8.
//   kind   0 background, 1 agent wall, 2 wall, 3 agent, 4 goal object, 5 movable
//   ob     border bits of the opaque layer (pw_zone_border_bits)
//   gb     zones covered by a goal outline (drawn last, puzzle.py:458)
PW_HD uint32_t pw_zone_entry(uint32_t kind, uint32_t ob, uint32_t gb) {
  const uint32_t fill = kind ? 2u * kind : (uint32_t)PW_C_BACKGROUND;
  const uint32_t edge = kind ? 2u * kind + 1u : (uint32_t)PW_C_BACKGROUND;
  uint32_t v = 0;
  for (int zx = 0; zx < 3; zx++) {
    uint32_t col = ((ob >> zx) & 1u) ? edge : fill;
    if ((gb >> zx) & 1u) col = PW_C_GOAL_BORDER;
    v |= col << (4 * zx);
  }
  return v;
}

// zones of an existing entry that show a goal outline (only goal outlines use that colour)
PW_HD uint32_t pw_entry_goal_bits(uint32_t e) {
  return (uint32_t)((e & 15u) == PW_C_GOAL_BORDER) | ((uint32_t)(((e >> 4) & 15u) == PW_C_GOAL_BORDER) << 1) |
         ((uint32_t)(((e >> 8) & 15u) == PW_C_GOAL_BORDER) << 2);
}

#endif  // PW_ZONE_H_
