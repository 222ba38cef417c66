// Packed puzzle tables shared by the host packer (pw_host.cpp) and the gfx950
// kernels (pw_kernels.hip).  One PwPuzzleHeader per puzzle + one byte blob.
//
// All grids are row bitboards: bit x of row y is cell (x, y), y grows downwards
// (puzzle.py:43-50).  W, H <= 64 so a row is one uint64 and a full board fits
// the 64 lanes of one CDNA wavefront (lane = row).
#ifndef PW_FORMAT_H_
#define PW_FORMAT_H_

#include <stdint.h>

#define PW_FMT_MAGIC 0x50573032u /* "PW02" */

// palette indices written by the render kernel (RGB values: puzzle.py:65-79)
enum PwColor {
  PW_C_PAD = 0,         // zero padding of env_utils.py:75-91
  PW_C_BACKGROUND = 1,  // white canvas, puzzle.py:451
  PW_C_AWALL = 2,       // kind 1: fill 2, border 3
  PW_C_AWALL_BORDER = 3,
  PW_C_WALL = 4,        // kind 2
  PW_C_WALL_BORDER = 5,
  PW_C_AGENT = 6,       // kind 3
  PW_C_AGENT_BORDER = 7,
  PW_C_GOALOBJ = 8,     // kind 4
  PW_C_GOALOBJ_BORDER = 9,
  PW_C_MOVABLE = 10,    // kind 5
  PW_C_MOVABLE_BORDER = 11,
  PW_C_GOAL_BORDER = 12,
  PW_NUM_COLORS = 13,
};

// "absent neighbour" bits of a cell inside its own object (puzzle.py:614-638):
// a set bit means the neighbour cell is NOT part of the object, so a border strip
// is drawn on that side / corner.
#define PW_NB_L 0x01u   // (x-1, y)
#define PW_NB_R 0x02u   // (x+1, y)
#define PW_NB_U 0x04u   // (x, y-1)
#define PW_NB_D 0x08u   // (x, y+1)
#define PW_NB_UL 0x10u  // (x-1, y-1)
#define PW_NB_UR 0x20u  // (x+1, y-1)
#define PW_NB_DL 0x40u  // (x-1, y+1)
#define PW_NB_DR 0x80u  // (x+1, y+1)

// Cell code (uint32) used by the render kernel:
//   bits  0..7   absent-neighbour mask of the top-most opaque layer in the cell
//   bits  8..11  kind of that layer: 0 background, 1 agent wall, 2 wall, 3 agent,
//                4 goal object, 5 movable          (fill colour 2*kind, border 2*kind+1)
//   bits 12..15  zero
//   bits 16..23  painter priority (0 static, 1 + object index for movables): the
//                per-env grid is composed with LDS atomicMax on (code & 0x00FFFFFF)
//   bits 24..31  OR of the absent-neighbour masks of all goal outlines covering the
//                cell (goals are static; drawn last, border only, puzzle.py:458)
#define PW_CODE_KIND_SHIFT 8
#define PW_CODE_PRIO_SHIFT 16
#define PW_CODE_GOAL_SHIFT 24

struct PwObjEntry {      // 4 bytes per movable
  uint8_t w, h;          // bounding box in cells
  uint16_t row_off;      // index of the first shape row (uint64 units) in the shape section
};

// Fixed 320-byte record per puzzle.  Everything a step needs besides the row bitboards
// (object table, goal and initial positions) sits inside the record so that a wavefront
// fetches it with ONE level of indirection (wave-uniform scalar loads); the bitboards and the
// render tables live in the blob at `base`.
struct PwPuzzleHeader {
  uint32_t base;         // byte offset of this puzzle's data in the blob (8 B aligned)
  uint8_t W, H, N, G;    // grid size incl. border walls, #movables (agent first), #goals
  uint32_t off_wall;     // uint64[H]  wall rows                     (offsets relative to base)
  uint32_t off_awall;    // uint64[H]  wall | agent-wall rows
  uint32_t off_shapes;   // uint64[sum h_j]  object shape rows, bit x = cell (x, row)
  uint32_t off_static;   // uint32[H*W] static cell codes (walls, agent walls, goal masks)
  uint32_t off_mcells;   // uint32[n_mcells]: cx | cy<<8 | absent mask<<16 | object<<24
  uint32_t n_mcells;
  uint32_t has_aw;
  uint32_t off_small;    // uint64[N]  8x8 bitboard of every movable with w, h <= 8 (bit 8 * row + column), 0 otherwise:
                         //            the push / wall tests of small objects run on it without touching the shape rows
  uint32_t reserved[6];
  PwObjEntry objtab[32];  // @64   bounding boxes + shape row offsets
  int8_t goal[32][2];     // @192  goal k is the target of movable k+1
  int8_t init[32][2];     // @256  initial positions
};

#ifdef __cplusplus
#include <cstddef>
// (pushworld_amd/_capi.py reads N out of the packed header array: PUZZLE_HEADER_BYTES / PUZZLE_HEADER_N_OFFSET)
static_assert(sizeof(PwPuzzleHeader) == 320 && offsetof(PwPuzzleHeader, N) == 6, "PwPuzzleHeader layout is part of the packed set format");
#endif

// the `off_small` entry of an object with bounding box w x h and shape rows `rows` (bit x of rows[r] = cell (x, r));
// host side (the device packer in pw_generate.inc builds the same value from its grid rows)
static inline uint64_t pw_small_board(const uint64_t* rows, int w, int h) {
  if (w > 8 || h > 8) return 0;
  uint64_t s = 0;
  for (int r = 0; r < h; r++) s |= (rows[r] & 0xffull) << (8 * r);
  return s;
}

#endif  // PW_FORMAT_H_
