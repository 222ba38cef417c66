/*
 * pushworld_amd.h -- C ABI of the MI355X-native batched PushWorld step engine.
 *
 * This is the drop-in boundary for the ONE hot path of google-deepmind/pushworld
 * that this library accelerates (agent move -> push-chain closure -> collision
 * test -> goal/reward -> RGB observation).  The reference has no FFI of its own;
 * each entry point below cites the reference function it replaces (paths are
 * relative to the reference checkout).  INTEGRATION.md shows the ctypes / C++
 * stubs a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types.
 *   - every function returns 0 on success or a negative PW_E* code; no C++
 *     exception crosses the ABI; pw_last_error() returns a thread-local message.
 *   - an empty batch (batch / num_states <= 0, num_steps == 0) is a no-op that returns PW_OK without looking at
 *     the buffer pointers (an empty tensor's data pointer is NULL);
 *   - "device pointers" are caller-owned HBM buffers (e.g. tensor.data_ptr());
 *     `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *     are asynchronous on that stream; the engine never allocates per call.  Its per-environment scratch (a 16 B
 *     record per environment for the page-ordered render, 4 B for pw_step_render_delta) is allocated by
 *     pw_engine_create when PwEngineConfig::max_batch is given (or by pw_obs_alloc for its batch); with max_batch 0 the
 *     first call of a batch size allocates it, and that one call must not sit inside a graph capture.  The records are
 *     written and read by the kernels of one call, so all calls on one engine must go to ONE stream (or be
 *     event-ordered by the caller).
 *   - no entry point changes the calling thread's current HIP device: everything an engine /
 *     set / search allocates or launches lands on the device of its puzzle set, and the
 *     caller's device is restored on return.
 *   - calls on one engine must be externally serialised (like the reference
 *     objects, puzzle.py:310 / pushworld_puzzle.h:178-180, which are not
 *     re-entrant); different engines are independent.
 *
 * State layout in HBM
 *   pos        int8  [B][NP][2]   (x, y) object origins, agent first, env-major;
 *                                 NP = pw_engine_npad() in {4, 8, 16, 32}
 *   puzzle_id  int32 [B]          index into the puzzle set
 *   steps      int32 [B]          steps since the last reset (gym_env.py:201)
 *   obs        uint8|float32 [B][Hp*ppc][Wp*ppc][3] with a caller-chosen env
 *                                 stride (bytes, multiple of 16)
 */
#ifndef PUSHWORLD_AMD_H_
#define PUSHWORLD_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PW_ABI_VERSION 4

/* error codes */
#define PW_OK 0
#define PW_EINVAL (-1)     /* bad argument                                   -> ValueError    */
#define PW_EPARSE (-2)     /* malformed puzzle text (ragged rows, no agent)  -> ValueError    */
#define PW_EGOAL (-3)      /* goal without movable (puzzle.py:230-232)        -> AssertionError */
#define PW_ELIMIT (-4)     /* puzzle exceeds engine limits (W,H<=64, N<=32)   -> ValueError    */
#define PW_EDEVICE (-5)    /* HIP runtime error / no device                   -> RuntimeError  */
#define PW_ENOMEM (-6)
#define PW_EELEMENT (-7)   /* empty element name ("M1++G1"): puzzle.py:218 elem_id[0]  -> IndexError     */

/* object orderings (SURVEY trap T1) */
#define PW_ORDER_PYTHON 0  /* puzzle.py:170-257: agent, goals descending, rest in file order */
#define PW_ORDER_CPP 1     /* pushworld_puzzle.cc:262-321: agent, goals ascending, rest ascending */

/* observation element types */
#define PW_OBS_U8 0        /* puzzle.py:426-469 image bytes, zero padded                        */
#define PW_OBS_F32 1       /* env_utils.py:44-91: uint8 -> float32 / 255, zero padded            */

/* pw_step flags */
#define PW_STEP_AUTORESET 1u /* next-step autoreset: envs whose terminated|truncated flag is
                                set on entry are reset (reward 0) instead of stepped          */

#define PW_MAX_DIM 64
#define PW_MAX_OBJECTS 32
#define PW_POSITION_LIMIT 10000 /* pushworld_puzzle.h:37 */

typedef struct PwPuzzle PwPuzzle;       /* one parsed puzzle (host)                     */
typedef struct PwPuzzleSet PwPuzzleSet; /* packed bitboard/render tables (host + HBM)   */
typedef struct PwEngine PwEngine;       /* step/render configuration bound to a set     */

typedef struct PwPuzzleInfo {
  int32_t width, height;      /* puzzle.py:160-161 (includes the border walls)            */
  int32_t num_movables;       /* puzzle.py:261                                           */
  int32_t num_goals;
  int32_t num_wall_cells;
  int32_t num_agent_wall_cells; /* raw "aw" cells (without the walls, see trap T2)        */
  int32_t has_agent_walls;
  int32_t order;
} PwPuzzleInfo;

typedef struct PwEngineConfig {
  int32_t max_steps;        /* < 0: no truncation (max_steps=None, gym_env.py:223); >= 0:
                               truncated = steps >= max_steps, so 0 truncates every step   */
  int32_t pixels_per_cell;  /* puzzle.py:26, >= 1 + 2 * border_width                      */
  int32_t border_width;     /* puzzle.py:22, >= 1                                         */
  int32_t obs_dtype;        /* PW_OBS_U8 | PW_OBS_F32                                     */
  int32_t pad_cell_height;  /* observation frame in cells (env_utils.py:44-91);           */
  int32_t pad_cell_width;   /*   0 = maximum over the puzzle set (gym_env.py:80-82)       */
  int32_t max_batch;        /* > 0: the engine-owned per-environment scratch (page records of the page-ordered
                               render, dirty-row records of pw_step_render_delta) is allocated by pw_engine_create
                               for this many environments: no entry point allocates afterwards for batches up to
                               it, so the FIRST pw_step_render call may already sit inside a HIP graph capture.
                               0: allocated by the first call of a batch size (never in a steady-state loop)      */
} PwEngineConfig;

const char* pw_last_error(void);
int pw_abi_version(void);
/* number of visible HIP devices, or PW_EDEVICE */
int pw_device_count(void);

/* ------------------------------------------------------------------ puzzles (host)
 * Replaces PushWorldPuzzle.__init__ parsing, puzzle.py:130-257 /
 * pushworld_puzzle.cc:191-321.  No collision hash-tables are built: the engine
 * evaluates the reference's collision predicate on row bitboards. */
int pw_puzzle_parse(const char* text, size_t len, int order, PwPuzzle** out);
void pw_puzzle_destroy(PwPuzzle* p);
int pw_puzzle_info(const PwPuzzle* p, PwPuzzleInfo* info);
/* xy: int32 [num_movables][2] -- puzzle.py:257 initial_state */
int pw_puzzle_initial_state(const PwPuzzle* p, int32_t* xy);
/* xy: int32 [num_goals][2] -- puzzle.py:228,256 goal_state */
int pw_puzzle_goal_state(const PwPuzzle* p, int32_t* xy);
/* cells of movable `obj` relative to its origin (PushWorldObject.cells, puzzle.py:90).
 * Returns the cell count (writes at most `cap` pairs) or a negative error. */
int pw_puzzle_object_cells(const PwPuzzle* p, int obj, int32_t* xy, int cap);
int pw_puzzle_goal_cells(const PwPuzzle* p, int goal, int32_t* xy, int cap);
int pw_puzzle_wall_cells(const PwPuzzle* p, int32_t* xy, int cap);       /* puzzle.py:254 */
int pw_puzzle_agent_wall_cells(const PwPuzzle* p, int32_t* xy, int cap); /* raw "aw" cells */
/* element id ("a", "m3", ...) of movable `obj`; returns length */
int pw_puzzle_object_name(const PwPuzzle* p, int obj, char* buf, int cap);

/* --------------------------------------------------------------- puzzle set (HBM)
 * device >= 0: tables are uploaded to that HIP device.  device < 0: host-only
 * packing (inspection/tests; engines cannot be created on it). */
int pw_puzzleset_create(const PwPuzzle* const* puzzles, int n, int device, PwPuzzleSet** out);
void pw_puzzleset_destroy(PwPuzzleSet* s);
int pw_puzzleset_size(const PwPuzzleSet* s);
int pw_puzzleset_max_dims(const PwPuzzleSet* s, int* max_w, int* max_h, int* max_n);
/* packed table image (host copy), for tests */
int pw_puzzleset_blob(const PwPuzzleSet* s, const void** data, size_t* bytes);
/* Packed puzzle-set file: the compiled pool, byte for byte what sits in HBM, behind a
 * 64-byte header with format version and FNV-1a checksum.  Loading replaces parsing the puzzle
 * texts again (puzzle.py:130-311 costs the reference 3 ms .. 1.5 s per puzzle).  pw_puzzleset_load
 * validates the checksum and every table offset; PW_EPARSE for a foreign / corrupt / truncated file. */
int pw_puzzleset_save(const PwPuzzleSet* s, const char* path);
int pw_puzzleset_load(const char* path, int device, PwPuzzleSet** out);

/* ------------------------------------------------ generation / transforms / packing on the device
 * A puzzle as a *symbol grid*: one byte per cell of the file's grid (without the border walls the parser adds),
 * row-major in a square slot of slot_w x slot_w bytes, one element per cell.  The generator, the transform and the
 * packer are kernels: generate -> (transform) -> pack -> train never goes through puzzle text.  The packer writes
 * exactly the tables pw_puzzle_parse + pw_puzzleset_create produce for the same puzzle. */
#define PW_SYM_EMPTY 0x00   /* "."                                                    */
#define PW_SYM_WALL 0x01    /* "W"                                                    */
#define PW_SYM_AWALL 0x02   /* "AW"                                                   */
#define PW_SYM_AGENT 0x03   /* "A"                                                    */
#define PW_SYM_MOVABLE 0x40 /* "M<k>" = PW_SYM_MOVABLE | k, k < 48                     */
#define PW_SYM_GOAL 0x80    /* "G<k>" = PW_SYM_GOAL | k                                */

typedef struct PwGenConfig {   /* generate_level0_puzzles, generate.py:136-259 (same meaning, inclusive ranges) */
  uint64_t seed;               /* puzzle i of a seed is a pure function of (seed, i): counter-based pw_mix64 stream */
  int32_t min_size, max_size;  /* width and height of the file grid, drawn independently; max_size = slot width   */
  int32_t min_walls, max_walls;
  int32_t min_obstacles, max_obstacles;
  int32_t min_goal_objects, max_goal_objects; /* 1 .. 2 */
  int32_t complex_shapes;      /* 0: single cells ("simple"), 1: + dominoes and trominoes ("complex")            */
} PwGenConfig;

/* Puzzles first .. first + count - 1 of the stream: grids uint8 [count][max_size * max_size], dims int32 [count][2]
 * (width, height; 0 0 if no puzzle could be completed).  device >= 0: device buffers, one thread per puzzle;
 * device < 0: the same function on the host (host buffers) -- identical output. */
int pw_generate_level0(int device, const PwGenConfig* cfg, uint64_t first, int32_t count, uint8_t* grids,
                       int32_t* dims, void* stream);
/* Host only, for the distribution tests: failed[i] = how many generate_puzzle attempts of puzzle first + i raised
 * FailedToGenerateError (generate.py:236-257 retries silently) before the one that was kept; 1000 = gave up. */
int pw_generate_level0_attempts(const PwGenConfig* cfg, uint64_t first, int32_t count, int32_t* failed);
/* The 8 dihedral variants of every grid (transform.py:21-48): out uint8 [count][8][slot_w * slot_w], out_dims int32
 * [count][8][2]; variant v = 4 * flipped + clockwise quarter turns (names r0 r90 r180 r270 r0_flipped ...), the
 * top-bottom flip applied before the rotation. */
int pw_transform_grids(int device, const uint8_t* grids, const int32_t* dims, int32_t count, int32_t slot_w,
                       uint8_t* out, int32_t* out_dims, void* stream);
/* Packs `count` device-resident grids into a puzzle set on `device` (one wavefront per puzzle) -- the device-side
 * pw_puzzle_parse + pw_puzzleset_create.  Errors as the parser's: PW_EPARSE (no agent), PW_EGOAL (goal without
 * movable), PW_ELIMIT.  Synchronises `stream`. */
int pw_puzzleset_from_grids(int device, const uint8_t* grids, const int32_t* dims, int32_t count, int32_t slot_w,
                            int order, PwPuzzleSet** out, void* stream);
/* .pwp text of a HOST grid (two spaces between tokens like generate.py:133); returns the length, writes <= cap - 1 chars */
int pw_grid_to_text(const uint8_t* grid, int32_t width, int32_t height, int32_t slot_w, char* buf, int32_t cap);
/* packed header array (host copy), for tests */
int pw_puzzleset_headers(const PwPuzzleSet* s, const void** data, size_t* bytes);

/* Overlap tables (PW_OPT_STEP_TABLES) of a set as the engine builds them, for tests / inspection (host only, works on
 * device < 0 sets).  The compressed form of the reference's collision tables (puzzle.py:259-311, :522-593):
 *   pair table (i, j), R rows:  bit (rx + w_i - 1) of row (ry + h_i - 1) = movable i at (rx, ry) relative to movable j
 *                               overlaps it;  "i pushes j with displacement d" = overlap at r + d and not at r
 *   wall table j, H + 2 rows:   bit (x + 1) of row (y + 1) = movable j at (x, y) overlaps a wall (j = 0: or agent wall)
 * mode: as PW_OPT_STEP_TABLES.  dir: uint32 [count][4] = {pair_off, wall_off, R | (H + 2) << 16, 0} in 8-byte words (pair_off 0 = no tables; pair
 * table (i, j) at pair_off + (i * N + j) * R, wall table j at wall_off + j * (H + 2)).  Returns the number of words
 * (writes them when cap_words suffices). */
int64_t pw_puzzleset_overlap_tables(const PwPuzzleSet* s, int mode, uint64_t* words, int64_t cap_words, uint32_t* dir);

/* ---------------------------------------------------------------------- engine */
int pw_engine_create(const PwPuzzleSet* s, const PwEngineConfig* cfg, PwEngine** out);
void pw_engine_destroy(PwEngine* e);
int pw_engine_npad(const PwEngine* e);                 /* NP of the pos layout           */
int pw_engine_obs_shape(const PwEngine* e, int* h, int* w, int* c); /* pixels             */
int64_t pw_engine_obs_bytes(const PwEngine* e);        /* h*w*c*sizeof(elem), unpadded    */
/* name of the kernel pw_render launches for this engine (profiling aid); returns its length */
int pw_engine_render_kernel(const PwEngine* e, char* buf, int cap);
int64_t pw_engine_obs_stride(const PwEngine* e);       /* recommended env stride (16 B aligned) */

/* Engine options: kernel selection for tests / A-B runs and profiling aids.  All 0 after
 * pw_engine_create; none changes a result, only which kernel computes it. */
#define PW_OPT_STEP_KERNEL 1       /* 0 lane group per env (default), 1 wavefront per env, 2 lane per env */
#define PW_OPT_FUSED_STEP_RENDER 2 /* 1: pw_step_render is ONE launch (step inside the per-env render workgroups) */
#define PW_OPT_RENDER_KERNEL 3     /* 0 automatic, 1 per-environment LDS kernel also where a page kernel applies */
#define PW_OPT_PAGE_SLICE_ENVS 4   /* page kernels: environments per launch (0 = as many as 2^31 chunks allow) */
#define PW_OPT_SEARCH_CHUNK 5      /* pw_search_create: parents per expansion pass (0 = 2^20) */
#define PW_OPT_PROFILE_RENDER 6    /* n > 0: time the next n render launches with HIP events on their stream */
/* launch configuration of the page-ordered render kernel (same bytes, different speed; see pw_engine_tune_render) */
#define PW_OPT_PAGE_ORDER 7        /* which 4 KiB page a workgroup writes: 0 page = workgroup index, 1 the buffer in 8 >> RUN_LOG2
                                      contiguous parts, XCD k sweeping part k mod parts (RUN_LOG2 0: one eighth per XCD,
                                      1: quarters shared by two XCDs), 2 every XCD writes runs of 2^RUN_LOG2 pages */
#define PW_OPT_PAGE_RUN_LOG2 8     /* log2 of the run length of order 2 (default 6) / parts selector of order 1 */
#define PW_OPT_PAGE_LDS_PAD_KB 9   /* KiB of unused dynamic LDS per workgroup: caps the workgroups per CU, i.e. the width
                                      of the chip-wide write front (default 7 for uint8, 8 for float32 observations) */
#define PW_OPT_STEP_LDS_TABLES 10   /* the lane-group step kernel copies the puzzle's wall / shape row bitboards into LDS first
                                      and reads them from there: 0 automatic (launches of >= 4 steps on sets with more than
                                      16 movables per puzzle, where it measures ~7 % faster), 1 always, 2 never */
#define PW_OPT_TUNED_NS 11          /* read-only: nanoseconds per render launch measured for the configuration the last
                                      pw_engine_tune_render kept (0 before the first call) */
#define PW_OPT_STEP_WIDE_GROUPS 12   /* sets with 17..32 movables per puzzle (N_pad 32): 0 (default) 16 lanes per environment,
                                      two movables per lane (4 environments per wavefront); 1: 32 lanes, one movable each */
#define PW_OPT_PAGE_LOAD_ALL 13      /* ppc-3 page kernel: 1 = every page loads its static-image chunks, also the all-zero pages of
                                      the frame padding (a more even write front; a candidate of pw_engine_tune_render) */
#define PW_OPT_OBS_CHUNK_MB 14       /* pw_obs_alloc: MiB per physical chunk (0 = default, 32 MiB) */
#define PW_OPT_OBS_ACCEPT_GBS 15     /* pw_obs_alloc_tuned: a candidate on which the tuned render reaches this many GB/s is
                                      kept without looking further (default 7050 = 0.88 of the 8 TB/s peak) */
#define PW_OPT_STEP_TABLES 16         /* overlap tables for the lane-group step / expansion / search kernels (the reference's
                                      collision tables, puzzle.py:259-311, with the four actions sharing one table; one or two
                                      8-byte loads instead of a loop over object rows): 0 (default) and 1 every puzzle that fits
                                      (grids up to 62 columns, movables up to 32 wide, 256 MB in all) -- when that is every puzzle
                                      of the set the kernels carry no row loops at all; 2 none; 3 only the puzzles with a movable
                                      beyond 8 x 8 cells (kernels with both paths).  Setting it rebuilds the tables (synchronises
                                      the device) */
#define PW_OPT_STEP_TABLE_BYTES 17   /* read-only: bytes of overlap tables (and, for engines of at most 64 puzzles, push tables) in HBM */
#define PW_OPT_STEP_TABLE_PUZZLES 18 /* read-only: puzzles of the set that have overlap tables */
#define PW_OPT_STEP_NARROW_GROUPS 19 /* sets with 9..16 movables per puzzle (N_pad 16): 8 lanes per environment, two movables per lane
                                      (8 environments per wavefront) instead of 16 lanes: 0 automatic (with the table-only
                                      kernels), 1 always, 2 never.  N_pad 32 sets with the table-only kernels: 0 / 1 workgroups of
                                      32 environments run 8-lane groups unless one of them has more than 16 movables, 2 never
                                      (16-lane groups for every environment) */
#define PW_OPT_STEP_BLOCK_ORDER 20   /* lane-group step kernels: 0 workgroups take the environments in index order, 1 in reverse --
                                      for batches sorted by puzzle whose expensive puzzles (many / big movables) come last:
                                      they then start first and the cheap ones fill the tail of the launch; + 2: a contiguous eighth of
                                      the blocks per XCD (A/B runs; measured slower: profiles/r05_step_block_order.txt) */
#define PW_OPT_STEP_LANE_BATCH 21    /* state-only launches (pw_step, pw_rollout) of at least this many environments run ONE LANE per
                                      environment (the table-only formulation, 64 environments per wavefront: ~3x fewer
                                      instructions per environment than a lane group, but an eighth of the wavefronts -- it wins
                                      once the batch alone fills the chip; needs overlap tables for every puzzle of the set).
                                      0 = default (sets with up to 16 movables per puzzle: 131 072 for one step per launch, 196 608
                                      for pw_rollout; N_pad 32 sets: 327 680 and 2 097 152); a threshold no batch reaches (2^31) = never.
                                      Also the number of states from which pw_expand4 and the passes of pw_search_expand run one
                                      lane per state (default 131 072; PW_OPT_STEP_KERNEL lane forces it for every size) */
#define PW_OPT_STEP_BOARDS 22        /* sets whose puzzles ALL fit into 8 x 8 cells (grid with its border walls: the 5 x 5 Level-0
                                      families) and have at most 8 movables: state-only launches (pw_step, pw_rollout) run one lane
                                      per environment on whole-grid uint64 boards -- registers only, no table lookups.
                                      0 (default) automatic, 2 never (the lane groups) */
#define PW_OPT_STEP_BOARD_SET 23     /* read-only: 1 when the engine's set qualifies for PW_OPT_STEP_BOARDS */
#define PW_OPT_EXPAND_LDS_TABLES 24  /* pw_expand4 with one lane per state: 0 (default) the kernel that keeps the puzzle's push tables in
                                      LDS (pw_expand4_v2_kernel: 2 .. 16 movables, tables up to 112 KB, 16-byte aligned output
                                      buffers) wherever it applies, 2 never (pw_expand4_lane_kernel: tables read from HBM through L1, round 3's
                                      staging of 1 / 2 actions at a time), 3 never + all four actions staged at once and non-temporal
                                      stores (what that kernel does by itself up to 14 movables when option 0 sends a launch to it) */
#define PW_OPT_EXPAND_TILE_ORDER 25    /* pw_expand4_v2_kernel: which 64-state tiles a wavefront takes: 0 interleaved over the workgroups,
                                      1 XCD x (workgroup index mod 8) sweeps the x-th contiguous eighth of the frontier; + 2: plain
                                      instead of non-temporal stores (A/B measurements only) */
#define PW_OPT_EXPAND_PREFETCH 26      /* pw_expand4_v2_kernel: 2 (and -1, the default, for persistent workgroups) software pipeline -- a tile's successors stay staged in
                                      LDS and leave, as non-temporal whole-line stores, after the NEXT tile's parent rows have been
                                      requested; 0 a tile loads its rows at its top and stores at its end (also what runs where the
                                      pipeline's staging does not fit in 78 KB of LDS) */
#define PW_OPT_EXPAND_GROUPS_PER_CU 27 /* pw_expand4_v2_kernel: persistent workgroups per CU, 1 .. 64 (0 = automatic: with small tables and up to 8
                                      movables one workgroup per 4 tiles of 64 states -- 8 tiles at 7 and 8 movables -- whatever the
                                      frontier's size; else 8 per CU, or what is resident (at most 2) with tables beyond 32 KB) */
#define PW_OPT_STEP_MIXED_GROUPS 30 /* N_pad 8 / 16 sets with overlap tables for every puzzle: 0 (default) the lanes per environment are chosen per
                                      workgroup of 32 environments (4 / 8 / 8 with two movables per lane), 2 never (one choice per set) */
#define PW_OPT_EXPAND_WG_WAVES 29 /* pw_expand4_v2_kernel with tables so large that one workgroup fits a CU: 4 or 8 wavefronts per workgroup (0 = automatic: 8
                                    where its LDS has staging room for them) */
#define PW_OPT_SEARCH_BATCH_GROUPS_PER_CU 28 /* pw_search_batch: persistent workgroups per CU (0 = automatic) */
#define PW_OPT_STEP_QUAD16 31        /* puzzles whose grid fits 16 x 16 cells with at most 8 movables of at most 16 x 8 cells (every Level-0
                                      puzzle): workgroups whose 32 environments all play such puzzles keep every board in registers, four
                                      lanes per environment, the puzzle in ONE 576-byte record (no table, no header walk): 0 automatic
                                      (sets with overlap tables for every puzzle: the fallback for states outside the grid), 2 never */
#define PW_OPT_SEARCH_KEYS 33        /* closed set of pw_search_create (read at creation): 0 (default) fingerprint + index entries; 1 where a
                                      state packs into 63 bits (bits per coordinate x movables) and the start state overlaps no wall, a
                                      published entry is the packed state itself (a visited state is recognised without reading the
                                      stored state).  Same results and, measured, the same HBM bytes; kept for the A/B */
#define PW_OPT_EXPAND_PAIR_DIMS 34   /* pw_expand4 with the tables in LDS: 0 automatic -- byte pair tables sized PER PAIR (h_i + h_j + 2 rows of
                                      w_i + w_j + 2 bytes) where the set-wide 2 max_h + 2 by 2 max_w + 2 tables exceed 16 KB, 7 .. 16
                                      movables --, 2 never (set-wide tables, or the lane kernel where those do not fit LDS) */
#define PW_OPT_STEP_QUAD16_PUZZLES 32 /* read-only: puzzles of the set with such a record */
#define PW_OPT_MAILBOX_MODE 35       /* pw_mailbox_open (A/B runs): bits 0-1 who reads the host's word across the link -- 0 every wavefront, 1 one
                                      wavefront per workgroup, 2 one wavefront of the first workgroup, which passes it on through device
                                      memory, 3 a workgroup of its own that does nothing else and runs ahead of the stepping
                                      ones; bit 2 (+4): system-scope fences around a step instead of system-scope accesses; 11 (default, round 6) =
                                      3 pipelined: the relay wavefront reads all the slots of the ring in one trip, a stepping wavefront asks
                                      for the word two steps ahead and the next step's actions before it steps, waits ONCE after the step (for them and for the
                                      previous step's stores), writes the number of steps it has completed into a word of its own -- a publisher workgroup
                                      publishes the smallest as done: no arrive counter -- and issues this step's stores without waiting for them; bit 4 (+16): wavefront 0 keeps the per-phase clock that
                                      pw_mailbox_close_profile returns (off by default: its clock reads made wavefront 0 the slowest of the launch).  Same results (C2 round trip 38 / 10.8 /
                                      6.2 / 6.0 us: profiles/r05_mailbox.json; 11: profiles/r06_mailbox.json). */
#define PW_OPT_BIND_MIN_ENVS 36      /* pw_batch_bind: a puzzle is bound when at least this many environments of the batch play it (0 = default 48) */
#define PW_OPT_BIND_FUSED 37         /* launches of a partly bound batch: 0 (default) single steps run segments and lane groups in ONE launch (where the
                                      per-workgroup lane-group kernel applies), launches of several steps as two kernels side by side on two
                                      streams (joined by events on the caller's stream); 2 two launches one after the other on the caller's
                                      stream (A/B runs) */
#define PW_OPT_BIND_PUZZLES 38       /* read-only: puzzles of the set that can be bound (their table block fits 16 KB of LDS) */
#define PW_OPT_BIND_MISMATCHES 39    /* read-only (synchronises the device): environments that bound launches found with a puzzle id other than
                                      the one they were bound to, since the engine was created -- 0 unless the caller changed puzzle_id
                                      behind the binding's back */
#define PW_OPT_BIND_LANES 42         /* pw_batch_bind (read when binding): lanes per environment of the segments -- 0 automatic (1 up to 7 movables, 2 from 8,
                                      4 from 12, 8 from 17: a step's latency grows with the movables and a launch lasts as long as its slowest
                                      segment), 1 / 2 / 3 / 4 = at most 1 / 2 / 4 / 8 (A/B runs) */
#define PW_OPT_BIND_ROLLOUTS 43      /* launches of several steps (pw_rollout) on a bound batch: 0 (default) the segments when EVERY environment of the batch
                                      is bound, else the lane groups for all of them (measured: next to lane groups that fill the chip the
                                      segment role does not finish sooner); 1 always (segments and lane groups side by side on two streams), 2 never */
#define PW_OPT_BIND_MAX_KB 44        /* pw_batch_bind (read when binding): puzzles whose table block exceeds this many KiB of LDS stay with the lane groups
                                      (0 = default 48: every benchmark puzzle is bound; the launches carry as much LDS as the largest BOUND block) */
#define PW_OPT_BIND_SPREAD 45         /* pw_batch_bind (read when binding): environments per wavefront of the segment kernels -- 0 automatic (= 64: measured,
                                      fewer do not help: DESIGN K1g), 1 .. 5 = at most 64 / 32 / 16 / 8 / 4 (lanes per environment included; the other
                                      lanes idle); + 16 x (1 .. 5): the same for the listed environments of pw_step_mseg_kernel alone (else they
                                      follow the segments' value).  A/B runs; results do not depend on it */
#define PW_OPT_STEP_ONE_FUSED 46     /* pw_step_render_delta on a batch of ONE with a completion word (pw_engine_set_step_signal; frames of at least 64 KiB,
                                      engines other than uint8 / ppc 3): 1 (default) one launch -- workgroup 0 steps and hands positions + changed rows to
                                      the other seven through device memory, and of the changed rows only the changed COLUMNS are written --, 2 one launch writing
                                      whole rows, 0 the step kernel and the redraw as two launches (rounds 4-5).  A/B runs: same observations. */
#define PW_OPT_MAILBOX_SEG 47        /* pw_mailbox_open on a BOUND batch (pw_batch_bind) whose every environment sits in a segment, pipelined mode: 0 (default)
                                      the resident kernel runs the segments -- the puzzle's tables copied into LDS once, at the open --, 2 never (one lane
                                      per environment over the tables in memory).  Same results. */
#define PW_OPT_MAILBOX_FORM 48       /* read-only: 0 no mailbox open, 1 its kernel steps one lane per environment (whole-grid boards / tables in memory), 2 the segments
                                      of the bound batch */
#define PW_OPT_STEP_ONE_APPLIES 49    /* read-only: 1 when pw_step_render_delta on a batch of one with a completion word takes the one-launch form on this engine */
#define PW_OPT_OBS_TUNE_MS 40        /* pw_obs_alloc_tuned: wall-clock budget of the candidate screen in milliseconds (0 = default 10 000): no
                                      further candidate is allocated once it is spent (the best so far is kept and tuned) -- bounds the
                                      constructor when several ranks of a node screen at the same time */
#define PW_OPT_OBS_SCREEN_MS 41      /* read-only: milliseconds the last pw_obs_alloc_tuned spent allocating, screening and releasing candidates */
int pw_engine_set_option(PwEngine* e, int32_t option, int64_t value);
int64_t pw_engine_get_option(const PwEngine* e, int32_t option);
/* Durations (milliseconds) of the render launches recorded since the last call, in launch order
 * (PW_OPT_PROFILE_RENDER).  Waits for them to finish, writes at most `cap` values, returns how many
 * were recorded and starts over. */
int pw_engine_profile_read(PwEngine* e, float* ms, int32_t cap);

/* Auto-tuning of the page-ordered render kernel's launch configuration (PW_OPT_PAGE_*): which page order and
 * occupancy is fastest depends on the physical backing of the observation buffer (measured: the same kernel
 * takes 0.545 .. 0.66 ms on buffers of identical size and alignment).  Renders (puzzle_id, pos) into `obs` with
 * every candidate configuration (a few launches each, HIP events on `stream`), keeps the fastest for this
 * engine and returns its index (0 = the default).  `obs` holds the correct observations on return; the
 * stream is synchronised.  Engines that do not use the page-ordered kernel just render.  Call once per
 * observation buffer, e.g. right after allocating it. */
int pw_engine_tune_render(PwEngine* e, const int32_t* puzzle_id, const int8_t* pos, void* obs,
                          int64_t env_stride_bytes, int32_t batch, void* stream);

/* Observation buffers owned by the library.  What the HBM-write-bound render kernel reaches on a buffer follows the
 * buffer's physical backing (two classes, 7-10 % apart, per allocation), so the library can
 * allocate the buffer itself: one reserved address range backed by physical chunks created and mapped with the HIP
 * virtual-memory API (hipMemCreate / hipMemAddressReserve / hipMemMap), batch * pw_engine_obs_stride(e) bytes,
 * zero-filled, env e at obs + e * pw_engine_obs_stride(e).  The page records of `batch` environments are allocated
 * with it, so that no later render call allocates.
 *   pw_obs_alloc        one buffer.
 *   pw_obs_alloc_tuned  allocates up to max_candidates buffers (all alive at once, so that each lands on other physical
 *                       memory), takes a quick look at each (the best of five launch configurations, three launches
 *                       each), keeps the first that is of the fast class (PW_OPT_OBS_ACCEPT_GBS; from the 16th
 *                       candidate on also one 8 % faster than the slowest seen; no further candidate once PW_OPT_OBS_TUNE_MS of wall
 *                       clock are spent) or else the fastest, RELEASES THE OTHERS TO THE DEVICE and runs
 *                       pw_engine_tune_render on the kept one.  It holds the observations of (puzzle_id, pos), the
 *                       engine its tuned launch configuration; returns the tuner's index (>= 0).  candidate_ms (host
 *                       float [max_candidates], may be NULL) receives every candidate's screened time, *tried how many
 *                       were made.
 *   (not shareable)     such a buffer is a hipMemMap range: hipIpcGetMemHandle does not apply to it.  A caller that shares
 *                       observations with other processes owns the buffer (hipMalloc) and calls pw_engine_tune_render on it.
 *   pw_obs_free         unmaps the buffer, releases its memory to the device and frees its address range (synchronises
 *                       the device first).  pw_engine_destroy frees what is left.
 * Address ranges are never handed out twice within a process (a range freed and reserved again showed stale contents
 * through views of the old one on this runtime): they are taken upwards from 32 TiB behind a watermark (O(1) state; a range
 * the runtime places below it is parked -- at most 8 per request -- and given back at process exit), ~20 000 buffers of the C3
 * size (one per candidate) in a 47-bit address space; after that the calls fail with PW_ENOMEM and the caller uses a buffer of
 * its own (VecPushWorld does, with a warning). */
int pw_obs_alloc(PwEngine* e, int32_t batch, void** obs);
int pw_obs_alloc_tuned(PwEngine* e, const int32_t* puzzle_id, const int8_t* pos, int32_t batch, int32_t max_candidates,
                       void** obs, float* candidate_ms, int32_t* tried, void* stream);
int pw_obs_free(PwEngine* e, void* obs);

/* Number of out-of-range actions (not in 0..3) the step kernels of this engine have seen since the
 * last call; reads and clears the counter, synchronises `stream`.  An asynchronous caller that never
 * looks at the 0xFF flags can poll this instead (an env flagged 0xFF is reset by the next
 * PW_STEP_AUTORESET step, where the reference raises ValueError, gym_env.py:195-196). */
int64_t pw_engine_bad_actions(PwEngine* e, void* stream);

/* Throughput counters kept ON THE DEVICE by the step kernels (the one collective of a multi-GPU job
 * sums this vector).  Every step kernel -- pw_step, pw_rollout, pw_step_render, pw_step_render_delta, whichever
 * formulation runs -- adds, per wavefront, popcount(ballot(...)) of its environments to engine-owned int64 counters
 * (64 slots a cache line apart, one atomic per wavefront and non-zero counter):
 *   out[0]  env-steps: (environment, step) pairs the kernels processed, autoreset steps and refused actions included
 *   out[1]  episodes ended: steps that returned terminated or truncated (gym_env.py:210-223)
 *   out[2]  episodes solved: steps that returned terminated (is_goal_state, puzzle.py:409-411)
 *   out[3]  actions outside 0..3 seen since the engine was created (pw_engine_bad_actions clears its own counter, not this)
 * pw_counters sums the slots (synchronises `stream`); pw_counters_reset zeroes them (asynchronous on `stream`). */
#define PW_NUM_COUNTERS 4
int pw_counters(PwEngine* e, int64_t out[PW_NUM_COUNTERS], void* stream);
int pw_counters_reset(PwEngine* e, void* stream);

/* Latency path of the single-state API (PushWorldPuzzle.get_next_state, puzzle.py:348-394; is_valid_plan :413-424;
 * render_plan :471-506): host memory in, host memory out, ONE launch and no copy command.  The state travels in the
 * kernel arguments, one wavefront computes the step(s), the result is written straight into pinned host memory mapped
 * into the device, and the call returns as soon as the kernel's completion word arrives (the calling thread polls it;
 * no stream synchronisation).  Runs on an engine-owned stream; as everything on an engine, not re-entrant.
 *   xy_in / xy_out  int8 [N][2] of puzzle `puzzle` (N = its num_movables)
 *   info            optional int32 [4]: moved-object bit mask (bit 0 = agent; 0 = nothing moved), goals achieved before,
 *                   goals achieved after, is_goal_state(after)
 * pw_plan_states replays `num_actions` actions from `xy_start` (NULL = the initial state) in one launch:
 *   states_out      int8 [num_actions + 1][N][2], state 0 = the start
 *   goal_out        optional uint8 [num_actions + 1]: is_goal_state of every state
 *   dev_states      optional DEVICE buffer int8 [num_actions + 1][NP][2] (NP = pw_engine_npad) that receives the same
 *                   states in the pos layout of pw_render, for one batched render of the whole plan
 * At most PW_PLAN_MAX_ACTIONS actions per call. */
#define PW_PLAN_MAX_ACTIONS 65536
int pw_next_state(PwEngine* e, int32_t puzzle, const int8_t* xy_in, int32_t action, int8_t* xy_out, int32_t* info);
int pw_plan_states(PwEngine* e, int32_t puzzle, const int8_t* xy_start, const uint8_t* actions, int32_t num_actions,
                   int8_t* states_out, uint8_t* goal_out, int8_t* dev_states);

/* Debug check of the preconditions of the step / render entry points: puzzle_id[i] inside the set; with `pos` != NULL
 * also every movable inside its puzzle's grid (0 <= x <= W - w, 0 <= y <= H - h) and zero padding beyond the puzzle's
 * movables.  The kernels are memory-safe without it -- a puzzle id outside the set is clamped into it, positions are
 * range-checked wherever they index a row or a table -- but then compute the step of a puzzle / state nobody meant.
 * Synchronises `stream`.  Returns the number of offending environments (0 = fine; the lowest
 * offending index goes to *first_bad when given) or a negative error. */
int64_t pw_validate_state(PwEngine* e, const int32_t* puzzle_id, const int8_t* pos, int32_t batch,
                          int32_t* first_bad, void* stream);

/* gym_env.py:150-186 reset(): pos <- initial state of puzzle_id[e], steps <- 0,
 * terminated/truncated <- 0 (when given), for envs with mask[e] != 0 (mask NULL = all). */
int pw_reset(PwEngine* e, const int32_t* puzzle_id, const uint8_t* mask, int8_t* pos,
             int32_t* steps, uint8_t* terminated, uint8_t* truncated, int32_t batch,
             void* stream);

/* Episode management on the device (the batched counterpart of
 * `self._current_puzzle = random.choice(self._puzzles)`, gym_env.py:172 / dm_env.py:172).
 * Every environment whose terminated[e] | truncated[e] flag is set (either pointer may be NULL;
 * both NULL = every environment) draws the puzzle of its next episode:
 *     episode[e] += 1
 *     r   = pw_mix64(seed, e, episode[e])          -- splitmix64 finaliser, see below
 *     idx = floor(r * n / 2^64)                     -- n = table_len, or the set size without a table
 *     puzzle_id[e] = table ? table[idx] : idx
 * `table` (device int32 [table_len], NULL = uniform over the set) holds puzzle indices, repeated to
 * weight them (e.g. the 50 % Level-0 / 50 % Level-1..4 mix of config C4).  The draw depends only on
 * (seed, e, episode[e]), not on launch geometry.  Call it before a pw_step(PW_STEP_AUTORESET): that
 * step then resets the finished environments to the initial state of their new puzzle.
 *     pw_mix64: z = seed + 0x9E3779B97F4A7C15 * (e + 1) + 0xD1B54A32D192ED03 * episode  (mod 2^64)
 *               z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9;  z = (z ^ z >> 27) * 0x94D049BB133111EB;
 *               r = z ^ z >> 31 */
uint64_t pw_mix64(uint64_t seed, uint64_t env, uint64_t episode); /* host copy of the hash above */
int pw_resample(PwEngine* e, int32_t* puzzle_id, const uint8_t* terminated, const uint8_t* truncated,
                const int32_t* table, int32_t table_len, uint64_t seed, uint32_t* episode,
                int32_t batch, void* stream);

/* gym_env.py:188-226 step() without the observation:
 *   pos <- get_next_state(pos, action)                 puzzle.py:348-394
 *   terminated <- is_goal_state                        puzzle.py:409-411
 *   reward <- 10.0 | d(count_achieved_goals) - 0.01    gym_env.py:212-221 (float64)
 *   steps += 1; truncated <- steps >= max_steps        gym_env.py:201,223
 * reward / dgoals may be NULL.  Actions outside 0..3 leave the env untouched, set
 * terminated = truncated = 0xFF for that env and bump the engine's sticky bad-action counter
 * (pw_engine_bad_actions; the wrappers raise ValueError like gym_env.py:195-196). */
int pw_step(PwEngine* e, const int32_t* puzzle_id, const uint8_t* actions, int8_t* pos,
            int32_t* steps, double* reward, int8_t* dgoals, uint8_t* terminated,
            uint8_t* truncated, int32_t batch, uint32_t flags, void* stream);

/* num_steps consecutive pw_step calls of every environment in ONE launch (state-only rollouts:
 * planner look-ahead, random-policy data, configs C2/C4).  actions is uint8 [num_steps][B],
 * step-major.  Semantics are exactly those of calling pw_step num_steps times with actions[t];
 * pos/steps/reward/dgoals/terminated/truncated receive the values after the last step.  The
 * optional histories reward_hist float64 [num_steps][B], terminated_hist / truncated_hist
 * uint8 [num_steps][B] (NULL to skip) receive every step's outputs. */
int pw_rollout(PwEngine* e, const int32_t* puzzle_id, const uint8_t* actions, int32_t num_steps,
               int8_t* pos, int32_t* steps, double* reward, int8_t* dgoals, uint8_t* terminated,
               uint8_t* truncated, double* reward_hist, uint8_t* terminated_hist,
               uint8_t* truncated_hist, int32_t batch, uint32_t flags, void* stream);

/* BATCH BINDING: stepping at LDS-table speed for batches whose puzzle ids are known in advance (configs C2 .. C4 state-only; the
 * step half of pw_step_render).  Every pw_step / pw_rollout call takes puzzle_id afresh and walks id -> header -> table rows through
 * the caches for every push test (get_next_state, puzzle.py:348-394).  pw_batch_bind reads the ids ONCE and cuts the batch into
 * segments -- up to 256 environments of one puzzle, consecutive or not --; from then on every stepping call that passes THE SAME
 * puzzle_id pointer and batch size launches one workgroup per segment, which copies its puzzle's push tables (the reference's collision
 * tables, puzzle.py:259-311, all four actions in one nibble per offset; median 3 KB) into LDS and steps one lane per environment.
 * Environments of puzzles played by fewer than PW_OPT_BIND_MIN_ENVS environments of the batch (a C4 shard's Level-0 half: 2 per
 * puzzle), or whose tables exceed 16 KB of LDS, keep the lane groups -- in the same launch.  Same results as unbound calls, bit for bit.
 *   - ONE binding per engine; binding again replaces it; every other call (other pointers / batch sizes, pw_step_render_delta,
 *     pw_mailbox_*) works unbound as before;
 *   - the binding is a promise about the CONTENTS of puzzle_id: change them only through pw_resample or before a pw_reset on that
 *     buffer (both rebuild the list behind themselves, on their stream, without a host round trip), or bind again.  An environment
 *     whose id changed otherwise is played on the puzzle it was bound to (memory-safe; counted: PW_OPT_BIND_MISMATCHES);
 *   - pw_batch_bind synchronises `stream` (it reads the number of segments back); info (optional, int64 [6]) receives the number of
 *     segments, bound environments, bound puzzles, how many of those needed an index list (their environments not consecutive), the
 *     environments NO segment holds that launches of several steps step one lane each all the same -- 64 puzzles per wavefront, every
 *     lane with its own copy of its puzzle's block (at most 8 movables, 1 KB) in LDS -- and the bytes of such a copy's slot.
 * Measured (one C4 shard, 65 536 environments): DESIGN.md section 4 K1g. */
int pw_batch_bind(PwEngine* e, const int32_t* puzzle_id, int32_t batch, int64_t* info, void* stream);
int pw_batch_unbind(PwEngine* e);

/* RESIDENT stepping of a small state-only batch ("mailbox"): gym_env.py:188-226 for a host that needs every step's verdicts
 * before it chooses the next actions, without a kernel launch and a stream synchronisation per step.  pw_mailbox_open starts a
 * kernel that keeps the batch in registers and waits; pw_mailbox_post hands it the actions of ONE step (a word in pinned memory:
 * no launch); the kernel performs exactly what pw_step(flags) performs -- the arrays given to pw_mailbox_open receive what
 * pw_step would have written, after every step -- and leaves the step's reward / terminated / truncated in pinned host memory,
 * which pw_mailbox_wait returns once the step is complete.  Up to `ring` steps may be posted ahead of the last complete one
 * (pw_mailbox_post waits beyond that); the pointers a wait returns stay valid until `ring` more steps have been posted.
 *   - any set, batches of up to 65 536 environments (every workgroup is resident): sets whose puzzles all fit 8 x 8 cells with at
 *     most 8 movables (PW_OPT_STEP_BOARD_SET = 1: the Level-0 families of 5 x 5 puzzles) are stepped by the whole-grid boards pw_step
 *     launches for them, the others one lane per environment over their overlap tables / row bitboards (pw_step_lane_kernel's step);
 *     a batch bound with pw_batch_bind whose every environment sits in a segment (and whose segments are all resident at once) is
 *     stepped by the segments -- the puzzle's tables in LDS, copied once at the open (PW_OPT_MAILBOX_SEG / PW_OPT_MAILBOX_FORM);
 *   - puzzle_id is read once, at the open: episodes restart (PW_STEP_AUTORESET) on the same puzzle;
 *   - everything queued on other streams for the arrays must be complete before the open, and while the mailbox is open the
 *     engine's other stepping calls fail (the environments live in the resident kernel); pw_counters is current after the close
 *     (the kernel adds its sums when it has to wait for a word, every 256 steps and at its end);
 *   - `actions`: device memory, or host memory (actions_on_host != 0: copied into a pinned staging slot, read by the kernel
 *     across the link);
 *   - the kernel ends by pw_mailbox_close, or BY ITSELF after idle_ms (0 = 1 000) without a post: a resident kernel would
 *     otherwise hold every device-wide synchronisation (hipDeviceSynchronize, torch.cuda.synchronize) for ever.  pw_mailbox_post
 *     fails with PW_EDEVICE once more than idle_ms / 2 have passed since the previous post (the mailbox has expired: close it
 *     and open a new one; the arrays hold the state after the last complete step -- the Python Mailbox does exactly that by itself
 *     when no step is in flight).  Consequences for the caller: (i) a host that pauses between steps (a learner update, logging)
 *     for longer than idle_ms / 2 loses the kernel -- choose idle_ms for the longest pause of the loop; (ii) ANY device-wide
 *     synchronisation while a mailbox is open (hipDeviceSynchronize, a hipFree / hipMalloc of another allocator on the device,
 *     pw_obs_free, destroying another engine) stalls until the kernel idles out, i.e. up to idle_ms, and ends the mailbox.
 *   - one host thread at a time per mailbox (post / wait / step / run / close are not synchronised against each other).
 * Measured (C2, 4 096 environments, tools/bench_mailbox.py, profiles/r06_mailbox.json): 6.1 - 6.6 us per synchronous step against
 * 17.8 us for pw_step + a stream synchronisation; 1.95 us with 8 steps in flight (2.1e9 env-steps/s; 3.2 - 3.45 us before round 6).
 * DESIGN.md K1f. */
typedef struct PwMailbox PwMailbox;
int pw_mailbox_open(PwEngine* e, const int32_t* puzzle_id, int8_t* pos, int32_t* steps, double* reward, int8_t* dgoals,
                    uint8_t* terminated, uint8_t* truncated, int32_t batch, uint32_t flags, int32_t ring /* 0 = 8 */,
                    int32_t idle_ms /* 0 = 1000 */, PwMailbox** out);
int pw_mailbox_post(PwMailbox* m, const uint8_t* actions, int32_t actions_on_host, uint64_t* seq /* out: 1, 2, ... */);
int pw_mailbox_wait(PwMailbox* m, uint64_t seq, const double** reward, const uint8_t** terminated, const uint8_t** truncated);
/* pw_mailbox_post + pw_mailbox_wait of that step in one call; pw_mailbox_layout: where the verdicts of step `seq` are --
 * base + ((seq - 1) mod ring) * stride: reward float64 [B] at 0, terminated / truncated uint8 [B] at the two offsets. */
int pw_mailbox_step(PwMailbox* m, const uint8_t* actions, int32_t actions_on_host, uint64_t* seq);
int pw_mailbox_layout(PwMailbox* m, const uint8_t** base, int64_t* stride, int64_t* off_terminated, int64_t* off_truncated,
                      int32_t* ring);
/* num_steps posts from one uint8 [num_steps][B] array with at most `ahead` steps in flight (<= 1: every step waits for the one
 * before, the cadence of a host that chooses the next actions from a step's verdicts); returns when all are complete. */
int pw_mailbox_run(PwMailbox* m, const uint8_t* actions, int32_t num_steps, int32_t actions_on_host, int32_t ahead,
                   uint64_t* last_seq);
/* pw_mailbox_close_profile: as the close, and (profile: int64 [8], or NULL) [0] steps completed, [1] how the kernel ended (2 stop
 * word, 3 idle limit), [2..6] ticks of the device's constant clock that wavefront 0 spent waiting for the word / reading its
 * actions / stepping / storing / counting itself in (zeros unless PW_OPT_MAILBOX_MODE has bit 4 set), [7] that clock's kHz. */
int pw_mailbox_close_profile(PwMailbox* m, int64_t* profile);
int pw_mailbox_close(PwMailbox* m);

/* puzzle.py:426-469 render() + env_utils.py:44-91 padding (+ /255 for PW_OBS_F32).
 * obs: device buffer, env e at obs + e * env_stride_bytes (multiple of 16, base 16 B aligned). */
int pw_render(PwEngine* e, const int32_t* puzzle_id, const int8_t* pos, void* obs,
              int64_t env_stride_bytes, int32_t batch, void* stream);

/* pw_step followed by pw_render of the new state on the same stream (PW_OPT_FUSED_STEP_RENDER selects a
 * single launch that runs the step inside the per-environment render workgroups; slower, kept for tests). */
int pw_step_render(PwEngine* e, const int32_t* puzzle_id, const uint8_t* actions, int8_t* pos,
                   int32_t* steps, double* reward, int8_t* dgoals, uint8_t* terminated,
                   uint8_t* truncated, void* obs, int64_t env_stride_bytes, int32_t batch,
                   uint32_t flags, void* stream);

/* Incremental form of pw_step_render for observation buffers that persist between steps.
 * PRECONDITION: on entry `obs` holds the observation of the state in `pos` (as left by pw_render,
 * pw_step_render or a previous pw_step_render_delta on the same buffers).  On return every output,
 * including every byte of `obs`, equals what pw_step_render produces -- but only the pixel rows swept
 * by the objects that moved are written (none for a blocked move; the whole image for environments
 * reset by PW_STEP_AUTORESET, which also covers a puzzle_id changed by pw_resample).  uint8 /
 * pixels_per_cell 3 engines; any other engine silently takes the pw_step_render path.
 * Returns PW_OK, or 1 / 2 when the launch will also write the engine's completion word (below) -- 2: the step and the redraw were ONE
 * launch (PW_OPT_STEP_ONE_FUSED) whose hand-over words carry a launch number -- a call made while its stream is being captured takes the
 * two launches (1): a graph may replay those. */
int pw_step_render_delta(PwEngine* e, const int32_t* puzzle_id, const uint8_t* actions, int8_t* pos,
                         int32_t* steps, double* reward, int8_t* dgoals, uint8_t* terminated,
                         uint8_t* truncated, void* obs, int64_t env_stride_bytes, int32_t batch,
                         uint32_t flags, void* stream);

/* A completion word for the single-environment adapters (gym / dm_env: batch 1, observation and state in pinned host memory):
 * `word` = 8 bytes of pinned, device-addressable host memory (NULL switches it off; the count restarts at 0).  A
 * pw_step_render_delta on a batch of ONE that redraws with the generic kernel (every engine but uint8 / pixels_per_cell 3) then
 * returns 1 instead of PW_OK and the last workgroup of its last kernel writes k -- the number of such calls since this call -- into the word after
 * everything else it wrote: the host polls the word instead of synchronising the stream (~8 us of runtime per step here). */
int pw_engine_set_step_signal(PwEngine* e, void* word);

/* For the same callers: the state arrays of their batch of one may live in DEVICE memory (a kernel that reads and writes pinned host
 * memory across the link lasts ~5 us whatever it computes) when the ONE-launch form of pw_step_render_delta applies
 * (PW_OPT_STEP_ONE_APPLIES = 1: returns 2) -- that launch then also writes a copy of what the step left into `block`, 16 + 2 * NP bytes of
 * pinned, device-addressable host memory (8-byte aligned; NULL switches it off), before the completion word:
 *   [0] reward float64 | [8] steps int32 | [12] terminated | [13] truncated | [14] dgoals int8 | [16 ...] (x, y) int8 pairs. */
int pw_engine_set_step_host_copy(PwEngine* e, void* block);

/* Planner successor expansion, best_first_search.h:76-78 calling
 * PushWorldPuzzle::getNextState (pushworld_puzzle.cc:386-460) and satisfiesGoal (:462-469)
 * for all 4 actions of F states of ONE puzzle (index `puzzle`, normally PW_ORDER_CPP).
 *   states int32 [F][N]      Position2D = x * 10000 + y   (pushworld_puzzle.h:32-37)
 *   succ   int32 [F][4][N]
 *   moved  uint32 [F][4]     bit k set <=> object k is in moved_object_indices (cc:446-457)
 *   goal   uint8 [F][4]      satisfiesGoal(successor)
 * Frontiers of >= 131 072 states run one LANE per state (engines of at most 64
 * puzzles: they carry the reference's collision tables with the four actions interleaved, 4 bits per relative offset, one
 * lookup per push test); smaller ones and other puzzles one lane group per state.  Same results either way; the buffers need
 * no particular alignment (16-byte aligned ones leave in wider stores).  PW_OPT_STEP_LANE_BATCH moves the threshold. */
int pw_expand4(PwEngine* e, int32_t puzzle, const int32_t* states, int32_t* succ,
               uint32_t* moved, uint8_t* goal, int32_t num_states, void* stream);

/* ---------------------------------------------------------- planner frontier services
 * Breadth-first exploration of ONE puzzle with the closed set on the device.  The reference
 * planner pops one node, calls getNextState x4 and looks each successor up in a
 * std::unordered_set<State> (best_first_search.h:72-93); here every call expands a whole layer:
 * pw_expand4 semantics for the successors, an open-addressing visited table in HBM, and the new
 * states appended to a store with (parent, action) links.  State numbering is deterministic and
 * equals the one of a sequential FIFO search trying the actions in the order 0..3 (state 0 = start).
 * All device memory is allocated by pw_search_create (about max_states * (2N + 40) bytes + scratch: the
 * visited table is 16 .. 32 bytes per state -- 8-byte slots of (fingerprint, entry) at a load factor below one half). */
/* Novelty tables: NoveltyHeuristic::estimate_cost_to_goal (cpp/src/heuristics/novelty.cc:30-77) for a
 * whole array of states.  novelty[k] is what the reference returns for state k when the states are fed
 * to it one after the other in index order, this call after all earlier calls: 1 = a moved object is at
 * a position it never had, 2 = a (moved object, other object) pair of positions is new, 3 = neither.
 *   states  device int32 [count][state_size] Position2D, inside the width x height grid
 *   moved   device uint32 [count], bit i = object i is in moved_object_indices (pw_expand4's `moved`)
 * Memory: state_size * (state_size - 1) / 2 * (width * height)^2 * 4 bytes for the pair table. */
typedef struct PwNovelty PwNovelty;
int pw_novelty_create(int device, int state_size, int width, int height, PwNovelty** out);
void pw_novelty_destroy(PwNovelty* n);
int pw_novelty_reset(PwNovelty* n, void* stream);
int pw_novelty_eval(PwNovelty* n, const int32_t* states, const uint32_t* moved, uint8_t* novelty,
                    int32_t count, void* stream);

typedef struct PwSearch PwSearch;
/* novelty_width 0: breadth-first search.  1 or 2: width-limited search IW(k) (Lipovetzky & Geffner) on
 * the reference's novelty definition: every new state gets its novelty (tables updated in discovery
 * order, the start state with all objects moved, best_first_search.h:58-67); states whose novelty exceeds
 * the width stay in the closed set but are not expanded. */
int pw_search_create(PwEngine* e, int32_t puzzle, int64_t max_states, int32_t novelty_width, PwSearch** out);
void pw_search_destroy(PwSearch* s);
/* start: host int32 [N] Position2D (x * 10000 + y), NULL = the puzzle's initial state */
int pw_search_begin(PwSearch* s, const int32_t* start, void* stream);
/* Expands the newest layer and synchronises the stream.
 *   info[0] depth of the new layer   info[1] states in it (0 = search space exhausted)
 *   info[2] states in the store      info[3] lowest index of a goal state found so far, or -1
 * PW_ELIMIT when the store is full (the new layer is then incomplete). */
int pw_search_expand(PwSearch* s, int64_t info[4], void* stream);
/* dst: device int32 [count][N] Position2D of states first .. first + count - 1 */
int pw_search_read_states(PwSearch* s, int64_t first, int64_t count, int32_t* dst, void* stream);
/* device int32 [count] parent indices (-1 for the start) and uint8 [count] actions; either may be NULL */
int pw_search_read_links(PwSearch* s, int64_t first, int64_t count, int32_t* parent, uint8_t* action,
                         void* stream);
/* device uint8 [count]: 1 = state was cut by the novelty width (never expanded), 0 otherwise */
int pw_search_read_flags(PwSearch* s, int64_t first, int64_t count, uint8_t* pruned, void* stream);
/* actions (host buffer) from the start to state `index`; returns the plan length (if > cap nothing
 * is written: call again with a larger buffer) or a negative error. */
int pw_search_plan(PwSearch* s, int64_t index, uint8_t* actions, int32_t cap, void* stream);

/* Batched search of SMALL puzzles: ONE launch decides solvability for `n` puzzles of the engine's set (the filter of
 * generate.py:262-297 over a generated set; best_first_search.h:45-98 with a FIFO frontier).  Persistent workgroups take
 * puzzles off a device counter and run the whole breadth-first loop inside the kernel -- closed set in LDS (4 096 states)
 * that moves to the workgroup's HBM slab when a layer could outgrow it, no launch per layer, no host readback.
 *   puzzles     device int32 [n] set indices, NULL = 0 .. n - 1
 *   verdict     device uint8 [n]: 1 solved, 0 unsolvable (reachable space exhausted), 2 unknown (more than
 *               max_states_each states), 3 not searched -- the kernel handles grids up to 16 x 16 (with their border) and up
 *               to 8 movables (every Level-0 recipe) of puzzles with overlap tables; search the others with pw_search_create
 *   plan_len    device int32 [n]: length of a shortest plan (depth of the first goal state), -1 unless solved
 *   num_states  optional device int32 [n]: states in the closed set when the search ended
 *   plans       optional device uint8 [n][plan_cap]: the actions of A shortest plan of every solved puzzle whose plan fits
 *               plan_cap (the layers are not numbered in FIFO order here: pw_search_plan's plan is the first in action order,
 *               this one is any of the same length); costs 4 more bytes per state of slab
 * novelty_width must be 0 (breadth-first; the width-limited IW(k) is pw_search_create's).  The slabs (8 bytes per state of
 * store + the tables, per persistent workgroup) are engine-owned, allocated on first use / growth (synchronises `stream`
 * then) and kept.  Asynchronous on `stream` otherwise. */
int pw_search_batch(PwEngine* e, const int32_t* puzzles, int32_t n, int64_t max_states_each, int32_t novelty_width,
                    uint8_t* verdict, int32_t* plan_len, int32_t* num_states, uint8_t* plans, int32_t plan_cap, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PUSHWORLD_AMD_H_ */
