// gfx950 (CDNA4 / MI355X) kernels of the batched PushWorld step engine and the engine half
// of the C ABI.  Written for 64-lane wavefronts; no other target is supported.
//
// Dynamics (reference: python3/src/pushworld/puzzle.py:348-394 get_next_state,
// cpp/src/pushworld_puzzle.cc:386-460 getNextState):
//   one WAVEFRONT per environment, lane r = grid row r (H <= 64).  Every object is a row
//   bitboard spread over the wave (one uint64 per lane); "object i pushes object j" is
//       ballot((shift(row_i, action) & row_j) != 0) != 0  &&  ballot((row_i & row_j) != 0) == 0
//   which is exactly membership of (pos_i - pos_j) in the reference's dynamic collision set
//   (puzzle.py:567-593), and likewise for walls (puzzle.py:522-564).  The push set is a
//   uint32 bit mask grown to a fixed point with wave ballots.
//
// Observation (reference: puzzle.py:426-469 render, :596-638 _draw_object,
// utils/env_utils.py:44-91 render_observation_padded):
//   one WORKGROUP per environment.  The painter's algorithm is evaluated per cell into an
//   LDS occupancy/code grid (static layers from the packed puzzle, movables composed with
//   LDS atomicMax in painter order), expanded to 3x3 zone colours per cell, and streamed to
//   HBM as coalesced 16-byte stores.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "pw_host.h"
#include "pw_device_guard.h"
#include "pw_zone.h"

#define PW_WAVE 64
#ifndef PW_RENDER_THREADS
#define PW_RENDER_THREADS 256  // workgroup size of the render kernels (one environment each)
#endif

// Overlap tables of one puzzle (engine-built, csrc/pw_engine.inc build_overlap_tables): what is left of the reference's
// collision tables (puzzle.py:259-311) once the four actions share one table.
//   pair table (i, j), R rows of uint64:   bit (rx + w_i - 1) of row (ry + h_i - 1) = movable i placed at (rx, ry)
//                                          relative to movable j overlaps it          ("i pushes j" = overlap after the
//                                          move and not before, puzzle.py:567-593)
//   wall table j, Hs = H + 2 rows:         bit (x + 1) of row (y + 1) = movable j at (x, y) overlaps a wall cell (the
//                                          agent: wall or agent wall)                 (puzzle.py:522-564)
// One or two 8-byte loads replace the row loops of lane_pushes / lane_blocked for objects that do not fit the 8 x 8
// boards.  16 bytes per puzzle, indexed by puzzle id (same dependency level as the puzzle header).
struct PwOvlDir {
  uint32_t pair_off;  // offset of pair table (0, 0) in 8-byte units; 0 = this puzzle has no tables
  uint32_t wall_off;  // offset of wall table 0
  uint16_t R;         // rows per pair table: max h_i + max h_j - 1
  uint16_t Hs;        // rows per wall table: H + 2
  uint32_t reserved;
};

// Push tables of one puzzle (engine-built from the overlap tables, sets of at most 64 puzzles: the planner's engines): the
// reference's collision tables themselves (puzzle.py:259-311), the four actions interleaved as one NIBBLE per relative
// offset -- bit a = "overlaps after action a and not now" -- so that one 4-byte load answers a push (or wall) test for all
// four actions (pw_expand4_lane_kernel).  32-bit words, 8 nibbles each, addressed from PwEngine::d_ovl.
//   pair table (i, j), R2 = 2 max_h + 1 rows of RW words:  nibble (rx + w_i) of row (ry + h_i), i at (rx, ry) relative to j
//   wall table j, Hs = H + 2 rows of WW words:             nibble (x + 1) of row (y + 1)
//   byte pair table (i, j), R2 + 1 rows of CWB = 2 max_w + 2 bytes: the same nibbles one per BYTE, with an all-zero guard row
//   (index R2) and guard column (index 2 max_w + 1) -- a lookup clamps its row / column with one v_min each instead of
//   testing them (pw_expand4_v2_kernel keeps this table in LDS)
struct PwPushDir {
  uint32_t pair_off;  // offset of pair table (0, 0) in 8-byte units from d_ovl; 0 = this puzzle has no push tables
  uint32_t wall_off;
  uint16_t R2, RW;
  uint16_t Hs, WW;
  uint32_t pairb_off;   // byte pair tables, 8-byte units from d_ovl; 0 = none (too big for LDS)
  uint16_t CWB;         // bytes per row of a byte pair table
  uint16_t reserved;
  uint32_t pairb_bytes; // N * N * (R2 + 1) * CWB rounded up to 16
  uint32_t reserved2;
  // the same byte tables sized PER PAIR (round 5): table (i, j) has h_i + h_j + 2 rows of w_i + w_j + 2 bytes (the last row and
  // column are the zero guards) instead of the set-wide 2 max_h + 2 by 2 max_w + 2 -- one big movable no longer inflates the
  // tables of every pair (`Pinhole Lock`, N = 14: 98.8 -> 10.1 KB; 44 benchmark puzzles beyond 16 KB become 1).  The block starts
  // with N * N descriptors: byte offset of the table inside the block | columns << 16 | guard row index << 24.
  uint32_t pairp_off;   // 8-byte units from d_ovl; 0 = none (uniform tables are small enough, or the block exceeds 64 KB)
  uint32_t pairp_bytes; // descriptors + tables, rounded up to 16
};

// An observation buffer owned by the library (pw_obs_alloc): one reserved address range backed by physical chunks
// created and mapped with the HIP virtual-memory API.  Unmapped, released and its address range freed by pw_obs_free /
// pw_engine_destroy: the memory goes back to the DEVICE, not to a caching allocator.
struct PwObsBuf {
  void* ptr;
  size_t bytes;   // mapped bytes (a multiple of the chunk size)
  size_t reserved;  // bytes of the reserved address range (0: none)
  size_t chunk;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};

struct PwMailbox;
struct PwSegDir;
struct PwSegDesc;
// A bound batch (pw_batch_bind): the segment list of ONE (puzzle_id buffer, batch size) pair per engine, built on the device.
struct PwBind {
  const int32_t* puzzle_id;  // the key: calls that pass this pointer and this batch size take the bound launches
  int32_t batch;
  int32_t* d_ints;           // one allocation: cnt / mn / mx / seg_base / perm_base [P each], rank / perm [B each], mlist [B][2], meta [16]
  int32_t *cnt, *mn, *mx, *seg_base, *perm_base, *rank, *perm, *mlist, *meta;
  uint8_t* d_bound;          // [B]
  PwSegDesc* d_segs;         // [seg_cap]
  int32_t seg_cap;           // upper bound of the number of segments of any assignment: B / 256 + min(P, B / min_envs) + 1
  int32_t seg_grid;          // workgroups the segment role is launched with: the exact number after pw_batch_bind (which reads
                             // it back), seg_cap after an asynchronous re-bind (pw_resample / pw_reset on the bound buffer)
  int32_t min_envs;
  int32_t mepw_log2;     // ... of pw_step_mseg_kernel
  int32_t epw_log2;      // PW_OPT_BIND_SPREAD when the list was built: at most 64 >> this environments per wavefront
  int64_t info[4];           // segments, bound environments, bound puzzles, of them with an index list (as of the last pw_batch_bind)
  int32_t multi_envs;        // environments no segment holds that pw_step_mseg_kernel can step (-1: unknown, after an asynchronous re-bind)
  uint32_t multi_slot;       // ... and the LDS bytes the blocks of a wavefront's 64 puzzles need at most
  uint32_t lds_bytes0;       // ... of the launches of one step (the blocks with wall bitmaps)
  uint32_t lds_bytes;        // dynamic LDS of the launches: the largest block among the bound puzzles (after an asynchronous re-bind:
                             // the largest block of the set)
};
struct PwEngine {
  const PwPuzzleSet* set;
  PwEngineConfig cfg;
  int np;            // padded object count of the pos layout
  int pad_h, pad_w;  // observation frame in cells
  int obs_h, obs_w;  // pixels
  int64_t obs_bytes;
  size_t render_lds;
  bool fast_u8_ppc3;       // uint8, pixels_per_cell 3, border_width 1: zones == pixels
  bool page_f32;           // float32, pixels_per_cell 3, border_width 1: page-ordered / delta kernels over a second,
                           // frame-layout zone table (the generic LDS kernel keeps its own layout)
  bool rowpage;            // any other frame with pixel rows of >= 512 bytes: the row-page kernel over per-puzzle
  bool rowpage_aligned;    //   static row tables (3 H distinct pixel rows + a zero row); aligned: rows are whole chunks
  uint8_t* d_srow;
  int64_t srow_stride;
  int srow_pitch;
  int row_bytes;
  uint16_t* d_estat_page;
  uint32_t* d_estat_page_off;
  // test / profiling knobs, pw_engine_set_option (all 0 by default)
  int step_kernel;         // PW_OPT_STEP_KERNEL: 0 lane group (default), 1 wavefront per env, 2 lane per env
  bool force_fused;        // PW_OPT_FUSED_STEP_RENDER: pw_step_render always uses the single fused launch
  unsigned long long step_one_seq;  // launches of pw_step_render_one_kernel so far (its hand-over word)
  uint8_t* step_host_copy; // pw_engine_set_step_host_copy: pinned block that receives a copy of what a one-launch step left (or NULL)
  int step_one_fused;      // PW_OPT_STEP_ONE_FUSED: pw_step_render_delta on a batch of one with a completion word is ONE launch (2: whole rows)
  bool force_lds_render;   // PW_OPT_RENDER_KERNEL = 1: per-environment LDS kernel even where the page kernel applies
  int64_t page_slice_envs; // PW_OPT_PAGE_SLICE_ENVS: environments per page-kernel launch (0 = what 2^31 chunks allow)
  int64_t search_chunk;    // PW_OPT_SEARCH_CHUNK: parents per pw_search_expand pass (0 = 2^20)
  int search_keys;         // PW_OPT_SEARCH_KEYS: closed set of pw_search_*: 0 fingerprinted entries, 1 exact 63-bit keys where the state packs
  int step_mixed;          // PW_OPT_STEP_MIXED_GROUPS: lanes per environment chosen per workgroup on N_pad 8 / 16 sets (0 auto, 2 never)
  int step_wide_groups;    // PW_OPT_STEP_WIDE_GROUPS: 32 lanes per environment for N_pad 32 instead of two movables per lane
  int64_t step_lane_batch; // PW_OPT_STEP_LANE_BATCH: state-only launches from this batch size on run one lane per environment
  int step_reverse;        // PW_OPT_STEP_BLOCK_ORDER: the lane-group step kernels start with the LAST environments
  int step_narrow_groups;  // PW_OPT_STEP_NARROW_GROUPS: N_pad 16 sets in 8-lane groups with two movables per lane
  int step_lds_tables;     // PW_OPT_STEP_LDS_TABLES: the group step kernel stages the puzzle's row tables in LDS:
                           // 0 automatic (launches of >= 4 steps), 1 always, 2 never
  bool lds_tables_fit;     // every puzzle of the set has <= 128 shape rows
  int step_tables;         // PW_OPT_STEP_TABLES: which puzzles get overlap tables: 0 automatic = 1 every puzzle that fits,
                           // 2 none, 3 only those with a movable beyond 8 x 8
  uint64_t* d_ovl;         // overlap tables of all puzzles that have them (word 0 unused)
  PwOvlDir* d_ovl_dir;     // [set size]
  uint64_t* d_boards;      // [set size][2] whole-grid wall / wall + agent-wall boards when EVERY puzzle fits 8 x 8 cells and has at
                           // most 8 movables (pw_step_board_kernel), else NULL
  int step_boards;         // PW_OPT_STEP_BOARDS: 0 automatic (state-only launches of such sets), 2 never
  uint8_t* d_qdesc;        // [set size] PwQuadDesc records (PW_QD_BYTES each): whole-grid 16 x 16 boards, four lanes per environment
                           // (step_quad16_body); a record with N = 0: the puzzle does not fit
  int quad_puzzles;        // puzzles of the set that fit
  int step_quad16;         // PW_OPT_STEP_QUAD16: 0 automatic (workgroups whose 32 environments all fit), 2 never
  PwPushDir* d_push_dir;   // [set size], or NULL (sets of more than 64 puzzles carry no push tables)
  std::vector<uint8_t> push_has;  // [set size] host copy: puzzle p has push tables
  std::vector<PwPushDir> push_host;  // [set size] host copy of d_push_dir
  int expand_lds_tables;   // PW_OPT_EXPAND_LDS_TABLES: 0 automatic (pw_expand4_v2_kernel where the tables fit LDS), 2 never
  int expand_tile_order;   // PW_OPT_EXPAND_TILE_ORDER: pw_expand4_v2_kernel: 0 tiles interleaved, 1 a contiguous eighth per XCD
  int expand_pair_dims;    // PW_OPT_EXPAND_PAIR_DIMS: pw_expand4_v2_kernel with pair tables sized per pair: 0 automatic, 2 never
  int expand_prefetch;     // PW_OPT_EXPAND_PREFETCH: ... loads the next tile's rows while it computes this one
  int expand_wg_waves;  // PW_OPT_EXPAND_WG_WAVES: cap on the wavefronts of a lone workgroup per CU (0 = automatic)
  int search_batch_groups_per_cu;  // PW_OPT_SEARCH_BATCH_GROUPS_PER_CU (0 = automatic)
  int expand_groups_per_cu;  // PW_OPT_EXPAND_GROUPS_PER_CU: persistent workgroups per CU (0 = as many as fit LDS, at most 8)
  int64_t ovl_bytes;
  int ovl_puzzles;         // puzzles with tables
  std::vector<uint8_t> ovl_has;  // [set size] host copy: puzzle p has tables
  // launch configuration of the page-ordered render kernel (CopyArgs::order / run_log2, dynamic LDS as an
  // occupancy cap); defaults are the robust optimum, pw_engine_tune_render measures the caller's buffer
  int page_order, page_run_log2, page_lds_pad_kb;
  int page_load_all;       // PW_OPT_PAGE_LOAD_ALL: the ppc-3 page kernel loads the static chunks of every page (kLoadAll)
  int64_t tuned_ns;        // nanoseconds per launch of the configuration pw_engine_tune_render kept (0: never tuned)
  bool tuning;             // inside pw_engine_tune_render: trial launches use the kTag = 1 symbol of the page kernel
  void* d_rec;             // page records (PageRec [rec_cap]): allocated with the first observation buffer / render
  int64_t rec_cap;         //   call of a batch size, grown (never shrunk) when a larger batch arrives
  std::vector<PwObsBuf> obs_bufs;  // pw_obs_alloc
  int obs_chunk_mb;        // PW_OPT_OBS_CHUNK_MB: physical chunk size of pw_obs_alloc in MiB (0 = the default, 32 MiB)
  int obs_accept_gbs;      // PW_OPT_OBS_ACCEPT_GBS: pw_obs_alloc_tuned keeps the first candidate that reaches this
  int64_t obs_tune_ms;     // PW_OPT_OBS_TUNE_MS: wall-clock budget of pw_obs_alloc_tuned's candidate screen (0 = 10 000)
  int64_t obs_screen_ms;   // PW_OPT_OBS_SCREEN_MS: what the last screen took
  // PW_OPT_PROFILE_RENDER: HIP event pairs around the dominant (render) launch, on the launch stream
  std::vector<hipEvent_t> prof_events;  // 2 per slot
  int prof_used;
  int num_cus;             // compute units of the device (persistent launches)
  uint8_t* d_search_slab;  // pw_search_batch: work counter + one slab per persistent workgroup (grown on demand, kept)
  size_t search_slab_bytes;
  uint32_t* d_scratch;     // 64 bytes of device scratch (pw_validate_state counters) + one 4 KiB page of zeros
  unsigned long long* d_counters;  // PW_COUNTER_SLOTS x 8 uint64: env-steps / episodes ended / solved per slot (pw_counters)
  int64_t bad_total;       // out-of-range actions pw_engine_bad_actions has read and cleared so far (pw_counters adds the rest)
  // latency path of the single-state API (pw_next_state / pw_plan_states): created on first use
  hipStream_t lat_stream;
  void* lat_host;          // pinned host memory mapped into the device: results + completion word
  void* lat_dev;           // the device's address of lat_host
  size_t lat_bytes;
  uint32_t lat_seq;        // completion word of the last launch
  std::mutex lat_mu;       // pw_next_state / pw_plan_states share lat_host and lat_seq: one call at a time per engine
  PwSegDir* d_seg_dir;     // [set size] LDS blocks of the segment kernels (pw_seg_kernels.inc), inside d_ovl; NULL without overlap tables
  int seg_puzzles;         // puzzles with a block
  uint32_t seg_max_bytes;  // the largest block of the set
  hipStream_t side_stream; // launches of several steps on a partly bound batch: the segments run here, next to the lane groups on the
  hipEvent_t ev_fork, ev_join;  // caller's stream (fork / join by events); created on first use
  PwBind* bind;            // pw_batch_bind, or NULL
  int bind_min_envs;       // PW_OPT_BIND_MIN_ENVS (0 = default)
  int bind_max_kb;         // PW_OPT_BIND_MAX_KB: largest block (KiB of LDS) that is bound (0 = 48)
  int bind_rollouts;       // PW_OPT_BIND_ROLLOUTS: launches of several steps take the segments: 0 when every environment is bound, 1 always, 2 never
  int bind_spread;         // PW_OPT_BIND_SPREAD: 0 automatic, k = at most 64 >> (k - 1) environments per wavefront of the segment kernels
  int bind_lanes;          // PW_OPT_BIND_LANES: 0 automatic, k = at most 2^(k - 1) lanes per environment
  int bind_fused;          // PW_OPT_BIND_FUSED: 0 automatic, 2 never (segments and lane groups as two launches)
  uint32_t* d_bind_mismatch;  // device counter (StepArgs::bind_mismatch)
  PwMailbox* mailbox;     // the open resident step kernel of this engine (pw_mailbox_open), or NULL
  int mailbox_mode;       // PW_OPT_MAILBOX_MODE
  int mailbox_seg;        // PW_OPT_MAILBOX_SEG: 0 the segment form for bound batches (default), 2 never
  unsigned long long* step_signal;     // pw_engine_set_step_signal: completion word of pw_step_render_delta on a batch of one
  unsigned long long step_signal_seq;  // ... and the last number written to it
  uint32_t* d_dirty;       // per-environment dirty row record of pw_step_render_delta (grown on demand)
  int64_t dirty_cap;
  uint8_t* d_simg;         // per puzzle: observation of the static layers only (page-ordered and delta kernels)
  bool simg_cached;        // the page-ordered full render applies: the bytes of the static images that are actually
                           // read (everything but the frame padding rows) stay cache resident or nearly so
  int64_t simg_stride;     // bytes between the static images of consecutive puzzles
  uint16_t* d_estat;       // per puzzle: static zone-colour table in this engine's frame layout
  uint32_t* d_estat_off;   // byte offset of puzzle p's table in d_estat (16 B aligned)
  uint32_t pal_rgb[16];
  float pal_f32[16][4];
};

// ------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// Row bitboard of the object displaced by one action.  LEFT/RIGHT are in-lane shifts,
// UP/DOWN move rows between neighbouring lanes.  Bits leaving the 64x64 frame are dropped,
// which reproduces the bounds clause of the static tables (puzzle.py:557-561).
__device__ __forceinline__ uint64_t shift_rows(uint64_t r, int act, int lane) {
  if (act == 0) return r >> 1;  // LEFT  (-1, 0)
  if (act == 1) return r << 1;  // RIGHT (+1, 0)
  if (act == 2) {               // UP    (0, -1): new row y holds old row y + 1
    uint64_t v = __shfl_down(r, 1, PW_WAVE);
    return lane == PW_WAVE - 1 ? 0ull : v;
  }
  uint64_t v = __shfl_up(r, 1, PW_WAVE);  // DOWN (0, +1)
  return lane == 0 ? 0ull : v;
}

struct PuzzleView {
  const PwPuzzleHeader* h;  // wave-uniform: fields are fetched with scalar loads
  const uint64_t* wall;
  const uint64_t* awall;
  const uint64_t* shapes;
  const uint32_t* stat;
  const uint32_t* mcells;
  int W, H, N, G, n_mcells;
};

__device__ __forceinline__ PuzzleView view_of(const PwPuzzleHeader* hdrs, const uint8_t* blob, int pid) {
  const PwPuzzleHeader* h = hdrs + pid;
  const uint8_t* b = blob + h->base;
  PuzzleView v;
  v.h = h;
  v.wall = reinterpret_cast<const uint64_t*>(b + h->off_wall);
  v.awall = reinterpret_cast<const uint64_t*>(b + h->off_awall);
  v.shapes = reinterpret_cast<const uint64_t*>(b + h->off_shapes);
  v.stat = reinterpret_cast<const uint32_t*>(b + h->off_static);
  v.mcells = reinterpret_cast<const uint32_t*>(b + h->off_mcells);
  v.W = h->W;
  v.H = h->H;
  v.N = h->N;
  v.G = h->G;
  v.n_mcells = static_cast<int>(h->n_mcells);
  return v;
}

// lane r's row of object j placed at the packed position p = x | y << 8 (int8 each)
__device__ __forceinline__ uint64_t object_row(const PuzzleView& pv, int j, int p, int lane) {
  const PwObjEntry e = pv.h->objtab[j];
  const int x = static_cast<int8_t>(p & 0xff), y = static_cast<int8_t>((p >> 8) & 0xff);
  const int rr = lane - y;
  uint64_t r = 0;
  if (static_cast<unsigned>(rr) < static_cast<unsigned>(e.h)) r = pv.shapes[e.row_off + rr];
  return (static_cast<unsigned>(x) < 64u) ? (r << x) : 0ull;
}

// What one wavefront keeps of an environment: the agent's row bitboard, the union of all
// other movables, and whether the state is overlap-free.  Individual object rows are only
// re-read (L1/L2 hits) in the ~1 % of steps in which the agent actually touches something.
struct EnvBoards {
  uint64_t agent;   // lane r = row r of the agent
  uint64_t others;  // union of movables 1..N-1
  uint64_t wall, awall;
  bool legal;       // no movable/movable overlap, no non-agent movable on a wall
};

__device__ __forceinline__ EnvBoards load_boards(const PuzzleView& pv, int xy, int lane) {
  EnvBoards b;
  b.wall = 0;
  b.awall = 0;
  if (lane < pv.H) {
    b.wall = pv.wall[lane];
    b.awall = pv.awall[lane];
  }
  b.agent = object_row(pv, 0, __builtin_amdgcn_readlane(xy, 0), lane);
  uint64_t acc = b.agent, overlap = 0, others = 0;
  for (int j = 1; j < pv.N; j++) {
    const uint64_t rj = object_row(pv, j, __builtin_amdgcn_readlane(xy, j), lane);
    overlap |= acc & rj;
    acc |= rj;
    others |= rj;
  }
  overlap |= others & b.wall;
  b.others = others;
  b.legal = !wave_any(overlap != 0);
  return b;
}

// Exact pairwise closure for states in which objects already overlap each other or a wall
// (never produced by legal play; pins the "not already overlapping" clause,
// puzzle.py:562,592).
__device__ __forceinline__ uint32_t closure_pairwise(const PuzzleView& pv, int xy, int act, int lane, uint64_t wall) {
  uint32_t pushed = 1u, frontier = 1u;
  while (frontier) {
    const int i = __ffs(frontier) - 1;
    frontier &= ~(1u << i);
    const uint64_t ri = object_row(pv, i, __builtin_amdgcn_readlane(xy, i), lane);
    const uint64_t si = shift_rows(ri, act, lane);
    for (int j = 1; j < pv.N; j++) {
      if ((pushed >> j) & 1u) continue;
      const uint64_t rj = object_row(pv, j, __builtin_amdgcn_readlane(xy, j), lane);
      if (wave_any((si & rj) != 0) && !wave_any((ri & rj) != 0)) {
        const uint64_t sj = shift_rows(rj, act, lane);
        if (wave_any((sj & wall) != 0) && !wave_any((rj & wall) != 0)) return 0u;  // transitive stopping
        pushed |= 1u << j;
        frontier |= 1u << j;
      }
    }
  }
  return pushed;
}

// Push-set fixed point for one environment held by one wavefront (puzzle.py:348-382).
//   xy  lane j = (x | y << 8) of object j
// Returns the bit mask of objects that move (bit 0 = agent), 0 when nothing moves.
__device__ __forceinline__ uint32_t push_closure(const PuzzleView& pv, const EnvBoards& b, int xy, int act,
                                                 int lane) {
  // agent vs walls + agent walls (puzzle.py:353; static table of the agent, :272-281)
  const uint64_t s0 = shift_rows(b.agent, act, lane);
  if (wave_any((s0 & b.awall) != 0) && !wave_any((b.agent & b.awall) != 0)) return 0u;
  // States reached by legal play never contain overlaps; then the pairwise rule collapses to
  // tests against the union of the displaced push set.
  if (!b.legal) return closure_pairwise(pv, xy, act, lane, b.wall);
  if (!wave_any((s0 & b.others) != 0)) return 1u;  // ~79 % of steps: the agent moves alone

  uint32_t pushed = 1u;
  uint64_t front = s0;  // displaced rows of the objects added in the previous sweep
  for (;;) {
    uint32_t fresh = 0u;
    uint64_t add = 0;
    for (int j = 1; j < pv.N; j++) {
      if ((pushed >> j) & 1u) continue;
      const uint64_t rj = object_row(pv, j, __builtin_amdgcn_readlane(xy, j), lane);
      if (wave_any((front & rj) != 0)) {
        fresh |= 1u << j;
        add |= rj;
      }
    }
    if (!fresh) break;
    pushed |= fresh;
    front = shift_rows(add, act, lane);
    if (wave_any((front & b.wall) != 0)) return 0u;  // a pushed object hits a wall: nothing moves
  }
  return pushed;
}

// Per-environment record for the page kernel: everything the pages that no movable reaches need -- the puzzle
// (-> static image), its size (-> position of the image inside the frame) and the grid rows covered by some
// movable's bounding box.  16 bytes, fetched with ONE scalar load; written by the group step kernel
// (pw_step_render) or by pw_page_records_kernel (pw_render).
struct PageRec {
  int32_t pid;
  uint8_t H, W;
  uint16_t reserved;
  uint64_t rows;  // bit y: some movable's bounding box covers grid row y
};

#include "pw_step_kernels.inc"
#include "pw_mailbox_kernels.inc"
#include "pw_expand_kernels.inc"
#include "pw_render_kernels.inc"
#include "pw_engine.inc"
#include "pw_mailbox.inc"
#include "pw_search.inc"
#include "pw_generate.inc"
