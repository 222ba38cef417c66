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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "pw_host.h"
#include "pw_zone.h"

#define PW_WAVE 64
#ifndef PW_RENDER_THREADS
#define PW_RENDER_THREADS 256  // workgroup size of the render kernels (one environment each)
#endif

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
  uint16_t* d_estat_page;
  uint32_t* d_estat_page_off;
  int step_kernel;         // 0 group (default), 1 wavefront per env, 2 lane per env (PUSHWORLD_AMD_STEP)
  bool force_fused;        // PUSHWORLD_AMD_FUSED=1: pw_step_render always uses the single fused launch
  uint32_t* d_dirty;       // per-environment dirty row record of pw_step_render_delta (grown on demand)
  int64_t dirty_cap;
  uint8_t* d_simg;         // per puzzle: observation of the static layers only (page-ordered and delta kernels)
  bool simg_cached;        // the static images of the whole set stay cache resident: page-ordered full render
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

// ------------------------------------------------------------------------------------
// K0 reset  (gym_env.py:150-186)
// ------------------------------------------------------------------------------------
struct ResetArgs {
  const PwPuzzleHeader* hdrs;
  const uint8_t* blob;
  const int32_t* puzzle_id;
  const uint8_t* mask;
  int8_t* pos;
  int32_t* steps;
  uint8_t* term;
  uint8_t* trunc;
  int32_t batch;
  int32_t np;
};

__global__ __launch_bounds__(256) void pw_reset_kernel(ResetArgs a) {
  // one thread per (env, object slot): coalesced int16 stores of the position rows
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int env = static_cast<int>(t / a.np);
  const int j = static_cast<int>(t - static_cast<int64_t>(env) * a.np);
  if (env >= a.batch) return;
  if (a.mask && !a.mask[env]) return;
  const PwPuzzleHeader& h = a.hdrs[a.puzzle_id[env]];
  int16_t v = 0;
  if (j < h.N) v = reinterpret_cast<const int16_t*>(h.init)[j];
  reinterpret_cast<int16_t*>(a.pos)[static_cast<int64_t>(env) * a.np + j] = v;
  if (j == 0) {
    a.steps[env] = 0;
    if (a.term) a.term[env] = 0;
    if (a.trunc) a.trunc[env] = 0;
  }
}

// ------------------------------------------------------------------------------------
// K1 step  (gym_env.py:188-226 minus the observation)
// ------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------
// Episode management: environments whose episode has ended draw the puzzle of their
// next episode on the device.  The reference draws with the host's Mersenne Twister
// (gym_env.py:172 random.choice); a batch has no sequential stream to share, so the draw is a
// counter-based hash of (seed, environment, episode number): reproducible, order independent.
// ------------------------------------------------------------------------------------
struct ResampleArgs {
  int32_t* puzzle_id;
  const uint8_t* term;
  const uint8_t* trunc;
  const int32_t* table;
  uint32_t* episode;
  uint64_t seed;
  int32_t table_len;
  int32_t batch;
};

// splitmix64 finaliser over the three counters (also restated in numpy by the tests)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t env, uint64_t episode) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (env + 1ull) + 0xD1B54A32D192ED03ull * episode;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void pw_resample_kernel(ResampleArgs a) {
  const int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= a.batch) return;
  const bool done = (!a.term && !a.trunc) || (a.term && a.term[env]) || (a.trunc && a.trunc[env]);
  if (!done) return;
  const uint32_t ep = a.episode[env] + 1u;
  a.episode[env] = ep;
  const uint64_t r = mix64(a.seed, static_cast<uint64_t>(env), ep);
  // floor(r * n / 2^64): unbiased to 2^-32 for n < 2^31
  const uint32_t idx = static_cast<uint32_t>(__umul64hi(r, static_cast<uint64_t>(a.table_len)));
  a.puzzle_id[env] = a.table ? a.table[idx] : static_cast<int32_t>(idx);
}

struct StepArgs {
  const PwPuzzleHeader* hdrs;
  const uint8_t* blob;
  const int32_t* puzzle_id;
  const uint8_t* actions;
  int8_t* pos;
  int32_t* steps;
  double* reward;
  int8_t* dgoals;
  uint8_t* term;
  uint8_t* trunc;
  int32_t batch;
  int32_t max_steps;
  uint32_t flags;
  int32_t np;
  uint32_t* dirty;  // optional [batch]: cell rows the step changed, lo | hi << 8 | puzzle height << 16
                    // (lo = hi = 0: none); group kernel only
};

// One wavefront advances one environment (all lanes of the wave must call this).
//   xy_out     lane j = packed (x | y << 8) of object j after the call
//   legal_out  true when the resulting state is known to be overlap-free (legal play keeps it
//              so: pushed objects move rigidly into free cells); false = unknown
__device__ __forceinline__ void step_one_env(const StepArgs& a, int env, int lane, const PuzzleView& pv, int& xy_out,
                                             bool& legal_out) {
  const int NP = a.np;
  // first-level loads, all independent
  const int act = __builtin_amdgcn_readfirstlane(static_cast<int>(a.actions[env]));
  const int was_done = __builtin_amdgcn_readfirstlane(static_cast<int>(a.term[env] | a.trunc[env]));
  const int steps_in = __builtin_amdgcn_readfirstlane(a.steps[env]);
  int16_t* prow = reinterpret_cast<int16_t*>(a.pos) + static_cast<int64_t>(env) * NP;
  int xy = 0;  // coalesced load of the packed (x, y) int8 pairs: lane j holds object j
  if (lane < NP) xy = static_cast<uint16_t>(prow[lane]);
  legal_out = false;

  if ((a.flags & PW_STEP_AUTORESET) && was_done) {
    // next-step autoreset: this call is the reset() of a finished episode
    xy = (lane < pv.N) ? static_cast<int>(reinterpret_cast<const uint16_t*>(pv.h->init)[lane]) : 0;
    if (lane < NP) prow[lane] = static_cast<int16_t>(xy);
    if (lane == 0) {
      a.steps[env] = 0;
      a.term[env] = 0;
      a.trunc[env] = 0;
      if (a.reward) a.reward[env] = 0.0;
      if (a.dgoals) a.dgoals[env] = 0;
    }
    xy_out = xy;
    return;
  }
  if (act > 3) {  // not in Discrete(4): flag and leave the env untouched (gym_env.py:195-196)
    if (lane == 0) {
      a.term[env] = 0xFF;
      a.trunc[env] = 0xFF;
    }
    xy_out = xy;
    return;
  }

  const EnvBoards b = load_boards(pv, xy, lane);
  const uint32_t pushed = push_closure(pv, b, xy, act, lane);

  // displaced state (puzzle.py:384-394) + goal bookkeeping (puzzle.py:396-411)
  const int dx = act == 0 ? -1 : (act == 1 ? 1 : 0);
  const int dy = act == 2 ? -1 : (act == 3 ? 1 : 0);
  int x = static_cast<int8_t>(xy & 0xff), y = static_cast<int8_t>((xy >> 8) & 0xff);
  const bool is_goal_lane = lane >= 1 && lane <= pv.G;
  int gxy = 0;
  if (is_goal_lane) gxy = reinterpret_cast<const uint16_t*>(pv.h->goal)[lane - 1];
  const int before = __popcll(__ballot(is_goal_lane && (xy & 0xffff) == gxy));
  if ((pushed >> lane) & 1u) {
    x += dx;
    y += dy;
  }
  const int nxy = (x & 0xff) | ((y & 0xff) << 8);
  const int after = __popcll(__ballot(is_goal_lane && nxy == gxy));
  if (pushed && lane < pv.N) prow[lane] = static_cast<int16_t>(nxy);

  if (lane == 0) {
    const bool terminated = after == pv.G;  // vacuously true without goals (trap T8)
    const int s = steps_in + 1;
    a.steps[env] = s;
    a.term[env] = terminated ? 1 : 0;
    a.trunc[env] = (a.max_steps > 0 && s >= a.max_steps) ? 1 : 0;
    // gym_env.py:212-221: python float arithmetic == IEEE double here
    if (a.reward) a.reward[env] = terminated ? 10.0 : static_cast<double>(after - before) - 0.01;
    if (a.dgoals) a.dgoals[env] = static_cast<int8_t>(after - before);
  }
  xy_out = lane < pv.N ? nxy : 0;
  legal_out = b.legal;
}

__global__ __launch_bounds__(256) void pw_step_kernel(StepArgs a) {
  const int lane = threadIdx.x & (PW_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int env = blockIdx.x * (256 / PW_WAVE) + wave;
  if (env >= a.batch) return;
  const int pid = __builtin_amdgcn_readfirstlane(a.puzzle_id[env]);
  const PuzzleView pv = view_of(a.hdrs, a.blob, pid);
  int xy;
  bool legal;
  step_one_env(a, env, lane, pv, xy, legal);
}

// ------------------------------------------------------------------------------------
// K1b step, one LANE per environment (state-only batches: configs C2 / C4)
//
// Same predicate as the wavefront kernel, evaluated by a single lane on the few rows where two
// bounding boxes can meet: object rows are fetched from the packed shape rows (L1 hits, the
// lanes of a wave mostly share a puzzle) instead of being spread over a wave.  64x less
// ALU work and coalesced per-env outputs.  These row helpers are what the lane-group kernels further down
// (pw_step / pw_rollout default, pw_expand4, pw_search) are built from; only the fused step+render launch
// keeps the wavefront formulation.
// ------------------------------------------------------------------------------------
struct LanePuzzle {
  const PwPuzzleHeader* h;
  const uint64_t* wall;
  const uint64_t* awall;
  const uint64_t* shapes;
  int H, N, G;
};

struct LaneObj {
  int x, y, w, h, off;
};

// ot = packed PwObjEntry (w | h << 8 | row_off << 16)
__device__ __forceinline__ LaneObj lane_obj(uint32_t ot, int xy) {
  LaneObj o;
  o.x = static_cast<int8_t>(xy & 0xff);
  o.y = static_cast<int8_t>((xy >> 8) & 0xff);
  o.w = static_cast<int>(ot & 0xffu);
  o.h = static_cast<int>((ot >> 8) & 0xffu);
  o.off = static_cast<int>(ot >> 16);
  return o;
}

// The agent's wall test with every load issued up front (one memory latency instead of one per
// row): window of 8 grid rows around the agent; taller agents use the generic loop.
__device__ __forceinline__ bool lane_agent_blocked(const LanePuzzle& p, const LaneObj& o, int act);

// row yy of the board that holds only object o (same conventions as object_row())
__device__ __forceinline__ uint64_t lane_row(const LanePuzzle& p, const LaneObj& o, int yy) {
  const int rr = yy - o.y;
  uint64_t r = 0;
  if (static_cast<unsigned>(rr) < static_cast<unsigned>(o.h) && static_cast<unsigned>(yy) < 64u) r = p.shapes[o.off + rr];
  return (static_cast<unsigned>(o.x) < 64u) ? (r << o.x) : 0ull;
}

// row yy of the board of object o displaced by the action (cf. shift_rows())
__device__ __forceinline__ uint64_t lane_row_shifted(const LanePuzzle& p, const LaneObj& o, int yy, int act) {
  if (act == 0) return lane_row(p, o, yy) >> 1;
  if (act == 1) return lane_row(p, o, yy) << 1;
  if (act == 2) return yy == 63 ? 0ull : lane_row(p, o, yy + 1);
  return yy == 0 ? 0ull : lane_row(p, o, yy - 1);
}

// moving o collides with the static rows and o does not overlap them now (puzzle.py:522-564)
__device__ __forceinline__ bool lane_blocked(const LanePuzzle& p, const LaneObj& o, const uint64_t* rows, int act) {
  uint64_t hit = 0, now = 0;
  for (int yy = o.y - 1; yy <= o.y + o.h; yy++) {
    if (static_cast<unsigned>(yy) >= static_cast<unsigned>(p.H)) continue;
    const uint64_t g = rows[yy];
    hit |= lane_row_shifted(p, o, yy, act) & g;
    now |= lane_row(p, o, yy) & g;
  }
  return hit != 0 && now == 0;
}

// moving a pushes b: they overlap after the move and do not overlap now (puzzle.py:567-593)
__device__ __forceinline__ bool lane_pushes(const LanePuzzle& p, const LaneObj& a, const LaneObj& b, int act, int dx,
                                            int dy) {
  // bounding boxes of the displaced pusher and of the pushee must meet
  if (a.x + dx >= b.x + b.w || b.x >= a.x + dx + a.w || a.y + dy >= b.y + b.h || b.y >= a.y + dy + a.h) return false;
  uint64_t hit = 0, now = 0;
  for (int yy = b.y; yy < b.y + b.h; yy++) {
    const uint64_t rb = lane_row(p, b, yy);
    hit |= lane_row_shifted(p, a, yy, act) & rb;
    now |= lane_row(p, a, yy) & rb;
  }
  return hit != 0 && now == 0;
}

__device__ __forceinline__ bool lane_agent_blocked(const LanePuzzle& p, const LaneObj& o, int act) {
  if (o.h > 6) return lane_blocked(p, o, p.awall, act);
  uint64_t g[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int yy = o.y - 1 + k;
    g[k] = (static_cast<unsigned>(yy) < static_cast<unsigned>(p.H)) ? p.awall[yy] : 0ull;
    // sh[k] = board row yy of the agent (k = 0 and k = 7 are outside for h <= 6)
    const int rr = k - 1;
    uint64_t r = 0;
    if (static_cast<unsigned>(rr) < static_cast<unsigned>(o.h) && static_cast<unsigned>(yy) < 64u) r = p.shapes[o.off + rr];
    sh[k] = (static_cast<unsigned>(o.x) < 64u) ? (r << o.x) : 0ull;
  }
  uint64_t hit = 0, now = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int yy = o.y - 1 + k;
    uint64_t moved;
    if (act == 0) moved = sh[k] >> 1;
    else if (act == 1) moved = sh[k] << 1;
    else if (act == 2) moved = (k < 7 && yy != 63) ? sh[k + 1] : 0ull;
    else moved = (k > 0 && yy != 0) ? sh[k - 1] : 0ull;
    hit |= moved & g[k];
    now |= sh[k] & g[k];
  }
  return hit != 0 && now == 0;
}

template <int NP>
__device__ __forceinline__ int lane_pos(const uint32_t (&P)[NP / 2], int j) {
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < NP / 2; k++) v = (k == (j >> 1)) ? P[k] : v;
  return static_cast<int>((v >> (16 * (j & 1))) & 0xffffu);
}

// Per-lane environment state kept in registers between the steps of a rollout.
template <int NP>
struct LaneEnv {
  uint32_t P[NP / 2];  // packed (x, y) int8 pairs, two objects per dword
  int steps;
  int term, trunc;     // flags of the last step (0 / 1 / 0xFF)
  double reward;
  int dgoals;
};

template <int NP>
__device__ __forceinline__ void lane_load(const StepArgs& a, int env, LaneEnv<NP>& s) {
  const uint32_t* prow = reinterpret_cast<const uint32_t*>(a.pos) + static_cast<int64_t>(env) * (NP / 2);
  if (NP == 4) {
    const uint2 v = *reinterpret_cast<const uint2*>(prow);
    s.P[0] = v.x;
    s.P[1] = v.y;
  } else {
#pragma unroll
    for (int k = 0; k < NP / 8; k++) {
      const uint4 v = reinterpret_cast<const uint4*>(prow)[k];
      s.P[4 * k + 0] = v.x;
      s.P[4 * k + 1] = v.y;
      s.P[4 * k + 2] = v.z;
      s.P[4 * k + 3] = v.w;
    }
  }
  s.steps = a.steps[env];
  s.term = a.term[env];
  s.trunc = a.trunc[env];
  s.reward = 0.0;
  s.dgoals = 0;
}

template <int NP>
__device__ __forceinline__ void lane_store_pos(const StepArgs& a, int env, const LaneEnv<NP>& s) {
  uint32_t* prow = reinterpret_cast<uint32_t*>(a.pos) + static_cast<int64_t>(env) * (NP / 2);
  if (NP == 4) {
    *reinterpret_cast<uint2*>(prow) = make_uint2(s.P[0], s.P[1]);
  } else {
#pragma unroll
    for (int k = 0; k < NP / 8; k++)
      reinterpret_cast<uint4*>(prow)[k] = make_uint4(s.P[4 * k], s.P[4 * k + 1], s.P[4 * k + 2], s.P[4 * k + 3]);
  }
}

// One pw_step of one environment on one lane (gym_env.py:188-226 minus the observation).
// Returns true when the positions changed.
template <int NP>
__device__ __forceinline__ bool lane_step(const LanePuzzle& p, const PwPuzzleHeader* h, const uint32_t (&OT)[NP],
                                          LaneEnv<NP>& s, int act, uint32_t flags, int max_steps) {
  if ((flags & PW_STEP_AUTORESET) && (s.term | s.trunc)) {
    // next-step autoreset: this step is the reset() of a finished episode
#pragma unroll
    for (int k = 0; k < NP / 2; k++) s.P[k] = reinterpret_cast<const uint32_t*>(h->init)[k];
    s.steps = 0;
    s.term = 0;
    s.trunc = 0;
    s.reward = 0.0;
    s.dgoals = 0;
    return true;
  }
  if (act > 3) {  // not in Discrete(4): flag and leave the env untouched (gym_env.py:195-196)
    s.term = 0xFF;
    s.trunc = 0xFF;
    return false;
  }
  const int dx = act == 0 ? -1 : (act == 1 ? 1 : 0);
  const int dy = act == 2 ? -1 : (act == 3 ? 1 : 0);

  uint32_t pushed = 0;
  const LaneObj agent = lane_obj(OT[0], static_cast<int>(s.P[0] & 0xffffu));
  if (!lane_agent_blocked(p, agent, act)) {  // puzzle.py:353
    pushed = 1u;
    uint32_t frontier = 0;
    bool blocked = false;
    // sweep of the agent over all other movables (positions in registers, static indices)
#pragma unroll
    for (int j = 1; j < NP; j++) {
      if (j < p.N && !blocked) {
        const LaneObj oj = lane_obj(OT[j], static_cast<int>((s.P[j >> 1] >> (16 * (j & 1))) & 0xffffu));
        if (lane_pushes(p, agent, oj, act, dx, dy)) {
          if (lane_blocked(p, oj, p.wall, act)) {
            blocked = true;  // transitive stopping
          } else {
            pushed |= 1u << j;
            frontier |= 1u << j;
          }
        }
      }
    }
    // pushed objects push further objects (about 0.1 % of steps)
    while (frontier && !blocked) {
      const int i = __ffs(frontier) - 1;
      frontier &= frontier - 1;
      const LaneObj oi = lane_obj(reinterpret_cast<const uint32_t*>(h->objtab)[i], lane_pos<NP>(s.P, i));
      for (int j = 1; j < p.N && !blocked; j++) {
        if ((pushed >> j) & 1u) continue;
        const LaneObj oj = lane_obj(reinterpret_cast<const uint32_t*>(h->objtab)[j], lane_pos<NP>(s.P, j));
        if (lane_pushes(p, oi, oj, act, dx, dy)) {
          if (lane_blocked(p, oj, p.wall, act)) {
            blocked = true;
          } else {
            pushed |= 1u << j;
            frontier |= 1u << j;
          }
        }
      }
    }
    if (blocked) pushed = 0;
  }

  // displaced state + goal bookkeeping (puzzle.py:384-411)
  int before = 0, after = 0;
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const uint32_t sh = 16 * (j & 1);
    const uint32_t cur = (s.P[j >> 1] >> sh) & 0xffffu;
    uint32_t nxt = cur;
    if ((pushed >> j) & 1u) {
      const int x = static_cast<int8_t>(cur & 0xff) + dx, y = static_cast<int8_t>((cur >> 8) & 0xff) + dy;
      nxt = static_cast<uint32_t>(x & 0xff) | (static_cast<uint32_t>(y & 0xff) << 8);
      s.P[j >> 1] = (s.P[j >> 1] & ~(0xffffu << sh)) | (nxt << sh);
    }
    if (j >= 1 && j <= p.G) {
      const uint32_t g = reinterpret_cast<const uint16_t*>(h->goal)[j - 1];
      before += cur == g;
      after += nxt == g;
    }
  }
  const bool terminated = after == p.G;  // vacuously true without goals (trap T8)
  s.steps += 1;
  s.term = terminated ? 1 : 0;
  s.trunc = (max_steps > 0 && s.steps >= max_steps) ? 1 : 0;
  // gym_env.py:212-221: python float arithmetic == IEEE double here
  s.reward = terminated ? 10.0 : static_cast<double>(after - before) - 0.01;
  s.dgoals = after - before;
  return pushed != 0;
}

template <int NP>
__device__ __forceinline__ void lane_puzzle(const StepArgs& a, int pid, LanePuzzle& p, uint32_t (&OT)[NP]) {
  const PwPuzzleHeader* h = a.hdrs + pid;
  p.h = h;
  const uint8_t* b = a.blob + h->base;
  p.wall = reinterpret_cast<const uint64_t*>(b + h->off_wall);
  p.awall = reinterpret_cast<const uint64_t*>(b + h->off_awall);
  p.shapes = reinterpret_cast<const uint64_t*>(b + h->off_shapes);
  p.H = h->H;
  p.N = h->N;
  p.G = h->G;
  // the whole object table (bounding boxes + shape-row offsets) with wide loads
#pragma unroll
  for (int k = 0; k < NP / 4; k++) {
    const uint4 v = reinterpret_cast<const uint4*>(h->objtab)[k];
    OT[4 * k + 0] = v.x;
    OT[4 * k + 1] = v.y;
    OT[4 * k + 2] = v.z;
    OT[4 * k + 3] = v.w;
  }
}

template <int NP>
__global__ __launch_bounds__(256) void pw_step_lane_kernel(StepArgs a) {
  const int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= a.batch) return;
  const int act = a.actions[env];
  LaneEnv<NP> s;
  lane_load<NP>(a, env, s);
  LanePuzzle p;
  uint32_t OT[NP];
  lane_puzzle<NP>(a, a.puzzle_id[env], p, OT);
  const int t0 = s.term, u0 = s.trunc;
  const bool changed = lane_step<NP>(p, p.h, OT, s, act, a.flags, a.max_steps);
  if (changed) lane_store_pos<NP>(a, env, s);
  if (act > 3 && !((a.flags & PW_STEP_AUTORESET) && (t0 | u0))) {
    a.term[env] = 0xFF;
    a.trunc[env] = 0xFF;
    return;
  }
  a.steps[env] = s.steps;
  a.term[env] = static_cast<uint8_t>(s.term);
  a.trunc[env] = static_cast<uint8_t>(s.trunc);
  if (a.reward) a.reward[env] = s.reward;
  if (a.dgoals) a.dgoals[env] = static_cast<int8_t>(s.dgoals);
}

// K1c rollout: T consecutive steps of every environment in ONE launch.  The environment lives in
// registers between steps; puzzle tables are L1 hits after the first step.
struct RolloutArgs {
  StepArgs s;              // actions = uint8 [T][B]
  int32_t num_steps;
  double* reward_hist;     // optional [T][B]
  uint8_t* term_hist;      // optional [T][B]
  uint8_t* trunc_hist;     // optional [T][B]
};

template <int NP>
__global__ __launch_bounds__(256) void pw_rollout_lane_kernel(RolloutArgs r) {
  const StepArgs& a = r.s;
  const int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= a.batch) return;
  LaneEnv<NP> s;
  lane_load<NP>(a, env, s);
  LanePuzzle p;
  uint32_t OT[NP];
  lane_puzzle<NP>(a, a.puzzle_id[env], p, OT);
  bool changed = false;
  for (int t = 0; t < r.num_steps; t++) {
    const int64_t o = static_cast<int64_t>(t) * a.batch + env;
    const int act = a.actions[o];
    changed = lane_step<NP>(p, p.h, OT, s, act, a.flags, a.max_steps) || changed;
    if (r.reward_hist) r.reward_hist[o] = s.reward;
    if (r.term_hist) r.term_hist[o] = static_cast<uint8_t>(s.term);
    if (r.trunc_hist) r.trunc_hist[o] = static_cast<uint8_t>(s.trunc);
  }
  if (changed) lane_store_pos<NP>(a, env, s);
  a.steps[env] = s.steps;
  a.term[env] = static_cast<uint8_t>(s.term);
  a.trunc[env] = static_cast<uint8_t>(s.trunc);
  if (a.reward) a.reward[env] = s.reward;
  if (a.dgoals) a.dgoals[env] = static_cast<int8_t>(s.dgoals);
}

// The agent's wall test spread over the lanes of its group: lane k tests grid row y - 1 + k of the window
// the agent can reach (h + 2 rows), the verdict is two ballots.  ~6x fewer VALU cycles than one lane walking
// the window (the step kernel is VALU bound, profiles/r01_sq2.txt).  Wave-uniform fallback for agents taller
// than the group.
template <int GS>
__device__ __forceinline__ bool group_agent_blocked(const LanePuzzle& p, int xy, uint32_t ot, int lj, int gbase,
                                                    unsigned long long gmask, bool play, int act) {
  const LaneObj ag = lane_obj(static_cast<uint32_t>(__shfl(static_cast<int>(ot), gbase, PW_WAVE)), __shfl(xy, gbase, PW_WAVE));
  if (__ballot(ag.h + 2 > GS) != 0ull) {
    bool blk = false;
    if (play && lj == 0) blk = lane_agent_blocked(p, ag, act);
    return (__ballot(blk) & gmask) != 0ull;
  }
  const int yy = ag.y - 1 + lj;
  bool hit = false, now = false;
  if (play && lj < ag.h + 2 && static_cast<unsigned>(yy) < static_cast<unsigned>(p.H)) {
    const uint64_t g = p.awall[yy];
    now = (lane_row(p, ag, yy) & g) != 0ull;
    hit = (lane_row_shifted(p, ag, yy, act) & g) != 0ull;
  }
  const unsigned long long hm = __ballot(hit) & gmask, nm = __ballot(now) & gmask;
  return hm != 0ull && nm == 0ull;
}

// The push set of one action for the environment / state held by a lane group (lane j = movable j):
// agent wall test, then the fixed point "some member pushes my object" by ballots; any wall-blocked member
// kills the move (transitive stopping, puzzle.py:376-379).  Returns the mask of the objects that move
// (bit 0 = agent), 0 when nothing moves.  Shared by the step, planner-expansion and search kernels.
template <int GS>
__device__ __forceinline__ uint32_t group_push_closure(const LanePuzzle& p, int xy, uint32_t ot, const LaneObj& me, int lj,
                                                       int gbase, unsigned long long gmask, bool play, bool dead, int act,
                                                       int dx, int dy) {
  uint32_t pushed = 1u, frontier = 0u;
  int cur = 0;
  bool active = play && !dead;
  while (__ballot(active) != 0ull) {  // every lane tests the group's current pusher against its own object
    const int pxy = __shfl(xy, gbase + cur, PW_WAVE);
    const uint32_t pot = static_cast<uint32_t>(__shfl(static_cast<int>(ot), gbase + cur, PW_WAVE));
    bool hit = false, blk = false;
    if (active && lj >= 1 && lj < p.N && !((pushed >> lj) & 1u)) {
      hit = lane_pushes(p, lane_obj(pot, pxy), me, act, dx, dy);
      if (hit) blk = lane_blocked(p, me, p.wall, act);
    }
    const unsigned long long hm = __ballot(hit), bk = __ballot(blk);
    const uint32_t fresh = static_cast<uint32_t>((hm & gmask) >> gbase);
    if ((bk & gmask) != 0ull) dead = true;
    pushed |= fresh;
    frontier |= fresh;
    active = active && !dead && frontier != 0u;
    if (active) {
      cur = __ffs(frontier) - 1;
      frontier &= frontier - 1u;
    }
  }
  return (play && !dead) ? pushed : 0u;
}

template <int GS>
__device__ __forceinline__ uint32_t group_push_set(const LanePuzzle& p, int xy, uint32_t ot, const LaneObj& me, int lj,
                                                   int gbase, unsigned long long gmask, bool play, int act, int dx, int dy) {
  const bool dead = group_agent_blocked<GS>(p, xy, ot, lj, gbase, gmask, play, act);
  return group_push_closure<GS>(p, xy, ot, me, lj, gbase, gmask, play, dead, act, dx, dy);
}

// The agent's wall test for all four actions of one state (planner expansion): the window rows are loaded
// once, LEFT / RIGHT are in-lane shifts, UP / DOWN take the neighbouring lane's row.  Bit a of the result =
// action a is blocked.  Same verdicts as four group_agent_blocked calls.
template <int GS>
__device__ __forceinline__ uint32_t group_agent_blocked4(const LanePuzzle& p, int xy, uint32_t ot, int lj, int gbase,
                                                         unsigned long long gmask, bool play) {
  const LaneObj ag = lane_obj(static_cast<uint32_t>(__shfl(static_cast<int>(ot), gbase, PW_WAVE)), __shfl(xy, gbase, PW_WAVE));
  if (__ballot(ag.h + 2 > GS) != 0ull) {
    uint32_t m = 0;
    for (int act = 0; act < 4; act++)
      if (group_agent_blocked<GS>(p, xy, ot, lj, gbase, gmask, play, act)) m |= 1u << act;
    return m;
  }
  const int yy = ag.y - 1 + lj;
  const bool in_win = play && lj < ag.h + 2;
  const uint64_t sh = in_win ? lane_row(p, ag, yy) : 0ull;  // 0 outside the object's rows / the 64-row frame
  const uint64_t g = (in_win && static_cast<unsigned>(yy) < static_cast<unsigned>(p.H)) ? p.awall[yy] : 0ull;
  // UP: new row yy holds old row yy + 1 (next lane); DOWN: old row yy - 1 (previous lane)
  uint64_t up = __shfl_down(sh, 1, PW_WAVE), dn = __shfl_up(sh, 1, PW_WAVE);
  if (lj == GS - 1 || yy == 63) up = 0ull;
  if (lj == 0 || yy == 0) dn = 0ull;
  const bool now = (sh & g) != 0ull;
  const bool overlapping = (__ballot(now) & gmask) != 0ull;  // already inside a wall: never "blocked" (puzzle.py:562)
  uint32_t m = 0;
  if ((__ballot(((sh >> 1) & g) != 0ull) & gmask) != 0ull) m |= 1u;
  if ((__ballot(((sh << 1) & g) != 0ull) & gmask) != 0ull) m |= 2u;
  if ((__ballot((up & g) != 0ull) & gmask) != 0ull) m |= 4u;
  if ((__ballot((dn & g) != 0ull) & gmask) != 0ull) m |= 8u;
  return overlapping ? 0u : m;
}

// ------------------------------------------------------------------------------------
// K1d step / rollout, GS lanes per environment (lane j of a group = movable j)
//
// The lane-per-env kernel above is bound by the latency of its serial loop over the objects with
// one wave per SIMD; here the objects of an environment are tested in parallel (each lane asks
// "does the current pusher push MY object, and am I wall-blocked?"), the push set grows by ballots
// inside the 16- or 32-lane group, and a 65 536-env batch is 16 k waves instead of 1 k, so the
// remaining L1/L2 latencies overlap.  Default kernel of pw_step and pw_rollout.
// ------------------------------------------------------------------------------------
template <int GS>
__global__ __launch_bounds__(256) void pw_step_group_kernel(RolloutArgs r) {
  const StepArgs& a = r.s;
  constexpr int kGroups = 256 / GS;
  const int lane = threadIdx.x & (PW_WAVE - 1);
  const int lj = threadIdx.x & (GS - 1);
  const int gbase = lane & ~(GS - 1);
  const unsigned long long gmask = ((1ull << (GS - 1) << 1) - 1ull) << gbase;
  const int env = blockIdx.x * kGroups + static_cast<int>(threadIdx.x) / GS;
  const bool live = env < a.batch;
  const int e = live ? env : 0;

  const int pid = a.puzzle_id[e];
  LanePuzzle p;
  const PwPuzzleHeader* h = a.hdrs + pid;
  p.h = h;
  {
    const uint8_t* b = a.blob + h->base;
    p.wall = reinterpret_cast<const uint64_t*>(b + h->off_wall);
    p.awall = reinterpret_cast<const uint64_t*>(b + h->off_awall);
    p.shapes = reinterpret_cast<const uint64_t*>(b + h->off_shapes);
  }
  p.H = h->H;
  p.N = live ? h->N : 0;
  p.G = h->G;
  const int N = p.N;
  int16_t* prow = reinterpret_cast<int16_t*>(a.pos) + static_cast<int64_t>(e) * a.np;
  int xy = (lj < N) ? static_cast<int>(static_cast<uint16_t>(prow[lj])) : 0;
  const uint32_t ot = (lj < N) ? reinterpret_cast<const uint32_t*>(h->objtab)[lj] : 0u;
  const bool is_goal_lane = live && lj >= 1 && lj <= p.G;
  const int gxy = is_goal_lane ? static_cast<int>(reinterpret_cast<const uint16_t*>(h->goal)[lj - 1]) : -1;
  int steps = a.steps[e], term = a.term[e], trunc = a.trunc[e];
  double reward = 0.0;
  int dgoals = 0;
  bool changed = false, any_played = false;
  int row_lo = 127, row_hi = -1;  // cell rows whose pixels the LAST step changed (a.dirty)

  for (int t = 0; t < r.num_steps; t++) {
    const int64_t o = static_cast<int64_t>(t) * a.batch + e;
    const int act = live ? static_cast<int>(a.actions[o]) : 0;
    const bool do_reset = live && (a.flags & PW_STEP_AUTORESET) && (term | trunc);
    const bool bad = live && !do_reset && act > 3;
    const bool play = live && !do_reset && !bad;
    const int dx = act == 0 ? -1 : (act == 1 ? 1 : 0);
    const int dy = act == 2 ? -1 : (act == 3 ? 1 : 0);
    const LaneObj me = lane_obj(ot, xy);

    const uint32_t moved = group_push_set<GS>(p, xy, ot, me, lj, gbase, gmask, play, act, dx, dy);

    // displaced state + goal bookkeeping (puzzle.py:384-411)
    int nxy = xy;
    if ((moved >> lj) & 1u) {
      const int x = static_cast<int8_t>(xy & 0xff) + dx, y = static_cast<int8_t>((xy >> 8) & 0xff) + dy;
      nxy = (x & 0xff) | ((y & 0xff) << 8);
    }
    const int before = __popcll(__ballot(is_goal_lane && xy == gxy) & gmask);
    const int after = __popcll(__ballot(is_goal_lane && nxy == gxy) & gmask);
    if (a.dirty) {  // rows swept by the moved objects, old and new position (pushed objects touch: one interval)
      int lo = 127, hi = -1;
      if ((moved >> lj) & 1u) {
        const int y1 = static_cast<int8_t>((nxy >> 8) & 0xff);
        lo = min(y1, y1 - dy);
        hi = max(y1, y1 - dy) + me.h;
      }
#pragma unroll
      for (int o = GS / 2; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, GS));
        hi = max(hi, __shfl_xor(hi, o, GS));
      }
      row_lo = do_reset ? 0 : lo;
      row_hi = do_reset ? PW_MAX_DIM : hi;
    }
    if (play) {
      xy = nxy;
      changed = changed || moved != 0u;
      const bool terminated = after == p.G;  // vacuously true without goals (trap T8)
      steps += 1;
      term = terminated ? 1 : 0;
      trunc = (a.max_steps > 0 && steps >= a.max_steps) ? 1 : 0;
      reward = terminated ? 10.0 : static_cast<double>(after - before) - 0.01;  // gym_env.py:212-221
      dgoals = after - before;
      any_played = true;
    } else if (do_reset) {
      xy = (lj < N) ? static_cast<int>(reinterpret_cast<const uint16_t*>(h->init)[lj]) : 0;
      changed = true;
      steps = 0;
      term = 0;
      trunc = 0;
      reward = 0.0;
      dgoals = 0;
      any_played = true;
    } else if (bad) {  // not in Discrete(4): flag, leave the env untouched (gym_env.py:195-196)
      term = 0xFF;
      trunc = 0xFF;
    }
    if (live && lj == 0) {
      if (r.reward_hist) r.reward_hist[o] = reward;
      if (r.term_hist) r.term_hist[o] = static_cast<uint8_t>(term);
      if (r.trunc_hist) r.trunc_hist[o] = static_cast<uint8_t>(trunc);
    }
  }
  if (!live) return;
  if (changed && lj < a.np) prow[lj] = static_cast<int16_t>(xy);
  if (lj == 0) {
    a.term[env] = static_cast<uint8_t>(term);
    a.trunc[env] = static_cast<uint8_t>(trunc);
    if (a.dirty) {
      const int lo = max(row_lo, 0), hi = min(row_hi, PW_MAX_DIM);
      a.dirty[env] = hi > lo ? static_cast<uint32_t>(lo | (hi << 8) | (p.H << 16)) : 0u;
    }
    if (any_played) {
      a.steps[env] = steps;
      if (a.reward) a.reward[env] = reward;
      if (a.dgoals) a.dgoals[env] = static_cast<int8_t>(dgoals);
    }
  }
}

// ------------------------------------------------------------------------------------
// K3 expand4  (best_first_search.h:76-78 -> pushworld_puzzle.cc:386-469)
// ------------------------------------------------------------------------------------
struct ExpandArgs {
  const PwPuzzleHeader* hdrs;
  const uint8_t* blob;
  int32_t puzzle;
  const int32_t* states;
  int32_t* succ;
  uint32_t* moved;
  uint8_t* goal;
  int32_t num_states;
};

template <int GS>
__global__ __launch_bounds__(256) void pw_expand4_kernel(ExpandArgs a) {
  constexpr int kGroups = 256 / GS;
  const int lane = threadIdx.x & (PW_WAVE - 1);
  const int lj = threadIdx.x & (GS - 1);
  const int gbase = lane & ~(GS - 1);
  const unsigned long long gmask = ((1ull << (GS - 1) << 1) - 1ull) << gbase;
  const int sidx = blockIdx.x * kGroups + static_cast<int>(threadIdx.x) / GS;
  const bool live = sidx < a.num_states;
  const int64_t sg = live ? sidx : 0;

  LanePuzzle p;
  const PwPuzzleHeader* h = a.hdrs + a.puzzle;
  p.h = h;
  {
    const uint8_t* b = a.blob + h->base;
    p.wall = reinterpret_cast<const uint64_t*>(b + h->off_wall);
    p.awall = reinterpret_cast<const uint64_t*>(b + h->off_awall);
    p.shapes = reinterpret_cast<const uint64_t*>(b + h->off_shapes);
  }
  p.H = h->H;
  p.N = h->N;
  p.G = h->G;
  const int N = p.N;
  // Position2D = x * 10000 + y (pushworld_puzzle.h:32-37)
  int p2d = 0, xy = 0;
  if (lj < N) {
    p2d = a.states[sg * N + lj];
    const int x = p2d / PW_POSITION_LIMIT;
    xy = (x & 0xff) | (((p2d - x * PW_POSITION_LIMIT) & 0xff) << 8);
  }
  const uint32_t ot = (lj < N) ? reinterpret_cast<const uint32_t*>(h->objtab)[lj] : 0u;
  const LaneObj me = lane_obj(ot, xy);
  const bool is_goal_lane = lj >= 1 && lj <= p.G;
  int g2d = -1;
  if (is_goal_lane) {
    const int gxy = reinterpret_cast<const uint16_t*>(h->goal)[lj - 1];
    g2d = (gxy & 0xff) * PW_POSITION_LIMIT + ((gxy >> 8) & 0xff);
  }
  const uint32_t agent_blocked = group_agent_blocked4<GS>(p, xy, ot, lj, gbase, gmask, live);
#pragma unroll 1
  for (int act = 0; act < 4; act++) {
    const int dx = act == 0 ? -1 : (act == 1 ? 1 : 0);
    const int dy = act == 2 ? -1 : (act == 3 ? 1 : 0);
    const uint32_t pushed =
        group_push_closure<GS>(p, xy, ot, me, lj, gbase, gmask, live, ((agent_blocked >> act) & 1u) != 0u, act, dx, dy);
    const int n2d = p2d + (((pushed >> lj) & 1u) ? dx * PW_POSITION_LIMIT + dy : 0);
    const int64_t o = sg * 4 + act;
    if (live && lj < N) a.succ[o * N + lj] = n2d;
    const int hits = __popcll(__ballot(is_goal_lane && n2d == g2d) & gmask);
    if (live && lj == 0) {
      a.moved[o] = pushed;  // = moved_object_indices (pushworld_puzzle.cc:446-457); empty when blocked
      a.goal[o] = hits == p.G ? 1 : 0;
    }
  }
}

// ------------------------------------------------------------------------------------
// K2 render  (puzzle.py:426-469, :596-638; env_utils.py:44-91)
// ------------------------------------------------------------------------------------
struct RenderArgs {
  const PwPuzzleHeader* hdrs;
  const uint8_t* blob;
  const int32_t* puzzle_id;
  const int8_t* pos;
  uint8_t* obs;
  const uint16_t* estat;      // static zone tables (engine frame layout)
  const uint32_t* estat_off;
  int64_t env_stride;
  int32_t batch;
  int32_t np;
  int32_t ppc, bw;
  int32_t pad_h, pad_w;    // frame in cells
  int32_t obs_bytes;       // bytes of one observation
  uint32_t pal_rgb[16];    // byte0 = R, byte1 = G, byte2 = B
  float pal_f32[16][4];    // uint8 -> float32 / 255 (env_utils.py:65-72), exact IEEE division
  int32_t do_step;         // fused pw_step_render: wave 0 advances the environment first
  int32_t skip_movables;   // draw the static layers only (engine setup: static images)
  const uint32_t* dirty_rows;  // generic kernel, pw_step_render_delta: per env cell rows to redraw (NULL = all)
  StepArgs step;
};

static void launch_render(PwEngine* e, const RenderArgs& ra, int32_t batch, hipStream_t st);

// LDS layout of the render kernels (dynamic):
//   [0, 16)        8 zero guard entries in front of E
//   E              zone table of this environment, uint16 entries, 16 B aligned; followed by >= 3
//                  zero entries (part of the static table image)
//   spos, pal, flag
struct RenderLds {
  uint16_t* E;
  int16_t* spos;
  uint32_t* pal;
  uint32_t* flag;
};

__device__ __forceinline__ RenderLds carve_lds(unsigned char* smem, int e_bytes) {
  RenderLds l;
  l.E = reinterpret_cast<uint16_t*>(smem + 16);
  l.spos = reinterpret_cast<int16_t*>(smem + 16 + e_bytes);
  l.pal = reinterpret_cast<uint32_t*>(smem + 16 + e_bytes + 64);
  l.flag = l.pal + 16;
  return l;
}

// Writes the three sub-row entries of one movable cell over the static table.
__device__ __forceinline__ void patch_cell(const PuzzleView& pv, uint16_t* E, const int16_t* spos, uint32_t c,
                                           int estride, int c0, int row_lo = 0, int row_hi = PW_MAX_DIM) {
  const int obj = c >> 24;
  const int p = static_cast<uint16_t>(spos[obj]);
  const int x = static_cast<int8_t>(p & 0xff) + static_cast<int>(c & 0xff);
  const int y = static_cast<int8_t>((p >> 8) & 0xff) + static_cast<int>((c >> 8) & 0xff);
  if (static_cast<unsigned>(x) >= static_cast<unsigned>(pv.W) || static_cast<unsigned>(y) >= static_cast<unsigned>(pv.H) ||
      y < row_lo || y >= row_hi)
    return;
  const uint32_t kind = obj == 0 ? 3u : (obj <= pv.G ? 4u : 5u);
  const uint32_t om = (c >> 16) & 0xffu;
#pragma unroll
  for (int zy = 0; zy < 3; zy++) {
    const int idx = (3 * y + zy) * estride + x + c0;
    const uint32_t gb = pw_entry_goal_bits(E[idx]);  // goal outlines stay on top (puzzle.py:458)
    E[idx] = static_cast<uint16_t>(pw_zone_entry(kind, pw_zone_border_bits(om, zy), gb));
  }
}

// Per-environment zone table: the puzzle's static table (walls, agent walls, background, goal
// outlines; precomputed per engine) is copied into LDS with 16-byte loads and the cells under
// the movables are patched in painter order (puzzle.py:453-458).  In an overlap-free state
// (always, under legal play) no two movables share a cell and the patches are independent;
// otherwise objects are applied one after the other so that a higher index wins.
// Only the cell rows [row_lo, row_hi) are needed by the caller (incremental redraw): the rest of the table
// is neither copied nor patched.
__device__ __forceinline__ void build_zone_table(const RenderArgs& a, const PuzzleView& pv, int pid, int env,
                                                 const RenderLds& l, int estride, int c0, int row_lo = 0,
                                                 int row_hi = PW_MAX_DIM) {
  const int tid = threadIdx.x;
  const int lane = tid & (PW_WAVE - 1);
  const int n_entries = 3 * pv.H * estride;
  const int n16 = (2 * (n_entries + 3) + 15) >> 4;
  const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(a.estat) + a.estat_off[pid]);
  uint4* dst = reinterpret_cast<uint4*>(l.E);
  const int i_lo = (2 * 3 * max(row_lo, 0) * estride) >> 4;
  const int i_hi = min(n16, (2 * (3 * min(row_hi, pv.H) * estride + 3) + 15) >> 4);
  for (int i = i_lo + tid; i < i_hi; i += blockDim.x) dst[i] = src[i];
  if (tid < 16) l.pal[tid] = a.pal_rgb[tid];
  if (tid >= 64 && tid < 72) l.E[tid - 72] = 0;  // guard entries E[-8..-1]
  if (tid < PW_WAVE) {
    // wave 0: (fused step,) positions to LDS + overlap check of the state to draw
    int xy = 0;
    bool legal;
    if (a.do_step) {
      step_one_env(a.step, env, lane, pv, xy, legal);
    } else {
      if (lane < a.np) xy = static_cast<uint16_t>(reinterpret_cast<const int16_t*>(a.pos)[static_cast<int64_t>(env) * a.np + lane]);
      legal = load_boards(pv, xy, lane).legal;
    }
    if (lane < 32) l.spos[lane] = static_cast<int16_t>(xy);
    if (lane == 0) l.flag[0] = legal ? 1u : 0u;
  }
  __syncthreads();
  if (a.skip_movables) return;
  if (l.flag[0]) {
    for (int m = tid; m < pv.n_mcells; m += blockDim.x)
      patch_cell(pv, l.E, l.spos, pv.mcells[m], estride, c0, row_lo, row_hi);
  } else {
    for (int j = 0; j < pv.N; j++) {
      for (int m = tid; m < pv.n_mcells; m += blockDim.x) {
        const uint32_t c = pv.mcells[m];
        if (static_cast<int>(c >> 24) == j) patch_cell(pv, l.E, l.spos, c, estride, c0, row_lo, row_hi);
      }
      __syncthreads();
    }
  }
  __syncthreads();
}

// Fast path: uint8 observation, pixels_per_cell = 3, border_width = 1 (zones == pixels).
// A zone-table entry is then exactly 3 pixels = 9 bytes of one image row, and because the
// frame is pad_w * 3 pixels wide the table (row stride pad_w) is the image itself in
// 9-byte units: output byte o belongs to entry floor((o - shift) / 9).
__global__ __launch_bounds__(PW_RENDER_THREADS) void pw_render_u8_ppc3_kernel(RenderArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int env = blockIdx.x;
  const int tid = threadIdx.x;
  const int pid = a.puzzle_id[env];
  const PuzzleView pv = view_of(a.hdrs, a.blob, pid);

  // pixel padding of env_utils.py:75-91 (left/top get the floor half)
  const int wpx = a.pad_w * 3;
  const int pady = (a.pad_h - pv.H) * 3 / 2;
  const int padx = (a.pad_w - pv.W) * 3 / 2;
  const int c0 = (padx + 2) / 3;  // virtual cell columns left of the puzzle
  const int n_entries = 3 * pv.H * a.pad_w;
  const RenderLds l = carve_lds(smem, ((2 * (n_entries + 3) + 15) >> 4) << 4);
#ifdef PW_STAGGER
  // De-correlate the first generation of workgroups: all workgroups take equally long, so
  // without this the whole chip alternates between "everyone in the preamble" and "everyone
  // streaming" for the entire launch.
  if (blockIdx.x < PW_STAGGER) {
    const unsigned h = (blockIdx.x * 2654435761u) >> 26;
    for (unsigned i = 0; i < h; i++) __builtin_amdgcn_s_sleep(8);
  }
#endif
  build_zone_table(a, pv, pid, env, l, a.pad_w, c0);
  const uint16_t* E = l.E;
  const uint32_t* pal = l.pal;

  const int shift_bytes = 3 * (pady * wpx + padx - 3 * c0);  // may be negative by < 9 bytes per row
  // bias keeps the dividend non-negative; multiple of 9
  const int bias_q = (shift_bytes > 0 ? shift_bytes / 9 : 0) + 2;
  const int n_chunks = (a.obs_bytes + 15) >> 4;
  uint8_t* out = a.obs + static_cast<int64_t>(env) * a.env_stride;

  for (int chunk = tid; chunk < n_chunks; chunk += PW_RENDER_THREADS) {
    const unsigned o3 = static_cast<unsigned>(chunk * 16 - shift_bytes + 9 * bias_q);
    const unsigned qb = o3 / 9u;
    const int b = static_cast<int>(o3 - qb * 9u);
    int q0 = static_cast<int>(qb) - bias_q;
    q0 = min(max(q0, -3), n_entries);
    const uint32_t e0 = E[q0], e1 = E[q0 + 1], e2 = E[q0 + 2];
    // 9-byte pixel triples -> the 24 bytes d0..d5 of the concatenation
    const uint32_t r00 = pal[e0 & 15u], r01 = pal[(e0 >> 4) & 15u], r02 = pal[(e0 >> 8) & 15u];
    const uint32_t r10 = pal[e1 & 15u], r11 = pal[(e1 >> 4) & 15u], r12 = pal[(e1 >> 8) & 15u];
    const uint32_t r20 = pal[e2 & 15u], r21 = pal[(e2 >> 4) & 15u];
    const uint32_t d0 = r00 | (r01 << 24);
    const uint32_t d1 = (r01 >> 8) | (r02 << 16);
    const uint32_t d2 = (r02 >> 16) | (r10 << 8);
    const uint32_t d3 = r11 | (r12 << 24);
    const uint32_t d4 = (r12 >> 8) | (r20 << 16);
    const uint32_t d5 = (r20 >> 16) | (r21 << 8);
    // byte-granular funnel shift by b (0..8): dword select by b >> 2, then v_alignbyte by b & 3
    const int sel = b >> 2;
    const uint32_t s0 = sel == 0 ? d0 : (sel == 1 ? d1 : d2);
    const uint32_t s1 = sel == 0 ? d1 : (sel == 1 ? d2 : d3);
    const uint32_t s2 = sel == 0 ? d2 : (sel == 1 ? d3 : d4);
    const uint32_t s3 = sel == 0 ? d3 : (sel == 1 ? d4 : d5);
    const uint32_t s4 = sel == 0 ? d4 : d5;  // sel == 2 implies b == 8: byte shift 0, s4 unused
    const uint32_t bs = static_cast<uint32_t>(b & 3);
    uint4 v;
    v.x = __builtin_amdgcn_alignbyte(s1, s0, bs);
    v.y = __builtin_amdgcn_alignbyte(s2, s1, bs);
    v.z = __builtin_amdgcn_alignbyte(s3, s2, bs);
    v.w = __builtin_amdgcn_alignbyte(s4, s3, bs);
    // plain store: nt stores measured 4-7 % slower for this access pattern
    *reinterpret_cast<uint4*>(out + static_cast<int64_t>(chunk) * 16) = v;
  }
}

// geometry of the page-ordered kernels below: the observation buffer as a flat run of 16-byte chunks
struct CopyArgs {
  const uint8_t* simg;
  const int32_t* puzzle_id;
  uint8_t* obs;
  int64_t simg_stride;
  int32_t batch;
  uint32_t chunks_per_env;  // env stride / 16
  uint32_t n_chunks;        // 16-byte chunks of one observation
  float inv_cpe;            // 1 / chunks_per_env
};

// ------------------------------------------------------------------------------------
// Page render (uint8, ppc 3): ONE pass, page ordered.  A workgroup is a single wavefront that
// owns one 4 KiB page of the observation buffer (address order = dispatch order, the pattern HBM
// sustains best).  Chunks that no movable touches (98.8 % on the Level-1 mix) are copied from the
// puzzle's L2-resident static image; the wave finds the movable-cell entries that intersect its
// page (one pass over the puzzle's movable-cell list, 64 cells at a time), keeps them in an LDS
// list and recomputes only the chunks they touch from zone entries -- every byte of the
// observation is written exactly once, as part of a full 16-byte store.
// ------------------------------------------------------------------------------------
#define PW_PAGE_ENTRIES 472  // zone entries a 4 KiB page can touch (4096 / 9 + margins), multiple of 8

// geometry of one environment's image inside the frame (workgroup-uniform)
struct PageEnv {
  const PwPuzzleHeader* h;
  const uint16_t* estat;
  int W, H, N, G, n_mcells, c0, shift_bytes, n_entries;
  int lo;     // byte offset of the page start inside this environment's image (may be negative)
  int q_lo;   // first entry kept in the page's LDS entry window
  uint32_t env;
};

// ES = bytes per channel value (1: uint8, 4: float32).  Offsets inside an image are kept in BYTES (`lo`);
// entries are addressed in channel units: entry q covers channels [9 q + shift, 9 q + shift + 9).
template <int ES>
__device__ __forceinline__ PageEnv page_env(const RenderArgs& a, int pid, uint32_t env, int lo) {
  PageEnv pe;
  pe.h = a.hdrs + pid;
  pe.estat = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(a.estat) + a.estat_off[pid]);
  pe.W = pe.h->W;
  pe.H = pe.h->H;
  pe.N = pe.h->N;
  pe.G = pe.h->G;
  pe.n_mcells = static_cast<int>(pe.h->n_mcells);
  const int pady = (a.pad_h - pe.H) * 3 / 2;
  const int padx = (a.pad_w - pe.W) * 3 / 2;
  pe.c0 = (padx + 2) / 3;
  pe.shift_bytes = 3 * (pady * a.pad_w * 3 + padx - 3 * pe.c0);
  pe.n_entries = 3 * pe.H * a.pad_w;
  pe.lo = lo;
  // entry holding the first channel of the page is floor((lo / ES - shift) / 9); two entries of margin
  const int t = (lo >= 0 ? lo / ES : -((-lo) / ES)) - pe.shift_bytes;
  pe.q_lo = (t >= 0 ? t / 9 : -((-t + 8) / 9)) - 2;
  pe.env = env;
  return pe;
}

// Movable cells of the environment that reach into the page [lo, lo + 4096): their zone entries go
// into the page's LDS entry window `win` (atomicMax on (object + 1) << 12 | entry: the highest
// object index wins = painter order, puzzle.py:457) and the chunks they touch are marked in
// `dirty`.  page_prefilter() tells beforehand whether any object's rows meet the page at all.
// packed positions of an environment's movables, lane j = movable j (zero padding beyond N); needs
// no puzzle data, so it is issued before the puzzle id is known
__device__ __forceinline__ int page_load_xy(const RenderArgs& a, uint32_t env, int lane) {
  int xy = 0;
  if (lane < a.np) xy = static_cast<uint16_t>(reinterpret_cast<const int16_t*>(a.pos)[static_cast<int64_t>(env) * a.np + lane]);
  return xy;
}

// `any` = some movable's rows meet the page
template <int ES>
__device__ __forceinline__ bool page_prefilter(const RenderArgs& a, const PageEnv& pe, int lane, int xy) {
  const int row_bytes = 9 * a.pad_w;  // one image row, in channels
  const int lo = pe.lo / ES, hi = lo + 4096 / ES;  // (exact: lo is a multiple of 16, also when negative)
  bool hit = false;
  if (lane < pe.N) {
    const int y = static_cast<int8_t>((xy >> 8) & 0xff);
    const int hh = pe.h->objtab[lane].h;
    hit = 3 * y * row_bytes + pe.shift_bytes < hi && 3 * (y + hh) * row_bytes + pe.shift_bytes + 9 > lo;
  }
  return __ballot(hit) != 0ull;
}

template <int ES>
__device__ __forceinline__ void page_mark(const RenderArgs& a, const PageEnv& pe, int lane, int xy, uint32_t* win, uint32_t* dirty) {
  const int row_bytes = 9 * a.pad_w;  // one image row, in channels
  const int lo = pe.lo / ES, hi = lo + 4096 / ES;
  const uint32_t* mcells = reinterpret_cast<const uint32_t*>(a.blob + pe.h->base + pe.h->off_mcells);
  for (int m0 = 0; m0 < pe.n_mcells; m0 += PW_WAVE) {
    const int m = m0 + lane;
    const uint32_t c = m < pe.n_mcells ? mcells[m] : 0u;
    const int obj = c >> 24;
    const int p = __shfl(xy, obj, PW_WAVE);
    const int x = static_cast<int8_t>(p & 0xff) + static_cast<int>(c & 0xff);
    const int y = static_cast<int8_t>((p >> 8) & 0xff) + static_cast<int>((c >> 8) & 0xff);
    const int q0 = 3 * y * a.pad_w + x + pe.c0;
    const int b00 = 9 * q0 + pe.shift_bytes;  // first byte of the cell's top sub-row
    if (m >= pe.n_mcells || static_cast<unsigned>(x) >= static_cast<unsigned>(pe.W) ||
        static_cast<unsigned>(y) >= static_cast<unsigned>(pe.H) || b00 >= hi || b00 + 2 * row_bytes + 9 <= lo)
      continue;
    const uint32_t kind = obj == 0 ? 3u : (obj <= pe.G ? 4u : 5u);
    const uint32_t om = (c >> 16) & 0xffu;
#pragma unroll
    for (int zy = 0; zy < 3; zy++) {
      const int q = q0 + zy * a.pad_w;
      const int b0 = b00 + zy * row_bytes;
      if (b0 + 9 <= lo || b0 >= hi) continue;
      const uint32_t e = pw_zone_entry(kind, pw_zone_border_bits(om, zy), pw_entry_goal_bits(pe.estat[q]));
      atomicMax(&win[q - pe.q_lo], (static_cast<uint32_t>(obj + 1) << 12) | e);
      const int cl = (max(b0 - lo, 0) * ES) >> 4, ch = (min(b0 + 8 - lo, 4096 / ES - 1) * ES) >> 4;
      if (ES == 1) {  // 9 bytes touch at most two 16-byte chunks
        atomicOr(&dirty[cl >> 5], 1u << (cl & 31));
        atomicOr(&dirty[ch >> 5], 1u << (ch & 31));
      } else {        // 36 bytes: up to four
        for (int cc = cl; cc <= ch; cc++) atomicOr(&dirty[cc >> 5], 1u << (cc & 31));
      }
    }
  }
}

// the 16 bytes of chunk c (inside the environment's image) from the page's entry window + static table
typedef unsigned int pw_u32x4 __attribute__((ext_vector_type(4)));
// float32: chunk c = channels 4 c .. 4 c + 3 of the environment's image = parts of at most two entries;
// palf = the 16 x 4 table of exact uint8 / 255 values (env_utils.py:65-72)
__device__ __forceinline__ pw_u32x4 page_chunk_f32(const PageEnv& pe, int c, const uint32_t* win, const float* palf) {
  const int bias_q = (pe.shift_bytes > 0 ? pe.shift_bytes / 9 : 0) + 2;
  const unsigned o3 = static_cast<unsigned>(c * 4 - pe.shift_bytes + 9 * bias_q);
  const unsigned qb = o3 / 9u;
  const int r0 = static_cast<int>(o3 - qb * 9u);
  const int q0 = static_cast<int>(qb) - bias_q;
  const uint32_t w0 = win[q0 - pe.q_lo], w1 = win[q0 + 1 - pe.q_lo];
  const uint32_t e0 = w0 ? (w0 & 0xFFFu) : ((static_cast<unsigned>(q0) < static_cast<unsigned>(pe.n_entries)) ? pe.estat[q0] : 0u);
  const uint32_t e1 = w1 ? (w1 & 0xFFFu) : ((static_cast<unsigned>(q0 + 1) < static_cast<unsigned>(pe.n_entries)) ? pe.estat[q0 + 1] : 0u);
  union {
    float f[4];
    pw_u32x4 v;
  } out;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int r = r0 + k;
    const uint32_t e = r < 9 ? e0 : e1;
    const int rr = r < 9 ? r : r - 9;
    const int px = rr >= 6 ? 2 : (rr >= 3 ? 1 : 0);
    out.f[k] = palf[((e >> (4 * px)) & 15u) * 4 + (rr - 3 * px)];
  }
  return out.v;
}

__device__ __forceinline__ pw_u32x4 page_chunk(const PageEnv& pe, int c, const uint32_t* win, const uint32_t* pal) {
  const int bias_q = (pe.shift_bytes > 0 ? pe.shift_bytes / 9 : 0) + 2;
  const unsigned o3 = static_cast<unsigned>(c * 16 - pe.shift_bytes + 9 * bias_q);
  const unsigned qb = o3 / 9u;
  const int b = static_cast<int>(o3 - qb * 9u);
  const int q0 = static_cast<int>(qb) - bias_q;
  uint32_t e0, e1, e2;
  {
    const uint32_t w0 = win[q0 - pe.q_lo], w1 = win[q0 + 1 - pe.q_lo], w2 = win[q0 + 2 - pe.q_lo];
    e0 = w0 ? (w0 & 0xFFFu) : ((static_cast<unsigned>(q0) < static_cast<unsigned>(pe.n_entries)) ? pe.estat[q0] : 0u);
    e1 = w1 ? (w1 & 0xFFFu) : ((static_cast<unsigned>(q0 + 1) < static_cast<unsigned>(pe.n_entries)) ? pe.estat[q0 + 1] : 0u);
    e2 = w2 ? (w2 & 0xFFFu) : ((static_cast<unsigned>(q0 + 2) < static_cast<unsigned>(pe.n_entries)) ? pe.estat[q0 + 2] : 0u);
  }
  const uint32_t r00 = pal[e0 & 15u], r01 = pal[(e0 >> 4) & 15u], r02 = pal[(e0 >> 8) & 15u];
  const uint32_t r10 = pal[e1 & 15u], r11 = pal[(e1 >> 4) & 15u], r12 = pal[(e1 >> 8) & 15u];
  const uint32_t r20 = pal[e2 & 15u], r21 = pal[(e2 >> 4) & 15u];
  const uint32_t d0 = r00 | (r01 << 24);
  const uint32_t d1 = (r01 >> 8) | (r02 << 16);
  const uint32_t d2 = (r02 >> 16) | (r10 << 8);
  const uint32_t d3 = r11 | (r12 << 24);
  const uint32_t d4 = (r12 >> 8) | (r20 << 16);
  const uint32_t d5 = (r20 >> 16) | (r21 << 8);
  const int sel = b >> 2;
  const uint32_t s0 = sel == 0 ? d0 : (sel == 1 ? d1 : d2);
  const uint32_t s1 = sel == 0 ? d1 : (sel == 1 ? d2 : d3);
  const uint32_t s2 = sel == 0 ? d2 : (sel == 1 ? d3 : d4);
  const uint32_t s3 = sel == 0 ? d3 : (sel == 1 ? d4 : d5);
  const uint32_t s4 = sel == 0 ? d4 : d5;
  const uint32_t bs = static_cast<uint32_t>(b & 3);
  return pw_u32x4{__builtin_amdgcn_alignbyte(s1, s0, bs), __builtin_amdgcn_alignbyte(s2, s1, bs),
                  __builtin_amdgcn_alignbyte(s3, s2, bs), __builtin_amdgcn_alignbyte(s4, s3, bs)};
}

#define PAGE_CHUNK(...) (ES == 1 ? page_chunk(__VA_ARGS__, pal) : page_chunk_f32(__VA_ARGS__, reinterpret_cast<const float*>(pal)))
template <typename T>
__global__ __launch_bounds__(64) void pw_render_page_kernel(RenderArgs a, CopyArgs ca) {
  typedef pw_u32x4 u32x4;
  constexpr int ES = static_cast<int>(sizeof(T));
  __shared__ uint32_t pal[ES == 1 ? 16 : 64];  // uint8: packed RGB; float32: 16 x 4 float bit patterns
  __shared__ uint32_t dirty[8];
  __shared__ __align__(16) uint32_t win0[PW_PAGE_ENTRIES];
  __shared__ __align__(16) uint32_t win1[PW_PAGE_ENTRIES];
  const int lane = threadIdx.x;

  // ---- workgroup-uniform bookkeeping -------------------------------------------------------------
  const uint32_t g0 = blockIdx.x * 256u;  // first 16-byte chunk of the page
  const uint32_t cpe = ca.chunks_per_env;
  // g0 / cpe without an integer division: float estimate (exact to +-1 for g0 < 2^31) + correction
  uint32_t env0 = static_cast<uint32_t>(static_cast<float>(g0) * ca.inv_cpe);
  {
    const int rr = static_cast<int>(g0 - env0 * cpe);
    if (rr < 0) env0 -= 1u;
    else if (rr >= static_cast<int>(cpe)) env0 += 1u;
  }
  env0 = __builtin_amdgcn_readfirstlane(env0);
  const uint32_t last = static_cast<uint32_t>(a.batch) - 1u;
  const int c_first = static_cast<int>(g0 - env0 * cpe);  // chunk index of the page start inside env0
  const int xy0 = page_load_xy(a, env0, lane);             // independent of the puzzle: issued first
  const int pid0 = a.puzzle_id[env0];
  uint8_t* dst = a.obs + static_cast<int64_t>(g0) * 16 + lane * 16;

  if (c_first + 256 <= static_cast<int>(ca.n_chunks)) {
    // ---- 13 of 14 pages: the whole page lies inside one environment's image ----------------------
    const uint8_t* src = ca.simg + static_cast<int64_t>(pid0) * ca.simg_stride + static_cast<int64_t>(c_first) * 16 + lane * 16;
    u32x4 v0 = *reinterpret_cast<const u32x4*>(src);
    u32x4 v1 = *reinterpret_cast<const u32x4*>(src + 1024);
    u32x4 v2 = *reinterpret_cast<const u32x4*>(src + 2048);
    u32x4 v3 = *reinterpret_cast<const u32x4*>(src + 3072);
    const PageEnv pe = page_env<ES>(a, pid0, env0, c_first * 16);
    const int xy = xy0;
    if (page_prefilter<ES>(a, pe, lane, xy)) {  // ~40 % of the pages: some movable's rows cross this page
      if (ES == 1) {
        if (lane < 16) pal[lane] = a.pal_rgb[lane];
      } else {
        pal[lane] = __float_as_uint(a.pal_f32[lane >> 2][lane & 3]);
      }
      if (lane < 8) dirty[lane] = 0;
      for (int i = lane; i < PW_PAGE_ENTRIES / 4; i += PW_WAVE) reinterpret_cast<uint4*>(win0)[i] = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
      page_mark<ES>(a, pe, lane, xy, win0, dirty);
      __syncthreads();
      if (dirty[0] | dirty[1] | dirty[2] | dirty[3] | dirty[4] | dirty[5] | dirty[6] | dirty[7]) {
        const int bit = lane & 31, wsel = lane >> 5;
        if ((dirty[0 + wsel] >> bit) & 1u) v0 = PAGE_CHUNK(pe, c_first + lane, win0);
        if ((dirty[2 + wsel] >> bit) & 1u) v1 = PAGE_CHUNK(pe, c_first + lane + 64, win0);
        if ((dirty[4 + wsel] >> bit) & 1u) v2 = PAGE_CHUNK(pe, c_first + lane + 128, win0);
        if ((dirty[6 + wsel] >> bit) & 1u) v3 = PAGE_CHUNK(pe, c_first + lane + 192, win0);
      }
    }
    __builtin_nontemporal_store(v0, reinterpret_cast<u32x4*>(dst));
    __builtin_nontemporal_store(v1, reinterpret_cast<u32x4*>(dst + 1024));
    __builtin_nontemporal_store(v2, reinterpret_cast<u32x4*>(dst + 2048));
    __builtin_nontemporal_store(v3, reinterpret_cast<u32x4*>(dst + 3072));
    return;
  }

  // ---- the page holds the tail of env0 (and usually the head of env0 + 1) ------------------------
  // Mostly bottom / top padding rows: same shape as above (4 loads in flight, LDS work only when a
  // movable's rows reach the page).
  const int split = static_cast<int>(cpe) - c_first;  // local chunk where the next environment starts
  const bool has_second = split < 256 && env0 + 1u <= last;
  const uint32_t env1 = min(env0 + 1u, last);
  const int xy1 = page_load_xy(a, env1, lane);
  const int pid1 = a.puzzle_id[env1];
  const int valid0 = static_cast<int>(ca.n_chunks) - c_first;  // local chunks [0, valid0) of env0 carry image bytes
  const uint8_t* src0 = ca.simg + static_cast<int64_t>(pid0) * ca.simg_stride + static_cast<int64_t>(c_first) * 16;
  const uint8_t* src1 = ca.simg + static_cast<int64_t>(pid1) * ca.simg_stride - static_cast<int64_t>(split) * 16;
  u32x4 v[4];
  bool ok[4], second[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int lc = lane + 64 * k;
    second[k] = lc >= split;
    ok[k] = second[k] ? (has_second && lc - split < static_cast<int>(ca.n_chunks)) : (lc < valid0);
    v[k] = u32x4{0u, 0u, 0u, 0u};
    if (ok[k]) v[k] = *reinterpret_cast<const u32x4*>((second[k] ? src1 : src0) + lc * 16);
  }
  const PageEnv pe0 = page_env<ES>(a, pid0, env0, c_first * 16);
  const PageEnv pe1 = page_env<ES>(a, pid1, env0 + 1u, -split * 16);
  const bool hit0 = page_prefilter<ES>(a, pe0, lane, xy0);
  const bool hit1 = has_second && page_prefilter<ES>(a, pe1, lane, xy1);
  if (hit0 || hit1) {
    if (ES == 1) {
        if (lane < 16) pal[lane] = a.pal_rgb[lane];
      } else {
        pal[lane] = __float_as_uint(a.pal_f32[lane >> 2][lane & 3]);
      }
    if (lane < 8) dirty[lane] = 0;
    for (int i = lane; i < PW_PAGE_ENTRIES / 4; i += PW_WAVE) {
      reinterpret_cast<uint4*>(win0)[i] = make_uint4(0u, 0u, 0u, 0u);
      reinterpret_cast<uint4*>(win1)[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    if (hit0) page_mark<ES>(a, pe0, lane, xy0, win0, dirty);
    if (hit1) page_mark<ES>(a, pe1, lane, xy1, win1, dirty);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int lc = lane + 64 * k;
      if (ok[k] && ((dirty[lc >> 5] >> (lc & 31)) & 1u))
        v[k] = second[k] ? PAGE_CHUNK(pe1, lc - split, win1) : PAGE_CHUNK(pe0, c_first + lc, win0);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (ok[k]) __builtin_nontemporal_store(v[k], reinterpret_cast<u32x4*>(dst + k * 1024));
}

// ------------------------------------------------------------------------------------
// Incremental render (uint8, ppc 3; pw_step_render_delta): the observation buffer already holds the
// observation of the state BEFORE the step, so only the pixel rows swept by the objects that moved
// (their old and new cells) change -- on the Level-1 mix ~5 % of an image, nothing at all for a
// blocked move.  One wavefront per environment reads the row interval the step kernel left in
// `dirty`, and rewrites that byte range with the page kernel's machinery (static image + LDS entry
// window over ALL movables that reach into the segment), 4 KiB at a time.  Environments that were
// reset carry the full interval and are redrawn completely.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void pw_render_delta_kernel(RenderArgs a, CopyArgs ca, const uint32_t* dirty_rows) {
  typedef pw_u32x4 u32x4;
  constexpr int ES = static_cast<int>(sizeof(T));
  __shared__ uint32_t pal[ES == 1 ? 16 : 64];
  __shared__ uint32_t dirty[8];
  __shared__ __align__(16) uint32_t win0[PW_PAGE_ENTRIES];
  const int lane = threadIdx.x;
  const uint32_t env = blockIdx.x;
  // first level of loads, all independent: the record (rows + puzzle height), positions, puzzle id
  const uint32_t d = dirty_rows[env];
  const int xy = page_load_xy(a, env, lane);
  const int pid = a.puzzle_id[env];
  const int ylo = static_cast<int>(d & 0xffu), yhi_raw = static_cast<int>((d >> 8) & 0xffu);
  if (yhi_raw <= ylo) return;  // nothing moved: the buffer is already right
  const int H = static_cast<int>(d >> 16);  // from the record: the chunk range needs no header load
  const int pady = (a.pad_h - H) * 3 / 2;
  const int row_bytes = 9 * a.pad_w * ES;
  const int yhi = min(yhi_raw, H);
  // a reset environment (interval 0 .. PW_MAX_DIM) may have changed puzzle: the whole frame, padding included
  const bool whole = yhi_raw >= PW_MAX_DIM;
  const int c_lo = whole ? 0 : ((pady + 3 * ylo) * row_bytes) >> 4;
  const int c_hi = whole ? static_cast<int>(ca.n_chunks)
                         : min(((pady + 3 * yhi) * row_bytes + 15) >> 4, static_cast<int>(ca.n_chunks));
  uint8_t* dst = a.obs + static_cast<int64_t>(env) * a.env_stride;
  const uint8_t* src = ca.simg + static_cast<int64_t>(pid) * ca.simg_stride;
  if (ES == 1) {
    if (lane < 16) pal[lane] = a.pal_rgb[lane];
  } else {
    pal[lane] = __float_as_uint(a.pal_f32[lane >> 2][lane & 3]);
  }
  for (int c0 = c_lo; c0 < c_hi; c0 += 256) {
    const PageEnv pe = page_env<ES>(a, pid, env, c0 * 16);
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = c0 + lane + 64 * k;
      v[k] = u32x4{0u, 0u, 0u, 0u};
      if (c < c_hi) v[k] = *reinterpret_cast<const u32x4*>(src + static_cast<int64_t>(c) * 16);
    }
    __syncthreads();  // the previous segment's window is no longer read
    if (page_prefilter<ES>(a, pe, lane, xy)) {
      if (lane < 8) dirty[lane] = 0;
      for (int i = lane; i < PW_PAGE_ENTRIES / 4; i += PW_WAVE) reinterpret_cast<uint4*>(win0)[i] = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
      page_mark<ES>(a, pe, lane, xy, win0, dirty);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int lc = lane + 64 * k;
        if (c0 + lc < c_hi && ((dirty[lc >> 5] >> (lc & 31)) & 1u)) v[k] = PAGE_CHUNK(pe, c0 + lc, win0);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = c0 + lane + 64 * k;
      if (c < c_hi) *reinterpret_cast<u32x4*>(dst + static_cast<int64_t>(c) * 16) = v[k];
    }
  }
}

#undef PAGE_CHUNK

// Generic path: any pixels_per_cell / border_width, uint8 or float32 elements.
// One thread produces 16 bytes (16 uint8 or 4 float32 channel values) per iteration.
template <typename T>
__global__ __launch_bounds__(PW_RENDER_THREADS) void pw_render_generic_kernel(RenderArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int env = blockIdx.x;
  const int tid = threadIdx.x;
  // incremental redraw (pw_step_render_delta): only the cell rows the step changed, nothing for a blocked
  // move, everything (frame padding included) for an environment that was reset
  int row_lo = 0, row_hi = PW_MAX_DIM;
  bool whole = true;
  if (a.dirty_rows) {
    const uint32_t d = a.dirty_rows[env];
    row_lo = static_cast<int>(d & 0xffu);
    row_hi = static_cast<int>((d >> 8) & 0xffu);
    if (row_hi <= row_lo) return;
    whole = row_hi >= PW_MAX_DIM;
  }
  const int pid = a.puzzle_id[env];
  const PuzzleView pv = view_of(a.hdrs, a.blob, pid);
  const int n_entries = 3 * pv.H * pv.W;
  const RenderLds l = carve_lds(smem, ((2 * (n_entries + 3) + 15) >> 4) << 4);
  // a 16-byte chunk at the edge of the range reaches into the neighbouring pixel row: one more cell row each side
  build_zone_table(a, pv, pid, env, l, pv.W, 0, whole ? 0 : row_lo - 1, whole ? PW_MAX_DIM : row_hi + 1);
  const uint16_t* E = l.E;
  const uint32_t* pal = l.pal;

  constexpr int kElems = 16 / sizeof(T);
  const int ppc = a.ppc, bw = a.bw;
  const int wpx = a.pad_w * ppc;
  const int own_w = pv.W * ppc, own_h = pv.H * ppc;
  const int pady = (a.pad_h * ppc - own_h) / 2;
  const int padx = (wpx - own_w) / 2;
  const int n_chunks = (a.obs_bytes + 15) >> 4;
  uint8_t* out = a.obs + static_cast<int64_t>(env) * a.env_stride;

  // float32: the palette as exact uint8/255 values, staged next to the uint8 palette in LDS
  float* palf = reinterpret_cast<float*>(l.flag + 4);
  if (sizeof(T) == 4 && tid < 64) palf[tid] = a.pal_f32[tid >> 2][tid & 3];
  __syncthreads();
  constexpr int kPix = sizeof(T) == 1 ? 7 : 2;  // pixels a 16-byte chunk can touch
  int chunk_lo = 0, chunk_hi = n_chunks;
  if (!whole) {
    const int64_t row_b = static_cast<int64_t>(wpx) * 3 * static_cast<int>(sizeof(T));
    chunk_lo = static_cast<int>(((pady + row_lo * ppc) * row_b) >> 4);
    chunk_hi = min(n_chunks, static_cast<int>(((pady + min(row_hi, pv.H) * ppc) * row_b + 15) >> 4));
  }
  for (int chunk = chunk_lo + tid; chunk < chunk_hi; chunk += PW_RENDER_THREADS) {
    const int elem0 = chunk * kElems;
    const int pix = elem0 / 3;
    const int ch0 = elem0 - pix * 3;  // channel of the first element
    // position of the first pixel: three divisions per chunk, then the walk below is incremental
    int Y = pix / wpx;
    int X = pix - Y * wpx;
    int yp = Y - pady, xp = X - padx;
    bool in_row = static_cast<unsigned>(yp) < static_cast<unsigned>(own_h);
    int cy = in_row ? yp / ppc : 0;
    int sy = yp - cy * ppc;
    int zy = sy < bw ? 0 : (sy >= ppc - bw ? 2 : 1);
    int cx = xp >= 0 ? xp / ppc : 0;
    int sx = xp - cx * ppc;  // negative while left of the puzzle
    uint32_t rgb[kPix];
#pragma unroll
    for (int j = 0; j < kPix; j++) {
      uint32_t col = PW_C_PAD;
      if (in_row && static_cast<unsigned>(xp) < static_cast<unsigned>(own_w)) {
        const int zx = sx < bw ? 0 : (sx >= ppc - bw ? 2 : 1);
        col = (E[(3 * cy + zy) * pv.W + cx] >> (4 * zx)) & 15u;
      }
      rgb[j] = sizeof(T) == 1 ? pal[col] : col;
      // next pixel
      xp++;
      if (++sx == ppc) {  // sx < 0 while left of the puzzle: reaches 0 together with xp
        sx = 0;
        cx++;
      }
      if (xp == wpx - padx) {  // row wrap (X == wpx)
        xp = -padx;
        cx = 0;
        sx = -padx;
        yp++;
        in_row = static_cast<unsigned>(yp) < static_cast<unsigned>(own_h);
        if (in_row && ++sy == ppc) {
          sy = 0;
          cy++;
        }
        if (yp == 0) {
          sy = 0;
          cy = 0;
        }
        zy = sy < bw ? 0 : (sy >= ppc - bw ? 2 : 1);
      }
    }
    union {
      uint4 v;
      uint32_t u32[4];
      float f32[4];
    } o;
    if (sizeof(T) == 1) {
      // 7 pixels = 21 bytes, the chunk is bytes [ch0, ch0 + 16) of them
      const uint32_t d0 = rgb[0] | (rgb[1] << 24);
      const uint32_t d1 = (rgb[1] >> 8) | (rgb[2] << 16);
      const uint32_t d2 = (rgb[2] >> 16) | (rgb[3] << 8);
      const uint32_t d3 = rgb[4] | (rgb[5] << 24);
      const uint32_t d4 = (rgb[5] >> 8) | (rgb[6 < kPix ? 6 : 0] << 16);
      const uint32_t bs = static_cast<uint32_t>(ch0);
      o.u32[0] = __builtin_amdgcn_alignbyte(d1, d0, bs);
      o.u32[1] = __builtin_amdgcn_alignbyte(d2, d1, bs);
      o.u32[2] = __builtin_amdgcn_alignbyte(d3, d2, bs);
      o.u32[3] = __builtin_amdgcn_alignbyte(d4, d3, bs);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int e = ch0 + k;  // element k belongs to pixel e / 3, channel e % 3
        o.f32[k] = palf[rgb[e >= 3 ? 1 : 0] * 4 + (e >= 3 ? e - 3 : e)];
      }
    }
    *reinterpret_cast<uint4*>(out + static_cast<int64_t>(chunk) * 16) = o.v;
  }
}

// ------------------------------------------------------------------------------------
// engine half of the C ABI
// ------------------------------------------------------------------------------------
namespace {

const uint8_t kPaletteRGB[PW_NUM_COLORS][3] = {
    {0, 0, 0},           // PW_C_PAD
    {255, 255, 255},     // PW_C_BACKGROUND      puzzle.py:451
    {0xFA, 0xC7, 0x1E},  // AGENT_WALL           puzzle.py:70
    {0x7D, 0x64, 0x0F},  // AGENT_WALL_BORDER    :71
    {0x0A, 0x0A, 0x0A},  // WALL                 :78
    {0x05, 0x05, 0x05},  // WALL_BORDER          :79
    {0x00, 0xDC, 0x00},  // AGENT                :68
    {0x00, 0x6E, 0x00},  // AGENT_BORDER         :69
    {0xDC, 0x00, 0x00},  // GOAL_OBJECT          :74
    {0x6E, 0x00, 0x00},  // GOAL_OBJECT_BORDER   :75
    {0x46, 0x9B, 0xFF},  // MOVABLE              :76
    {0x23, 0x48, 0x7F},  // MOVABLE_BORDER       :77
    {0xB9, 0x00, 0x00},  // GOAL_BORDER          :73
};

int check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pw_fail(PW_EDEVICE, std::string(what) + ": " + hipGetErrorString(err));
  return PW_OK;
}

int fill_render_args(const PwEngine* e, const int32_t* puzzle_id, const int8_t* pos, void* obs, int64_t stride,
                     int32_t batch, RenderArgs* ra) {
  if (!puzzle_id || !pos || !obs) return pw_fail(PW_EINVAL, "null device pointer");
  if (stride < e->obs_bytes || (stride & 15)) return pw_fail(PW_EINVAL, "env_stride_bytes must be >= obs bytes rounded up to 16 and a multiple of 16");
  if (stride < ((e->obs_bytes + 15) & ~int64_t(15))) return pw_fail(PW_EINVAL, "env_stride_bytes too small");
  if (reinterpret_cast<uintptr_t>(obs) & 15) return pw_fail(PW_EINVAL, "obs must be 16-byte aligned");
  ra->hdrs = e->set->d_headers;
  ra->blob = e->set->d_blob;
  ra->puzzle_id = puzzle_id;
  ra->pos = pos;
  ra->obs = static_cast<uint8_t*>(obs);
  ra->env_stride = stride;
  ra->batch = batch;
  ra->np = e->np;
  ra->ppc = e->cfg.pixels_per_cell;
  ra->bw = e->cfg.border_width;
  ra->pad_h = e->pad_h;
  ra->pad_w = e->pad_w;
  ra->estat = e->d_estat;
  ra->estat_off = e->d_estat_off;
  ra->obs_bytes = static_cast<int32_t>(e->obs_bytes);
  ra->do_step = 0;
  ra->skip_movables = 0;
  ra->dirty_rows = nullptr;
  for (int i = 0; i < 16; i++) {
    ra->pal_rgb[i] = e->pal_rgb[i];
    for (int c = 0; c < 4; c++) ra->pal_f32[i][c] = e->pal_f32[i][c];
  }
  return PW_OK;
}

}  // namespace

extern "C" {

int pw_engine_create(const PwPuzzleSet* s, const PwEngineConfig* cfg, PwEngine** out) {
  if (!s || !cfg || !out) return pw_fail(PW_EINVAL, "null argument");
  if (s->device < 0 || !s->d_blob) return pw_fail(PW_EDEVICE, "puzzle set has no device tables (created with device < 0)");
  if (cfg->border_width < 1) return pw_fail(PW_EINVAL, "border_width must be >= 1");
  if (cfg->pixels_per_cell < 3) return pw_fail(PW_EINVAL, "pixels_per_cell must be >= 3");
  if (cfg->pixels_per_cell < 1 + 2 * cfg->border_width)
    return pw_fail(PW_EINVAL, "pixels_per_cell must be >= 1 + 2*border_width");
  if (cfg->obs_dtype != PW_OBS_U8 && cfg->obs_dtype != PW_OBS_F32) return pw_fail(PW_EINVAL, "bad obs_dtype");
  PwEngine* e = new (std::nothrow) PwEngine();
  if (!e) return pw_fail(PW_ENOMEM, "out of memory");
  e->set = s;
  e->cfg = *cfg;
  e->np = s->max_n <= 4 ? 4 : (s->max_n <= 8 ? 8 : (s->max_n <= 16 ? 16 : 32));
  e->pad_h = cfg->pad_cell_height > 0 ? cfg->pad_cell_height : s->max_h;
  e->pad_w = cfg->pad_cell_width > 0 ? cfg->pad_cell_width : s->max_w;
  if (e->pad_h < s->max_h || e->pad_w < s->max_w) {
    delete e;
    return pw_fail(PW_EINVAL, "observation frame is smaller than the largest puzzle in the set");
  }
  e->obs_h = e->pad_h * cfg->pixels_per_cell;
  e->obs_w = e->pad_w * cfg->pixels_per_cell;
  e->obs_bytes = static_cast<int64_t>(e->obs_h) * e->obs_w * 3 * (cfg->obs_dtype == PW_OBS_F32 ? 4 : 1);
  if (e->obs_bytes > (int64_t(1) << 30)) {
    delete e;
    return pw_fail(PW_ELIMIT, "observation larger than 1 GiB");
  }
  e->fast_u8_ppc3 = cfg->obs_dtype == PW_OBS_U8 && cfg->pixels_per_cell == 3 && cfg->border_width == 1;
  e->page_f32 = cfg->obs_dtype == PW_OBS_F32 && cfg->pixels_per_cell == 3 && cfg->border_width == 1;
  e->d_estat_page = nullptr;
  e->d_estat_page_off = nullptr;
  e->d_estat = nullptr;
  e->d_estat_off = nullptr;
  {
    const char* sel = getenv("PUSHWORLD_AMD_STEP");
    e->step_kernel = (sel && std::string(sel) == "wave") ? 1 : ((sel && std::string(sel) == "lane") ? 2 : 0);
    const char* ff = getenv("PUSHWORLD_AMD_FUSED");
    e->force_fused = ff && std::string(ff) == "1";
  }
  // Static zone-colour tables (walls, agent walls, background, goal outlines) of every puzzle in
  // the layout the render kernel of this engine streams from: row stride pad_w with the puzzle
  // shifted by c0 virtual columns for the 3-pixel fast path, row stride W otherwise.
  std::vector<uint16_t> estat, estat_page;
  std::vector<uint32_t> estat_off(s->count), estat_page_off(s->count);
  size_t max_e_bytes = 0;
  auto build_tables = [&](bool frame_layout, std::vector<uint16_t>& tab, std::vector<uint32_t>& off, size_t* max_bytes) {
    for (int p = 0; p < s->count; p++) {
      const PwPuzzleHeader& h = s->headers[p];
      const int W = h.W, H = h.H;
      const int estride = frame_layout ? e->pad_w : W;
      const int c0 = frame_layout ? ((e->pad_w - W) * 3 / 2 + 2) / 3 : 0;
      const uint32_t* codes = reinterpret_cast<const uint32_t*>(s->blob.data() + h.base + h.off_static);
      const size_t n_entries = static_cast<size_t>(3) * H * estride;
      const size_t e_bytes = ((2 * (n_entries + 3) + 15) >> 4) << 4;
      if (max_bytes) *max_bytes = std::max(*max_bytes, e_bytes);
      off[p] = static_cast<uint32_t>(tab.size() * 2);
      const size_t first = tab.size();
      tab.resize(first + e_bytes / 2, 0);
      for (int cy = 0; cy < H; cy++)
        for (int cx = 0; cx < W; cx++) {
          const uint32_t code = codes[cy * W + cx];
          const uint32_t kind = (code >> PW_CODE_KIND_SHIFT) & 0xfu;
          const uint32_t om = code & 0xffu, gm = code >> PW_CODE_GOAL_SHIFT;
          for (int zy = 0; zy < 3; zy++)
            tab[first + static_cast<size_t>(3 * cy + zy) * estride + cx + c0] =
                static_cast<uint16_t>(pw_zone_entry(kind, pw_zone_border_bits(om, zy), pw_zone_border_bits(gm, zy)));
        }
    }
  };
  build_tables(e->fast_u8_ppc3, estat, estat_off, &max_e_bytes);
  if (e->page_f32) build_tables(true, estat_page, estat_page_off, nullptr);
  e->render_lds = 16 + max_e_bytes + 64 + 64 + 16 + 256;  // guard, E, spos, pal, flag, float palette
  for (int i = 0; i < 16; i++) {
    e->pal_rgb[i] = 0;
    for (int c = 0; c < 4; c++) e->pal_f32[i][c] = 0.0f;
  }
  for (int i = 0; i < PW_NUM_COLORS; i++) {
    e->pal_rgb[i] = kPaletteRGB[i][0] | (kPaletteRGB[i][1] << 8) | (kPaletteRGB[i][2] << 16);
    for (int c = 0; c < 3; c++) {
      // env_utils.py:65-72: uint8.astype(float32) / 255 -- one correctly rounded binary32 division
      volatile float num = static_cast<float>(kPaletteRGB[i][c]);
      volatile float den = 255.0f;
      e->pal_f32[i][c] = num / den;
    }
  }
  hipError_t err = hipSetDevice(s->device);
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&e->d_estat), estat.size() * 2);
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&e->d_estat_off), estat_off.size() * 4);
  if (err == hipSuccess) err = hipMemcpy(e->d_estat, estat.data(), estat.size() * 2, hipMemcpyHostToDevice);
  if (err == hipSuccess)
    err = hipMemcpy(e->d_estat_off, estat_off.data(), estat_off.size() * 4, hipMemcpyHostToDevice);
  if (e->page_f32) {
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&e->d_estat_page), estat_page.size() * 2);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&e->d_estat_page_off), estat_page_off.size() * 4);
    if (err == hipSuccess)
      err = hipMemcpy(e->d_estat_page, estat_page.data(), estat_page.size() * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess)
      err = hipMemcpy(e->d_estat_page_off, estat_page_off.data(), estat_page_off.size() * 4, hipMemcpyHostToDevice);
  }
  if (err == hipSuccess)
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(pw_render_u8_ppc3_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(e->render_lds));
  if (err == hipSuccess)
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(pw_render_generic_kernel<uint8_t>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(e->render_lds));
  if (err == hipSuccess)
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(pw_render_generic_kernel<float>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(e->render_lds));
  // static images of all puzzles for the page-ordered kernels, drawn once by the LDS kernel itself
  e->d_simg = nullptr;
  e->simg_stride = (e->obs_bytes + 255) & ~int64_t(255);
  const char* rsel = getenv("PUSHWORLD_AMD_RENDER");
  e->d_dirty = nullptr;
  e->dirty_cap = 0;
  // Default for uint8 / ppc 3: the page-ordered kernel (static-image copy + LDS entry window), as long
  // as the static images of the whole puzzle set stay cache resident (64 MB; they are read once per
  // observation).  PUSHWORLD_AMD_RENDER=lds forces the per-environment LDS kernel.
  const bool want_lds = rsel && std::string(rsel) == "lds";
  // The delta kernel reads only the rows a step changed, so for it the images may live in HBM (<= 4 GB).
  const int64_t simg_total = static_cast<int64_t>(s->count) * e->simg_stride;
  e->simg_cached = !want_lds && simg_total <= (int64_t(64) << 20);
  const bool want_page = (e->fast_u8_ppc3 || e->page_f32) && simg_total <= (int64_t(4) << 30);
  if (err == hipSuccess && want_page) {
    int32_t* d_ids = nullptr;
    int8_t* d_pos = nullptr;
    std::vector<int32_t> ids(s->count);
    for (int p = 0; p < s->count; p++) ids[p] = p;
    err = hipMalloc(reinterpret_cast<void**>(&e->d_simg), static_cast<size_t>(s->count) * e->simg_stride);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&d_ids), ids.size() * 4);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&d_pos), static_cast<size_t>(s->count) * e->np * 2);
    if (err == hipSuccess) err = hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemset(d_pos, 0, static_cast<size_t>(s->count) * e->np * 2);
    if (err == hipSuccess) {
      RenderArgs ra;
      uint8_t* simg = e->d_simg;
      e->d_simg = nullptr;  // fill_render_args / launch_render must take the LDS path here
      if (fill_render_args(e, d_ids, d_pos, simg, e->simg_stride, s->count, &ra) == PW_OK) {
        ra.skip_movables = 1;
        launch_render(e, ra, s->count, nullptr);
        err = hipGetLastError();
        if (err == hipSuccess) err = hipDeviceSynchronize();
      } else {
        err = hipErrorInvalidValue;
      }
      e->d_simg = simg;
    }
    if (d_ids) (void)hipFree(d_ids);
    if (d_pos) (void)hipFree(d_pos);
  }
  if (err != hipSuccess) {
    std::string msg = std::string("engine setup failed: ") + hipGetErrorString(err);
    pw_engine_destroy(e);
    return pw_fail(PW_EDEVICE, msg);
  }
  *out = e;
  return PW_OK;
}

void pw_engine_destroy(PwEngine* e) {
  if (!e) return;
  if (e->d_estat) (void)hipFree(e->d_estat);
  if (e->d_estat_off) (void)hipFree(e->d_estat_off);
  if (e->d_estat_page) (void)hipFree(e->d_estat_page);
  if (e->d_estat_page_off) (void)hipFree(e->d_estat_page_off);
  if (e->d_simg) (void)hipFree(e->d_simg);
  if (e->d_dirty) (void)hipFree(e->d_dirty);
  delete e;
}

int pw_engine_npad(const PwEngine* e) { return e ? e->np : pw_fail(PW_EINVAL, "null engine"); }

int pw_engine_obs_shape(const PwEngine* e, int* h, int* w, int* c) {
  if (!e) return pw_fail(PW_EINVAL, "null engine");
  if (h) *h = e->obs_h;
  if (w) *w = e->obs_w;
  if (c) *c = 3;
  return PW_OK;
}

int pw_engine_render_kernel(const PwEngine* e, char* buf, int cap) {
  if (!e) return pw_fail(PW_EINVAL, "null engine");
  const char* name = "pw_render_generic_kernel";
  if (e->fast_u8_ppc3) {
    if (!e->d_simg || !e->simg_cached) name = "pw_render_u8_ppc3_kernel";
    else name = "pw_render_page_kernel";
  } else if (e->page_f32 && e->d_simg && e->simg_cached) {
    name = "pw_render_page_kernel";
  }
  const int n = static_cast<int>(strlen(name));
  if (buf && cap > 0) {
    const int m = n < cap - 1 ? n : cap - 1;
    memcpy(buf, name, m);
    buf[m] = 0;
  }
  return n;
}

int64_t pw_engine_obs_bytes(const PwEngine* e) { return e ? e->obs_bytes : pw_fail(PW_EINVAL, "null engine"); }

int64_t pw_engine_obs_stride(const PwEngine* e) {
  if (!e) return pw_fail(PW_EINVAL, "null engine");
  // multiple of 512 B: 32 chunks of 16 B, so a 4 KiB page of the buffer starts on a word boundary of
  // an environment's dirty-chunk bitmap (overlay render path); 1 KiB / 4 KiB measured no faster
  return (e->obs_bytes + 511) & ~int64_t(511);
}

int pw_reset(PwEngine* e, const int32_t* puzzle_id, const uint8_t* mask, int8_t* pos, int32_t* steps,
             uint8_t* terminated, uint8_t* truncated, int32_t batch, void* stream) {
  if (!e || !puzzle_id || !pos || !steps) return pw_fail(PW_EINVAL, "null argument");
  if (batch <= 0) return PW_OK;
  ResetArgs a{e->set->d_headers, e->set->d_blob, puzzle_id, mask, pos, steps, terminated, truncated, batch, e->np};
  const int64_t threads = static_cast<int64_t>(batch) * e->np;
  const unsigned blocks = static_cast<unsigned>((threads + 255) / 256);
  hipLaunchKernelGGL(pw_reset_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch("pw_reset");
}

uint64_t pw_mix64(uint64_t seed, uint64_t env, uint64_t episode) { return mix64(seed, env, episode); }

int pw_resample(PwEngine* e, int32_t* puzzle_id, const uint8_t* terminated, const uint8_t* truncated,
                const int32_t* table, int32_t table_len, uint64_t seed, uint32_t* episode, int32_t batch,
                void* stream) {
  if (!e || !puzzle_id || !episode) return pw_fail(PW_EINVAL, "null argument");
  const int n = e->set->count;
  if (table ? table_len <= 0 : (table_len != 0 && table_len != n))
    return pw_fail(PW_EINVAL, "pw_resample: table_len must be > 0 with a table, 0 or the set size without");
  if (batch <= 0) return PW_OK;
  ResampleArgs a{puzzle_id, terminated, truncated, table, episode, seed, table ? table_len : n, batch};
  hipLaunchKernelGGL(pw_resample_kernel, dim3((batch + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch("pw_resample");
}

static int fill_step_args(PwEngine* e, const int32_t* puzzle_id, const uint8_t* actions, int8_t* pos, int32_t* steps,
                          double* reward, int8_t* dgoals, uint8_t* terminated, uint8_t* truncated, int32_t batch,
                          uint32_t flags, StepArgs* a) {
  if (!e || !puzzle_id || !actions || !pos || !steps || !terminated || !truncated)
    return pw_fail(PW_EINVAL, "null argument");
  *a = StepArgs{e->set->d_headers, e->set->d_blob, puzzle_id, actions, pos, steps, reward, dgoals,
                terminated, truncated, batch, e->cfg.max_steps, flags, e->np, nullptr};
  return PW_OK;
}

static void launch_render(PwEngine* e, const RenderArgs& ra, int32_t batch, hipStream_t st) {
  // page kernels: a 4 KiB page may hold the tail of one environment and the head of the next, not
  // more -- observations smaller than a page take the per-environment LDS kernel
  if (e->d_simg && e->simg_cached && !ra.do_step && !ra.skip_movables && ra.env_stride >= 4096) {
    CopyArgs ca;
    ca.simg = e->d_simg;
    ca.puzzle_id = ra.puzzle_id;
    ca.obs = ra.obs;
    ca.simg_stride = e->simg_stride;
    ca.batch = batch;
    ca.chunks_per_env = static_cast<uint32_t>(ra.env_stride / 16);
    ca.n_chunks = static_cast<uint32_t>((e->obs_bytes + 15) / 16);
    ca.inv_cpe = 1.0f / static_cast<float>(ca.chunks_per_env);
    const uint64_t total = static_cast<uint64_t>(batch) * ca.chunks_per_env;
    const dim3 pgrid(static_cast<unsigned>((total + 255) / 256));
    if (e->page_f32) {
      RenderArgs rp = ra;  // the page kernels index the frame-layout zone table
      rp.estat = e->d_estat_page;
      rp.estat_off = e->d_estat_page_off;
      hipLaunchKernelGGL(pw_render_page_kernel<float>, pgrid, dim3(64), 0, st, rp, ca);
    } else {
      hipLaunchKernelGGL(pw_render_page_kernel<uint8_t>, pgrid, dim3(64), 0, st, ra, ca);
    }
    return;
  }
  const dim3 grid(static_cast<unsigned>(batch)), block(PW_RENDER_THREADS);
  if (e->fast_u8_ppc3)
    hipLaunchKernelGGL(pw_render_u8_ppc3_kernel, grid, block, e->render_lds, st, ra);
  else if (e->cfg.obs_dtype == PW_OBS_U8)
    hipLaunchKernelGGL(pw_render_generic_kernel<uint8_t>, grid, block, e->render_lds, st, ra);
  else
    hipLaunchKernelGGL(pw_render_generic_kernel<float>, grid, block, e->render_lds, st, ra);
}

static void launch_group(PwEngine* e, const RolloutArgs& r, int32_t batch, hipStream_t st) {
  if (e->np <= 16)
    hipLaunchKernelGGL(pw_step_group_kernel<16>, dim3(static_cast<unsigned>((batch + 15) / 16)), dim3(256), 0, st, r);
  else
    hipLaunchKernelGGL(pw_step_group_kernel<32>, dim3(static_cast<unsigned>((batch + 7) / 8)), dim3(256), 0, st, r);
}

int pw_step(PwEngine* e, const int32_t* puzzle_id, const uint8_t* actions, int8_t* pos, int32_t* steps,
            double* reward, int8_t* dgoals, uint8_t* terminated, uint8_t* truncated, int32_t batch,
            uint32_t flags, void* stream) {
  StepArgs a;
  int rc = fill_step_args(e, puzzle_id, actions, pos, steps, reward, dgoals, terminated, truncated, batch, flags, &a);
  if (rc != PW_OK) return rc;
  if (batch <= 0) return PW_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (e->step_kernel == 1) {  // PUSHWORLD_AMD_STEP=wave: one wavefront per environment
    hipLaunchKernelGGL(pw_step_kernel, dim3(static_cast<unsigned>((batch + 3) / 4)), dim3(256), 0, st, a);
    return check_launch("pw_step");
  }
  if (e->step_kernel == 2) {  // PUSHWORLD_AMD_STEP=lane: one lane per environment
    const dim3 grid(static_cast<unsigned>((batch + 255) / 256)), block(256);
    switch (e->np) {
      case 4: hipLaunchKernelGGL(pw_step_lane_kernel<4>, grid, block, 0, st, a); break;
      case 8: hipLaunchKernelGGL(pw_step_lane_kernel<8>, grid, block, 0, st, a); break;
      case 16: hipLaunchKernelGGL(pw_step_lane_kernel<16>, grid, block, 0, st, a); break;
      default: hipLaunchKernelGGL(pw_step_lane_kernel<32>, grid, block, 0, st, a); break;
    }
    return check_launch("pw_step");
  }
  RolloutArgs r;
  r.s = a;
  r.num_steps = 1;
  r.reward_hist = nullptr;
  r.term_hist = nullptr;
  r.trunc_hist = nullptr;
  launch_group(e, r, batch, st);
  return check_launch("pw_step");
}

int pw_rollout(PwEngine* e, const int32_t* puzzle_id, const uint8_t* actions, int32_t num_steps, int8_t* pos,
               int32_t* steps, double* reward, int8_t* dgoals, uint8_t* terminated, uint8_t* truncated,
               double* reward_hist, uint8_t* terminated_hist, uint8_t* truncated_hist, int32_t batch, uint32_t flags,
               void* stream) {
  RolloutArgs r;
  int rc = fill_step_args(e, puzzle_id, actions, pos, steps, reward, dgoals, terminated, truncated, batch, flags, &r.s);
  if (rc != PW_OK) return rc;
  if (num_steps < 0) return pw_fail(PW_EINVAL, "num_steps must be >= 0");
  if (batch <= 0 || num_steps == 0) return PW_OK;
  r.num_steps = num_steps;
  r.reward_hist = reward_hist;
  r.term_hist = terminated_hist;
  r.trunc_hist = truncated_hist;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (e->step_kernel == 2) {
    const dim3 grid(static_cast<unsigned>((batch + 255) / 256)), block(256);
    switch (e->np) {
      case 4: hipLaunchKernelGGL(pw_rollout_lane_kernel<4>, grid, block, 0, st, r); break;
      case 8: hipLaunchKernelGGL(pw_rollout_lane_kernel<8>, grid, block, 0, st, r); break;
      case 16: hipLaunchKernelGGL(pw_rollout_lane_kernel<16>, grid, block, 0, st, r); break;
      default: hipLaunchKernelGGL(pw_rollout_lane_kernel<32>, grid, block, 0, st, r); break;
    }
  } else {
    launch_group(e, r, batch, st);
  }
  return check_launch("pw_rollout");
}

int pw_render(PwEngine* e, const int32_t* puzzle_id, const int8_t* pos, void* obs, int64_t env_stride_bytes,
              int32_t batch, void* stream) {
  if (!e) return pw_fail(PW_EINVAL, "null engine");
  if (batch <= 0) return PW_OK;
  RenderArgs ra;
  int rc = fill_render_args(e, puzzle_id, pos, obs, env_stride_bytes, batch, &ra);
  if (rc != PW_OK) return rc;
  launch_render(e, ra, batch, static_cast<hipStream_t>(stream));
  return check_launch("pw_render");
}

// Step + observation.  With the page-ordered render the step kernel and the render kernel are two
// launches on the caller's stream (splitting the batch over two streams to hide the 26 us step
// kernel behind the render of the other half measured 2 % SLOWER); otherwise ONE launch: wave 0
// of every per-environment workgroup advances its environment, the workgroup then draws it.
int pw_step_render(PwEngine* e, const int32_t* puzzle_id, const uint8_t* actions, int8_t* pos, int32_t* steps,
                   double* reward, int8_t* dgoals, uint8_t* terminated, uint8_t* truncated, void* obs,
                   int64_t env_stride_bytes, int32_t batch, uint32_t flags, void* stream) {
  if (!e) return pw_fail(PW_EINVAL, "null engine");
  RenderArgs ra;
  int rc = fill_step_args(e, puzzle_id, actions, pos, steps, reward, dgoals, terminated, truncated, batch, flags,
                          &ra.step);
  if (rc != PW_OK) return rc;
  if (batch <= 0) return PW_OK;
  const StepArgs sa = ra.step;
  rc = fill_render_args(e, puzzle_id, pos, obs, env_stride_bytes, batch, &ra);
  if (rc != PW_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // two launches (lane-group step kernel, then the render): measured faster than the single fused launch
  // for every engine -- the wavefront formulation of the step sits on each workgroup's critical path there
  if (!e->force_fused) {
    RolloutArgs r;
    r.s = sa;
    r.num_steps = 1;
    r.reward_hist = nullptr;
    r.term_hist = nullptr;
    r.trunc_hist = nullptr;
    launch_group(e, r, batch, st);
    launch_render(e, ra, batch, st);
    return check_launch("pw_step_render");
  }
  ra.step = sa;
  ra.do_step = 1;
  launch_render(e, ra, batch, st);
  return check_launch("pw_step_render");
}

int pw_step_render_delta(PwEngine* e, const int32_t* puzzle_id, const uint8_t* actions, int8_t* pos, int32_t* steps,
                         double* reward, int8_t* dgoals, uint8_t* terminated, uint8_t* truncated, void* obs,
                         int64_t env_stride_bytes, int32_t batch, uint32_t flags, void* stream) {
  if (!e) return pw_fail(PW_EINVAL, "null engine");
  // uint8 / ppc 3 engines patch from their static images, every other engine redraws the changed rows with
  // the generic LDS kernel; both need the group step kernel (it reports the rows).  Otherwise: full render.
  // float32 / ppc 3 also has static images, but its changed rows are 4x the bytes: one wavefront walking them
  // 4 KiB at a time (0.29 ms) loses to the 256-thread generic redraw (0.27 ms), so only uint8 patches from images
  const bool page_path = e->fast_u8_ppc3 && e->d_simg;
  if ((e->fast_u8_ppc3 && !e->d_simg) || e->step_kernel != 0 || e->force_fused)
    return pw_step_render(e, puzzle_id, actions, pos, steps, reward, dgoals, terminated, truncated, obs, env_stride_bytes,
                          batch, flags, stream);
  RenderArgs ra;
  int rc = fill_step_args(e, puzzle_id, actions, pos, steps, reward, dgoals, terminated, truncated, batch, flags, &ra.step);
  if (rc != PW_OK) return rc;
  if (batch <= 0) return PW_OK;
  StepArgs sa = ra.step;
  rc = fill_render_args(e, puzzle_id, pos, obs, env_stride_bytes, batch, &ra);
  if (rc != PW_OK) return rc;
  if (e->dirty_cap < batch) {
    if (e->d_dirty) (void)hipFree(e->d_dirty);
    e->d_dirty = nullptr;
    e->dirty_cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(&e->d_dirty), static_cast<size_t>(batch) * sizeof(uint32_t)) != hipSuccess)
      return pw_fail(PW_ENOMEM, "pw_step_render_delta: cannot allocate the dirty-row buffer");
    e->dirty_cap = batch;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  sa.dirty = e->d_dirty;
  RolloutArgs r;
  r.s = sa;
  r.num_steps = 1;
  r.reward_hist = nullptr;
  r.term_hist = nullptr;
  r.trunc_hist = nullptr;
  launch_group(e, r, batch, st);
  if (!page_path) {
    ra.dirty_rows = e->d_dirty;
    const dim3 grid(static_cast<unsigned>(batch)), block(PW_RENDER_THREADS);
    if (e->cfg.obs_dtype == PW_OBS_U8)
      hipLaunchKernelGGL(pw_render_generic_kernel<uint8_t>, grid, block, e->render_lds, st, ra);
    else
      hipLaunchKernelGGL(pw_render_generic_kernel<float>, grid, block, e->render_lds, st, ra);
    return check_launch("pw_step_render_delta");
  }
  CopyArgs ca;
  ca.simg = e->d_simg;
  ca.puzzle_id = puzzle_id;
  ca.obs = ra.obs;
  ca.simg_stride = e->simg_stride;
  ca.batch = batch;
  ca.chunks_per_env = static_cast<uint32_t>(env_stride_bytes / 16);
  ca.n_chunks = static_cast<uint32_t>((e->obs_bytes + 15) / 16);
  ca.inv_cpe = 1.0f / static_cast<float>(ca.chunks_per_env);
  hipLaunchKernelGGL(pw_render_delta_kernel<uint8_t>, dim3(static_cast<unsigned>(batch)), dim3(64), 0, st, ra, ca, e->d_dirty);
  return check_launch("pw_step_render_delta");
}

int pw_expand4(PwEngine* e, int32_t puzzle, const int32_t* states, int32_t* succ, uint32_t* moved, uint8_t* goal,
               int32_t num_states, void* stream) {
  if (!e || !states || !succ || !moved || !goal) return pw_fail(PW_EINVAL, "null argument");
  if (puzzle < 0 || puzzle >= e->set->count) return pw_fail(PW_EINVAL, "puzzle index out of range");
  if (num_states <= 0) return PW_OK;
  ExpandArgs a{e->set->d_headers, e->set->d_blob, puzzle, states, succ, moved, goal, num_states};
  if (e->set->headers[puzzle].N <= 16)
    hipLaunchKernelGGL(pw_expand4_kernel<16>, dim3(static_cast<unsigned>((num_states + 15) / 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(pw_expand4_kernel<32>, dim3(static_cast<unsigned>((num_states + 7) / 8)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  return check_launch("pw_expand4");
}

}  // extern "C"

#include "pw_search.inc"
