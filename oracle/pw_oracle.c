/*
 * pw_oracle.c -- plain-C CPU restatement of the PushWorld hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Used only by tests/ (as a checker) and by the
 * `cpu_baseline` leg of bench.py (kind "port").  Nothing under pushworld_amd/ links or
 * loads it.  Parity status: pinned -- tests/test_oracle_golden.py checks it against the
 * fixtures captured from the reference (tests/golden/make_golden.py).
 *
 * It follows the reference ALGORITHM (hash-set collision tables + LIFO push frontier +
 * per-cell painter), not the bitboard formulation of the HIP kernels, so that the two are
 * independent implementations:
 *   tables     python3/src/pushworld/puzzle.py:259-308, :522-593
 *              (== cpp/src/pushworld_puzzle.cc:123-172, :324-359); sets are stored as bit
 *              sets indexed by position (static) / relative offset (dynamic)
 *   step       puzzle.py:348-394 (== pushworld_puzzle.cc:386-460)
 *   goal       puzzle.py:396-411
 *   reward     python3/src/pushworld/gym_env.py:201-223
 *   render     puzzle.py:426-469, :596-638
 *   padding    python3/src/pushworld/utils/env_utils.py:44-91
 *
 * Parsing is done by the Python oracle (oracle/pw_oracle.py), which hands over cell lists.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* sched_setaffinity, CPU_SET */
#endif
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 32
#define NUM_ACTIONS 4
static const int DX[4] = {-1, 1, 0, 0}; /* puzzle.py:43-50 */
static const int DY[4] = {0, 0, -1, 1};

typedef struct {
  int n;         /* cells */
  int *x, *y;    /* relative to the origin (absolute for walls) */
  int w, h;      /* bounding box */
  uint8_t* mem;  /* (w+2) x (h+2) membership raster with a 1-cell apron (the reference uses a
                    hash set; O(1) lookups keep this baseline honest) */
} Shape;

typedef struct {
  int ox, oy, w, h; /* offsets dx in [ox, ox+w), dy in [oy, oy+h) */
  uint8_t* bits;    /* w*h flags */
} OffsetSet;

typedef struct OrPuzzle {
  int W, H, N, G, has_aw;
  Shape obj[MAXN];
  Shape goal_shape[MAXN];
  Shape walls, awalls; /* awalls = AW u W (puzzle.py:273) */
  int init[MAXN][2], goal[MAXN][2];
  uint8_t* stat[NUM_ACTIONS][MAXN];    /* W*H flags: position is in the static set */
  OffsetSet dyn[NUM_ACTIONS][MAXN][MAXN];
  uint8_t* occ_tmp;
} OrPuzzle;

static void shape_init(Shape* s, int n, const int* xy) {
  s->n = n;
  s->x = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
  s->y = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
  s->w = s->h = 0;
  for (int i = 0; i < n; i++) {
    s->x[i] = xy[2 * i];
    s->y[i] = xy[2 * i + 1];
    if (s->x[i] + 1 > s->w) s->w = s->x[i] + 1;
    if (s->y[i] + 1 > s->h) s->h = s->y[i] + 1;
  }
  s->mem = (uint8_t*)calloc((size_t)(s->w + 2) * (s->h + 2), 1);
  for (int i = 0; i < n; i++) s->mem[(s->y[i] + 1) * (s->w + 2) + s->x[i] + 1] = 1;
}

static void shape_free(Shape* s) {
  free(s->x);
  free(s->y);
  free(s->mem);
}

static int shape_has(const Shape* s, int x, int y) {
  if (x < -1 || y < -1 || x > s->w || y > s->h) return 0;
  return s->mem[(y + 1) * (s->w + 2) + x + 1];
}

/* points_overlap(a, b, offset): exists p in a with p + offset in b (puzzle.py:509-513);
 * `grid` is a membership raster of b over [0,gw)x[0,gh). */
static int overlap_raster(const Shape* a, const uint8_t* grid, int gw, int gh, int dx, int dy) {
  for (int i = 0; i < a->n; i++) {
    int x = a->x[i] + dx, y = a->y[i] + dy;
    if (x >= 0 && y >= 0 && x < gw && y < gh && grid[y * gw + x]) return 1;
  }
  return 0;
}

static uint8_t* rasterize(const Shape* s, int gw, int gh) {
  uint8_t* g = (uint8_t*)calloc((size_t)gw * gh, 1);
  for (int i = 0; i < s->n; i++)
    if (s->x[i] >= 0 && s->y[i] >= 0 && s->x[i] < gw && s->y[i] < gh) g[s->y[i] * gw + s->x[i]] = 1;
  return g;
}

/* puzzle.py:522-564 */
static uint8_t* build_static(const OrPuzzle* p, int a, const Shape* obj, const Shape* obst) {
  const int W = p->W, H = p->H;
  uint8_t* set = (uint8_t*)calloc((size_t)W * H, 1);
  uint8_t* grid = rasterize(obst, W, H);
  const int maxx = W - obj->w, maxy = H - obj->h;
  for (int i = 0; i < obj->n; i++)
    for (int k = 0; k < obst->n; k++) {
      int dx = -DX[a] + obst->x[k] - obj->x[i];
      int dy = -DY[a] + obst->y[k] - obj->y[i];
      if (dx >= 0 && dy >= 0 && dx <= maxx && dy <= maxy && !set[dy * W + dx] &&
          !overlap_raster(obj, grid, W, H, dx, dy))
        set[dy * W + dx] = 1;
    }
  free(grid);
  return set;
}

/* puzzle.py:567-593 */
static void build_dynamic(OffsetSet* out, int a, const Shape* pusher, const Shape* pushee) {
  out->ox = -pusher->w - 1;
  out->oy = -pusher->h - 1;
  out->w = pusher->w + pushee->w + 3;
  out->h = pusher->h + pushee->h + 3;
  out->bits = (uint8_t*)calloc((size_t)out->w * out->h, 1);
  uint8_t* grid = rasterize(pushee, pushee->w, pushee->h);
  for (int i = 0; i < pusher->n; i++)
    for (int k = 0; k < pushee->n; k++) {
      int dx = -DX[a] + pushee->x[k] - pusher->x[i];
      int dy = -DY[a] + pushee->y[k] - pusher->y[i];
      /* (dx, dy) = pusher_pos - pushee_pos (looked up at puzzle.py:370-372): a pusher cell
       * shifted by it is expressed in the pushee's frame. */
      if (!overlap_raster(pusher, grid, pushee->w, pushee->h, dx, dy))
        out->bits[(dy - out->oy) * out->w + (dx - out->ox)] = 1;
    }
  free(grid);
}

static int dyn_contains(const OffsetSet* s, int dx, int dy) {
  int x = dx - s->ox, y = dy - s->oy;
  if (x < 0 || y < 0 || x >= s->w || y >= s->h) return 0;
  return s->bits[y * s->w + x];
}

OrPuzzle* or_puzzle_create(int W, int H, int N, int G, int has_aw, const int* obj_ncells, const int* obj_cells,
                           const int* init_xy, const int* goal_xy, const int* goal_ncells, const int* goal_cells,
                           int n_walls, const int* walls_xy, int n_awalls, const int* awalls_xy) {
  if (N > MAXN || G > MAXN) return NULL;
  OrPuzzle* p = (OrPuzzle*)calloc(1, sizeof(OrPuzzle));
  p->W = W;
  p->H = H;
  p->N = N;
  p->G = G;
  p->has_aw = has_aw;
  const int* c = obj_cells;
  for (int j = 0; j < N; j++) {
    shape_init(&p->obj[j], obj_ncells[j], c);
    c += 2 * obj_ncells[j];
    p->init[j][0] = init_xy[2 * j];
    p->init[j][1] = init_xy[2 * j + 1];
  }
  c = goal_cells;
  for (int g = 0; g < G; g++) {
    shape_init(&p->goal_shape[g], goal_ncells[g], c);
    c += 2 * goal_ncells[g];
    p->goal[g][0] = goal_xy[2 * g];
    p->goal[g][1] = goal_xy[2 * g + 1];
  }
  shape_init(&p->walls, n_walls, walls_xy);
  shape_init(&p->awalls, n_awalls, awalls_xy);
  /* the obstacle shapes are absolute: their bounding box is irrelevant */
  for (int a = 0; a < NUM_ACTIONS; a++) {
    p->stat[a][0] = build_static(p, a, &p->obj[0], &p->awalls); /* puzzle.py:272-281 */
    for (int m = 1; m < N; m++) p->stat[a][m] = build_static(p, a, &p->obj[m], &p->walls); /* :284-293 */
    for (int i = 0; i < N; i++)
      for (int j = 1; j < N; j++) build_dynamic(&p->dyn[a][i][j], a, &p->obj[i], &p->obj[j]); /* :298-308 */
  }
  return p;
}

void or_puzzle_destroy(OrPuzzle* p) {
  if (!p) return;
  for (int j = 0; j < p->N; j++) shape_free(&p->obj[j]);
  for (int g = 0; g < p->G; g++) shape_free(&p->goal_shape[g]);
  shape_free(&p->walls);
  shape_free(&p->awalls);
  for (int a = 0; a < NUM_ACTIONS; a++) {
    for (int m = 0; m < p->N; m++) free(p->stat[a][m]);
    for (int i = 0; i < p->N; i++)
      for (int j = 1; j < p->N; j++) free(p->dyn[a][i][j].bits);
  }
  free(p);
}

/* table sizes, for the golden cross-check */
int or_static_size(const OrPuzzle* p, int a, int i) {
  int n = 0;
  for (int k = 0; k < p->W * p->H; k++) n += p->stat[a][i][k];
  return n;
}

int or_dynamic_size(const OrPuzzle* p, int a, int i, int j) {
  if (j < 1) return 0;
  const OffsetSet* s = &p->dyn[a][i][j];
  int n = 0;
  for (int k = 0; k < s->w * s->h; k++) n += s->bits[k];
  return n;
}

static int in_static(const OrPuzzle* p, int a, int i, int x, int y) {
  if (x < 0 || y < 0 || x >= p->W || y >= p->H) return 0;
  return p->stat[a][i][y * p->W + x];
}

/* puzzle.py:348-394.  state: int[N][2], updated in place.  Returns the moved-object bit mask
 * (0 when nothing moves), i.e. the C++ moved_object_indices (pushworld_puzzle.cc:446-457). */
uint32_t or_step(const OrPuzzle* p, int* state, int action) {
  const int N = p->N;
  if (in_static(p, action, 0, state[0], state[1])) return 0;
  int pushed[MAXN] = {0};
  int stack[MAXN], sp = 0;
  pushed[0] = 1;
  stack[sp++] = 0;
  while (sp) {
    const int i = stack[--sp]; /* LIFO */
    const int xi = state[2 * i], yi = state[2 * i + 1];
    for (int j = 1; j < N; j++) {
      if (pushed[j]) continue;
      const int xj = state[2 * j], yj = state[2 * j + 1];
      if (!dyn_contains(&p->dyn[action][i][j], xi - xj, yi - yj)) continue;
      if (in_static(p, action, j, xj, yj)) return 0; /* transitive stopping */
      pushed[j] = 1;
      stack[sp++] = j;
    }
  }
  uint32_t mask = 0;
  for (int k = 0; k < N; k++)
    if (pushed[k]) {
      state[2 * k] += DX[action];
      state[2 * k + 1] += DY[action];
      mask |= 1u << k;
    }
  return mask;
}

int or_count_goals(const OrPuzzle* p, const int* state) { /* puzzle.py:396-407 */
  int n = 0;
  for (int g = 0; g < p->G; g++)
    if (state[2 * (g + 1)] == p->goal[g][0] && state[2 * (g + 1) + 1] == p->goal[g][1]) n++;
  return n;
}

/* gym_env.py:201-223.  Returns terminated; *reward as a double. */
int or_env_step(const OrPuzzle* p, int* state, int action, double* reward) {
  const int before = or_count_goals(p, state);
  or_step(p, state, action);
  const int after = or_count_goals(p, state);
  const int terminated = after == p->G;
  *reward = terminated ? 10.0 : (double)(after - before) - 0.01;
  return terminated;
}

/* ------------------------------------------------------------------ render */
static void fill_rect(uint8_t* img, int iw, int r1, int r2, int c1, int c2, const uint8_t* rgb) {
  for (int r = r1; r < r2; r++)
    for (int c = c1; c < c2; c++) {
      uint8_t* px = img + ((size_t)r * iw + c) * 3;
      px[0] = rgb[0];
      px[1] = rgb[1];
      px[2] = rgb[2];
    }
}

/* puzzle.py:596-638 */
static void draw_object(uint8_t* img, int iw, const Shape* s, int px, int py, const uint8_t* fill,
                        const uint8_t* border, int ppc, int bw) {
  for (int i = 0; i < s->n; i++) {
    const int r = (py + s->y[i]) * ppc, c = (px + s->x[i]) * ppc;
    if (fill) fill_rect(img, iw, r, r + ppc, c, c + ppc, fill);
    for (int dr = -1; dr <= 1; dr++)
      for (int dc = -1; dc <= 1; dc++) {
        if (!dr && !dc) continue;
        if (shape_has(s, s->x[i] + dc, s->y[i] + dr)) continue;
        const int r1 = r + (dr > 0 ? ppc - bw : 0), r2 = dr == 0 ? r1 + ppc : r1 + bw;
        const int c1 = c + (dc > 0 ? ppc - bw : 0), c2 = dc == 0 ? c1 + ppc : c1 + bw;
        fill_rect(img, iw, r1, r2, c1, c2, border);
      }
  }
}

static const uint8_t C_AGENT[3] = {0x00, 0xDC, 0x00}, C_AGENT_B[3] = {0x00, 0x6E, 0x00};
static const uint8_t C_AW[3] = {0xFA, 0xC7, 0x1E}, C_AW_B[3] = {0x7D, 0x64, 0x0F};
static const uint8_t C_GOAL_B[3] = {0xB9, 0x00, 0x00};
static const uint8_t C_GOBJ[3] = {0xDC, 0x00, 0x00}, C_GOBJ_B[3] = {0x6E, 0x00, 0x00};
static const uint8_t C_MOV[3] = {0x46, 0x9B, 0xFF}, C_MOV_B[3] = {0x23, 0x48, 0x7F};
static const uint8_t C_WALL[3] = {0x0A, 0x0A, 0x0A}, C_WALL_B[3] = {0x05, 0x05, 0x05};

/* puzzle.py:426-469: img uint8 [H*ppc][W*ppc][3] */
void or_render(const OrPuzzle* p, const int* state, int ppc, int bw, uint8_t* img) {
  const int iw = p->W * ppc, ih = p->H * ppc;
  memset(img, 255, (size_t)iw * ih * 3);
  if (p->has_aw) draw_object(img, iw, &p->awalls, 0, 0, C_AW, C_AW_B, ppc, bw);
  draw_object(img, iw, &p->walls, 0, 0, C_WALL, C_WALL_B, ppc, bw);
  for (int j = 0; j < p->N; j++) {
    const uint8_t* f = j == 0 ? C_AGENT : (j <= p->G ? C_GOBJ : C_MOV);
    const uint8_t* b = j == 0 ? C_AGENT_B : (j <= p->G ? C_GOBJ_B : C_MOV_B);
    draw_object(img, iw, &p->obj[j], state[2 * j], state[2 * j + 1], f, b, ppc, bw);
  }
  for (int g = 0; g < p->G; g++)
    draw_object(img, iw, &p->goal_shape[g], p->goal[g][0], p->goal[g][1], NULL, C_GOAL_B, ppc, bw);
}

/* env_utils.py:44-91 with uint8 kept (pad = 0).  out: [pad_h*ppc][pad_w*ppc][3]; scratch: H*W*ppc^2*3 */
void or_observation_u8(const OrPuzzle* p, const int* state, int pad_h, int pad_w, int ppc, int bw, uint8_t* out,
                       uint8_t* scratch) {
  const int iw = p->W * ppc, ih = p->H * ppc, ow = pad_w * ppc, oh = pad_h * ppc;
  or_render(p, state, ppc, bw, scratch);
  const int top = (oh - ih) / 2, left = (ow - iw) / 2;
  memset(out, 0, (size_t)ow * oh * 3);
  for (int r = 0; r < ih; r++) memcpy(out + ((size_t)(r + top) * ow + left) * 3, scratch + (size_t)r * iw * 3, (size_t)iw * 3);
}

void or_observation_f32(const OrPuzzle* p, const int* state, int pad_h, int pad_w, int ppc, int bw, float* out,
                        uint8_t* scratch) {
  const int iw = p->W * ppc, ih = p->H * ppc, ow = pad_w * ppc, oh = pad_h * ppc;
  or_render(p, state, ppc, bw, scratch);
  const int top = (oh - ih) / 2, left = (ow - iw) / 2;
  memset(out, 0, (size_t)ow * oh * 3 * sizeof(float));
  for (int r = 0; r < ih; r++) {
    float* dst = out + ((size_t)(r + top) * ow + left) * 3;
    const uint8_t* src = scratch + (size_t)r * iw * 3;
    for (int k = 0; k < iw * 3; k++) dst[k] = (float)src[k] / 255.0f; /* env_utils.py:65-72 */
  }
}

/* ------------------------------------------------------------------ batched baseline driver
 * B environments (env b uses puzzles[pid[b]]), T steps of pre-generated actions [T][B], with
 * the reference's reset-on-done handled like the GPU bench (next-step autoreset).  When
 * `render` != 0 every step also renders the padded observation into a per-thread buffer (1: uint8,
 * 2: float32 = uint8 / 255 as env_utils.py:65-72).  Returns a checksum so the work cannot be optimised away.  OpenMP over envs. */
/* Thread placement of the baseline runs: OpenMP thread t of or_rollout pins itself to CPU cpus[t % n] (n = 0: no pinning).
 * Set from Python (tools/cpu_baselines.py) with one hardware thread of every physical core first -- OMP_PROC_BIND in the
 * environment would also bind the process's main thread, i.e. the thread that launches the GPU kernels. */
static int g_cpus[1024];
static int g_ncpus = 0;
void or_set_thread_cpus(const int* cpus, int n) {
  g_ncpus = n < 0 ? 0 : (n > 1024 ? 1024 : n);
  for (int i = 0; i < g_ncpus; i++) g_cpus[i] = cpus[i];
}
static void or_pin_self(int tid) {
  if (g_ncpus <= 0) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  CPU_SET(g_cpus[tid % g_ncpus], &set);
  (void)sched_setaffinity(0, sizeof(set), &set);
}

uint64_t or_rollout(OrPuzzle* const* puzzles, const int32_t* pid, int B, int T, const uint8_t* actions,
                    int max_steps, int render, int pad_h, int pad_w, int ppc, int bw, int num_threads,
                    int* threads_used) {
  uint64_t total = 0;
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = num_threads > 0 ? num_threads : omp_get_max_threads(); /* num_threads <= 0: all */
#endif
  if (threads_used) *threads_used = nthreads;
#pragma omp parallel num_threads(nthreads) reduction(+ : total)
  {
#ifdef _OPENMP
    or_pin_self(omp_get_thread_num());
#endif
    uint8_t* obs = NULL;
    uint8_t* scratch = NULL;
    if (render) {
      obs = (uint8_t*)malloc((size_t)pad_h * pad_w * ppc * ppc * 3 * (render == 2 ? sizeof(float) : 1));
      scratch = (uint8_t*)malloc((size_t)pad_h * pad_w * ppc * ppc * 3);
    }
#pragma omp for schedule(dynamic, 16)
    for (int b = 0; b < B; b++) {
      const OrPuzzle* p = puzzles[pid[b]];
      int state[2 * MAXN];
      for (int j = 0; j < p->N; j++) {
        state[2 * j] = p->init[j][0];
        state[2 * j + 1] = p->init[j][1];
      }
      int steps = 0, done = 0;
      for (int t = 0; t < T; t++) {
        double reward = 0.0;
        if (done) {
          for (int j = 0; j < p->N; j++) {
            state[2 * j] = p->init[j][0];
            state[2 * j + 1] = p->init[j][1];
          }
          steps = 0;
          done = 0;
        } else {
          const int term = or_env_step(p, state, actions[(size_t)t * B + b] & 3, &reward);
          steps++;
          done = term || (max_steps >= 0 && steps >= max_steps);
        }
        if (render == 2) {
          or_observation_f32(p, state, pad_h, pad_w, ppc, bw, (float*)obs, scratch);
          total += (uint64_t)(((float*)obs)[((size_t)pad_h * ppc / 2) * pad_w * ppc * 3 + (size_t)pad_w * ppc / 2 * 3] > 0.5f);
        } else if (render) {
          or_observation_u8(p, state, pad_h, pad_w, ppc, bw, obs, scratch);
          total += obs[((size_t)pad_h * ppc / 2) * pad_w * ppc * 3 + (size_t)pad_w * ppc / 2 * 3];
        }
        total += (uint64_t)(state[0] * 131 + state[1]) + (uint64_t)(reward > 0.5);
      }
    }
    free(obs);
    free(scratch);
  }
  return total;
}

/* ------------------------------------------------------------------ batched checkers for the full-size tests
 * (tests/test_gpu_configs.py, tests/test_gpu_expand.py): the same per-environment calls as above, OpenMP over
 * environments / states, every step's outputs kept. */

/* T steps of B environments from their initial states.  autoreset != 0: next-step autoreset exactly as
 * pw_step(PW_STEP_AUTORESET) -- an environment whose terminated | truncated flag is set on entry is reset
 * instead of stepped (reward 0, flags 0, step counter 0).  Outputs per step: pos int8 [T][B][np][2] (zero
 * beyond the puzzle's movables), reward f64 [T][B], terminated / truncated uint8 [T][B], steps int32 [T][B]. */
void or_rollout_trace(OrPuzzle* const* puzzles, const int32_t* pid, int B, int T, const uint8_t* actions, int max_steps,
                      int autoreset, int np, int8_t* pos, double* reward, uint8_t* term, uint8_t* trunc, int32_t* steps) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int b = 0; b < B; b++) {
    const OrPuzzle* p = puzzles[pid[b]];
    int state[2 * MAXN];
    for (int j = 0; j < p->N; j++) {
      state[2 * j] = p->init[j][0];
      state[2 * j + 1] = p->init[j][1];
    }
    int n = 0, te = 0, tr = 0;
    for (int t = 0; t < T; t++) {
      double r = 0.0;
      if (autoreset && (te | tr)) {
        for (int j = 0; j < p->N; j++) {
          state[2 * j] = p->init[j][0];
          state[2 * j + 1] = p->init[j][1];
        }
        n = 0;
        te = 0;
        tr = 0;
      } else {
        te = or_env_step(p, state, actions[(size_t)t * B + b] & 3, &r);
        n++;
        tr = (max_steps >= 0 && n >= max_steps) ? 1 : 0;
      }
      const size_t o = (size_t)t * B + b;
      int8_t* row = pos + o * np * 2;
      for (int j = 0; j < np; j++) {
        row[2 * j] = j < p->N ? (int8_t)state[2 * j] : 0;
        row[2 * j + 1] = j < p->N ? (int8_t)state[2 * j + 1] : 0;
      }
      reward[o] = r;
      term[o] = (uint8_t)te;
      trunc[o] = (uint8_t)tr;
      steps[o] = n;
    }
  }
}

/* or_rollout_trace for long runs of big batches: instead of every position row, a 64-bit LINEAR digest of it per step and
 * environment -- digest[t][b] = sum_k row[k] * weights[k] (mod 2^64) over the np * 2 int8 values of the row -- which the test
 * forms from the engine's positions with the same weights; the full rows only after the last step (pos_last int8 [B][np][2]). */
void or_rollout_digest(OrPuzzle* const* puzzles, const int32_t* pid, int B, int T, const uint8_t* actions, int max_steps,
                       int autoreset, int np, const int64_t* weights, int64_t* digest, double* reward, uint8_t* term,
                       uint8_t* trunc, int32_t* steps, int8_t* pos_last) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int b = 0; b < B; b++) {
    const OrPuzzle* p = puzzles[pid[b]];
    int state[2 * MAXN];
    for (int j = 0; j < p->N; j++) {
      state[2 * j] = p->init[j][0];
      state[2 * j + 1] = p->init[j][1];
    }
    int n = 0, te = 0, tr = 0;
    for (int t = 0; t < T; t++) {
      double r = 0.0;
      if (autoreset && (te | tr)) {
        for (int j = 0; j < p->N; j++) {
          state[2 * j] = p->init[j][0];
          state[2 * j + 1] = p->init[j][1];
        }
        n = 0;
        te = 0;
        tr = 0;
      } else {
        te = or_env_step(p, state, actions[(size_t)t * B + b] & 3, &r);
        n++;
        tr = (max_steps >= 0 && n >= max_steps) ? 1 : 0;
      }
      const size_t o = (size_t)t * B + b;
      uint64_t d = 0;
      for (int k = 0; k < 2 * p->N && k < 2 * np; k++) d += (uint64_t)(int64_t)(int8_t)state[k] * (uint64_t)weights[k];
      digest[o] = (int64_t)d;
      reward[o] = r;
      term[o] = (uint8_t)te;
      trunc[o] = (uint8_t)tr;
      steps[o] = n;
    }
    int8_t* row = pos_last + (size_t)b * np * 2;
    for (int j = 0; j < np; j++) {
      row[2 * j] = j < p->N ? (int8_t)state[2 * j] : 0;
      row[2 * j + 1] = j < p->N ? (int8_t)state[2 * j + 1] : 0;
    }
  }
}

/* Padded observations (uint8, or float32 when f32 != 0) of the K environments sel[0..K-1] of a batch whose
 * positions are pos int8 [B][np][2]; out is [K][pad_h * ppc][pad_w * ppc][3]. */
void or_observe_batch(OrPuzzle* const* puzzles, const int32_t* pid, const int8_t* pos, int np, const int32_t* sel, int K,
                      int pad_h, int pad_w, int ppc, int bw, int f32, void* out) {
  const size_t elems = (size_t)pad_h * ppc * pad_w * ppc * 3;
#pragma omp parallel
  {
    uint8_t* scratch = (uint8_t*)malloc(elems);
#pragma omp for schedule(dynamic, 4)
    for (int k = 0; k < K; k++) {
      const int b = sel[k];
      const OrPuzzle* p = puzzles[pid[b]];
      int state[2 * MAXN];
      for (int j = 0; j < p->N; j++) {
        state[2 * j] = pos[((size_t)b * np + j) * 2];
        state[2 * j + 1] = pos[((size_t)b * np + j) * 2 + 1];
      }
      if (f32) or_observation_f32(p, state, pad_h, pad_w, ppc, bw, (float*)out + (size_t)k * elems, scratch);
      else or_observation_u8(p, state, pad_h, pad_w, ppc, bw, (uint8_t*)out + (size_t)k * elems, scratch);
    }
    free(scratch);
  }
}

/* The four successors of F states of one puzzle in the planner's wire format (pushworld_puzzle.h:32-37,
 * Position2D = x * 10000 + y): succ int32 [F][4][N], moved uint32 [F][4] (bit k = object k moved,
 * pushworld_puzzle.cc:446-457), goal uint8 [F][4] (satisfiesGoal of the successor, cc:462-469). */
void or_expand4_batch(const OrPuzzle* p, const int32_t* states, int64_t F, int32_t* succ, uint32_t* moved, uint8_t* goal) {
  const int N = p->N;
#pragma omp parallel
  {
#ifdef _OPENMP
  or_pin_self(omp_get_thread_num());
#endif
#pragma omp for schedule(static, 1024)
  for (int64_t f = 0; f < F; f++) {
    for (int a = 0; a < 4; a++) {
      int state[2 * MAXN];
      for (int j = 0; j < N; j++) {
        state[2 * j] = states[f * N + j] / 10000;
        state[2 * j + 1] = states[f * N + j] % 10000;
      }
      const uint32_t m = or_step(p, state, a);
      const int64_t o = f * 4 + a;
      for (int j = 0; j < N; j++) succ[o * N + j] = state[2 * j] * 10000 + state[2 * j + 1];
      moved[o] = m;
      goal[o] = (uint8_t)(or_count_goals(p, state) == p->G);
    }
  }
  }
}
