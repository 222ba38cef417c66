// Host side of libpushworld_amd: .pwp parser, bitboard/render table packer and the
// puzzle / puzzle-set half of the C ABI.
//
// Replaces the table construction of the reference constructor
// (python3/src/pushworld/puzzle.py:130-311, cpp/src/pushworld_puzzle.cc:191-362).  The
// reference enumerates hash sets of colliding positions (O(cells^2 * overlap), up to
// 1.5 s per puzzle); here a puzzle is packed into row bitboards in microseconds and the
// kernels evaluate the same predicate
//     collides(i, j) <=> shift(mask_i, action) & mask_j != 0  and  mask_i & mask_j == 0
// directly (puzzle.py:562,592 / pushworld_puzzle.cc:135,168).
#include "pw_host.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <set>

#include "pw_device_guard.h"

static thread_local std::string g_last_error;

void pw_set_error(const std::string& msg) { g_last_error = msg; }
int pw_fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

int pw_current_exception() noexcept {
  try {
    try {
      throw;
    } catch (const std::bad_alloc&) {
      g_last_error = "out of host memory";
      return PW_ENOMEM;
    } catch (const std::exception& ex) {
      g_last_error = std::string("unexpected error: ") + ex.what();
      return PW_EINVAL;
    } catch (...) {
      g_last_error = "unexpected error";
      return PW_EINVAL;
    }
  } catch (...) {  // the message itself could not be stored
    return PW_ENOMEM;
  }
}

namespace {

// Splits on any whitespace like Python's str.split() (puzzle.py:137).  The C++ reference
// splits on ' ' only (pushworld_puzzle.cc:209-211); the two differ for tabs, which no
// shipped puzzle contains.
std::vector<std::string> split_ws(const std::string& line) {
  std::vector<std::string> out;
  size_t i = 0, n = line.size();
  while (i < n) {
    while (i < n && std::isspace(static_cast<unsigned char>(line[i]))) i++;
    size_t j = i;
    while (j < n && !std::isspace(static_cast<unsigned char>(line[j]))) j++;
    if (j > i) out.push_back(line.substr(i, j - i));
    i = j;
  }
  return out;
}

std::vector<PwCell> sorted_cells(const std::set<PwCell>& s) {
  return std::vector<PwCell>(s.begin(), s.end());
}

uint32_t absent_mask(const std::set<PwCell>& cells, int x, int y) {
  // puzzle.py:614-638: one border strip per missing 8-neighbour
  uint32_t m = 0;
  if (!cells.count({x - 1, y})) m |= PW_NB_L;
  if (!cells.count({x + 1, y})) m |= PW_NB_R;
  if (!cells.count({x, y - 1})) m |= PW_NB_U;
  if (!cells.count({x, y + 1})) m |= PW_NB_D;
  if (!cells.count({x - 1, y - 1})) m |= PW_NB_UL;
  if (!cells.count({x + 1, y - 1})) m |= PW_NB_UR;
  if (!cells.count({x - 1, y + 1})) m |= PW_NB_DL;
  if (!cells.count({x + 1, y + 1})) m |= PW_NB_DR;
  return m;
}

int parse_puzzle(const char* text, size_t len, int order, PwPuzzle* pz) {
  if (order != PW_ORDER_PYTHON && order != PW_ORDER_CPP)
    return pw_fail(PW_EINVAL, "order must be PW_ORDER_PYTHON or PW_ORDER_CPP");

  bool saw_empty_element = false;
  // element id -> absolute cells; `first_seen` keeps the file order of appearance, which
  // is the dict insertion order the Python reference iterates (puzzle.py:235-237).
  std::map<std::string, std::set<PwCell>> cells;
  std::vector<std::string> first_seen;

  int n_cols = -1, n_rows = 0;
  size_t pos = 0;
  while (pos < len) {
    size_t eol = pos;
    while (eol < len && text[eol] != '\n') eol++;
    std::string line(text + pos, eol - pos);
    pos = eol + 1;
    n_rows++;
    std::vector<std::string> toks = split_ws(line);
    if (n_cols < 0) {
      n_cols = static_cast<int>(toks.size());
    } else if (static_cast<int>(toks.size()) != n_cols) {
      return pw_fail(PW_EPARSE, "Row " + std::to_string(n_rows) +
                                    " does not have the same number of elements as the first row.");
    }
    for (int col = 1; col <= static_cast<int>(toks.size()); col++) {
      const std::string& tok = toks[col - 1];
      size_t s = 0;
      while (s <= tok.size()) {
        size_t e = tok.find('+', s);
        if (e == std::string::npos) e = tok.size();
        std::string id = tok.substr(s, e - s);
        s = e + 1;
        for (auto& ch : id) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
        if (id.empty()) {  // "M1++G1", "M1+", "+": the reference stores the element "" (see below)
          saw_empty_element = true;
          continue;
        }
        if (id == ".") continue;
        if (!cells.count(id)) first_seen.push_back(id);
        cells[id].insert({col, n_rows});
      }
    }
  }
  if (n_cols <= 0 || n_rows <= 0) return pw_fail(PW_EPARSE, "empty puzzle");
  if (!cells.count("a"))
    return pw_fail(PW_EPARSE, "Every puzzle must have an agent object, indicated by 'a'.");

  const int W = n_cols + 2, H = n_rows + 2;  // puzzle.py:160-161
  if (W > PW_MAX_DIM || H > PW_MAX_DIM)
    return pw_fail(PW_ELIMIT, "puzzle is " + std::to_string(W) + "x" + std::to_string(H) +
                                  " cells; the engine supports at most 64x64");
  std::set<PwCell>& wall = cells["w"];
  for (int x = 0; x < W; x++) {
    wall.insert({x, 0});
    wall.insert({x, H - 1});
  }
  for (int y = 0; y < H; y++) {
    wall.insert({0, y});
    wall.insert({W - 1, y});
  }

  // goal ids in scan order: descending strings for Python (puzzle.py:178-179), ascending
  // for C++ (std::map iteration, pushworld_puzzle.cc:274).
  std::vector<std::string> goal_ids;
  for (const auto& kv : cells)
    if (kv.first[0] == 'g') goal_ids.push_back(kv.first);  // ascending
  if (order == PW_ORDER_PYTHON) std::reverse(goal_ids.begin(), goal_ids.end());

  std::vector<std::string> movables = {"a"};
  for (const auto& g : goal_ids) {
    std::string m = "m" + g.substr(1);
    if (!cells.count(m)) return pw_fail(PW_EGOAL, "Goal has no associated movable object: " + m);
    movables.push_back(m);
  }
  auto listed = [&](const std::string& id) {
    return std::find(movables.begin(), movables.end(), id) != movables.end();
  };
  if (order == PW_ORDER_PYTHON) {
    for (const auto& id : first_seen)
      if (id[0] == 'm' && !listed(id)) movables.push_back(id);
  } else {
    std::vector<std::string> rest;
    for (const auto& kv : cells)
      if (kv.first[0] == 'm' && !listed(kv.first)) rest.push_back(kv.first);  // ascending
    for (const auto& id : rest) movables.push_back(id);
  }
  // puzzle.py:218 evaluates elem_id[0] for every element in descending order; "" sorts last, so the
  // reference raises IndexError only after every goal has found its movable
  if (saw_empty_element)
    return pw_fail(PW_EELEMENT, "empty element name in a cell ('+' without a name on one side)");
  if (movables.size() > PW_MAX_OBJECTS)
    return pw_fail(PW_ELIMIT, "puzzle has " + std::to_string(movables.size()) +
                                  " movables; the engine supports at most 32");

  auto origin_of = [&](const std::set<PwCell>& s) {
    int mx = 1 << 30, my = 1 << 30;
    for (const auto& c : s) {
      mx = std::min(mx, c.first);
      my = std::min(my, c.second);
    }
    return PwCell(mx, my);
  };
  auto relative = [&](const std::set<PwCell>& s, PwCell o) {
    std::set<PwCell> r;
    for (const auto& c : s) r.insert({c.first - o.first, c.second - o.second});
    return r;
  };

  pz->width = W;
  pz->height = H;
  pz->order = order;
  pz->names = movables;
  for (const auto& id : movables) {
    PwCell o = origin_of(cells[id]);
    pz->initial.push_back(o);
    pz->shapes.push_back(sorted_cells(relative(cells[id], o)));
  }
  pz->goal_names = goal_ids;
  for (const auto& g : goal_ids) {
    PwCell o = origin_of(cells[g]);
    pz->goal.push_back(o);
    pz->goal_shapes.push_back(sorted_cells(relative(cells[g], o)));
  }
  pz->walls = sorted_cells(wall);
  pz->has_agent_walls = cells.count("aw") > 0;
  if (pz->has_agent_walls) pz->agent_walls = sorted_cells(cells["aw"]);
  return PW_OK;
}

template <typename T>
uint32_t append(std::vector<uint8_t>& blob, const std::vector<T>& v) {
  while (blob.size() % 8) blob.push_back(0);
  uint32_t off = static_cast<uint32_t>(blob.size());
  const uint8_t* p = reinterpret_cast<const uint8_t*>(v.data());
  blob.insert(blob.end(), p, p + v.size() * sizeof(T));
  return off;
}

void pack_puzzle(const PwPuzzle& pz, PwPuzzleHeader* hdr, std::vector<uint8_t>& blob) {
  const int W = pz.width, H = pz.height;
  const int N = static_cast<int>(pz.names.size()), G = static_cast<int>(pz.goal.size());
  while (blob.size() % 8) blob.push_back(0);
  const uint32_t base = static_cast<uint32_t>(blob.size());
  std::memset(hdr, 0, sizeof(*hdr));
  hdr->base = base;
  hdr->W = static_cast<uint8_t>(W);
  hdr->H = static_cast<uint8_t>(H);
  hdr->N = static_cast<uint8_t>(N);
  hdr->G = static_cast<uint8_t>(G);
  hdr->has_aw = pz.has_agent_walls ? 1 : 0;

  std::set<PwCell> wall(pz.walls.begin(), pz.walls.end());
  std::set<PwCell> awall(wall);  // trap T2: the agent-wall layer is AW u W (puzzle.py:273)
  awall.insert(pz.agent_walls.begin(), pz.agent_walls.end());

  std::vector<uint64_t> wall_rows(H, 0), awall_rows(H, 0);
  for (const auto& c : wall) wall_rows[c.second] |= 1ull << c.first;
  for (const auto& c : awall) awall_rows[c.second] |= 1ull << c.first;
  hdr->off_wall = append(blob, wall_rows) - base;
  hdr->off_awall = append(blob, awall_rows) - base;

  std::vector<uint64_t> shape_rows;
  for (int j = 0; j < N; j++) {
    int w = 0, h = 0;
    for (const auto& c : pz.shapes[j]) {
      w = std::max(w, c.first + 1);
      h = std::max(h, c.second + 1);
    }
    hdr->objtab[j].w = static_cast<uint8_t>(w);
    hdr->objtab[j].h = static_cast<uint8_t>(h);
    hdr->objtab[j].row_off = static_cast<uint16_t>(shape_rows.size());
    std::vector<uint64_t> rows(h, 0);
    for (const auto& c : pz.shapes[j]) rows[c.second] |= 1ull << c.first;
    shape_rows.insert(shape_rows.end(), rows.begin(), rows.end());
  }
  hdr->off_shapes = append(blob, shape_rows) - base;
  for (int j = 0; j < N; j++) {
    hdr->init[j][0] = static_cast<int8_t>(pz.initial[j].first);
    hdr->init[j][1] = static_cast<int8_t>(pz.initial[j].second);
  }
  for (int g = 0; g < G; g++) {
    hdr->goal[g][0] = static_cast<int8_t>(pz.goal[g].first);
    hdr->goal[g][1] = static_cast<int8_t>(pz.goal[g].second);
  }

  // static render codes: painter's order AW(uW) -> W, goal outlines kept separately in
  // the top byte (puzzle.py:453-458).
  std::vector<uint32_t> codes(static_cast<size_t>(W) * H, 0);
  if (pz.has_agent_walls)
    for (const auto& c : awall)
      codes[c.second * W + c.first] = (1u << PW_CODE_KIND_SHIFT) | absent_mask(awall, c.first, c.second);
  for (const auto& c : wall)
    codes[c.second * W + c.first] = (2u << PW_CODE_KIND_SHIFT) | absent_mask(wall, c.first, c.second);
  for (int g = 0; g < G; g++) {
    std::set<PwCell> gs(pz.goal_shapes[g].begin(), pz.goal_shapes[g].end());
    for (const auto& c : gs) {
      int x = pz.goal[g].first + c.first, y = pz.goal[g].second + c.second;
      codes[y * W + x] |= absent_mask(gs, c.first, c.second) << PW_CODE_GOAL_SHIFT;
    }
  }
  hdr->off_static = append(blob, codes) - base;

  std::vector<uint32_t> mcells;
  for (int j = 0; j < N; j++) {
    std::set<PwCell> s(pz.shapes[j].begin(), pz.shapes[j].end());
    for (const auto& c : s)
      mcells.push_back(static_cast<uint32_t>(c.first) | (static_cast<uint32_t>(c.second) << 8) |
                       (absent_mask(s, c.first, c.second) << 16) | (static_cast<uint32_t>(j) << 24));
  }
  hdr->n_mcells = static_cast<uint32_t>(mcells.size());
  if (mcells.empty()) mcells.push_back(0);
  hdr->off_mcells = append(blob, mcells) - base;

  std::vector<uint64_t> small(N, 0);
  for (int j = 0; j < N; j++)
    small[j] = pw_small_board(shape_rows.data() + hdr->objtab[j].row_off, hdr->objtab[j].w, hdr->objtab[j].h);
  hdr->off_small = append(blob, small) - base;
}

int copy_cells(const std::vector<PwCell>& v, int32_t* xy, int cap) {
  int n = static_cast<int>(v.size());
  if (xy)
    for (int i = 0; i < n && i < cap; i++) {
      xy[2 * i] = v[i].first;
      xy[2 * i + 1] = v[i].second;
    }
  return n;
}

}  // namespace

// copies the packed tables of `s` to its HIP device (no-op for host-only sets); destroys `s` on failure
static int upload_set(PwPuzzleSet* s) {
  if (s->device < 0) return PW_OK;
  const size_t n = static_cast<size_t>(s->count);
  PwDeviceGuard guard(s->device);
  hipError_t err = guard.status();
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&s->d_headers), n * sizeof(PwPuzzleHeader));
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&s->d_blob), s->blob.size());
  if (err == hipSuccess)
    err = hipMemcpy(s->d_headers, s->headers.data(), n * sizeof(PwPuzzleHeader), hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(s->d_blob, s->blob.data(), s->blob.size(), hipMemcpyHostToDevice);
  if (err != hipSuccess) {
    std::string msg = std::string("puzzle set upload failed: ") + hipGetErrorString(err);
    pw_puzzleset_destroy(s);
    return pw_fail(PW_EDEVICE, msg);
  }
  return PW_OK;
}

extern "C" {

const char* pw_last_error(void) { return g_last_error.c_str(); }
int pw_abi_version(void) { return PW_ABI_VERSION; }

int pw_device_count(void) try {
  int n = 0;
  hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess) return pw_fail(PW_EDEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(err));
  return n;
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzle_parse(const char* text, size_t len, int order, PwPuzzle** out) try {
  if (!text || !out) return pw_fail(PW_EINVAL, "null argument");
  PwPuzzle* pz = new (std::nothrow) PwPuzzle();
  if (!pz) return pw_fail(PW_ENOMEM, "out of memory");
  int rc;
  try {
    rc = parse_puzzle(text, len, order, pz);
  } catch (const std::exception& e) {
    rc = pw_fail(PW_ENOMEM, e.what());
  }
  if (rc != PW_OK) {
    delete pz;
    return rc;
  }
  *out = pz;
  return PW_OK;
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

void pw_puzzle_destroy(PwPuzzle* p) { delete p; }

int pw_puzzle_info(const PwPuzzle* p, PwPuzzleInfo* info) try {
  if (!p || !info) return pw_fail(PW_EINVAL, "null argument");
  info->width = p->width;
  info->height = p->height;
  info->num_movables = static_cast<int32_t>(p->names.size());
  info->num_goals = static_cast<int32_t>(p->goal.size());
  info->num_wall_cells = static_cast<int32_t>(p->walls.size());
  info->num_agent_wall_cells = static_cast<int32_t>(p->agent_walls.size());
  info->has_agent_walls = p->has_agent_walls ? 1 : 0;
  info->order = p->order;
  return PW_OK;
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzle_initial_state(const PwPuzzle* p, int32_t* xy) try {
  if (!p || !xy) return pw_fail(PW_EINVAL, "null argument");
  copy_cells(p->initial, xy, static_cast<int>(p->initial.size()));
  return PW_OK;
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzle_goal_state(const PwPuzzle* p, int32_t* xy) try {
  if (!p) return pw_fail(PW_EINVAL, "null argument");
  copy_cells(p->goal, xy, static_cast<int>(p->goal.size()));
  return PW_OK;
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzle_object_cells(const PwPuzzle* p, int obj, int32_t* xy, int cap) try {
  if (!p || obj < 0 || obj >= static_cast<int>(p->shapes.size())) return pw_fail(PW_EINVAL, "bad object index");
  return copy_cells(p->shapes[obj], xy, cap);
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzle_goal_cells(const PwPuzzle* p, int goal, int32_t* xy, int cap) try {
  if (!p || goal < 0 || goal >= static_cast<int>(p->goal_shapes.size())) return pw_fail(PW_EINVAL, "bad goal index");
  return copy_cells(p->goal_shapes[goal], xy, cap);
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzle_wall_cells(const PwPuzzle* p, int32_t* xy, int cap) try {
  if (!p) return pw_fail(PW_EINVAL, "null argument");
  return copy_cells(p->walls, xy, cap);
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzle_agent_wall_cells(const PwPuzzle* p, int32_t* xy, int cap) try {
  if (!p) return pw_fail(PW_EINVAL, "null argument");
  return copy_cells(p->agent_walls, xy, cap);
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzle_object_name(const PwPuzzle* p, int obj, char* buf, int cap) try {
  if (!p || obj < 0 || obj >= static_cast<int>(p->names.size())) return pw_fail(PW_EINVAL, "bad object index");
  const std::string& s = p->names[obj];
  if (buf && cap > 0) {
    int n = std::min<int>(cap - 1, static_cast<int>(s.size()));
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return static_cast<int>(s.size());
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzleset_create(const PwPuzzle* const* puzzles, int n, int device, PwPuzzleSet** out) try {
  if (!puzzles || !out || n <= 0) return pw_fail(PW_EINVAL, "empty puzzle set");
  PwPuzzleSet* s = new (std::nothrow) PwPuzzleSet();
  if (!s) return pw_fail(PW_ENOMEM, "out of memory");
  s->device = device;
  s->count = n;
  s->headers.resize(n);
  for (int i = 0; i < n; i++) {
    if (!puzzles[i]) {
      delete s;
      return pw_fail(PW_EINVAL, "null puzzle in set");
    }
    pack_puzzle(*puzzles[i], &s->headers[i], s->blob);
    s->max_w = std::max(s->max_w, puzzles[i]->width);
    s->max_h = std::max(s->max_h, puzzles[i]->height);
    s->max_n = std::max(s->max_n, static_cast<int>(puzzles[i]->names.size()));
  }
  while (s->blob.size() % 16) s->blob.push_back(0);
  if (int rc = upload_set(s)) return rc;
  *out = s;
  return PW_OK;
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

void pw_puzzleset_destroy(PwPuzzleSet* s) {
  if (!s) return;
  if (s->d_headers) (void)hipFree(s->d_headers);
  if (s->d_blob) (void)hipFree(s->d_blob);
  delete s;
}

int pw_puzzleset_size(const PwPuzzleSet* s) try { return s ? s->count : pw_fail(PW_EINVAL, "null set"); } catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzleset_max_dims(const PwPuzzleSet* s, int* max_w, int* max_h, int* max_n) try {
  if (!s) return pw_fail(PW_EINVAL, "null set");
  if (max_w) *max_w = s->max_w;
  if (max_h) *max_h = s->max_h;
  if (max_n) *max_n = s->max_n;
  return PW_OK;
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzleset_blob(const PwPuzzleSet* s, const void** data, size_t* bytes) try {
  if (!s || !data || !bytes) return pw_fail(PW_EINVAL, "null argument");
  *data = s->blob.data();
  *bytes = s->blob.size();
  return PW_OK;
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

}  // extern "C"

// ---- packed puzzle-set file: the compiled form of a puzzle pool, so that a training
// job loads ~15 k puzzles with two reads instead of parsing text.  Little endian, 64-byte header,
// then the PwPuzzleHeader array and the table blob exactly as they sit in HBM (mmap-able).
struct PwSetFileHeader {
  char magic[8];          // "PWSET\0\0\0"
  uint32_t version;       // PW_SETFILE_VERSION
  uint32_t header_bytes;  // sizeof(PwPuzzleHeader): layout guard
  int32_t count, max_w, max_h, max_n;
  uint64_t blob_bytes;
  uint64_t checksum;      // FNV-1a 64 over headers + blob
  uint8_t reserved[16];
};
static_assert(sizeof(PwSetFileHeader) == 64, "file header is 64 bytes");
#define PW_SETFILE_VERSION 2u  // 2: PwPuzzleHeader::off_small

static uint64_t fnv1a64(const void* data, size_t n, uint64_t h) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001B3ull;
  return h;
}

// nullptr when the packed record is self-consistent, otherwise what is wrong with it
static const char* pw_validate_packed_puzzle(const PwPuzzleHeader& h, const uint8_t* blob, uint64_t lim) {
  if (h.W < 1 || h.W > PW_MAX_DIM || h.H < 1 || h.H > PW_MAX_DIM || h.N < 1 || h.N > PW_MAX_OBJECTS || h.G >= h.N)
    return "puzzle dimensions out of range";
  if (h.base > lim || (h.base & 7u)) return "table offsets out of range";
  const uint64_t room = lim - h.base;
  auto inside = [&](uint64_t off, uint64_t bytes, uint64_t align) {
    return off <= room && bytes <= room - off && (off % align) == 0;
  };
  uint32_t shape_rows = 0;
  for (int j = 0; j < h.N; j++) {
    const PwObjEntry& o = h.objtab[j];
    if (o.w < 1 || o.h < 1 || o.w > h.W || o.h > h.H) return "object bounding box out of range";
    shape_rows = std::max<uint32_t>(shape_rows, static_cast<uint32_t>(o.row_off) + o.h);
    const int x = h.init[j][0], y = h.init[j][1];
    if (x < 0 || y < 0 || x + o.w > h.W || y + o.h > h.H) return "initial position outside the grid";
  }
  for (int g = 0; g < h.G; g++) {
    const int x = h.goal[g][0], y = h.goal[g][1];
    if (x < 0 || y < 0 || x >= h.W || y >= h.H) return "goal position outside the grid";
  }
  if (!inside(h.off_wall, 8ull * h.H, 8) || !inside(h.off_awall, 8ull * h.H, 8) ||
      !inside(h.off_shapes, 8ull * shape_rows, 8) || !inside(h.off_static, 4ull * h.W * h.H, 4) ||
      !inside(h.off_mcells, 4ull * std::max<uint32_t>(h.n_mcells, 1u), 4) || h.n_mcells > 64u * 64u ||
      !inside(h.off_small, 8ull * h.N, 8))
    return "table offsets out of range";
  // the LDS-staged step kernel copies (off_static - off_shapes) / 8 shape rows: the packer's section order
  if (h.off_static < h.off_shapes || h.off_static - h.off_shapes < 8ull * shape_rows) return "table sections out of order";
  const uint64_t width_mask = h.W >= 64 ? ~0ull : ((1ull << h.W) - 1ull);
  const uint64_t* wall = reinterpret_cast<const uint64_t*>(blob + h.base + h.off_wall);
  const uint64_t* awall = reinterpret_cast<const uint64_t*>(blob + h.base + h.off_awall);
  for (int y = 0; y < h.H; y++)
    if ((wall[y] & ~width_mask) || (awall[y] & ~width_mask)) return "wall rows reach outside the grid";
  const uint64_t* shapes = reinterpret_cast<const uint64_t*>(blob + h.base + h.off_shapes);
  for (int j = 0; j < h.N; j++) {
    const PwObjEntry& o = h.objtab[j];
    const uint64_t m = o.w >= 64 ? ~0ull : ((1ull << o.w) - 1ull);
    for (int r = 0; r < o.h; r++)
      if (shapes[o.row_off + r] & ~m) return "shape rows reach outside the bounding box";
    if (reinterpret_cast<const uint64_t*>(blob + h.base + h.off_small)[j] != pw_small_board(shapes + o.row_off, o.w, o.h))
      return "small-object bitboards do not match the shape rows";
  }
  const uint32_t* mcells = reinterpret_cast<const uint32_t*>(blob + h.base + h.off_mcells);
  for (uint32_t i = 0; i < h.n_mcells; i++) {
    const uint32_t c = mcells[i];
    const uint32_t obj = c >> 24, cx = c & 0xffu, cy = (c >> 8) & 0xffu;
    if (obj >= h.N || cx >= h.objtab[obj].w || cy >= h.objtab[obj].h) return "movable cell list out of range";
  }
  const uint32_t* codes = reinterpret_cast<const uint32_t*>(blob + h.base + h.off_static);
  for (int i = 0; i < h.W * h.H; i++)
    if (((codes[i] >> PW_CODE_KIND_SHIFT) & 0xfu) > 2u || ((codes[i] >> 12) & 0xfffu)) return "static cell codes out of range";
  return nullptr;
}

extern "C" {

int pw_puzzleset_save(const PwPuzzleSet* s, const char* path) try {
  if (!s || !path) return pw_fail(PW_EINVAL, "null argument");
  PwSetFileHeader fh;
  std::memset(&fh, 0, sizeof(fh));
  std::memcpy(fh.magic, "PWSET", 5);
  fh.version = PW_SETFILE_VERSION;
  fh.header_bytes = sizeof(PwPuzzleHeader);
  fh.count = s->count;
  fh.max_w = s->max_w;
  fh.max_h = s->max_h;
  fh.max_n = s->max_n;
  fh.blob_bytes = s->blob.size();
  const size_t hb = static_cast<size_t>(s->count) * sizeof(PwPuzzleHeader);
  fh.checksum = fnv1a64(s->blob.data(), s->blob.size(), fnv1a64(s->headers.data(), hb, 0xCBF29CE484222325ull));
  FILE* f = std::fopen(path, "wb");
  if (!f) return pw_fail(PW_EINVAL, std::string("cannot open for writing: ") + path);
  bool ok = std::fwrite(&fh, sizeof(fh), 1, f) == 1 && std::fwrite(s->headers.data(), hb, 1, f) == 1 &&
            (s->blob.empty() || std::fwrite(s->blob.data(), s->blob.size(), 1, f) == 1);
  ok = (std::fclose(f) == 0) && ok;
  return ok ? PW_OK : pw_fail(PW_EINVAL, std::string("short write: ") + path);
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

int pw_puzzleset_load(const char* path, int device, PwPuzzleSet** out) try {
  if (!path || !out) return pw_fail(PW_EINVAL, "null argument");
  // owners: whatever throws below (the host vectors of a multi-GB set), the file and the set are released
  struct FileOwner {
    FILE* f;
    ~FileOwner() { if (f) std::fclose(f); }
  } file{std::fopen(path, "rb")};
  FILE* f = file.f;
  if (!f) return pw_fail(PW_EINVAL, std::string("cannot open: ") + path);
  PwSetFileHeader fh;
  auto bad = [&](const char* why) { return pw_fail(PW_EPARSE, std::string(path) + ": " + why); };
  if (std::fread(&fh, sizeof(fh), 1, f) != 1) return bad("truncated file header");
  if (std::memcmp(fh.magic, "PWSET\0\0\0", 8) != 0) return bad("not a packed puzzle set");
  if (fh.version != PW_SETFILE_VERSION || fh.header_bytes != sizeof(PwPuzzleHeader)) return bad("unsupported format version");
  if (fh.count <= 0 || fh.blob_bytes % 16 != 0 || fh.blob_bytes > (1ull << 34) || fh.max_w > PW_MAX_DIM ||
      fh.max_h > PW_MAX_DIM || fh.max_n > PW_MAX_OBJECTS)
    return bad("corrupt file header");
  {  // the sizes in the header must be the file's: nothing is allocated on the word of a crafted count
    const uint64_t want = sizeof(fh) + static_cast<uint64_t>(fh.count) * sizeof(PwPuzzleHeader) + fh.blob_bytes;
    if (std::fseek(f, 0, SEEK_END) != 0) return bad("cannot seek");
    const long size = std::ftell(f);
    if (size < 0 || static_cast<uint64_t>(size) != want) return bad(static_cast<uint64_t>(size) < want ? "truncated file" : "trailing bytes");
    if (std::fseek(f, static_cast<long>(sizeof(fh)), SEEK_SET) != 0) return bad("cannot seek");
  }
  struct SetOwner {
    PwPuzzleSet* s;
    ~SetOwner() { delete s; }
  } owner{new (std::nothrow) PwPuzzleSet()};
  PwPuzzleSet* s = owner.s;
  if (!s) return pw_fail(PW_ENOMEM, "out of memory");
  s->device = device;
  s->count = fh.count;
  s->max_w = fh.max_w;
  s->max_h = fh.max_h;
  s->max_n = fh.max_n;
  s->headers.resize(fh.count);
  s->blob.resize(fh.blob_bytes);
  const size_t hb = static_cast<size_t>(fh.count) * sizeof(PwPuzzleHeader);
  const bool ok = std::fread(s->headers.data(), hb, 1, f) == 1 &&
                  (s->blob.empty() || std::fread(s->blob.data(), s->blob.size(), 1, f) == 1);
  if (!ok || fnv1a64(s->blob.data(), s->blob.size(), fnv1a64(s->headers.data(), hb, 0xCBF29CE484222325ull)) != fh.checksum)
    return bad(ok ? "checksum mismatch" : "truncated file");
  // The kernels index with every field below unchecked (LDS writes included), and FNV-1a is no protection
  // against a crafted file: validate the full extent of every section and every index stored in them.
  for (const PwPuzzleHeader& h : s->headers) {
    if (const char* why = pw_validate_packed_puzzle(h, s->blob.data(), s->blob.size()))
      return pw_fail(PW_EPARSE, std::string(path) + ": " + why);
  }
  owner.s = nullptr;  // upload_set destroys the set itself when it fails
  if (int rc = upload_set(s)) return rc;
  *out = s;
  return PW_OK;
} catch (...) {
  return pw_current_exception();  // nothing C++ leaves the C ABI
}

}  // extern "C"
