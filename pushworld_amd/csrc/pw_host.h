// Host-side structures behind the opaque C-ABI handles (pushworld_amd.h).
#ifndef PW_HOST_H_
#define PW_HOST_H_

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pushworld_amd.h"
#include "pw_format.h"

typedef std::pair<int, int> PwCell;  // (x, y)

struct PwPuzzle {
  int width = 0, height = 0;
  int order = PW_ORDER_PYTHON;
  bool has_agent_walls = false;
  std::vector<std::string> names;             // movables in state order, agent first
  std::vector<std::vector<PwCell>> shapes;    // cells relative to the object origin, sorted
  std::vector<PwCell> initial;                // origins (min x, min y), puzzle.py:187-188
  std::vector<std::string> goal_names;
  std::vector<std::vector<PwCell>> goal_shapes;
  std::vector<PwCell> goal;                   // goal k is the target of movable k+1
  std::vector<PwCell> walls;                  // absolute, includes the border
  std::vector<PwCell> agent_walls;            // absolute, raw "aw" cells
};

struct PwPuzzleSet {
  int device = -1;
  int count = 0;
  int max_w = 0, max_h = 0, max_n = 0;
  std::vector<PwPuzzleHeader> headers;
  std::vector<uint8_t> blob;
  PwPuzzleHeader* d_headers = nullptr;  // HBM copies (device >= 0)
  uint8_t* d_blob = nullptr;
};

void pw_set_error(const std::string& msg);
int pw_fail(int code, const std::string& msg);
// Inside a catch (...) handler: the status code of the exception in flight (std::bad_alloc -> PW_ENOMEM, anything
// else PW_EINVAL with its what()).  Every extern "C" entry point is a function-try-block that ends in it, so that
// no C++ exception crosses the C ABI (host vectors and strings are the only things that can throw).
int pw_current_exception() noexcept;

// Makes `device` the calling thread's current HIP device for the lifetime of the guard and restores the caller's
// device afterwards: no entry point of the C ABI changes the caller's current device (and therefore
// torch.cuda.current_device()), and everything an entry point allocates or launches lands on the device of the
// puzzle set it works on.  pw_host.cpp / pw_kernels.hip define it after including the HIP runtime.
struct PwDeviceGuard;

#endif  // PW_HOST_H_
