// Does the ARRANGEMENT of the same physical chunks inside the buffer decide the write class?  (round 3)
// One set of n physical 32 MiB (or 2 MiB) chunks, mapped into one address range in many orders; for each order the
// pure-write form of the page-ordered render (one single-wavefront workgroup per 4 KiB page, non-temporal 16-byte
// stores) in the "eighths" order (XCD k sweeps part k) and the "quarters" order, ms per sweep of the whole buffer.
// Re-mapping costs milliseconds and no memory: if arrangements differ, candidates can be arrangements, not allocations.
//   arrange_probe <MB> <chunk MB> <sets> [lds pad bytes]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#define CK(x)                                                           \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      printf("FAILED %s: %s\n", #x, hipGetErrorString(e_));             \
      exit(2);                                                          \
    }                                                                   \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// order 1 of pw_render_page_kernel: 8 >> run_log2 contiguous parts, XCD k writes in part k mod parts
__global__ __launch_bounds__(64) void write_parts(char* base, unsigned n_pages, unsigned run_log2) {
  extern __shared__ char pad[];
  unsigned page = blockIdx.x;
  const unsigned pl = 3u - run_log2, parts = 1u << pl, k = page & 7u, j = page >> 3;
  const unsigned per = ((n_pages + 7u) >> 3) << run_log2;
  page = (k & (parts - 1u)) * per + ((j << run_log2) | (k >> pl));
  if (page >= n_pages) return;
  char* p = base + static_cast<size_t>(page) * 4096 + threadIdx.x * 16;
  const u32x4 v = {page, k, 0u, 0u};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + 1024));
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + 2048));
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + 3072));
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 3616) << 20;
  const size_t chunk = (size_t)(argc > 2 ? atol(argv[2]) : 32) << 20;
  const int sets = argc > 3 ? atoi(argv[3]) : 3;
  const size_t lds = argc > 4 ? atol(argv[4]) : 0;
  const int n = static_cast<int>((bytes + chunk - 1) / chunk);
  const size_t total = static_cast<size_t>(n) * chunk;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = 0;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const unsigned n_pages = static_cast<unsigned>(bytes / 4096);
  auto measure = [&](char* va, unsigned run_log2) {
    const unsigned grid = (n_pages + 7u) & ~7u;
    write_parts<<<grid, 64, lds>>>(va, n_pages, run_log2);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; r++) write_parts<<<grid, 64, lds>>>(va, n_pages, run_log2);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 4.0f;
  };
  // arrangements: slot s <- chunk perm[s]
  struct Arr {
    std::string name;
    std::vector<int> perm;
  };
  auto make = [&](int set) {
    std::vector<Arr> v;
    std::vector<int> id(n);
    std::iota(id.begin(), id.end(), 0);
    v.push_back({"identity", id});
    {
      std::vector<int> p(id.rbegin(), id.rend());
      v.push_back({"reverse", p});
    }
    for (int parts : {8, 4, 2}) {  // transpose: the `parts` chunks under the fronts at any time are physically consecutive
      const int per = (n + parts - 1) / parts;
      std::vector<int> p(n, -1);
      std::vector<char> used(n, 0);
      int next = 0;
      for (int t = 0; t < per; t++)
        for (int k = 0; k < parts; k++) {
          const int slot = k * per + t;
          if (slot < n) {
            p[slot] = next;
            used[next++] = 1;
          }
        }
      v.push_back({"transpose-" + std::to_string(parts), p});
    }
    for (int q : {7, 11, 13, 31, 59}) {
      if (std::gcd(q, n) != 1) continue;
      std::vector<int> p(n);
      for (int s = 0; s < n; s++) p[s] = static_cast<int>((static_cast<long long>(s) * q) % n);
      v.push_back({"stride-" + std::to_string(q), p});
    }
    for (int seed = 0; seed < 4; seed++) {
      std::vector<int> p = id;
      std::mt19937 rng(1000 * set + seed);
      std::shuffle(p.begin(), p.end(), rng);
      v.push_back({"random-" + std::to_string(seed), p});
    }
    for (int r : {1, 7}) {
      std::vector<int> p(n);
      for (int s = 0; s < n; s++) p[s] = (s + r) % n;
      v.push_back({"rotate-" + std::to_string(r), p});
    }
    v.push_back({"identity again", id});
    return v;
  };
  printf("%zu MB in %d chunks of %zu MB, LDS pad %zu B; ms per sweep (eighths / quarters), GB/s of the better\n", bytes >> 20, n, chunk >> 20, lds);
  std::vector<std::vector<hipMemGenericAllocationHandle_t>> keep;  // earlier sets stay allocated: every set is other memory
  for (int set = 0; set < sets; set++) {
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (int i = 0; i < n; i++) CK(hipMemCreate(&h[i], chunk, &prop, 0));
    printf("set %d\n", set);
    for (const Arr& a : make(set)) {
      char* va = nullptr;  // a fresh address range per arrangement (never one that was mapped before)
      CK(hipMemAddressReserve(reinterpret_cast<void**>(&va), total, 0, nullptr, 0));
      for (int s = 0; s < n; s++) {
        CK(hipMemMap(va + static_cast<size_t>(s) * chunk, chunk, 0, h[a.perm[s]], 0));
        CK(hipMemSetAccess(va + static_cast<size_t>(s) * chunk, chunk, &acc, 1));
      }
      const float e8 = measure(va, 0), e4 = measure(va, 1);
      printf("  %-16s %.4f %.4f   %5.0f\n", a.name.c_str(), e8, e4, bytes / (std::min(e8, e4) * 1e-3) / 1e9);
      fflush(stdout);
      CK(hipDeviceSynchronize());
      for (int s = 0; s < n; s++) CK(hipMemUnmap(va + static_cast<size_t>(s) * chunk, chunk));
    }
    keep.push_back(h);
  }
  return 0;
}
