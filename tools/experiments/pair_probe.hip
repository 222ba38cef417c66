// Which physical chunks can eight write fronts share at full speed?  (time-boxed probe, round 3)
// K chunks of 32 MB (hipMemCreate), each mapped at its own slot of one reserved range.  One launch = eight write fronts
// (XCD k = workgroups with index k mod 8, the dispatch order the render kernel relies on), front k sweeping chunk c[k]
// `reps` times with non-temporal 16-byte stores, one 4 KiB page per single-wavefront workgroup -- the store pattern of
// pw_render_page_kernel in its contiguous-parts order.
//   E1: random 8-subsets of the K chunks -> GB/s per set; least squares for pair penalties; how much variance do chunk
//       pairs explain, and are the penalties clustered?
//   E2: all fronts inside ONE chunk (eight 4 MB parts) -> per-chunk GB/s.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/pair_probe.hip -o tools/experiments/bin/pair_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("FAILED %s: %s\n", #x, hipGetErrorString(e_));                \
      exit(2);                                                             \
    }                                                                      \
  } while (0)

struct Fronts {
  char* base[8];
  unsigned pages;  // pages per sweep of one front
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void write8(Fronts f) {
  extern __shared__ char pad[];
  const unsigned k = blockIdx.x & 7u, j = blockIdx.x >> 3;
  char* p = f.base[k] + static_cast<size_t>(j % f.pages) * 4096 + threadIdx.x * 16;
  const u32x4 v = {j, k, 0u, 0u};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + 1024));
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + 2048));
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + 3072));
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 24;
  const int sets = argc > 2 ? atoi(argv[2]) : 600;
  const int reps = argc > 3 ? atoi(argv[3]) : 8;
  const size_t lds = argc > 4 ? atoi(argv[4]) : 0;
  const size_t chunk = size_t(32) << 20;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = 0;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  char* va = nullptr;
  CK(hipMemAddressReserve(reinterpret_cast<void**>(&va), chunk * K, 0, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> h(K);
  // spread the chunks over the device memory: a 3 GB spacer between them (released at the end)
  std::vector<void*> spacers;
  for (int i = 0; i < K; i++) {
    CK(hipMemCreate(&h[i], chunk, &prop, 0));
    CK(hipMemMap(va + i * chunk, chunk, 0, h[i], 0));
    CK(hipMemSetAccess(va + i * chunk, chunk, &acc, 1));
    if (argc > 5 && atoi(argv[5])) {
      void* s = nullptr;
      if (hipMalloc(&s, size_t(atoi(argv[5])) << 20) == hipSuccess) spacers.push_back(s);
    }
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const Fronts& f, unsigned pages_total) {
    write8<<<8 * pages_total, 64, lds>>>(f);  // warm
    CK(hipEventRecord(e0));
    write8<<<8 * pages_total, 64, lds>>>(f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 8.0 * pages_total * 4096.0 / (ms * 1e-3) / 1e9;  // GB/s
  };
  const unsigned ppc = static_cast<unsigned>(chunk / 4096);
  // ---- E2: all eight fronts inside one chunk
  printf("E2 per chunk (eight 4 MB parts of one chunk), GB/s:\n");
  for (int i = 0; i < K; i++) {
    Fronts f;
    for (int k = 0; k < 8; k++) f.base[k] = va + i * chunk + k * (chunk / 8);
    f.pages = ppc / 8;
    printf(" %5.0f", run(f, ppc / 8 * reps * 8));
  }
  printf("\n");
  // ---- E1: random 8-subsets
  std::mt19937 rng(12345);
  std::vector<std::vector<int>> S(sets);
  std::vector<double> bw(sets);
  for (int s = 0; s < sets; s++) {
    std::vector<int> perm(K);
    for (int i = 0; i < K; i++) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), rng);
    S[s].assign(perm.begin(), perm.begin() + 8);
    Fronts f;
    for (int k = 0; k < 8; k++) f.base[k] = va + S[s][k] * chunk;
    f.pages = ppc;
    bw[s] = run(f, ppc * reps);
  }
  std::vector<double> sorted = bw;
  std::sort(sorted.begin(), sorted.end());
  printf("E1 %d random 8-subsets of %d chunks: GB/s min %.0f p10 %.0f median %.0f p90 %.0f max %.0f\n", sets, K, sorted.front(),
         sorted[sets / 10], sorted[sets / 2], sorted[sets * 9 / 10], sorted.back());
  printf("E1 histogram (100 GB/s bins from 5500):");
  for (int b = 0; b < 22; b++) {
    int c = 0;
    for (double v : bw) c += v >= 5500 + 100 * b && v < 5600 + 100 * b;
    printf(" %d", c);
  }
  printf("\n");
  // repeatability: the first 10 sets again
  printf("E1 repeat of the first 10 sets (first / again):");
  for (int s = 0; s < 10 && s < sets; s++) {
    Fronts f;
    for (int k = 0; k < 8; k++) f.base[k] = va + S[s][k] * chunk;
    f.pages = ppc;
    printf(" %.0f/%.0f", bw[s], run(f, ppc * reps));
  }
  printf("\n");
  // order dependence: same set, fronts permuted
  printf("E1 same chunks, other XCD assignment (set 0, five permutations):");
  for (int t = 0; t < 5; t++) {
    std::vector<int> q = S[0];
    std::shuffle(q.begin(), q.end(), rng);
    Fronts f;
    for (int k = 0; k < 8; k++) f.base[k] = va + q[k] * chunk;
    f.pages = ppc;
    printf(" %.0f", run(f, ppc * reps));
  }
  printf("\n");
  // ---- additive chunk model: bw = mu - sum_k a[c_k]  (least squares by coordinate descent), then pair residuals
  std::vector<double> a(K, 0.0);
  double mu = 0;
  for (double v : bw) mu += v;
  mu /= sets;
  for (int it = 0; it < 200; it++) {
    for (int c = 0; c < K; c++) {
      double num = 0;
      int cnt = 0;
      for (int s = 0; s < sets; s++) {
        bool in = false;
        double pred = mu;
        for (int x : S[s]) {
          pred -= a[x];
          in = in || x == c;
        }
        if (!in) continue;
        num += (pred + a[c]) - bw[s];
        cnt++;
      }
      if (cnt) a[c] = num / cnt;
    }
  }
  double ss_tot = 0, ss_res = 0;
  for (int s = 0; s < sets; s++) {
    double pred = mu;
    for (int x : S[s]) pred -= a[x];
    ss_tot += (bw[s] - mu) * (bw[s] - mu);
    ss_res += (bw[s] - pred) * (bw[s] - pred);
  }
  printf("additive per-chunk model: R^2 = %.3f; per-chunk cost (GB/s):", 1.0 - ss_res / ss_tot);
  for (int c = 0; c < K; c++) printf(" %.0f", a[c]);
  printf("\n");
  // pair residual means
  std::vector<double> pr(K * K, 0.0);
  std::vector<int> pn(K * K, 0);
  for (int s = 0; s < sets; s++) {
    double pred = mu;
    for (int x : S[s]) pred -= a[x];
    const double r = bw[s] - pred;
    for (int x : S[s])
      for (int y : S[s])
        if (x < y) {
          pr[x * K + y] += r;
          pn[x * K + y]++;
        }
  }
  std::vector<double> pm;
  for (int i = 0; i < K * K; i++)
    if (pn[i] >= 5) pm.push_back(pr[i] / pn[i]);
  std::sort(pm.begin(), pm.end());
  if (!pm.empty())
    printf("pair residual means (%zu pairs, >= 5 sets each): min %.0f p10 %.0f median %.0f p90 %.0f max %.0f  (sd of set bw %.0f)\n", pm.size(),
           pm.front(), pm[pm.size() / 10], pm[pm.size() / 2], pm[pm.size() * 9 / 10], pm.back(), std::sqrt(ss_tot / sets));
  for (void* s : spacers) (void)hipFree(s);
  return 0;
}
