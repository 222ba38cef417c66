// How long does a kernel of N single-wave workgroups take as a function of the number of DEPENDENT
// global loads each wave performs?  (latency model for the per-environment kernels)
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/launch_rate.hip -o tools/experiments/bin/launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K>
__global__ void chain_kernel(const uint32_t* __restrict__ tab, uint32_t tab_mask, uint32_t* __restrict__ out, int n_items,
                             int items_per_wave_group) {
  const int item = (blockIdx.x * blockDim.x + threadIdx.x) / items_per_wave_group;
  if (item >= n_items) return;
  uint32_t v = static_cast<uint32_t>(item) * 2654435761u;
#pragma unroll
  for (int k = 0; k < K; k++) v = tab[(v + threadIdx.x) & tab_mask] + v;  // dependent: address from previous value
  if ((threadIdx.x & (items_per_wave_group - 1)) == 0) out[item] = v;
}

template <int K>
float run(const uint32_t* tab, uint32_t mask, uint32_t* out, int n_items, int block, int lanes_per_item) {
  const long threads = static_cast<long>(n_items) * lanes_per_item;
  const int grid = static_cast<int>((threads + block - 1) / block);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(chain_kernel<K>, dim3(grid), dim3(block), 0, 0, tab, mask, out, n_items, lanes_per_item);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(chain_kernel<K>, dim3(grid), dim3(block), 0, 0, tab, mask, out, n_items, lanes_per_item);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1000.0f;
}

int main() {
  const int n_items = 65536;
  const uint32_t tab_n = 1u << 20;  // 4 MB table: L2 resident
  std::vector<uint32_t> h(tab_n);
  for (uint32_t i = 0; i < tab_n; i++) h[i] = i * 747796405u + 2891336453u;
  uint32_t *tab, *out;
  hipMalloc(&tab, tab_n * 4);
  hipMalloc(&out, n_items * 4);
  hipMemcpy(tab, h.data(), tab_n * 4, hipMemcpyHostToDevice);
  printf("65536 items; microseconds per launch (back-to-back launches, so launch overhead is included)\n");
  printf("%-34s %8s %8s %8s %8s %8s\n", "layout", "K=0", "K=1", "K=2", "K=4", "K=6");
  struct L { const char* name; int block, lanes; } ls[] = {
      {"64-thread WG = 1 item (64 lanes)", 64, 64},
      {"256-thread WG = 4 items (64 lanes)", 256, 64},
      {"256-thread WG = 16 items (16 lanes)", 256, 16},
      {"256-thread WG = 256 items (1 lane)", 256, 1},
  };
  for (auto& l : ls) {
    printf("%-34s %8.1f %8.1f %8.1f %8.1f %8.1f\n", l.name, run<0>(tab, tab_n - 1, out, n_items, l.block, l.lanes),
           run<1>(tab, tab_n - 1, out, n_items, l.block, l.lanes), run<2>(tab, tab_n - 1, out, n_items, l.block, l.lanes),
           run<4>(tab, tab_n - 1, out, n_items, l.block, l.lanes), run<6>(tab, tab_n - 1, out, n_items, l.block, l.lanes));
  }
  return 0;
}
