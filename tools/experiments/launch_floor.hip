// Round-4 experiment: what a launch of the C4 step's SHAPE costs before it does anything (2 048 workgroups x 256 threads,
// back to back on one stream), and what each dependent memory level adds -- the floor under pw_step_group_mixed_kernel<true>.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/bin/launch_floor tools/experiments/launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k_empty() {}

// `levels` dependent loads per thread (index chain through `idx`, 64-byte strides so every lane's load is its own sector),
// then one 2-byte store per thread -- a C4 step without its arithmetic
__global__ __launch_bounds__(256) void k_chain(const int* __restrict__ idx, short* __restrict__ out, int n, int levels) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  int v = t;
  for (int l = 0; l < levels; l++) v = idx[(static_cast<long>(v) * 16 + l) % n];
  out[t] = static_cast<short>(v);
}

// `valu` dependent integer multiply-adds per thread (the issue time of the step's arithmetic, no memory)
__global__ __launch_bounds__(256) void k_valu(short* __restrict__ out, int valu) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  unsigned v = t;
  for (int i = 0; i < valu; i++) v = v * 1664525u + 1013904223u;
  out[t] = static_cast<short>(v);
}

template <typename F>
static float per_launch_us(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 20; i++) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}

int main() {
  const int n = 1 << 26;  // 256 MB of indices: beyond the L2s
  std::vector<int> h(n);
  unsigned s = 12345u;
  for (int i = 0; i < n; i++) {
    s = s * 1664525u + 1013904223u;
    h[i] = static_cast<int>(s >> 8) % (n / 16);
  }
  int* d_idx;
  short* d_out;
  hipMalloc(&d_idx, sizeof(int) * n);
  hipMalloc(&d_out, sizeof(short) * 2048 * 256 * 4);
  hipMemcpy(d_idx, h.data(), sizeof(int) * n, hipMemcpyHostToDevice);
  for (int wgs : {64, 256, 2048, 8192}) {
    printf("grid %5d x 256: empty %6.2f us", wgs, per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, 0); }, 2000));
    for (int lv : {1, 2, 3, 4})
      printf("  chain%d %6.2f", lv, per_launch_us([&] { hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(256), 0, 0, d_idx, d_out, n, lv); }, 1000));
    for (int v : {100, 400, 800})
      printf("  valu%d %6.2f", v, per_launch_us([&] { hipLaunchKernelGGL(k_valu, dim3(wgs), dim3(256), 0, 0, d_out, v); }, 1000));
    printf("\n");
  }
  return 0;
}
