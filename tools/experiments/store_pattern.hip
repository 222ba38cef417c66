// Micro-benchmark: HBM write throughput of different block -> address mappings (gfx950).
//   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// each block writes `bytes_per_block` contiguous bytes at block * stride, T threads, 16 B per thread per iteration
template <int T>
__global__ __launch_bounds__(T) void k_contig(uint8_t* out, long stride, int bytes_per_block, int spin) {
  uint8_t* base = out + (long)blockIdx.x * stride;
  const int n = bytes_per_block >> 4;
  uint4 v = {blockIdx.x, 1u, 2u, 3u};
  for (int c = threadIdx.x; c < n; c += T) {
    // optional ALU work between stores
    unsigned acc = c;
    for (int s = 0; s < spin; s++) acc = acc * 1664525u + 1013904223u;
    v.y = acc;
    *reinterpret_cast<uint4*>(base + (long)c * 16) = v;
  }
}

// same bytes per env, but each wave owns a contiguous quarter of the env
template <int T>
__global__ __launch_bounds__(T) void k_wave_contig(uint8_t* out, long stride, int bytes_per_block, int spin) {
  uint8_t* base = out + (long)blockIdx.x * stride;
  const int n = bytes_per_block >> 4;
  const int waves = T / 64, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per = (n + waves - 1) / waves;
  const int lo = wave * per, hi = min(n, lo + per);
  uint4 v = {blockIdx.x, 1u, 2u, 3u};
  for (int c = lo + lane; c < hi; c += 64) {
    unsigned acc = c;
    for (int s = 0; s < spin; s++) acc = acc * 1664525u + 1013904223u;
    v.y = acc;
    *reinterpret_cast<uint4*>(base + (long)c * 16) = v;
  }
}

template <typename F>
float time_ms(F f, int reps = 8) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  f(); f();
  CHECK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int i = 0; i < reps; i++) {
    CHECK(hipEventRecord(a));
    f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float t; CHECK(hipEventElapsedTime(&t, a, b));
    ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

// block b -> env b / ppe, piece b % ppe; env base = env * stride, piece base = piece * piece_bytes,
// bytes = min(piece_bytes, env_bytes - piece base)
template <int T>
__global__ __launch_bounds__(T) void k_env_pieces(uint8_t* out, long stride, int env_bytes, int piece_bytes, int ppe, int spin) {
  const int env = blockIdx.x / ppe, piece = blockIdx.x - env * ppe;
  const int lo = piece * piece_bytes;
  const int nbytes = min(piece_bytes, env_bytes - lo);
  if (nbytes <= 0) return;
  uint8_t* base = out + (long)env * stride + lo;
  const int n = (nbytes + 15) >> 4;
  uint4 v = {blockIdx.x, 1u, 2u, 3u};
  for (int c = threadIdx.x; c < n; c += T) {
    unsigned acc = c;
    for (int s = 0; s < spin; s++) acc = acc * 1664525u + 1013904223u;
    v.y = acc;
    *reinterpret_cast<uint4*>(base + (long)c * 16) = v;
  }
}

// (1) per-env block, waves kept in lockstep by a barrier per iteration
template <int T>
__global__ __launch_bounds__(T) void k_env_sync(uint8_t* out, long stride, int bytes_per_block) {
  uint8_t* base = out + (long)blockIdx.x * stride;
  const int n = bytes_per_block >> 4;
  uint4 v = {blockIdx.x, 1u, 2u, 3u};
  for (int c0 = 0; c0 < n; c0 += T) {
    const int c = c0 + threadIdx.x;
    if (c < n) *reinterpret_cast<uint4*>(base + (long)c * 16) = v;
    __syncthreads();
  }
}

// (3) grid-stride sweep: persistent blocks, page p = it * gridDim + blockIdx
template <int T>
__global__ __launch_bounds__(T) void k_gridstride(uint8_t* out, long npages, int page_bytes) {
  uint4 v = {blockIdx.x, 1u, 2u, 3u};
  const int n = page_bytes >> 4;
  for (long p = blockIdx.x; p < npages; p += gridDim.x) {
    uint8_t* base = out + p * page_bytes;
    for (int c = threadIdx.x; c < n; c += T) *reinterpret_cast<uint4*>(base + (long)c * 16) = v;
  }
}

// (4) flat pages, permuted block -> page mapping (multiplicative hash, odd multiplier mod 2^k pages)
template <int T>
__global__ __launch_bounds__(T) void k_permuted(uint8_t* out, long npages_pow2, int page_bytes, unsigned mult) {
  const unsigned long p = ((unsigned long)blockIdx.x * mult) & (npages_pow2 - 1);
  uint8_t* base = out + p * page_bytes;
  uint4 v = {blockIdx.x, 1u, 2u, 3u};
  const int n = page_bytes >> 4;
  for (int c = threadIdx.x; c < n; c += T) *reinterpret_cast<uint4*>(base + (long)c * 16) = v;
}

// (5) block handles K consecutive pages (one after the other), blocks in address order
template <int T>
__global__ __launch_bounds__(T) void k_kpages(uint8_t* out, int K, int page_bytes, int sync) {
  uint4 v = {blockIdx.x, 1u, 2u, 3u};
  const int n = page_bytes >> 4;
  for (int k = 0; k < K; k++) {
    uint8_t* base = out + ((long)blockIdx.x * K + k) * page_bytes;
    for (int c = threadIdx.x; c < n; c += T) *reinterpret_cast<uint4*>(base + (long)c * 16) = v;
    if (sync) __syncthreads();
  }
}

int main() {
  const long envs = 65536, bytes = 57840;
  uint8_t* buf;
  const long cap = envs * 65536;
  CHECK(hipMalloc((void**)&buf, cap));
  CHECK(hipMemset(buf, 0, cap));
  const double total = (double)envs * bytes;
  for (int rep = 0; rep < 2; rep++) {
    float t = time_ms([&] { hipLaunchKernelGGL(k_contig<256>, dim3(envs), dim3(256), 0, 0, buf, 61440L, (int)bytes, 0); });
    printf("per-env block stride=61440        : %.4f ms  %.0f GB/s\n", t, total / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(k_env_sync<256>, dim3(envs), dim3(256), 0, 0, buf, 61440L, (int)bytes); });
    printf("per-env block, barrier per iter   : %.4f ms  %.0f GB/s\n", t, total / t / 1e6);
    const long npages = 1 << 20;  // 4 GiB
    t = time_ms([&] { hipLaunchKernelGGL(k_contig<256>, dim3(npages), dim3(256), 0, 0, buf, 4096L, 4096, 0); });
    printf("flat 4 KiB pages (1M blocks)       : %.4f ms  %.0f GB/s\n", t, (double)npages * 4096 / t / 1e6);
    for (int g : {2048, 4096, 8192}) {
      t = time_ms([&] { hipLaunchKernelGGL(k_gridstride<256>, dim3(g), dim3(256), 0, 0, buf, npages, 4096); });
      printf("grid-stride sweep grid=%5d         : %.4f ms  %.0f GB/s\n", g, t, (double)npages * 4096 / t / 1e6);
    }
    for (unsigned mult : {1u, 15u, 4097u, 0x9E3779B1u}) {
      t = time_ms([&] { hipLaunchKernelGGL(k_permuted<256>, dim3(npages), dim3(256), 0, 0, buf, npages, 4096, mult); });
      printf("flat pages permuted mult=%10u : %.4f ms  %.0f GB/s\n", mult, t, (double)npages * 4096 / t / 1e6);
    }
    for (int K : {2, 4, 8, 16}) {
      for (int sync : {0, 1}) {
        t = time_ms([&] { hipLaunchKernelGGL(k_kpages<256>, dim3(npages / K), dim3(256), 0, 0, buf, K, 4096, sync); });
        printf("K=%2d consecutive pages/block sync=%d : %.4f ms  %.0f GB/s\n", K, sync, t, (double)npages * 4096 / t / 1e6);
      }
    }
  }
  return 0;
}
