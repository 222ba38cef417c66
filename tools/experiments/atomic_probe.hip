// Round-5 probe: what the closed set of pw_search_* can expect from this chip -- random 8-byte loads, random 64-byte line loads
// (four 16-byte loads of one lane), random 64-bit atomicCAS and "line load, then CAS into it", one access per thread, over a
// table of 128 MB (Infinity-Cache sized) and of 2 GB.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/bin/atomic_probe tools/experiments/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; return x ^ (x >> 16);
}
__global__ __launch_bounds__(256) void k_load8(const unsigned long long* t, uint32_t mask, unsigned long long* out, uint32_t salt) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long v = t[mix(i ^ salt) & mask];
  if (v == 0x123456789ull) out[0] = v;
}
__global__ __launch_bounds__(256) void k_line(const unsigned long long* t, uint32_t mask, unsigned long long* out, uint32_t salt) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint4* l = reinterpret_cast<const uint4*>(t + (mix(i ^ salt) & mask & ~7u));
  const uint4 a = l[0], b = l[1], c = l[2], d = l[3];
  if ((a.x ^ b.y ^ c.z ^ d.w) == 0x12345678u) out[0] = a.x;
}
__global__ __launch_bounds__(256) void k_cas(unsigned long long* t, uint32_t mask, unsigned long long* out, uint32_t salt) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long old = atomicCAS(&t[mix(i ^ salt) & mask], 0ull, static_cast<unsigned long long>(i) + 1ull);
  if (old == 0x123456789ull) out[0] = old;
}
__global__ __launch_bounds__(256) void k_line_cas(unsigned long long* t, uint32_t mask, unsigned long long* out, uint32_t salt) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t s = mix(i ^ salt) & mask & ~7u;
  const uint4* l = reinterpret_cast<const uint4*>(t + s);
  const uint4 a = l[0], b = l[1], c = l[2], d = l[3];
  const uint32_t k = (a.x ^ b.y ^ c.z ^ d.w) & 7u;
  const unsigned long long old = atomicCAS(&t[s + k], 0ull, static_cast<unsigned long long>(i) + 1ull);
  if (old == 0x123456789ull) out[0] = old;
}
__global__ __launch_bounds__(256) void k_min(unsigned long long* t, uint32_t mask, unsigned long long* out, uint32_t salt) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long old = atomicMin(&t[mix(i ^ salt) & mask], static_cast<unsigned long long>(i) + 1ull);
  if (old == 0x123456789ull) out[0] = old;
}

template <typename K>
static double rate(K kernel, unsigned long long* t, uint32_t mask, unsigned long long* out, uint32_t n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(n / 256), dim3(256), 0, 0, t, mask, out, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (uint32_t r = 0; r < 5; r++) hipLaunchKernelGGL(kernel, dim3(n / 256), dim3(256), 0, 0, t, mask, out, 77u + r);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return 5.0 * n / (ms * 1e-3) / 1e9;
}

int main() {
  const uint32_t n = 1u << 24;  // 16 M accesses per launch
  unsigned long long* out;
  hipMalloc(&out, 64);
  for (int lg : {24, 28}) {  // 2^24 slots = 128 MB, 2^28 = 2 GB
    const size_t slots = size_t(1) << lg;
    unsigned long long* t;
    if (hipMalloc(&t, slots * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(t, 0, slots * 8);
    const uint32_t mask = static_cast<uint32_t>(slots - 1);
    printf("table %4zu MB: load8 %6.1f G/s", slots * 8 >> 20, rate(k_load8, t, mask, out, n));
    printf("  line64 %6.1f G/s", rate(k_line, t, mask, out, n));
    hipMemset(t, 0, slots * 8);
    printf("  cas %6.1f G/s", rate(k_cas, t, mask, out, n));
    hipMemset(t, 0, slots * 8);
    printf("  line64+cas %6.1f G/s", rate(k_line_cas, t, mask, out, n));
    hipMemset(t, 0xff, slots * 8);
    printf("  min %6.1f G/s\n", rate(k_min, t, mask, out, n));
    hipFree(t);
  }
  return 0;
}
