// Experiment: write bandwidth of ONE XCD (workgroups with blockIdx % 8 == xcd; the others exit at once) into one region of
// a buffer, non-temporal 16-byte stores, one single-wavefront workgroup per 4 KiB page -- the store pattern of the
// page-ordered render kernel.  Is there an XCD x region structure behind the allocation classes?
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/experiments/xcd_probe.hip -o tools/experiments/bin/libxcdprobe.so
#include <hip/hip_runtime.h>

#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void probe_kernel(uint8_t* base, uint32_t pages, uint32_t xcd_mask) {
  const uint32_t k = blockIdx.x & 7u;
  if (!((xcd_mask >> k) & 1u)) return;
  // the active XCDs share the region's pages round-robin
  const uint32_t n_active = __builtin_popcount(xcd_mask), rank = __builtin_popcount(xcd_mask & ((1u << k) - 1u));
  const uint32_t page = (blockIdx.x >> 3) * n_active + rank;
  if (page >= pages) return;
  u32x4* p = reinterpret_cast<u32x4*>(base + static_cast<size_t>(page) * 4096) + threadIdx.x;
  const u32x4 v = {page, k, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 4; i++) __builtin_nontemporal_store(v, p + 64 * i);
}

// every XCD k with pages[k] > 0 writes its OWN region base[k] sequentially (the contiguous-parts page orders by hand)
struct MultiArgs {
  uint8_t* base[8];
  uint32_t pages[8];
};

__global__ __launch_bounds__(64) void probe_multi_kernel(MultiArgs a) {
  const uint32_t k = blockIdx.x & 7u, page = blockIdx.x >> 3;
  if (page >= a.pages[k]) return;
  u32x4* p = reinterpret_cast<u32x4*>(a.base[k] + static_cast<size_t>(page) * 4096) + threadIdx.x;
  const u32x4 v = {page, k, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 4; i++) __builtin_nontemporal_store(v, p + 64 * i);
}

// P parts of `per` pages each; XCD k owns the parts k, k + 8, ... and interleaves them page by page (P = 8: the eighths
// order; P = 64: every XCD keeps 8 write fronts going)
__global__ __launch_bounds__(64) void probe_parts_kernel(uint8_t* base, uint32_t per, uint32_t m, uint32_t n_pages) {
  const uint32_t k = blockIdx.x & 7u, j = blockIdx.x >> 3;  // j-th page of XCD k
  const uint32_t part = k + 8u * (j % m), off = j / m;
  const uint32_t page = part * per + off;
  if (off >= per || page >= n_pages) return;
  u32x4* p = reinterpret_cast<u32x4*>(base + static_cast<size_t>(page) * 4096) + threadIdx.x;
  const u32x4 v = {page, k, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 4; i++) __builtin_nontemporal_store(v, p + 64 * i);
}

extern "C" int xcd_probe_parts_per(void* base, uint32_t n_pages, uint32_t parts, uint32_t per, int reps, int lds_pad, float* ms_out);

extern "C" int xcd_probe_parts(void* base, uint32_t n_pages, uint32_t parts, int reps, int lds_pad, float* ms_out) {
  return xcd_probe_parts_per(base, n_pages, parts, (n_pages + parts - 1) / parts, reps, lds_pad, ms_out);
}

// the same with a chosen part size (pages); pages beyond parts * per are not written (callers keep the loss tiny)
extern "C" int xcd_probe_parts_per(void* base, uint32_t n_pages, uint32_t parts, uint32_t per, int reps, int lds_pad, float* ms_out) {
  if (parts < 8 || parts % 8) return -1;
  const uint32_t m = parts / 8;
  const uint32_t grid = per * m * 8;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe_parts_kernel, dim3(grid), dim3(64), lds_pad, 0, static_cast<uint8_t*>(base), per, m, n_pages);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL(probe_parts_kernel, dim3(grid), dim3(64), lds_pad, 0, static_cast<uint8_t*>(base), per, m, n_pages);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// the eighths order as a COPY: every page is first read from a small, cache-resident source (`src_pages` pages, wrapped),
// like the render kernel reads its static images, then stored; mode 1: load all four chunks, then store; mode 2: two
// pages per workgroup, the second page's loads issued before the first page's stores
__global__ __launch_bounds__(64) void probe_copy_kernel(uint8_t* base, const uint8_t* src, uint32_t src_pages, uint32_t per,
                                                        uint32_t n_pages, int mode) {
  const uint32_t k = blockIdx.x & 7u, j = blockIdx.x >> 3;
  if (mode == 2) {
    const uint32_t o0 = 2 * j, o1 = 2 * j + 1;
    if (o0 >= per) return;
    const uint32_t p0 = k * per + o0, p1 = k * per + o1;
    if (p0 >= n_pages) return;
    const bool two = o1 < per && p1 < n_pages;
    const u32x4* s0 = reinterpret_cast<const u32x4*>(src + static_cast<size_t>(p0 % src_pages) * 4096) + threadIdx.x;
    const u32x4* s1 = reinterpret_cast<const u32x4*>(src + static_cast<size_t>(p1 % src_pages) * 4096) + threadIdx.x;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = s0[64 * i];
    if (two) {
#pragma unroll
      for (int i = 0; i < 4; i++) b[i] = s1[64 * i];
    }
    u32x4* d0 = reinterpret_cast<u32x4*>(base + static_cast<size_t>(p0) * 4096) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; i++) __builtin_nontemporal_store(a[i], d0 + 64 * i);
    if (two) {
      u32x4* d1 = reinterpret_cast<u32x4*>(base + static_cast<size_t>(p1) * 4096) + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 4; i++) __builtin_nontemporal_store(b[i], d1 + 64 * i);
    }
    return;
  }
  if (j >= per) return;
  const uint32_t page = k * per + j;
  if (page >= n_pages) return;
  const u32x4* s = reinterpret_cast<const u32x4*>(src + static_cast<size_t>(page % src_pages) * 4096) + threadIdx.x;
  u32x4 v[4];
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = s[64 * i];
  u32x4* d = reinterpret_cast<u32x4*>(base + static_cast<size_t>(page) * 4096) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; i++) __builtin_nontemporal_store(v[i], d + 64 * i);
}

extern "C" int xcd_probe_copy(void* base, const void* src, uint32_t src_pages, uint32_t n_pages, int mode, int reps,
                              int lds_pad, float* ms_out) {
  const uint32_t per = (n_pages + 7) / 8;
  const uint32_t grid = (mode == 2 ? (per + 1) / 2 : per) * 8;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe_copy_kernel, dim3(grid), dim3(64), lds_pad, 0, static_cast<uint8_t*>(base),
                     static_cast<const uint8_t*>(src), src_pages, per, n_pages, mode);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL(probe_copy_kernel, dim3(grid), dim3(64), lds_pad, 0, static_cast<uint8_t*>(base),
                       static_cast<const uint8_t*>(src), src_pages, per, n_pages, mode);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int xcd_probe_multi(void* const* base, const uint32_t* pages, int reps, int lds_pad, float* ms_out) {
  MultiArgs a;
  uint32_t most = 0;
  for (int k = 0; k < 8; k++) {
    a.base[k] = static_cast<uint8_t*>(base[k]);
    a.pages[k] = pages[k];
    most = pages[k] > most ? pages[k] : most;
  }
  if (!most) return -1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe_multi_kernel, dim3(most * 8), dim3(64), lds_pad, 0, a);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(probe_multi_kernel, dim3(most * 8), dim3(64), lds_pad, 0, a);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int xcd_probe(void* base, size_t region_bytes, uint32_t xcd_mask, int reps, float* ms_out) {
  const uint32_t pages = static_cast<uint32_t>(region_bytes / 4096);
  const uint32_t n_active = __builtin_popcount(xcd_mask);
  if (!n_active || !pages) return -1;
  const uint32_t groups = (pages + n_active - 1) / n_active;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipLaunchKernelGGL(probe_kernel, dim3(groups * 8), dim3(64), 7168, 0, static_cast<uint8_t*>(base), pages, xcd_mask);
  (void)hipEventRecord(a, 0);
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL(probe_kernel, dim3(groups * 8), dim3(64), 7168, 0, static_cast<uint8_t*>(base), pages, xcd_mask);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  *ms_out = ms / reps;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
