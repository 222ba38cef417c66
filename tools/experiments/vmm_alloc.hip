// Experiment: does the physical backing decide the "allocation class" of the page-ordered render kernel?
// A buffer built with the HIP virtual-memory API: one address range, physical memory created in chunks and mapped
// in a chosen order.  mode 0: one chunk (like hipMalloc); 1: chunks mapped in creation order; 2: chunks mapped in a
// pseudo-random permutation; 3: every second chunk first, then the others (two interleaved physical runs).
//   hipcc --offload-arch=gfx950 -shared -fPIC tools/experiments/vmm_alloc.hip -o tools/experiments/bin/libvmm.so
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

struct VmmBuf {
  void* ptr;
  size_t bytes;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};

extern "C" {

int vmm_alloc(int device, size_t bytes, size_t chunk, int mode, uint64_t seed, void** out_ptr, void** out_handle) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) return -1;
  if (mode == 0 || chunk == 0) chunk = bytes;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;
  VmmBuf* b = new VmmBuf();
  b->bytes = total;
  if (hipMemAddressReserve(&b->ptr, total, 0, nullptr, 0) != hipSuccess) return -2;
  b->handles.resize(n);
  for (size_t i = 0; i < n; i++)
    if (hipMemCreate(&b->handles[i], chunk, &prop, 0) != hipSuccess) return -3;
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = i;
  if (mode == 2) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (size_t i = n - 1; i > 0; i--) {
      s ^= s << 13;
      s ^= s >> 7;
      s ^= s << 17;
      std::swap(order[i], order[s % (i + 1)]);
    }
  } else if (mode == 4) {  // reverse creation order
    for (size_t i = 0; i < n; i++) order[i] = n - 1 - i;
  } else if (mode == 3) {
    size_t k = 0;
    for (size_t i = 0; i < n; i += 2) order[k++] = i;
    for (size_t i = 1; i < n; i += 2) order[k++] = i;
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (size_t i = 0; i < n; i++) {  // virtual slot i <- physical chunk order[i]; access is granted per mapping
    char* at = static_cast<char*>(b->ptr) + i * chunk;
    if (hipMemMap(at, chunk, 0, b->handles[order[i]], 0) != hipSuccess) return -4;
    if (hipMemSetAccess(at, chunk, &acc, 1) != hipSuccess) return -5;
  }
  if (hipDeviceSynchronize() != hipSuccess) return -6;
  *out_ptr = b->ptr;
  *out_handle = b;
  return static_cast<int>(n);
}

void vmm_free(void* handle) {
  VmmBuf* b = static_cast<VmmBuf*>(handle);
  if (!b) return;
  (void)hipMemUnmap(b->ptr, b->bytes);
  for (auto h : b->handles) (void)hipMemRelease(h);
  (void)hipMemAddressFree(b->ptr, b->bytes);
  delete b;
}

}  // extern "C"
