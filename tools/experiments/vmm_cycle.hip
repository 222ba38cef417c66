// Lifecycle probe of the HIP virtual-memory API on the GPU box: which call patterns are safe and how long they take.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/vmm_cycle.hip -o tools/experiments/bin/vmm_cycle
//   vmm_cycle <bytes_mb> <chunk_mb> <per_chunk_access 0|1> <per_chunk_unmap 0|1> <free_va 0|1> <rounds>
// Every round: allocate A and B (both alive), write both with a kernel, free A, allocate C, write C, free B, free C.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      printf("FAILED %s: %s\n", #x, hipGetErrorString(e_));                                     \
      fflush(stdout);                                                                           \
      exit(2);                                                                                  \
    }                                                                                           \
  } while (0)

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Buf {
  void* ptr = nullptr;
  size_t bytes = 0, chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> h;
};

__global__ void fill(uint4* p, size_t n, unsigned v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, v, v, v);
}
__global__ void check(const uint4* p, size_t n, unsigned v, unsigned* bad) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (p[i].x != v || p[i].w != v) atomicAdd(bad, 1u);
}

static bool g_per_access, g_per_unmap, g_free_va;

static Buf alloc(size_t bytes, size_t chunk) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  chunk = (chunk + gran - 1) / gran * gran;
  Buf b;
  b.chunk = chunk;
  const size_t n = (bytes + chunk - 1) / chunk;
  b.bytes = n * chunk;
  double t0 = now();
  CK(hipMemAddressReserve(&b.ptr, b.bytes, 0, nullptr, 0));
  double t1 = now();
  b.h.resize(n);
  for (size_t i = 0; i < n; i++) CK(hipMemCreate(&b.h[i], chunk, &prop, 0));
  double t2 = now();
  for (size_t i = 0; i < n; i++) CK(hipMemMap((char*)b.ptr + i * chunk, chunk, 0, b.h[i], 0));
  double t3 = now();
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = 0;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (g_per_access) {
    for (size_t i = 0; i < n; i++) CK(hipMemSetAccess((char*)b.ptr + i * chunk, chunk, &acc, 1));
  } else {
    CK(hipMemSetAccess(b.ptr, b.bytes, &acc, 1));
  }
  double t4 = now();
  printf("  alloc %p %zu chunks: reserve %.3f s, create %.3f s, map %.3f s, access %.3f s\n", b.ptr, n, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
  fflush(stdout);
  return b;
}

static void release(Buf& b) {
  double t0 = now();
  CK(hipDeviceSynchronize());
  if (g_per_unmap) {
    for (size_t i = 0; i < b.h.size(); i++) CK(hipMemUnmap((char*)b.ptr + i * b.chunk, b.chunk));
  } else {
    CK(hipMemUnmap(b.ptr, b.bytes));
  }
  double t1 = now();
  for (auto h : b.h) CK(hipMemRelease(h));
  double t2 = now();
  if (g_free_va) CK(hipMemAddressFree(b.ptr, b.bytes));
  printf("  free  %p: unmap %.3f s, release %.3f s%s\n", b.ptr, t1 - t0, t2 - t1, g_free_va ? ", va freed" : "");
  fflush(stdout);
  b.ptr = nullptr;
}

static void touch(Buf& b, unsigned v, unsigned* d_bad) {
  double t0 = now();
  fill<<<4096, 256>>>((uint4*)b.ptr, b.bytes / 16, v);
  check<<<4096, 256>>>((const uint4*)b.ptr, b.bytes / 16, v, d_bad);
  CK(hipDeviceSynchronize());
  unsigned bad = 0;
  CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
  printf("  write+check %p: %.3f s, bad %u\n", b.ptr, now() - t0, bad);
  fflush(stdout);
}

int main(int argc, char** argv) {
  if (argc < 7) return 1;
  const size_t bytes = (size_t)atol(argv[1]) << 20, chunk = (size_t)atol(argv[2]) << 20;
  g_per_access = atoi(argv[3]);
  g_per_unmap = atoi(argv[4]);
  g_free_va = atoi(argv[5]);
  const int rounds = atoi(argv[6]);
  unsigned* d_bad;
  CK(hipMalloc(&d_bad, 4));
  CK(hipMemset(d_bad, 0, 4));
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("bytes %zu MB chunk %zu MB per_access %d per_unmap %d free_va %d; hipMemGetInfo free %zu MB total %zu MB\n", bytes >> 20,
         chunk >> 20, g_per_access, g_per_unmap, g_free_va, fr >> 20, tot >> 20);
  for (int r = 0; r < rounds; r++) {
    printf("round %d\n", r);
    fflush(stdout);
    Buf A = alloc(bytes, chunk);
    Buf B = alloc(bytes, chunk);
    touch(A, 1 + r, d_bad);
    touch(B, 100 + r, d_bad);
    CK(hipMemGetInfo(&fr, &tot));
    printf("  hipMemGetInfo with two buffers: free %zu MB\n", fr >> 20);
    release(A);
    Buf C = alloc(bytes, chunk);
    touch(C, 200 + r, d_bad);
    touch(B, 300 + r, d_bad);
    release(B);
    release(C);
    CK(hipMemGetInfo(&fr, &tot));
    printf("  hipMemGetInfo after the round: free %zu MB\n", fr >> 20);
  }
  printf("done\n");
  return 0;
}
