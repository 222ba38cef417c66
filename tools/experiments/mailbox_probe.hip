// Round-5 probe for pw_mailbox_*: the floor of a host <-> resident-kernel round trip on this box.
//   one wavefront polls a 64-bit word and answers into pinned host memory; the host bumps the word and spins on the answer.
//   word in (a) pinned host memory (the GPU polls across the link), (b) fine-grained device memory written by the CPU through the
//   BAR (hipExtMallocWithFlags(hipDeviceMallocFinegrained); a forked child tries the CPU store first: no large BAR = SIGSEGV there),
//   (c) plain hipMalloc memory written by the CPU (same check).  N pollers for (a): 1, 4, 16, 64 wavefronts.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/bin/mailbox_probe tools/experiments/mailbox_probe.hip
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

__global__ void k_echo(const unsigned long long* word, unsigned long long* answer, unsigned long long* arrive, unsigned long long last, unsigned long long timeout_ticks) {
  if ((threadIdx.x & 63) != 0) return;
  const unsigned long long t0 = wall_clock64();
  const unsigned waves = gridDim.x * (blockDim.x >> 6);
  for (unsigned long long want = 1; want <= last; want++) {
    for (;;) {
      const unsigned long long w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (w >= want) break;
      if (wall_clock64() - t0 > timeout_ticks) return;
      __builtin_amdgcn_s_sleep(1);
    }
    if (waves == 1) {
      __hip_atomic_store(answer, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      const unsigned long long before = atomicAdd(arrive, 1ull);
      if (before + 1 == want * waves) __hip_atomic_store(answer, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static bool cpu_can_store(void* p) {  // in a child: a fault stays there
  fflush(stdout);
  const pid_t pid = fork();
  if (pid == 0) {
    *reinterpret_cast<volatile unsigned long long*>(p) = 0ull;
    _exit(*reinterpret_cast<volatile unsigned long long*>(p) == 0ull ? 0 : 1);
  }
  int st = 0;
  waitpid(pid, &st, 0);
  return WIFEXITED(st) && WEXITSTATUS(st) == 0;
}

static void run(const char* name, volatile unsigned long long* word_host, const unsigned long long* word_dev, int waves, int rounds) {
  unsigned long long *answer = nullptr, *answer_dev = nullptr, *arrive = nullptr;
  hipHostMalloc(reinterpret_cast<void**>(&answer), 64, hipHostMallocMapped);
  hipHostGetDevicePointer(reinterpret_cast<void**>(&answer_dev), answer, 0);
  hipMalloc(reinterpret_cast<void**>(&arrive), 64);
  hipMemset(arrive, 0, 64);
  *answer = 0;
  *word_host = 0;
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipDeviceSynchronize();
  const int blocks = waves >= 4 ? waves / 4 : 1, threads = waves >= 4 ? 256 : 64 * waves;
  hipLaunchKernelGGL(k_echo, dim3(blocks), dim3(threads), 0, st, word_dev, answer_dev, arrive, static_cast<unsigned long long>(rounds), 300000000ull /* 3 s at 100 MHz */);
  volatile unsigned long long* ans = answer;
  const auto t0 = std::chrono::steady_clock::now();
  bool ok = true;
  for (int r = 1; r <= rounds && ok; r++) {
    *word_host = static_cast<unsigned long long>(r);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    const auto s0 = std::chrono::steady_clock::now();
    while (*ans < static_cast<unsigned long long>(r)) {
      if ((r & 1023) == 0 && std::chrono::steady_clock::now() - s0 > std::chrono::seconds(2)) {
        ok = false;
        break;
      }
    }
  }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
  hipStreamSynchronize(st);
  printf("%-64s %3d poller(s): %s %.3f us per round trip\n", name, waves, ok ? "" : "(TIMED OUT)", us);
  hipStreamDestroy(st);
  hipFree(arrive);
  hipHostFree(answer);
}

int main() {
  const int rounds = 20000;
  unsigned long long *pin = nullptr, *pin_dev = nullptr;
  hipHostMalloc(reinterpret_cast<void**>(&pin), 64, hipHostMallocMapped);
  hipHostGetDevicePointer(reinterpret_cast<void**>(&pin_dev), pin, 0);
  for (int waves : {1, 4, 16, 64}) run("word in pinned host memory (GPU polls across the link)", pin, pin_dev, waves, rounds);
  void* fine = nullptr;
  if (hipExtMallocWithFlags(&fine, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
    if (cpu_can_store(fine)) {
      for (int waves : {1, 16, 64}) run("word in fine-grained DEVICE memory (CPU stores through the BAR)", static_cast<volatile unsigned long long*>(fine), static_cast<unsigned long long*>(fine), waves, rounds);
    } else {
      printf("fine-grained device memory: the CPU cannot store into it (no large BAR mapping)\n");
    }
  } else {
    printf("hipExtMallocWithFlags(hipDeviceMallocFinegrained) failed\n");
  }
  void* plain = nullptr;
  hipMalloc(&plain, 4096);
  if (cpu_can_store(plain)) {
    for (int waves : {1, 64}) run("word in plain hipMalloc memory (CPU stores through the BAR)", static_cast<volatile unsigned long long*>(plain), static_cast<unsigned long long*>(plain), waves, rounds);
  } else {
    printf("plain hipMalloc memory: the CPU cannot store into it\n");
  }
  return 0;
}
