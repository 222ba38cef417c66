// What a clock read costs the host on the GPU box (round 6: pw_mailbox_post / mailbox_wait each called steady_clock::now() per step).
//   gcc -O2 -o tools/experiments/bin/clock_cost tools/experiments/clock_cost.c && tools/experiments/bin/clock_cost
#include <stdio.h>
#include <time.h>
static double run(clockid_t id, int n) {
  struct timespec a, b, t;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < n; i++) clock_gettime(id, &t);
  clock_gettime(CLOCK_MONOTONIC, &b);
  return ((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec)) / n;
}
int main(void) {
  printf("CLOCK_MONOTONIC        %.1f ns per call\n", run(CLOCK_MONOTONIC, 200000));
  printf("CLOCK_MONOTONIC_COARSE %.1f ns per call\n", run(CLOCK_MONOTONIC_COARSE, 200000));
  struct timespec r;
  clock_getres(CLOCK_MONOTONIC_COARSE, &r);
  printf("coarse resolution      %.3f ms\n", r.tv_nsec / 1e6);
  return 0;
}
