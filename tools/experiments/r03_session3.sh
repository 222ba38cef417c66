#!/bin/bash
# GPU session 3 of round 3: pair probe (time-boxed forensics), allocator tests, table-only kernels A/B, bench line.
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
P=tools/experiments/bin/pair_probe
{
  echo "=== pair_probe 24 chunks, 600 sets, 8 sweeps, no LDS pad"; timeout 120 $P 24 600 8 0
  echo "=== pair_probe 24 chunks, 600 sets, 8 sweeps, 7 KB LDS pad"; timeout 120 $P 24 600 8 7168
  echo "=== pair_probe 24 chunks spread with 3 GB spacers"; timeout 120 $P 24 400 8 0 3000
} > $O/r03_pair_probe.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_obs_alloc.py -m gpu -q > $O/r03_t_alloc.txt 2>&1
timeout 900 python tools/experiments/step_tables_xp.py > $O/r03_step_tables.txt 2>&1
timeout 400 python tools/bench_expand.py > $O/r03_expand4.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_expand.py tests/test_gpu_shapes.py tests/test_gpu_search.py tests/test_gpu_abi.py tests/test_gpu_vector.py -m gpu -q > $O/r03_t_tables.txt 2>&1
timeout 600 python bench.py > $O/r03_bench_a.json 2> $O/r03_bench_a.err
timeout 900 python tools/experiments/obs_alloc_xp.py many c3 6 > $O/r03_many_c3.txt 2>&1
timeout 600 python tools/experiments/obs_alloc_xp.py many c4 4 > $O/r03_many_c4.txt 2>&1
tail -n 3 $O/r03_t_alloc.txt $O/r03_t_tables.txt
