#!/bin/bash
# GPU session 2 of round 3: VMM lifecycle probe, allocator tests + lottery statistics, overlap-table A/B.
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
V=tools/experiments/bin/vmm_cycle
{
  for args in "474 2 0 0 0 3" "474 2 1 1 0 3" "3800 32 1 1 0 3" "3800 32 0 0 0 2" "3800 32 1 1 1 3" "3800 2 1 1 0 1" "3800 2 0 0 0 1"; do
    echo "=== vmm_cycle $args"
    timeout 150 $V $args 2>&1 | tail -40
    echo "rc ${PIPESTATUS[0]}"
  done
} > $O/r03_vmm_cycle.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_obs_alloc.py -m gpu -q > $O/r03_t_alloc.txt 2>&1
timeout 900 python tools/experiments/obs_alloc_xp.py many c3 8 > $O/r03_many_c3.txt 2>&1
timeout 600 python tools/experiments/obs_alloc_xp.py many c4 5 > $O/r03_many_c4.txt 2>&1
timeout 900 python tools/experiments/step_tables_xp.py > $O/r03_step_tables.txt 2>&1
timeout 400 python tools/bench_expand.py > $O/r03_expand4.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_expand.py tests/test_gpu_shapes.py tests/test_gpu_search.py tests/test_gpu_abi.py -m gpu -q > $O/r03_t_tables.txt 2>&1
tail -3 $O/r03_t_alloc.txt $O/r03_t_tables.txt
