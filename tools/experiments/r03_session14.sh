#!/bin/bash
# GPU session 14 of round 3: screened candidates (up to 12): tests, fresh-process statistics, bench.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_obs_alloc.py tests/test_gpu_bench.py tests/test_gpu_abi.py -m gpu -q > $O/r03_t_screen.txt 2>&1
timeout 900 python tools/experiments/obs_alloc_xp.py many c3 6 > $O/r03_many_c3b.txt 2>&1
timeout 600 python tools/experiments/obs_alloc_xp.py many c4 3 > $O/r03_many_c4b.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/r03_bench_final4.json 2> $O/r03_bench_final4.err
tail -n 3 $O/r03_t_screen.txt; grep -v amdgpu $O/r03_many_c3b.txt $O/r03_many_c4b.txt; cut -c1-200 $O/r03_bench_final4.json
