#!/bin/bash
# GPU session 11 of round 3: every observation setting with library-owned buffers, configs C1 / C2 / C4, bench sanity.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 900 python tools/render_sweep.py > $O/r03_render_sweep.txt 2>&1
timeout 900 python tools/bench_configs.py > $O/r03_configs.json 2> $O/r03_configs.err
timeout 600 python bench.py --no-cpu-baseline > $O/r03_bench_final3.json 2> $O/r03_bench_final3.err
timeout 300 python -m pytest tests/test_gpu_obs_alloc.py tests/test_gpu_bench.py -m gpu -q > $O/r03_t_last.txt 2>&1
cat $O/r03_render_sweep.txt; tail -n 3 $O/r03_t_last.txt; cut -c1-200 $O/r03_bench_final3.json
