#!/bin/bash
# GPU session 8 of round 3: actions prefetched in multi-step launches.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_soak.py tests/test_gpu_vector.py -m gpu -q > $O/r03_t_prefetch.txt 2>&1
timeout 900 python tools/experiments/step_tables_xp.py --modes auto > $O/r03_step_tables4.txt 2>&1
tail -n 3 $O/r03_t_prefetch.txt; cat $O/r03_step_tables4.txt
