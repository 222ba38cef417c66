#!/bin/bash
# GPU session 4 of round 3: allocator tests (fresh address ranges), table policy A/B, search with fingerprints.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_obs_alloc.py -m gpu -q > $O/r03_t_alloc.txt 2>&1
timeout 900 python tools/experiments/step_tables_xp.py > $O/r03_step_tables2.txt 2>&1
timeout 400 python tools/bench_expand.py > $O/r03_expand4b.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_expand.py tests/test_gpu_parity.py -m gpu -q > $O/r03_t_search.txt 2>&1
timeout 600 python tools/bench_search.py > $O/r03_search.json 2> $O/r03_search.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_search -o search -- python tools/bench_search.py --max-states 20000000 > $O/r03_search_trace.log 2>&1
python tools/rocprof_summary.py $O/prof_search/search_results.db > $O/r03_search_trace.txt 2>&1
rm -rf $O/prof_search
tail -n 3 $O/r03_t_alloc.txt $O/r03_t_search.txt
