#!/bin/bash
# GPU session 9 of round 3: kernel trace of the GPU search with the final code.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_search -o search -- python tools/bench_search.py --max-states 20000000 > $O/r03_search_trace2.log 2>&1
python tools/rocprof_summary.py $O/prof_search/search_results.db > $O/r03_search_trace2.txt 2>&1
rm -rf $O/prof_search
head -14 $O/r03_search_trace2.txt | cut -c1-150
