#!/bin/bash
# GPU session 15 of round 3: search with wide slots: tests, rates, trace.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_generate.py tests/test_gpu_expand.py -m gpu -q > $O/r03_t_search2.txt 2>&1
timeout 600 python tools/bench_search.py > $O/r03_search3.json 2> $O/r03_search3.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_search -o search -- python tools/bench_search.py --max-states 20000000 > $O/r03_search_trace3.log 2>&1
python tools/rocprof_summary.py $O/prof_search/search_results.db > $O/r03_search_trace3.txt 2>&1
rm -rf $O/prof_search
tail -n 3 $O/r03_t_search2.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_search3.json'))
for k,v in d.items(): print(k, v['status'], v['states'], '%.3e'%v['parents_per_s'])
PY
head -10 $O/r03_search_trace3.txt | cut -c1-140
