#!/bin/bash
# GPU session 16 of round 3: final code (8-lane groups for single steps too): the whole GPU suite, smoke, bench.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r03_smoke.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gputest.txt 2>&1
timeout 600 python bench.py > $O/r03_bench_final5.json 2> $O/r03_bench_final5.err
cp profiles/bench_n1_latest.json $O/bench_n1_latest.json 2>/dev/null
tail -n 1 $O/r03_smoke.txt; tail -n 3 $O/r03_gputest.txt; cut -c1-200 $O/r03_bench_final5.json
