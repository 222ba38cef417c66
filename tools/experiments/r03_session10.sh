#!/bin/bash
# GPU session 10 of round 3: final kernels (puzzle-id clamp, action prefetch): tests, profiles, bench.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gputest.txt 2>&1
timeout 1500 bash tools/collect_profiles.sh r03b > $O/r03b_collect.log 2>&1
cp $O/prof_r03b/pmc_render_latest.json profiles/pmc_render_latest.json
timeout 600 python bench.py > $O/r03_bench_final2.json 2> $O/r03_bench_final2.err
cp profiles/bench_n1_latest.json $O/bench_n1_latest.json 2>/dev/null
tail -n 3 $O/r03_gputest.txt; cut -c1-200 $O/r03_bench_final2.json; cat $O/prof_r03b/pmc_render_latest.json | head -30
