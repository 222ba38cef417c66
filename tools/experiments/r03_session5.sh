#!/bin/bash
# GPU session 5 of round 3: the whole GPU suite, C5 / search rates, rocprofv3 profiles of the C3 and C4 hot paths.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/r03_gputest.txt 2>&1
timeout 400 python tools/bench_expand.py > $O/r03_expand4c.txt 2>&1
timeout 600 python tools/bench_search.py > $O/r03_search2.json 2> $O/r03_search2.err
timeout 900 python tools/experiments/step_tables_xp.py --modes none+fwd,auto > $O/r03_step_tables3.txt 2>&1
timeout 1500 bash tools/collect_profiles.sh r03 > $O/r03_collect.log 2>&1
timeout 400 python bench.py --config c4 --no-cpu-baseline > $O/r03_bench_c4_state.json 2> $O/r03_bench_c4_state.err
timeout 400 python bench.py --config c4 --obs uint8 --no-cpu-baseline > $O/r03_bench_c4_u8.json 2> $O/r03_bench_c4_u8.err
P=$O/prof_r03c4
mkdir -p $P
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o fetch -- python tools/profile_hotpath.py --config c4 --steps 6 > $P/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o write -- python tools/profile_hotpath.py --config c4 --steps 6 > $P/write.log 2>&1
for f in fetch write; do python tools/rocprof_summary.py $P/${f}_results.db > $P/${f}_summary.txt 2>&1; done
rm -f $P/*.db
tail -n 3 $O/r03_gputest.txt
