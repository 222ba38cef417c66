#!/bin/bash
# kernel shares of the 40 M-state searches of tools/bench_search.py (round 4)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_search_big
mkdir -p $P
rocprofv3 --kernel-trace --stats -d $P -o trace -- python tools/bench_search.py > $P/bench_search.json 2> $P/trace.log
python tools/rocprof_summary.py $P/trace_results.db 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd" > $P/trace_summary.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o fetch -- python tools/bench_search.py > /dev/null 2> $P/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o write -- python tools/bench_search.py > /dev/null 2> $P/write.log
for f in fetch write; do python tools/rocprof_summary.py $P/${f}_results.db 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd" > $P/${f}_summary.txt; done
rm -f $P/*.db
cat $P/trace_summary.txt
