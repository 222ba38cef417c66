#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + HBM traffic counters (separate --pmc passes, never with sys/hip traces) of the
# state-only stepping of BOUND batches (pw_batch_bind): C4 shard and C3 set, single steps and 64-step launches.
# Usage: tools/collect_bind_profiles.sh <tag>   -> gpurun_out/prof_<tag>/bind_*
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_$TAG
mkdir -p $P
T="timeout 300"
summ() { python tools/rocprof_summary.py "$1" 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd\|elementwise\|vectorized"; }
for c in c4 c3; do
  for v in bound unbound; do
    $T rocprofv3 --kernel-trace --stats -d $P -o bind_${c}_${v}_trace -- python tools/bench_bind.py --config $c --variants $v > $P/bind_${c}_${v}_trace.log 2>&1
    $T rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o bind_${c}_${v}_fetch -- python tools/bench_bind.py --config $c --variants $v > $P/bind_${c}_${v}_fetch.log 2>&1
    $T rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o bind_${c}_${v}_write -- python tools/bench_bind.py --config $c --variants $v > $P/bind_${c}_${v}_write.log 2>&1
    for f in trace fetch write; do summ $P/bind_${c}_${v}_${f}_results.db > $P/bind_${c}_${v}_${f}_summary.txt; done
  done
done
grep -h "pw_step\|pw_bind" $P/bind_*_summary.txt | cut -c1-200
