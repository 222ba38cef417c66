#!/bin/bash
# FETCH_SIZE / WRITE_SIZE records of the C4 shard's step kernels only, merged into profiles/pmc_kernels_latest.json (same kernel source)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_r04c4
mkdir -p $P
rocprofv3 --kernel-trace --stats -d $P -o c4_trace -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 > $P/c4_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o c4_fetch -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 > $P/c4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o c4_write -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 > $P/c4_write.log 2>&1
for f in trace fetch write; do python tools/rocprof_summary.py $P/c4_${f}_results.db 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd" > $P/c4_${f}_summary.txt; done
python tools/make_kernel_pmc_record.py "tools/collect_profiles.sh r04" --merge=profiles/pmc_kernels_latest.json \
  "C4_state:pw_step_group_mixed_kernel<true,:65536:$P/c4_fetch_results.db:$P/c4_write_results.db" \
  "C4_rollout:pw_step_group_mixed_kernel<false,:4194304:$P/c4_fetch_results.db:$P/c4_write_results.db" \
  > $P/pmc_kernels_latest.json 2> $P/pmc_kernels.err
cat $P/pmc_kernels.err
rm -f $P/*.db
head -8 $P/c4_trace_summary.txt
