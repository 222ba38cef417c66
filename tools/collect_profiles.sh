#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the C3 hot path.
# Usage: tools/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/*.txt (+ .db)
# Counters are collected in their own runs with --kernel-trace only (never with sys/hip traces).
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_$TAG
mkdir -p $P
python bench.py --steps 50 --warmup 5 > $P/bench.json 2> $P/bench.err
rocprofv3 --kernel-trace --stats -d $P -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $P/trace_bench_line.json 2> $P/trace.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $P -o sq1 -- python tools/profile_hotpath.py --steps 6 > $P/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $P -o sq2 -- python tools/profile_hotpath.py --steps 6 > $P/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o fetch -- python tools/profile_hotpath.py --steps 6 > $P/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o write -- python tools/profile_hotpath.py --steps 6 > $P/write.log 2>&1
for f in trace sq1 sq2 fetch write; do
  python tools/rocprof_summary.py $P/${f}_results.db > $P/${f}_summary.txt 2>&1
done
python tools/make_pmc_record.py $P/fetch_results.db $P/write_results.db pw_render_page_kernel 65536 57834 "tools/collect_profiles.sh $TAG" > $P/pmc_render_latest.json 2> $P/pmc_record.err
rm -f $P/*.db
tail -1 $P/bench.json | cut -c1-400
