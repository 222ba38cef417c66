cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/s13; mkdir -p $P
for k in auto never; do
rocprofv3 --kernel-trace --stats -d $P -o trace_$k -- python tools/bench_search.py --only Pull --keys $k > $P/bench_$k.json 2> $P/trace_$k.log
python tools/rocprof_summary.py $P/trace_${k}_results.db 2>&1 | grep "pw_search" | cut -c1-140 | sed "s/^/$k /"
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --kernel-trace -d $P -o pmc -- python tools/bench_search.py --only Pull --keys auto > /dev/null 2> $P/pmc.log
python tools/rocprof_summary.py $P/pmc_results.db 2>&1 | grep "claim" | cut -c1-140
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_ATOMIC_sum --kernel-trace -d $P -o pmc2 -- python tools/bench_search.py --only Pull --keys auto > /dev/null 2> $P/pmc2.log
python tools/rocprof_summary.py $P/pmc2_results.db 2>&1 | grep "claim" | cut -c1-140; tail -2 $P/pmc2.log
rm -f $P/*.db
