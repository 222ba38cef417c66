cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/s16; mkdir -p $P
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum --kernel-trace -d $P -o pmc2 -- python tools/bench_search.py --only Pull --keys auto > /dev/null 2> $P/pmc2.log
python tools/rocprof_summary.py $P/pmc2_results.db 2>&1 | grep "claim\|publish" | cut -c1-140; grep -i "error\|invalid\|not" $P/pmc2.log | head -5
rm -f $P/*.db
