cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/s12; mkdir -p $P
for k in auto never; do
rocprofv3 --kernel-trace --stats -d $P -o trace_$k -- python tools/bench_search.py --only Pull --keys $k > $P/bench_$k.json 2> $P/trace_$k.log
python tools/rocprof_summary.py $P/trace_${k}_results.db 2>&1 | grep "pw_search" | cut -c1-140 | sed "s/^/$k /"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o fetch_$k -- python tools/bench_search.py --only Pull --keys $k > /dev/null 2> $P/fetch_$k.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o write_$k -- python tools/bench_search.py --only Pull --keys $k > /dev/null 2> $P/write_$k.log
for f in fetch write; do python tools/rocprof_summary.py $P/${f}_${k}_results.db 2>&1 | grep "pw_search.*_SIZE" | cut -c1-160 | sed "s/^/$k /"; done
done
rm -f $P/*.db
