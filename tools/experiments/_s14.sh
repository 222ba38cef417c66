cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/s14; mkdir -p $P
for k in auto never; do
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o fetch_$k -- python tools/bench_search.py --only Pull --keys $k > /dev/null 2> $P/fetch_$k.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o write_$k -- python tools/bench_search.py --only Pull --keys $k > /dev/null 2> $P/write_$k.log
for f in fetch write; do python tools/rocprof_summary.py $P/${f}_${k}_results.db 2>&1 | grep "pw_search.*_SIZE" | cut -c1-160 | sed "s/^/$k /"; done
done
rm -f $P/*.db
