cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/s11; mkdir -p $P
timeout 1200 python -m pytest tests/test_gpu_search.py -x -q > $P/search_tests.log 2>&1; tail -4 $P/search_tests.log
for k in auto never; do python tools/bench_search.py --keys $k --only "Pull" 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); [print('$k', r, v['status'], v['states'], '%.3g parents/s' % v['parents_per_s']) for r,v in d.items()]"; done
python tools/bench_search.py --only "Four" 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); [print(r, v['status'], v['states'], '%.3g parents/s' % v['parents_per_s']) for r,v in d.items()]"
