cd "$GRAFT_REPO_ROOT"
for a in "--max-states 4000000" "--max-states 40000000 --stop-at 3500000" "--max-states 12000000" "--max-states 40000000 --stop-at 11000000"; do
python tools/bench_search.py --only Pull $a 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); [print('$a', v['status'], v['states'], v['parents_expanded'], '%.3g parents/s' % v['parents_per_s']) for r,v in d.items()]"; done
