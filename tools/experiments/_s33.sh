cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -2; done
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['render_launch']['candidates_ms'])"; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
