#!/bin/bash
# usage: tools/ab_bench.sh [N]   -- N default bench runs, prints value / ms_per_step / render avg ms
for i in $(seq 1 ${1:-3}); do python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
