#!/bin/bash
# usage: tools/ab_compare.sh LIB_A LIB_B [ROUNDS] -- interleaved bench runs of two builds of the library
# (box-to-box and minute-to-minute drift is larger than most kernel tweaks; only interleaved medians count)
A=$1; B=$2; N=${3:-5}
run() { PUSHWORLD_AMD_LIB=$1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
for i in $(seq 1 $N); do run $A A; run $B B; done | tee /tmp/ab.txt
python - <<'PY'
import statistics as st
r = {"A": [], "B": []}
for l in open("/tmp/ab.txt"):
    k, ms, rd = l.split()
    r[k].append((float(ms), float(rd)))
for k, v in r.items():
    print(k, "median ms/step %.4f  render %.4f   min %.4f / %.4f" % (st.median(x[0] for x in v), st.median(x[1] for x in v), min(x[0] for x in v), min(x[1] for x in v)))
PY
