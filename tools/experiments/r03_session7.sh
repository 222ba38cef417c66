#!/bin/bash
# GPU session 7 of round 3: final code -- smoke, the whole GPU suite, the bench line (N = 1 record), two self-spawned ranks.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r03_smoke.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gputest.txt 2>&1
timeout 600 python bench.py > $O/r03_bench_final.json 2> $O/r03_bench_final.err
cp profiles/bench_n1_latest.json $O/bench_n1_latest.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --shared-device --no-cpu-baseline --no-extras > $O/r03_bench_n2_shared.json 2> $O/r03_bench_n2_shared.err
rocm-smi --showclocks --showpower --showperflevel > $O/r03_rocm_smi.txt 2>&1
tail -n 2 $O/r03_smoke.txt; tail -n 3 $O/r03_gputest.txt; cut -c1-300 $O/r03_bench_final.json
