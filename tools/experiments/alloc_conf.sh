#!/bin/bash
# Which of k consecutive 3.8 GB allocations of a fresh process the render kernel is fast on (bench.py's
# config.render_launch.candidates_ms), several processes on one box.
cd $GRAFT_REPO_ROOT
K=${1:-16}
for i in 1 2 3 4 5; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 1 --tune-allocations $K 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('process $i: %.4e env-steps/s ' % d['value'], d['config']['render_launch']['candidates_ms'])"
done
