cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/s7; mkdir -p $P
timeout 900 python -m pytest tests/test_gpu_quad16.py -x -q > $P/quad.log 2>&1; tail -3 $P/quad.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lane or big" > $P/parity.log 2>&1; tail -3 $P/parity.log
for set in hi l0 c4 c3; do
  rocprofv3 --kernel-trace --stats -d $P -o ${set} -- python tools/experiments/step_quad_xp.py --sets $set --modes auto:lanes > $P/${set}.log 2>&1
  python tools/rocprof_summary.py $P/${set}_results.db 2>&1 | grep "pw_step\|pw_rollout" | sed "s/^/$set lanes  /"
done
rm -f $P/*.db
