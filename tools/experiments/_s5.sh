cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/s5; mkdir -p $P
timeout 900 python -m pytest tests/test_gpu_quad16.py -x -q > $P/quad.log 2>&1; tail -3 $P/quad.log
for set in l0 c4 hi c3; do for m in auto:groups never:groups; do
  rocprofv3 --kernel-trace --stats -d $P -o ${set}_${m/:/_} -- python tools/experiments/step_quad_xp.py --sets $set --modes $m > $P/${set}_${m/:/_}.log 2>&1
  python tools/rocprof_summary.py $P/${set}_${m/:/_}_results.db 2>&1 | grep "pw_step\|pw_rollout" | sed "s/^/$set $m  /"
done; done
rm -f $P/*.db
