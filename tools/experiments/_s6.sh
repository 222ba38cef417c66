cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/s6; mkdir -p $P
for B in 16384 65536 262144; do for set in hi l0; do
  rocprofv3 --kernel-trace --stats -d $P -o ${set}_$B -- python tools/experiments/step_quad_xp.py --sets $set --modes auto:lanes --envs $B > $P/${set}_$B.log 2>&1
  python tools/rocprof_summary.py $P/${set}_${B}_results.db 2>&1 | grep "pw_step\|pw_rollout" | sed "s/^/$set $B lanes  /"
done; done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace -d $P -o hi_pmc -- python tools/experiments/step_quad_xp.py --sets hi --modes auto:lanes > $P/hi_pmc.log 2>&1
python tools/rocprof_summary.py $P/hi_pmc_results.db 2>&1 | grep "lane_kernel"
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_UTCL1_TRANSLATION_MISS_sum --kernel-trace -d $P -o hi_pmc2 -- python tools/experiments/step_quad_xp.py --sets hi --modes auto:lanes > $P/hi_pmc2.log 2>&1
python tools/rocprof_summary.py $P/hi_pmc2_results.db 2>&1 | grep "lane_kernel"; tail -3 $P/hi_pmc2.log
rm -f $P/*.db
