cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/s4; mkdir -p $P
timeout 600 python -m pytest tests/test_gpu_quad16.py -x -q > $P/quad.log 2>&1; tail -3 $P/quad.log
rocprofv3 --kernel-trace --stats -d $P -o lanes -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 --lanes > $P/lanes.log 2>&1
rocprofv3 --kernel-trace --stats -d $P -o groups -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 > $P/groups.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $P -o lanes_sq -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 --lanes > $P/lanes_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH --kernel-trace -d $P -o lanes_sq2 -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 --lanes > $P/lanes_sq2.log 2>&1
for f in lanes groups lanes_sq lanes_sq2; do python tools/rocprof_summary.py $P/${f}_results.db 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd" > $P/${f}_summary.txt; done
rm -f $P/*.db
cat $P/lanes_summary.txt $P/groups_summary.txt | grep "pw_step\|pw_rollout\|kernel " ; cat $P/lanes_sq_summary.txt $P/lanes_sq2_summary.txt | grep "lane_kernel"
