cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_c4n
mkdir -p $P
rocprofv3 --kernel-trace --stats -d $P -o c4_trace -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 > $P/c4_trace.log 2>&1
python tools/rocprof_summary.py $P/c4_trace_results.db 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd" > $P/c4_trace_summary.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace -d $P -o c4_sq2 -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 > $P/c4_sq2.log 2>&1
python tools/rocprof_summary.py $P/c4_sq2_results.db 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd" > $P/c4_sq2_summary.txt
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace -d $P -o c4_sq1 -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 > $P/c4_sq1.log 2>&1
python tools/rocprof_summary.py $P/c4_sq1_results.db 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd" > $P/c4_sq1_summary.txt
rm -f $P/*.db
cat $P/c4_trace_summary.txt; grep "mixed" $P/c4_sq2_summary.txt $P/c4_sq1_summary.txt
