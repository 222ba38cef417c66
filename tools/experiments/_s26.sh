cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_expand
mkdir -p $P
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $P -o sq1 -- python tools/bench_expand.py --sizes 1000000 > $P/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $P -o sq2 -- python tools/bench_expand.py --sizes 1000000 > $P/sq2.log 2>&1
for f in sq1 sq2; do
  python tools/rocprof_summary.py $P/${f}_results.db 2>&1 | grep -i "expand4_lane\|^kernel" > $P/${f}_summary.txt
done
rm -f $P/*.db
cat $P/sq1_summary.txt $P/sq2_summary.txt | cut -c1-200
