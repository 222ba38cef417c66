set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_x1
mkdir -p $P
python -m pytest tests/test_gpu_search.py -x -q -k "batch" > $P/t_batch.log 2>&1; tail -5 $P/t_batch.log
for pz in "level4/Four Pistons.pwp" "level2/Pull Dont Push.pwp"; do
  tag=$(echo "$pz" | tr ' /.' '___')
  rocprofv3 --kernel-trace --stats -d $P -o tr_$tag -- python tools/profile_kernels.py --what expand --puzzle "$pz" --steps 12 > $P/tr_$tag.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $P -o sq1_$tag -- python tools/profile_kernels.py --what expand --puzzle "$pz" --steps 6 > $P/sq1_$tag.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $P -o sq2_$tag -- python tools/profile_kernels.py --what expand --puzzle "$pz" --steps 6 > $P/sq2_$tag.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o fetch_$tag -- python tools/profile_kernels.py --what expand --puzzle "$pz" --steps 6 > $P/fetch_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o write_$tag -- python tools/profile_kernels.py --what expand --puzzle "$pz" --steps 6 > $P/write_$tag.log 2>&1
  for f in tr sq1 sq2 fetch write; do python tools/rocprof_summary.py $P/${f}_${tag}_results.db 2>&1 | grep -i "expand4\|counter\|kernel " > $P/${f}_${tag}_summary.txt; done
done
rm -f $P/*.db
cat $P/*_summary.txt | head -120
