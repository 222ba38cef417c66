#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the C3 hot path AND of the other configurations' kernels.
# Usage: tools/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/*.txt, pmc_render_latest.json, pmc_kernels_latest.json
# Counters are collected in their own runs with --kernel-trace only (never with sys/hip traces).
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_$TAG
mkdir -p $P
python bench.py --steps 50 --warmup 5 > $P/bench.json 2> $P/bench.err
rocprofv3 --kernel-trace --stats -d $P -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $P/trace_bench_line.json 2> $P/trace.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $P -o sq1 -- python tools/profile_hotpath.py --steps 6 > $P/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $P -o sq2 -- python tools/profile_hotpath.py --steps 6 > $P/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o fetch -- python tools/profile_hotpath.py --steps 6 > $P/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o write -- python tools/profile_hotpath.py --steps 6 > $P/write.log 2>&1
for f in trace sq1 sq2 fetch write; do
  python tools/rocprof_summary.py $P/${f}_results.db > $P/${f}_summary.txt 2>&1
done
python tools/make_pmc_record.py $P/fetch_results.db $P/write_results.db pw_render_page_kernel 65536 57834 "tools/collect_profiles.sh $TAG" > $P/pmc_render_latest.json 2> $P/pmc_record.err

# ---- the other configurations: one (trace, FETCH_SIZE, WRITE_SIZE) triple per launch shape
prof3() {  # name, then the profile_kernels.py arguments
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $P -o ${name}_trace -- python tools/profile_kernels.py "$@" > $P/${name}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o ${name}_fetch -- python tools/profile_kernels.py "$@" > $P/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o ${name}_write -- python tools/profile_kernels.py "$@" > $P/${name}_write.log 2>&1
  for f in trace fetch write; do
    python tools/rocprof_summary.py $P/${name}_${f}_results.db 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd" > $P/${name}_${f}_summary.txt
  done
}
prof3 c4 --what c4_step --steps 40 --rollouts 4
prof3 c2s --what c2_step --steps 200 --rollouts 0
prof3 c2r --what c2_step --steps 0 --rollouts 20
prof3 x2ob --what expand --puzzle "level1/2 Obstacle.pwp" --steps 12
prof3 xpdp --what expand --puzzle "level2/Pull Dont Push.pwp" --steps 12
prof3 x4p --what expand --puzzle "level4/Four Pistons.pwp" --steps 12
prof3 search --what search --puzzle "level2/Pull Dont Push.pwp" --states 4000000
prof3 batch --what batch --states 20000 --steps 8
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $P -o x4p_sq1 -- python tools/profile_kernels.py --what expand --puzzle "level4/Four Pistons.pwp" --steps 6 > $P/x4p_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $P -o x4p_sq2 -- python tools/profile_kernels.py --what expand --puzzle "level4/Four Pistons.pwp" --steps 6 > $P/x4p_sq2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $P -o x2ob_sq1 -- python tools/profile_kernels.py --what expand --puzzle "level1/2 Obstacle.pwp" --steps 6 > $P/x2ob_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $P -o xpdp_sq1 -- python tools/profile_kernels.py --what expand --puzzle "level2/Pull Dont Push.pwp" --steps 6 > $P/xpdp_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $P -o c4_sq1 -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 > $P/c4_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace -d $P -o c4_sq2 -- python tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4 > $P/c4_sq2.log 2>&1
for f in x4p_sq1 x4p_sq2 c4_sq1 c4_sq2 x2ob_sq1 xpdp_sq1; do python tools/rocprof_summary.py $P/${f}_results.db 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd" > $P/${f}_summary.txt; done
# the other render settings of the configuration suite (tools/profile_hotpath.py builds them as bench.py does)
hot3() {  # name, then the profile_hotpath.py arguments
  local name=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o ${name}_fetch -- python tools/profile_hotpath.py "$@" > $P/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o ${name}_write -- python tools/profile_hotpath.py "$@" > $P/${name}_write.log 2>&1
  for f in fetch write; do
    python tools/rocprof_summary.py $P/${name}_${f}_results.db 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd" > $P/${name}_${f}_summary.txt
  done
}
hot3 c3f3 --steps 6 --obs float32
hot3 c3f20 --steps 4 --obs float32 --ppc 20 --bw 2 --envs 8192
hot3 c4u8 --steps 6 --config c4
python tools/make_kernel_pmc_record.py "tools/collect_profiles.sh $TAG" \
  "C3_f32_ppc3:pw_render_page_kernel<float:65536:$P/c3f3_fetch_results.db:$P/c3f3_write_results.db" \
  "C3_f32_ppc20_8192:pw_render_rowpage_kernel<float:8192:$P/c3f20_fetch_results.db:$P/c3f20_write_results.db::3" \
  "C4_u8_ppc3:pw_render_page_kernel<unsigned char:65536:$P/c4u8_fetch_results.db:$P/c4u8_write_results.db" \
  "C4_state:pw_step_group_mixed_kernel<true,:65536:$P/c4_fetch_results.db:$P/c4_write_results.db" \
  "C4_rollout:pw_step_group_mixed_kernel<false,:4194304:$P/c4_fetch_results.db:$P/c4_write_results.db" \
  "C2_step:pw_step_board_kernel:4096:$P/c2s_fetch_results.db:$P/c2s_write_results.db" \
  "C2_rollout:pw_step_board_kernel:262144:$P/c2r_fetch_results.db:$P/c2r_write_results.db" \
  "C5_2_obstacle:pw_expand4_v2_kernel<3,:4000000:$P/x2ob_fetch_results.db:$P/x2ob_write_results.db" \
  "C5_pull_dont_push:pw_expand4_v2_kernel<6,:4000000:$P/xpdp_fetch_results.db:$P/xpdp_write_results.db" \
  "C5_four_pistons:pw_expand4_v2_kernel<12,:4000000:$P/x4p_fetch_results.db:$P/x4p_write_results.db" \
  > $P/pmc_kernels_latest.json 2> $P/pmc_kernels.err
rm -f $P/*.db
tail -1 $P/bench.json | cut -c1-400
