#!/bin/bash
# Runs on the GPU box (via gpurun): the round's bench line, its kernel trace, and PMC passes of the kernels whose SOURCES
# CHANGED since their committed record was measured (tools/prof_state.py; the records carry a hash of the files they depend
# on) -- a tweak to one kernel re-spends GPU minutes on that kernel's configurations only.
# Usage: tools/collect_profiles.sh <tag> [all]   -> gpurun_out/prof_<tag>/*; `all` re-measures everything
# Counters are collected in their own runs with --kernel-trace only (never with sys/hip traces); every profiled command runs
# under `timeout` (a counter set that the tool cannot schedule has hung a box for 20 minutes).
set -u
TAG=${1:-r05}
ALL=${2:-}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_$TAG
mkdir -p $P
T="timeout 300"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
summ() { python tools/rocprof_summary.py "$1" 2>&1 | grep -v "at::\|rocclr\|hipMem\|Cijk\|__amd"; }
stale() { [ -n "$ALL" ] && { echo "$*"; return; }; python tools/prof_state.py stale "$@"; }

# ---- the headline render kernel's counters: only when pw_render_kernels.inc changed
RSTALE=$(python - <<'PY'
import json, sys
sys.path.insert(0, "tools")
from make_pmc_record import kernel_source_sha
try:
    print("" if json.load(open("profiles/pmc_render_latest.json")).get("kernel_source_sha16") == kernel_source_sha() else "stale")
except Exception:
    print("stale")
PY
)
if [ -n "$ALL$RSTALE" ]; then
  $T rocprofv3 --pmc $SQ1 --kernel-trace -d $P -o sq1 -- python tools/profile_hotpath.py --steps 6 > $P/sq1.log 2>&1
  $T rocprofv3 --pmc $SQ2 --kernel-trace -d $P -o sq2 -- python tools/profile_hotpath.py --steps 6 > $P/sq2.log 2>&1
  $T rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o fetch -- python tools/profile_hotpath.py --steps 6 > $P/fetch.log 2>&1
  $T rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o write -- python tools/profile_hotpath.py --steps 6 > $P/write.log 2>&1
  for f in sq1 sq2 fetch write; do summ $P/${f}_results.db > $P/${f}_summary.txt; done
  python tools/make_pmc_record.py $P/fetch_results.db $P/write_results.db pw_render_page_kernel 65536 57834 "tools/collect_profiles.sh $TAG" > $P/pmc_render_latest.json 2> $P/pmc_record.err
else
  echo "render record current: skipped" > $P/render_skipped.txt
fi

# ---- the other configurations: one (trace, FETCH_SIZE, WRITE_SIZE) triple per launch shape, only for stale records
SPECS=()
prof3() {  # name, then the driver and its arguments
  local name=$1; shift
  $T rocprofv3 --kernel-trace --stats -d $P -o ${name}_trace -- python "$@" > $P/${name}_trace.log 2>&1
  $T rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o ${name}_fetch -- python "$@" > $P/${name}_fetch.log 2>&1
  $T rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o ${name}_write -- python "$@" > $P/${name}_write.log 2>&1
  for f in trace fetch write; do summ $P/${name}_${f}_results.db > $P/${name}_${f}_summary.txt; done
}
sq() {  # name, counters, then the driver and its arguments
  local name=$1 ctr=$2; shift 2
  $T rocprofv3 --pmc $ctr --kernel-trace -d $P -o $name -- python "$@" > $P/$name.log 2>&1
  summ $P/${name}_results.db > $P/${name}_summary.txt
}
if [ -n "$(stale C4_state C4_rollout)" ]; then
  prof3 c4 tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4
  sq c4_sq1 "$SQ1" tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4
  sq c4_sq2 "$SQ2" tools/profile_kernels.py --what c4_step --steps 40 --rollouts 4
  SPECS+=("C4_state:pw_step_group_mixed_kernel<true,:65536:$P/c4_fetch_results.db:$P/c4_write_results.db"
          "C4_rollout:pw_step_seg_kernel<+pw_step_mseg_kernel<:4194304:$P/c4_fetch_results.db:$P/c4_write_results.db")  # (bound: two kernels side by side)
fi
if [ -n "$(stale C2_step C2_rollout)" ]; then
  prof3 c2s tools/profile_kernels.py --what c2_step --steps 200 --rollouts 0
  prof3 c2r tools/profile_kernels.py --what c2_step --steps 0 --rollouts 20
  SPECS+=("C2_step:pw_step_seg_kernel<:4096:$P/c2s_fetch_results.db:$P/c2s_write_results.db"  # (bound: every environment in a segment)
          "C2_rollout:pw_step_seg_kernel<:262144:$P/c2r_fetch_results.db:$P/c2r_write_results.db")
fi
if [ -n "$(stale C5_2_obstacle C5_pull_dont_push C5_four_pistons)" ]; then
  prof3 x2ob tools/profile_kernels.py --what expand --puzzle "level1/2 Obstacle.pwp" --steps 12
  prof3 xpdp tools/profile_kernels.py --what expand --puzzle "level2/Pull Dont Push.pwp" --steps 12
  prof3 x4p tools/profile_kernels.py --what expand --puzzle "level4/Four Pistons.pwp" --steps 12
  sq x4p_sq1 "$SQ1" tools/profile_kernels.py --what expand --puzzle "level4/Four Pistons.pwp" --steps 6
  SPECS+=("C5_2_obstacle:pw_expand4_v2_kernel<3,:4000000:$P/x2ob_fetch_results.db:$P/x2ob_write_results.db"
          "C5_pull_dont_push:pw_expand4_v2_kernel<6,:4000000:$P/xpdp_fetch_results.db:$P/xpdp_write_results.db"
          "C5_four_pistons:pw_expand4_v2_kernel<12,:4000000:$P/x4p_fetch_results.db:$P/x4p_write_results.db")
fi
hot3() {  # name, then the profile_hotpath.py arguments
  local name=$1; shift
  $T rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o ${name}_fetch -- python tools/profile_hotpath.py "$@" > $P/${name}_fetch.log 2>&1
  $T rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o ${name}_write -- python tools/profile_hotpath.py "$@" > $P/${name}_write.log 2>&1
  for f in fetch write; do summ $P/${name}_${f}_results.db > $P/${name}_${f}_summary.txt; done
}
if [ -n "$(stale C3_f32_ppc3 C3_f32_ppc20_8192 C4_u8_ppc3)" ]; then
  hot3 c3f3 --steps 6 --obs float32
  hot3 c3f20 --steps 4 --obs float32 --ppc 20 --bw 2 --envs 8192
  hot3 c4u8 --steps 6 --config c4
  SPECS+=("C3_f32_ppc3:pw_render_page_kernel<float:65536:$P/c3f3_fetch_results.db:$P/c3f3_write_results.db"
          "C3_f32_ppc20_8192:pw_render_rowpage_kernel<float:8192:$P/c3f20_fetch_results.db:$P/c3f20_write_results.db::3"
          "C4_u8_ppc3:pw_render_page_kernel<unsigned char:65536:$P/c4u8_fetch_results.db:$P/c4u8_write_results.db")
fi
python tools/make_kernel_pmc_record.py "tools/collect_profiles.sh $TAG" --merge=profiles/pmc_kernels_latest.json "${SPECS[@]}" \
  > $P/pmc_kernels_latest.json 2> $P/pmc_kernels.err

# ---- the big searches (pw_search_*): kernel shares and bytes per parent, when pw_search.inc changed since the committed summary
SSHA=$(sha256sum pushworld_amd/csrc/pw_search.inc | cut -c1-16)
if [ -n "$ALL" ] || ! grep -q "$SSHA" profiles/${TAG}_search_big_trace.txt 2>/dev/null; then
  $T rocprofv3 --kernel-trace --stats -d $P -o search_big_trace -- python tools/bench_search.py > $P/bench_search.json 2> $P/search_big_trace.log
  $T rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o search_big_fetch -- python tools/bench_search.py > /dev/null 2> $P/search_big_fetch.log
  $T rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o search_big_write -- python tools/bench_search.py > /dev/null 2> $P/search_big_write.log
  for f in trace fetch write; do { echo "# pw_search.inc sha256 $SSHA"; summ $P/search_big_${f}_results.db; } > $P/search_big_${f}_summary.txt; done
fi
# ---- always, and LAST: the bench line of this code and the kernel trace of the same command -- after the records above, so that
# the line's measured-traffic fields (hbm_frac, traffic_ratio) come from records of THIS code (bench.py reads them from profiles/)
python - <<PY
import json, shutil
for name in ("pmc_kernels_latest.json", "pmc_render_latest.json"):
    try:
        json.load(open("$P/" + name))
        shutil.copy("$P/" + name, "profiles/" + name)
    except Exception:
        pass
PY
$T python bench.py --steps 20 --warmup 5 > $P/bench.json 2> $P/bench.err
cp gpurun_out/bench_full.json $P/bench_full.json 2>/dev/null
$T rocprofv3 --kernel-trace --stats -d $P -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $P/trace_bench_line.json 2> $P/trace.log
summ $P/trace_results.db > $P/trace_summary.txt
rm -f $P/*.db
tail -1 $P/bench.json | cut -c1-400
