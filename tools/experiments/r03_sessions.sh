#!/bin/bash
# The gpurun sessions of round 3, as run:  gpurun -- "bash tools/experiments/r03_sessions.sh <n>"
# (each block is the script of one session; outputs under gpurun_out/, summaries copied to profiles/r03_*)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
case "${1:-}" in
2)  # GPU session 2 of round 3: VMM lifecycle probe, allocator tests + lottery statistics, overlap-table A/B.
V=tools/experiments/bin/vmm_cycle
{
  for args in "474 2 0 0 0 3" "474 2 1 1 0 3" "3800 32 1 1 0 3" "3800 32 0 0 0 2" "3800 32 1 1 1 3" "3800 2 1 1 0 1" "3800 2 0 0 0 1"; do
    echo "=== vmm_cycle $args"
    timeout 150 $V $args 2>&1 | tail -40
    echo "rc ${PIPESTATUS[0]}"
  done
} > $O/r03_vmm_cycle.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_obs_alloc.py -m gpu -q > $O/r03_t_alloc.txt 2>&1
timeout 900 python tools/experiments/obs_alloc_xp.py many c3 8 > $O/r03_many_c3.txt 2>&1
timeout 600 python tools/experiments/obs_alloc_xp.py many c4 5 > $O/r03_many_c4.txt 2>&1
timeout 900 python tools/experiments/step_tables_xp.py > $O/r03_step_tables.txt 2>&1
timeout 400 python tools/bench_expand.py > $O/r03_expand4.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_expand.py tests/test_gpu_shapes.py tests/test_gpu_search.py tests/test_gpu_abi.py -m gpu -q > $O/r03_t_tables.txt 2>&1
tail -3 $O/r03_t_alloc.txt $O/r03_t_tables.txt
;;
3)  # GPU session 3 of round 3: pair probe (time-boxed forensics), allocator tests, table-only kernels A/B, bench line.
P=tools/experiments/bin/pair_probe
{
  echo "=== pair_probe 24 chunks, 600 sets, 8 sweeps, no LDS pad"; timeout 120 $P 24 600 8 0
  echo "=== pair_probe 24 chunks, 600 sets, 8 sweeps, 7 KB LDS pad"; timeout 120 $P 24 600 8 7168
  echo "=== pair_probe 24 chunks spread with 3 GB spacers"; timeout 120 $P 24 400 8 0 3000
} > $O/r03_pair_probe.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_obs_alloc.py -m gpu -q > $O/r03_t_alloc.txt 2>&1
timeout 900 python tools/experiments/step_tables_xp.py > $O/r03_step_tables.txt 2>&1
timeout 400 python tools/bench_expand.py > $O/r03_expand4.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_expand.py tests/test_gpu_shapes.py tests/test_gpu_search.py tests/test_gpu_abi.py tests/test_gpu_vector.py -m gpu -q > $O/r03_t_tables.txt 2>&1
timeout 600 python bench.py > $O/r03_bench_a.json 2> $O/r03_bench_a.err
timeout 900 python tools/experiments/obs_alloc_xp.py many c3 6 > $O/r03_many_c3.txt 2>&1
timeout 600 python tools/experiments/obs_alloc_xp.py many c4 4 > $O/r03_many_c4.txt 2>&1
tail -n 3 $O/r03_t_alloc.txt $O/r03_t_tables.txt
;;
4)  # GPU session 4 of round 3: allocator tests (fresh address ranges), table policy A/B, search with fingerprints.
timeout 600 python -m pytest tests/test_gpu_obs_alloc.py -m gpu -q > $O/r03_t_alloc.txt 2>&1
timeout 900 python tools/experiments/step_tables_xp.py > $O/r03_step_tables2.txt 2>&1
timeout 400 python tools/bench_expand.py > $O/r03_expand4b.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_expand.py tests/test_gpu_parity.py -m gpu -q > $O/r03_t_search.txt 2>&1
timeout 600 python tools/bench_search.py > $O/r03_search.json 2> $O/r03_search.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_search -o search -- python tools/bench_search.py --max-states 20000000 > $O/r03_search_trace.log 2>&1
python tools/rocprof_summary.py $O/prof_search/search_results.db > $O/r03_search_trace.txt 2>&1
rm -rf $O/prof_search
tail -n 3 $O/r03_t_alloc.txt $O/r03_t_search.txt
;;
5)  # GPU session 5 of round 3: the whole GPU suite, C5 / search rates, rocprofv3 profiles of the C3 and C4 hot paths.
timeout 1800 python -m pytest tests -m gpu -q -x > $O/r03_gputest.txt 2>&1
timeout 400 python tools/bench_expand.py > $O/r03_expand4c.txt 2>&1
timeout 600 python tools/bench_search.py > $O/r03_search2.json 2> $O/r03_search2.err
timeout 900 python tools/experiments/step_tables_xp.py --modes none+fwd,auto > $O/r03_step_tables3.txt 2>&1
timeout 1500 bash tools/collect_profiles.sh r03 > $O/r03_collect.log 2>&1
timeout 400 python bench.py --config c4 --no-cpu-baseline > $O/r03_bench_c4_state.json 2> $O/r03_bench_c4_state.err
timeout 400 python bench.py --config c4 --obs uint8 --no-cpu-baseline > $O/r03_bench_c4_u8.json 2> $O/r03_bench_c4_u8.err
P=$O/prof_r03c4
mkdir -p $P
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P -o fetch -- python tools/profile_hotpath.py --config c4 --steps 6 > $P/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P -o write -- python tools/profile_hotpath.py --config c4 --steps 6 > $P/write.log 2>&1
for f in fetch write; do python tools/rocprof_summary.py $P/${f}_results.db > $P/${f}_summary.txt 2>&1; done
rm -f $P/*.db
tail -n 3 $O/r03_gputest.txt
;;
6)  # GPU session 6 of round 3: arrangement probe (same chunks, other order).
A=tools/experiments/bin/arrange_probe
{
  echo "=== C3 size, 32 MiB chunks"; timeout 200 $A 3616 32 4 0
  echo "=== C3 size, 32 MiB chunks, 7 KB LDS pad"; timeout 200 $A 3616 32 2 7168
  echo "=== C4 size, 32 MiB chunks"; timeout 200 $A 4288 32 3 0
  echo "=== C3 size, 256 MiB chunks"; timeout 200 $A 3616 256 2 0
  echo "=== C3 size, 2 MiB chunks"; timeout 300 $A 3616 2 2 0
} > $O/r03_arrange.txt 2>&1
tail -n 30 $O/r03_arrange.txt
;;
7)  # GPU session 7 of round 3: final code -- smoke, the whole GPU suite, the bench line (N = 1 record), two self-spawned ranks.
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r03_smoke.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gputest.txt 2>&1
timeout 600 python bench.py > $O/r03_bench_final.json 2> $O/r03_bench_final.err
cp profiles/bench_n1_latest.json $O/bench_n1_latest.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --shared-device --no-cpu-baseline --no-extras > $O/r03_bench_n2_shared.json 2> $O/r03_bench_n2_shared.err
rocm-smi --showclocks --showpower --showperflevel > $O/r03_rocm_smi.txt 2>&1
tail -n 2 $O/r03_smoke.txt; tail -n 3 $O/r03_gputest.txt; cut -c1-300 $O/r03_bench_final.json
;;
8)  # GPU session 8 of round 3: actions prefetched in multi-step launches.
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_soak.py tests/test_gpu_vector.py -m gpu -q > $O/r03_t_prefetch.txt 2>&1
timeout 900 python tools/experiments/step_tables_xp.py --modes auto > $O/r03_step_tables4.txt 2>&1
tail -n 3 $O/r03_t_prefetch.txt; cat $O/r03_step_tables4.txt
;;
9)  # GPU session 9 of round 3: kernel trace of the GPU search with the final code.
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_search -o search -- python tools/bench_search.py --max-states 20000000 > $O/r03_search_trace2.log 2>&1
python tools/rocprof_summary.py $O/prof_search/search_results.db > $O/r03_search_trace2.txt 2>&1
rm -rf $O/prof_search
head -14 $O/r03_search_trace2.txt | cut -c1-150
;;
10)  # GPU session 10 of round 3: final kernels (puzzle-id clamp, action prefetch): tests, profiles, bench.
timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gputest.txt 2>&1
timeout 1500 bash tools/collect_profiles.sh r03b > $O/r03b_collect.log 2>&1
cp $O/prof_r03b/pmc_render_latest.json profiles/pmc_render_latest.json
timeout 600 python bench.py > $O/r03_bench_final2.json 2> $O/r03_bench_final2.err
cp profiles/bench_n1_latest.json $O/bench_n1_latest.json 2>/dev/null
tail -n 3 $O/r03_gputest.txt; cut -c1-200 $O/r03_bench_final2.json; cat $O/prof_r03b/pmc_render_latest.json | head -30
;;
11)  # GPU session 11 of round 3: every observation setting with library-owned buffers, configs C1 / C2 / C4, bench sanity.
timeout 900 python tools/render_sweep.py > $O/r03_render_sweep.txt 2>&1
timeout 900 python tools/bench_configs.py > $O/r03_configs.json 2> $O/r03_configs.err
timeout 600 python bench.py --no-cpu-baseline > $O/r03_bench_final3.json 2> $O/r03_bench_final3.err
timeout 300 python -m pytest tests/test_gpu_obs_alloc.py tests/test_gpu_bench.py -m gpu -q > $O/r03_t_last.txt 2>&1
cat $O/r03_render_sweep.txt; tail -n 3 $O/r03_t_last.txt; cut -c1-200 $O/r03_bench_final3.json
;;
14)  # GPU session 14 of round 3: screened candidates (up to 12): tests, fresh-process statistics, bench.
timeout 600 python -m pytest tests/test_gpu_obs_alloc.py tests/test_gpu_bench.py tests/test_gpu_abi.py -m gpu -q > $O/r03_t_screen.txt 2>&1
timeout 900 python tools/experiments/obs_alloc_xp.py many c3 6 > $O/r03_many_c3b.txt 2>&1
timeout 600 python tools/experiments/obs_alloc_xp.py many c4 3 > $O/r03_many_c4b.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/r03_bench_final4.json 2> $O/r03_bench_final4.err
tail -n 3 $O/r03_t_screen.txt; grep -v amdgpu $O/r03_many_c3b.txt $O/r03_many_c4b.txt; cut -c1-200 $O/r03_bench_final4.json
;;
15)  # GPU session 15 of round 3: search with wide slots: tests, rates, trace.
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_generate.py tests/test_gpu_expand.py -m gpu -q > $O/r03_t_search2.txt 2>&1
timeout 600 python tools/bench_search.py > $O/r03_search3.json 2> $O/r03_search3.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_search -o search -- python tools/bench_search.py --max-states 20000000 > $O/r03_search_trace3.log 2>&1
python tools/rocprof_summary.py $O/prof_search/search_results.db > $O/r03_search_trace3.txt 2>&1
rm -rf $O/prof_search
tail -n 3 $O/r03_t_search2.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_search3.json'))
for k,v in d.items(): print(k, v['status'], v['states'], '%.3e'%v['parents_per_s'])
PY
head -10 $O/r03_search_trace3.txt | cut -c1-140
;;
16)  # GPU session 16 of round 3: final code (8-lane groups for single steps too): the whole GPU suite, smoke, bench.
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r03_smoke.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gputest.txt 2>&1
timeout 600 python bench.py > $O/r03_bench_final5.json 2> $O/r03_bench_final5.err
cp profiles/bench_n1_latest.json $O/bench_n1_latest.json 2>/dev/null
tail -n 1 $O/r03_smoke.txt; tail -n 3 $O/r03_gputest.txt; cut -c1-200 $O/r03_bench_final5.json
;;
20)  # GPU sessions 20-24: one lane per environment (table-driven) against the lane groups over batch sizes; RCCL with one rank.
for b in 65536 131072 262144 524288 1048576; do
  timeout 900 python tools/experiments/step_tables_xp.py --which l0,c3,l14 --modes all+nolane,all+lane --reps 100 --batch $b >> $O/r03_lane_xp.txt 2>&1
done
timeout 600 python -m pytest tests/test_gpu_bench.py -x -q -k rccl 2>&1 | tail -n 3
;;
25)  # GPU sessions 25-35: pw_expand4 and the search with one lane per state: parity, rates, counters, staging variants.
timeout 900 python -m pytest tests/test_gpu_expand.py tests/test_gpu_search.py -x -q 2>&1 | tail -n 3
timeout 900 python tools/bench_expand.py --sizes 65536,131072,1000000 > $O/r03_expand4_lanes.txt 2>&1
timeout 900 python tools/bench_search.py --max-states 20000000 --groups > $O/r03_search_groups.json 2>/dev/null
timeout 900 python tools/bench_search.py --max-states 20000000 > $O/r03_search_lanes.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && \
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
            --kernel-trace -d $O/prof_expand -o sq1 -- python tools/bench_expand.py --sizes 1000000 > $O/prof_expand_sq1.log 2>&1 && \
  python tools/rocprof_summary.py $O/prof_expand/sq1_results.db | grep -i "expand4\|^kernel"; rm -rf $O/prof_expand )
;;
36)  # GPU sessions 36-40: the whole-grid board kernel: parity, rates against the lane groups, configs.
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py tests/test_gpu_configs.py -x -q -k "boards or tiny or c2 or bad or garbage or rollout" 2>&1 | tail -n 3
for b in 65536 1048576; do
  timeout 900 python tools/experiments/step_tables_xp.py --which c2,l0tiny --modes all+noboards,all --reps 200 --batch $b >> $O/r03_boards_xp.txt 2>&1
done
timeout 600 python tools/bench_configs.py > $O/r03_configs2.json 2> $O/r03_configs2.err
;;
41)  # GPU session 41: final code: the whole GPU suite, profiles (trace, counters, PMC record), loop-gap experiment.
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 2
bash tools/collect_profiles.sh r03d 2>&1 | tail -n 2
timeout 600 python tools/experiments/loop_gap_xp.py > $O/r03_loop_gap.txt 2>&1
;;
*) echo "usage: $0 <session number>"; exit 2 ;;
esac
