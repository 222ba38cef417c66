set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_r04x
mkdir -p $P
hot3() {
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
python tools/make_kernel_pmc_record.py "tools/collect_profiles.sh r04" --merge=profiles/pmc_kernels_latest.json \
  "C3_f32_ppc3:pw_render_page_kernel<float:65536:$P/c3f3_fetch_results.db:$P/c3f3_write_results.db" \
  "C3_f32_ppc20_8192:pw_render_rowpage_kernel<float:8192:$P/c3f20_fetch_results.db:$P/c3f20_write_results.db::3" \
  "C4_u8_ppc3:pw_render_page_kernel<unsigned char:65536:$P/c4u8_fetch_results.db:$P/c4u8_write_results.db" \
  > $P/pmc_kernels_latest.json 2> $P/pmc_kernels.err
cat $P/pmc_kernels.err
rm -f $P/*.db
grep -h "SIZE" $P/*_summary.txt | grep render
