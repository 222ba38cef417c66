for cfg in "--obs float32" "--obs float32 --ppc 20 --bw 2 --envs-per-gpu 8192" "--obs uint8 --ppc 20 --bw 2 --envs-per-gpu 32768" "--obs uint8 --ppc 8 --bw 2"; do
  echo "== $cfg"
  python bench.py $cfg --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full: %.3e steps/s %.3f ms  %s %.0f GB/s' % (d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'])); print('incremental:', d['incremental_render'].get('env_steps_per_s'), d['incremental_render'].get('ms_per_step'), d['incremental_render'].get('error'))"
done
