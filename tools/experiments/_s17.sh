cd "$GRAFT_REPO_ROOT"
P=gpurun_out/s17; mkdir -p $P
timeout 900 python -m pytest tests/test_gpu_expand.py -x -q > $P/expand_tests.log 2>&1; tail -5 $P/expand_tests.log
timeout 600 python tools/experiments/expand_all.py > $P/expand_all.txt 2>&1; tail -16 $P/expand_all.txt | cut -c1-220
