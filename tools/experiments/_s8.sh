cd "$GRAFT_REPO_ROOT"
python tools/experiments/step_quad_xp.py --sets c3 --envs 1048576 --modes auto:lanes,auto:groups 2>&1 | grep -v amdgpu
python tools/experiments/step_quad_xp.py --sets c4 --envs 524288 --modes auto:lanes,auto:groups 2>&1 | grep -v amdgpu
