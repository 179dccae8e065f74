cd "${GRAFT_REPO_ROOT:-.}"
export SGK_NO_BUILD=1
bash tools/gpu_ab_step.sh libsgk_before.so libsgk.so
for g in 1536 2048 3072 4096; do echo "SGK_MAX_GRID=$g"; SGK_MAX_GRID=$g python tools/bench_stream.py --envs BoatRace-v0 --sizes 262144,524288,1048576,4194304 --modes launch --reps 8 2>&1 | grep -v amdgpu | cut -c1-70; done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_parity or stepwise_graph or streamed" 2>&1 | grep -E "passed|failed" | tail -1
