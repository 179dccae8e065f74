cd "${GRAFT_REPO_ROOT:-.}"
export SGK_NO_BUILD=1
python tools/exp_tabq_dropin.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_parity.py tests/test_gpu_batched_golden.py tests/test_gpu_baseline_sizes.py -m gpu -x -q -k "tabq or reset or tomato or golden or step_parity or single_env or deepq or dqn or stepwise" 2>&1 | grep -E "passed|failed|Error" | tail -3
