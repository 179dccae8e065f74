#!/bin/bash
# The graph-capture race of round 5 (EXPERIMENTS R5.11) on one box: the regression test against the library as it was
# (safe-grid-agents_amd/lib/libsgk_before.so, built from the parent commit: must FAIL) and as it is (must pass), then the two
# thread tests twenty times over.  bash tools/gpu_capture_race.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp SGK_NO_BUILD=1
T="tests/test_gpu_parity.py"
K="a_graph_capture_in_one_thread or handles_driven_from_concurrent_threads"
if [ -f safe-grid-agents_amd/lib/libsgk_before.so ]; then
  SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/libsgk_before.so python -m pytest $T -m gpu -q -x -k "a_graph_capture_in_one_thread" 2>&1 | grep -E "^E  .*Error|passed|failed" | cut -c1-400 | sed "s/^/library before the fix: /"
fi
for i in $(seq 1 20); do
  python -m pytest $T -m gpu -q -k "$K" 2>&1 | grep -E "^E  .*Error| (passed|failed)" | cut -c1-400 | sed "s/^/library at HEAD, run $i: /"
done
