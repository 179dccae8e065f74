#!/bin/bash
# A/B of two builds of libsgk.so on ONE box: bench.py --config 2, the per-step launch at four batch sizes, the levels with irregular
# episode ends.  bash tools/gpu_ab_step.sh <lib A> <lib B>   (file names under safe-grid-agents_amd/lib/)
cd "${GRAFT_REPO_ROOT:-.}"
export SGK_NO_BUILD=1
A=${1:-libsgk_before.so}; B=${2:-libsgk.so}
for lib in $A $B; do
  [ -f "safe-grid-agents_amd/lib/$lib" ] || { echo "gpu_ab_step.sh: missing safe-grid-agents_amd/lib/$lib (build the 'before' library first: make -C safe-grid-agents_amd/csrc OUT=../lib/libsgk_before.so OBJDIR=build_before at the commit to compare with)"; exit 1; }
done
for r in 1 2; do for lib in $A $B; do
  SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/$lib python bench.py --config 2 --no-cpu-baseline | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config 2', '$lib', round(d['us_per_lockstep_step'],3))"; done; done
for lib in $A $B; do echo "== $lib"
  SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/$lib python tools/bench_stream.py --envs BoatRace-v0 --sizes 1024,65536,262144,1048576 --modes launch --reps 8 2>&1 | grep -v amdgpu | cut -c1-70
  SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/$lib python tools/exp_island_step.py 2>&1 | grep -v amdgpu | cut -c1-110
done
