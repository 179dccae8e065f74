#!/bin/bash
# On the GPU box: build the -DSGK_LEARN_TIMELINE variant of the library next to the product's and print dqn_sgd_kernel's
# barrier-to-barrier timeline (lane 0's wall_clock64 behind every barrier). Usage: tools/gpu_dqn_timeline.sh [out.log] [multi]
#   multi: also build in the four-workgroup experiment (-DSGK_DQN_MULTI_WG, sgk_learn.hip) and print its timeline first
set -e
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/dqn_timeline.log}
mkdir -p "$(dirname "$out")"
extra="-DSGK_LEARN_TIMELINE"
[ "$2" = multi ] && extra="$extra -DSGK_DQN_MULTI_WG"
make -s -j6 -C safe-grid-agents_amd/csrc OUT=../lib/libsgk_tl.so OBJDIR=build_tl EXTRA="$extra"
: > "$out"
if [ "$2" = multi ]; then
  SGK_DQN_WORKGROUPS=4 SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/libsgk_tl.so SGK_NO_BUILD=1 python tools/exp_dqn_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a "$out"
fi
SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/libsgk_tl.so SGK_NO_BUILD=1 python tools/exp_dqn_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a "$out"
SGK_DQN_ONE_LAUNCH=1 SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/libsgk_tl.so SGK_NO_BUILD=1 python tools/exp_dqn_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a "$out"
