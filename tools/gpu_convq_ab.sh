#!/bin/bash
# A/B of the conv Q-body kernel's shape: groups per wave and pass (CQ_GPW) x waves per SIMD aimed at (CQ_MIN_WAVES).
cd "${GRAFT_REPO_ROOT:-.}"
export SGK_NO_BUILD=1
timeout 600 python -m pytest tests/test_gpu_convq.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/exp_convq.py gpw2_cap3 2>&1 | grep -v amdgpu.ids
for v in "3 2" "4 2" "2 4"; do
  set -- $v
  make -C safe-grid-agents_amd/csrc clean > /dev/null; make -j8 -C safe-grid-agents_amd/csrc EXTRA="-DCQ_GPW=$1 -DCQ_MIN_WAVES=$2" > /dev/null 2>&1
  timeout 300 python -m pytest tests/test_gpu_convq.py -x -q -m gpu -k "Sokoban or BoatRace" 2>&1 | tail -1
  timeout 300 python tools/exp_convq.py gpw$1_cap$2 2>&1 | grep -v amdgpu.ids
done
