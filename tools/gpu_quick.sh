#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
echo "baseline"; timeout 600 python tools/sweep.py BoatRace-v0 IslandNavigation-v0 65536 1048576 4194304 2>&1 | grep compact | cut -c1-130
for g in 768 1024 1536 2048; do
  echo "pair, SGK_MAX_GRID=$g"
  SGK_STEP_PAIR=1 SGK_MAX_GRID=$g timeout 600 python tools/sweep.py BoatRace-v0 IslandNavigation-v0 65536 1048576 4194304 2>&1 | grep compact | cut -c1-130
done
SGK_STEP_PAIR=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "random or graph or step" 2>&1 | tail -2
