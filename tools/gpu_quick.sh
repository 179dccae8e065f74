#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for G in 2048 4096 8192 16384; do
  echo "== SGK_MAX_GRID=$G"
  SGK_MAX_GRID=$G python tools/sweep.py BoatRace-v0 262144 1048576 4194304 2>&1 | grep compact | cut -c1-150
done
