#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for g in 1280 1408 1536 1664 1792 2048; do
  echo "SGK_MAX_GRID=$g"
  SGK_MAX_GRID=$g timeout 600 python tools/sweep.py BoatRace-v0 IslandNavigation-v0 262144 1048576 4194304 2>&1 | grep compact | cut -c1-130
done
