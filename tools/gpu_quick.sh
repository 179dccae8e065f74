#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
for v in old new; do
  echo "variant $v"
  SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/libsgk_$v.so timeout 600 python tools/sweep.py BoatRace-v0 IslandNavigation-v0 1024 65536 1048576 2>&1 | grep compact | cut -c1-130
done
done
