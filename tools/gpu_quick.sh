#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for P in 1 2; do
  echo "== SGK_PARTITIONS=$P"
  SGK_PARTITIONS=$P python tools/sweep.py BoatRace-v0 1048576 4194304 2>&1 | grep compact | cut -c1-118
done
