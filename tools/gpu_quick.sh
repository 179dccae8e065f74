#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for NG in 0 1; do
  echo "== SGK_NO_GRAPH=$NG"
  SGK_NO_GRAPH=$NG python tools/sweep.py BoatRace-v0 65536 262144 524288 1048576 4194304 2>&1 | grep -E "compact" | cut -c1-118
done
