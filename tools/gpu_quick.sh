#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4
python tools/sweep.py BoatRace-v0 65536 262144 1048576 4194304 2>&1 | grep -E "compact" | cut -c1-118
