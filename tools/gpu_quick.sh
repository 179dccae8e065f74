#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
python tools/sweep.py BoatRace-v0 65536 1048576 2>&1 | grep compact | cut -c120-260
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
