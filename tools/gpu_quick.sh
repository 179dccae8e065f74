#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
python tools/sweep.py 1048576 2>&1 | grep -E "compact" | cut -c1-24,175-260
