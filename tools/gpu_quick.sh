#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1200 python -m pytest tests -q -m gpu -x -k "tabq or tabular or train or smoke or demo" 2>&1 | tail -3
timeout 900 python tools/bench_configs.py 2>&1 | grep '"config": 3' | cut -c1-220
