#!/bin/bash
# scratch: the last ad-hoc command sequence sent to the GPU box
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -5
