#!/bin/bash
# scratch: the last ad-hoc command sequence sent to the GPU box
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for i in 1 2 3; do timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
