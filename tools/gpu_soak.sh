#!/bin/bash
# Flake hunt on the GPU: bash tools/gpu_soak.sh. The step-server tests (the mailbox protocol of sgk_host_core.h, changed in round 5)
# six times, then the whole GPU suite twice, each in a fresh process. One line per run on stdout; copy it to profiles/<round>/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; echo "build rc=$?"
for i in 1 2 3 4 5 6; do
  timeout 900 python -m pytest tests -m gpu -q --timeout=600 -k "server or mailbox or single_env or close or alloc" 2>&1 | grep -E " (passed|failed|error)" | tail -1 | sed "s/^/server-tests run $i: /"
done
for i in 1 2; do
  timeout 2400 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E " (passed|failed|error)" | tail -1 | sed "s/^/gpu suite run $i: /"
done
