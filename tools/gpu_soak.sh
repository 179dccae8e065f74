#!/bin/bash
# Flake hunt on the GPU: bash tools/gpu_soak.sh [suite runs, default 2]. The step-server tests (the mailbox protocol of sgk_host_core.h,
# changed in round 5) six times, then the whole GPU suite N times, each in a fresh process. One line per run on stdout plus the FAILED
# lines; the full logs of failing runs stay in gpurun_out/soak/. Copy the stdout to profiles/<round>/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/soak; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; echo "build rc=$?"
summary() { grep -E "^FAILED|^ERROR" $1 | cut -c1-300; grep -E " (passed|failed|error)" $1 | tail -1 | sed "s/^/$2: /"; }
for i in 1 2 3 4 5 6; do
  timeout 900 python -m pytest tests -m gpu -q -rf --timeout=600 -k "server or mailbox or single_env or close or alloc" > $O/server_$i.log 2>&1
  summary $O/server_$i.log "server-tests run $i"; grep -q " failed" $O/server_$i.log || rm -f $O/server_$i.log
done
for i in $(seq 1 ${1:-2}); do
  timeout 2400 python -m pytest tests -m gpu -q -rf --timeout=900 > $O/suite_$i.log 2>&1
  summary $O/suite_$i.log "gpu suite run $i"; grep -q " failed" $O/suite_$i.log || rm -f $O/suite_$i.log
done
