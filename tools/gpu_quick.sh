#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_deepq.py -m gpu -q 2>&1 | tail -2
rm -rf gpurun_out/prof_dq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dq -- python3 tools/prof_deepq.py > gpurun_out/prof_dq.log 2>&1
f=$(find gpurun_out/prof_dq -name "*kernel_stats.csv"); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:4]:
    print("%-60s calls=%4s avg=%8.1f us  pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find gpurun_out/prof_dq -name "*.csv" -size +1M -delete
