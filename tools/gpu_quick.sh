#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_deepq.py tests/test_gpu_ppo.py -q -m gpu -x 2>&1 | tail -2
rm -rf gpurun_out/prof_policy
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_policy -- python3 tools/prof_policy.py > gpurun_out/prof_policy.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_policy/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "policy" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
for name, sl in (("4096", d[3:20]), ("32768", d[23:40]), ("1048576", d[43:60])):
    print(name, "launches", len(sl), "avg us %.2f min %.2f" % (sum(sl) / len(sl), min(sl)))
PY
find gpurun_out/prof_policy -name "*.csv" -size +1M -delete
