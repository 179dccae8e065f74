#!/bin/bash
# HBM-side bytes of the four per-step tabular-Q kernels at the config-3 shape (separate --pmc passes; FETCH_SIZE is raw KB here:
# multiply by the ~1.98 calibration of profiles/traffic.json)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/tabq_traffic; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
export SGK_NO_BUILD=1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -- python3 tools/prof_tabq_stepwise.py 262144 calls > $O/pmc_$ctr.log 2>&1
  python tools/pmc_summary.py $O/pmc_$ctr > $O/${ctr}_summary.json
  find $O/pmc_$ctr -name "*.csv" -size +1M -delete
done
python - <<'PY'
import json
f = json.load(open("gpurun_out/tabq_traffic/FETCH_SIZE_summary.json")); w = json.load(open("gpurun_out/tabq_traffic/WRITE_SIZE_summary.json"))
for k in f:
    if "tabq" in k or "step_kernel" in k or "reset_kernel" in k:
        fr = f[k]["FETCH_SIZE"]["avg_per_dispatch"] * 1024 * 1.98 / 1e6
        wr = w.get(k, {}).get("WRITE_SIZE", {}).get("avg_per_dispatch", 0) * 1024 / 1e6
        print("%-60s fetch %.1f MB  write %.1f MB  per env: %.0f + %.0f B" % (k[:60], fr, wr, fr * 1e6 / 262144, wr * 1e6 / 262144))
PY
