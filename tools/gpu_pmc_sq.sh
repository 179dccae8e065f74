#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
rm -rf gpurun_out/pmc2_*
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc2_trace -- python3 tools/pmc_tabq_run.py > gpurun_out/pmc2_trace.log 2>&1
for f in $(find gpurun_out/pmc2_trace -name "*kernel_stats.csv"); do head -5 $f; done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc2_sq -- python3 tools/pmc_tabq_run.py > gpurun_out/pmc2_sq.log 2>&1
tail -1 gpurun_out/pmc2_sq.log
python tools/pmc_summary.py gpurun_out/pmc2_sq > gpurun_out/pmc2_sq_summary.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc2_sq_summary.json"))
for k, v in d.items():
    if "tabq_rollout" in k or "step_kernel" in k:
        print(k, {c: round(x["avg_per_dispatch"]) for c, x in v.items()})
PY
find gpurun_out/pmc2_sq gpurun_out/pmc2_trace -name "*.csv" -size +1M -delete
