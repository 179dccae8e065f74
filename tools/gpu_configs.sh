#!/bin/bash
# BASELINE.json configs 2 / 3 / 4 on one GPU: bench.py --config k (one JSON object each, with roofline + cpu_baseline) -> configs.json;
# the MFMA counters of config 4's acting kernel; the SQ counters + kernel trace of config 3's kernel.
#   bash tools/gpu_configs.sh [out dir under gpurun_out]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_configs}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for k in 2 3 4; do timeout 600 python bench.py --config $k > $O/config$k.log 2> $O/config$k.err; tail -1 $O/config$k.log | cut -c1-400; done
python - $O <<'PY'
import json, sys
O = sys.argv[1]
out = []
for k in (2, 3, 4):
    lines = [ln for ln in open("%s/config%d.log" % (O, k)) if ln.startswith("{")]
    if lines:
        out.append(json.loads(lines[-1]))
json.dump(out, open("%s/configs.json" % O, "w"), indent=1)
PY
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_policy -- python3 tools/prof_policy_rollout.py > $O/trace_policy.log 2>&1
for f in $(find $O/trace_policy -name "*kernel_stats.csv"); do cp $f $O/policy_rollout_kernel_stats.csv; head -3 $f | cut -c1-200; done
rm -rf $O/trace_policy
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_policy -- python3 tools/prof_policy_rollout.py > $O/pmc_policy.log 2>&1
python tools/pmc_summary.py $O/pmc_policy > $O/pmc_mfma_policy_rollout.json
rm -rf $O/pmc_policy
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/pmc_mfma_policy_rollout.json"))
for k, v in d.items():
    if "policy" in k:
        print(k[:70], {c: round(x["avg_per_dispatch"]) for c, x in v.items()})
PY
