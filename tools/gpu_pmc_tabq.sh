#!/bin/bash
# SQ counter passes + kernel trace of the fused tabular-Q rollout at config 3's shape (IslandNavigation, 262 144 agents).
#   bash tools/gpu_pmc_tabq.sh [out dir under gpurun_out] [tag]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-tabq_pmc}; T=${2:-head}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/pmc_tabq_run.py > $O/trace_$T.log 2>&1
for f in $(find $O/trace -name "*kernel_stats.csv"); do cp $f $O/tabq_rollout_kernel_stats_$T.csv; head -4 $f | cut -c1-220; done
rm -rf $O/trace
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
i=1
for P in "$P1" "$P2"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/sq$i -- python3 tools/pmc_tabq_run.py > $O/sq${i}_$T.log 2>&1
  python tools/pmc_summary.py $O/sq$i > $O/pmc_sq${i}_tabq_rollout_$T.json
  rm -rf $O/sq$i; i=$((i+1))
done
python - $O $T <<'PY'
import json, sys
O, T = sys.argv[1], sys.argv[2]
for i in (1, 2):
    d = json.load(open("%s/pmc_sq%d_tabq_rollout_%s.json" % (O, i, T)))
    for k, v in d.items():
        if "tabq_rollout" in k:
            print(k[:60], {c: round(x["avg_per_dispatch"]) for c, x in v.items()})
PY
