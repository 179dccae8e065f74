#!/bin/bash
# SQ counters of the per-step tabular-Q kernels (eager calls): where do tabq_act_kernel's ~18 us go
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
export SGK_NO_BUILD=1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -- python3 tools/prof_tabq_stepwise.py 262144 calls > $O/pmc_sq.log 2>&1
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq_summary.json
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 tools/prof_tabq_stepwise.py 262144 calls > $O/pmc_sq2.log 2>&1
python tools/pmc_summary.py $O/pmc_sq2 > $O/pmc_sq2_summary.json
python - <<'PY'
import json
for f in ("gpurun_out/r02k/pmc_sq_summary.json", "gpurun_out/r02k/pmc_sq2_summary.json"):
    d = json.load(open(f))
    for k, v in d.items():
        if "tabq" in k or "step_kernel" in k or "reset_kernel" in k:
            print(k[:40], {c: round(x["avg_per_dispatch"]) for c, x in v.items()})
PY
tail -2 $O/pmc_sq2.log
find $O -name "*.csv" -size +1M -delete
