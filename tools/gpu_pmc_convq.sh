#!/bin/bash
# PMC passes over the conv Q-body kernel (tools/prof_convq.py): MFMA busy / waits, then LDS. Output: gpurun_out/pmc_convq_summary.json
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1  # never let a build start under the profiler
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_convq_a gpurun_out/pmc_convq_b gpurun_out/convq_trace
export SGK_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/convq_trace -- python3 tools/prof_convq.py > gpurun_out/convq_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_convq_a -- python3 tools/prof_convq.py > gpurun_out/pmc_convq_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/pmc_convq_b -- python3 tools/prof_convq.py > gpurun_out/pmc_convq_b.log 2>&1
tail -2 gpurun_out/pmc_convq_a.log gpurun_out/pmc_convq_b.log
python - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ("a", "b"):
    fs = glob.glob("gpurun_out/pmc_convq_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "convq" in r["Kernel_Name"]]
    disp = collections.OrderedDict()
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    ds = [disp[k] for k in sorted(disp)]
    for name, sl in (("32768", ds[3:20]), ("1048576", ds[23:40])):
        if sl:
            avg = {k: sum(d[k] for d in sl) / len(sl) for k in sl[0]}
            out.setdefault(name, {}).update({("us_under_pmc_" + tag if k == "us" else k): v for k, v in avg.items()})
fs = glob.glob("gpurun_out/convq_trace/**/*kernel_stats.csv", recursive=True)
if fs:
    out["kernel_stats"] = [r for r in csv.DictReader(open(fs[0])) if "convq" in r["Name"]]
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_convq_summary.json", "w"), indent=1)
PY
find gpurun_out/pmc_convq_a gpurun_out/pmc_convq_b gpurun_out/convq_trace -name "*.csv" -size +1M -delete
