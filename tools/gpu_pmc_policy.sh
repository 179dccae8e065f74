#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1  # never let a build start under the profiler
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_policy
export SGK_NO_BUILD=1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_policy -- python3 tools/prof_policy.py > gpurun_out/pmc_policy.log 2>&1
tail -3 gpurun_out/pmc_policy.log
python - <<'PY'
import csv, glob, collections, json
fs = glob.glob("gpurun_out/pmc_policy/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(fs[0])) if "policy" in r["Kernel_Name"]]
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
ds = [disp[k] for k in sorted(disp)]
out = {}
for name, n, sl in (("4096", 4096, ds[3:20]), ("32768", 32768, ds[23:40]), ("1048576", 1048576, ds[43:60])):
    avg = {k: sum(d[k] for d in sl) / len(sl) for k in sl[0]}
    clk_ghz = avg["GRBM_GUI_ACTIVE"] / 8 / (avg["us"] * 1e3)
    avg.update(n_envs=n, launches=len(sl), shader_clock_ghz_from_GRBM_over_8_xcds=clk_ghz,
               mfma_instructions=avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 32,
               mfma_busy_fraction_of_1024_simds=avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * avg["GRBM_GUI_ACTIVE"] / 8),
               note="duration under PMC collection is longer than in the plain kernel trace")
    out[name] = avg
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_policy_summary.json", "w"), indent=1)
PY
find gpurun_out/pmc_policy -name "*.csv" -size +1M -delete
