#!/bin/bash
# the gaps between back-to-back streamed launches (rocprofv3 kernel trace of bench.py's timed region)
cd "$GRAFT_REPO_ROOT" || exit 1
export SGK_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/gaps; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-fused > $O/bench.log 2>&1
for f in $(find $O/tr -name "*kernel_trace.csv"); do cp $f $O/kernel_trace.csv; done
rm -rf $O/tr
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/gaps/kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
prev=None
out=[]
for r in rows:
    name=r["Kernel_Name"][:60]; s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    gap=(s-prev)/1e3 if prev else 0
    out.append("%-60s dur %9.1f us  gap before %8.1f us"%(name,(e-s)/1e3,gap))
    prev=e
open("gpurun_out/gaps/gaps.log","w").write("\n".join(out)+"\n")
print("\n".join(out[-40:]))
PY
