#!/usr/bin/env python3
"""profiles/traffic.json from the rocprofv3 --pmc summaries of one round (tools/gpu_r03_final.sh): HBM-side bytes per launch of
the bench's kernels, collected and corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE
in separate passes; WRITE_SIZE (KB) taken as is (16-byte-per-lane streaming stores); FETCH_SIZE calibrated on a kernel of the
same access pattern with a KNOWN byte count (reset_kernel in mode 1 reads n state words of 8 B and nothing else: the guide's
"reports exactly half for wide coalesced reads", re-measured here instead of assumed).

    python tools/make_traffic_json.py profiles/r03 1048576 [524288 262144 131072]

The extra sizes are the per-GPU shares of the 1 M-env batch at 2 / 4 / 8 GPUs (launch, stream and ring forms; summaries named
pmc_<mode>_n<size>_<counter>_summary.json), so that `bench.py --gpus N` can fill `roofline.traffic` at every N.
"""
import json
import os
import sys

d, n = sys.argv[1], int(sys.argv[2])
extra_sizes = [int(x) for x in sys.argv[3:]]
out = {"_how": {"source": "%s/pmc_{launch,stream,ring}_{FETCH,WRITE}_SIZE_summary.json" % d, "units": "bytes per launch",
                "write": "WRITE_SIZE KB x 1024 (exact for 16-B-per-lane streaming stores, MI355X_MICROARCH.md)",
                "fetch": "FETCH_SIZE KB x 1024 x calibration; calibration = known reset_kernel reads (8 B x n) / its FETCH_SIZE"}}


def load(mode, ctr, size=None):
    name = "pmc_%s_%s_summary.json" % (mode, ctr) if size is None else "pmc_%s_n%d_%s_summary.json" % (mode, size, ctr)
    with open(os.path.join(d, name)) as f:
        return json.load(f)


def pick(summary, needle, ctr):
    for k, v in summary.items():
        if needle in k and ctr in v:
            return v[ctr]["avg_per_dispatch"]
    raise KeyError(needle)


for mode, needle, key in (("launch", "step_kernel", "BoatRace-v0/compact/%d" % n),
                          ("stream", "rollout_random_kernel", "BoatRace-v0/compact/%d/stream100" % n),
                          ("ring", "rollout_random_kernel", "BoatRace-v0/compact/%d/ring100" % n)):
    f, w = load(mode, "FETCH_SIZE"), load(mode, "WRITE_SIZE")
    cal = 8.0 * n / 1024.0 / pick(f, "reset_kernel", "FETCH_SIZE")
    fetch_kb, write_kb = pick(f, needle, "FETCH_SIZE"), pick(w, needle, "WRITE_SIZE")
    out[key] = int(round((fetch_kb * cal + write_kb) * 1024))
    out["_how"][key] = {"fetch_kb_raw": fetch_kb, "fetch_calibration": cal, "write_kb": write_kb}
for m in extra_sizes:
    for mode, needle, key in (("launch", "step_kernel", "BoatRace-v0/compact/%d" % m),
                              ("stream", "rollout_random_kernel", "BoatRace-v0/compact/%d/stream100" % m),
                              ("ring", "rollout_random_kernel", "BoatRace-v0/compact/%d/ring100" % m)):
        if not os.path.exists(os.path.join(d, "pmc_%s_n%d_FETCH_SIZE_summary.json" % (mode, m))):
            continue
        f, w = load(mode, "FETCH_SIZE", m), load(mode, "WRITE_SIZE", m)
        cal = 8.0 * m / 1024.0 / pick(f, "reset_kernel", "FETCH_SIZE")
        fetch_kb, write_kb = pick(f, needle, "FETCH_SIZE"), pick(w, needle, "WRITE_SIZE")
        out[key] = int(round((fetch_kb * cal + write_kb) * 1024))
        out["_how"][key] = {"fetch_kb_raw": fetch_kb, "fetch_calibration": cal, "write_kb": write_kb}
with open(os.path.join(os.path.dirname(d.rstrip("/")), "traffic.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
