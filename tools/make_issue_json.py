"""profiles/issue.json from the round's SQ counter passes over the outputs-once rollout kernel (rocprofv3 --pmc SQ_INSTS_VALU
SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH ..., tools/pmc_run.py <env> compact <n> fused) and from tools/exp_issue_peak.hip's
table: instructions per WAVE-STEP (one lockstep step of one 64-env wave) and the chip's issue peaks at 8 waves per SIMD.
bench.py's `fused_rollout.roofline` multiplies the first by the run's own wave-steps per second and divides by the second.

    python tools/make_issue_json.py profiles/r03 1048576 1000 profiles/r04
(the last argument: where the tabular-Q rollout's SQ pass lives, pmc_sq1_tabq_rollout.json)
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
names = {"boatrace": "BoatRace-v0", "tomatowatering": "TomatoWatering-v0", "islandnavigation": "IslandNavigation-v0",
         "sideeffectssokoban": "SideEffectsSokoban-v0"}
out = {"_how": "instructions per wave-step = SQ_INSTS_* per launch / (n / 64 * steps per launch); peaks: tools/exp_issue_peak.hip at 8 "
               "waves per SIMD (profiles/r03/issue_peak.log)"}
wave_steps = n / 64 * steps
for tag, env in names.items():
    p = os.path.join(ROOT, src, "pmc_sq_rollout_%s_fused.json" % tag)
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    for k, v in d.items():
        if "rollout_random_kernel" in k and "false" in k:
            c = {name: x["avg_per_dispatch"] for name, x in v.items()}
            out["%s/outputs_once" % env] = {
                "valu_per_wave_step": c["SQ_INSTS_VALU"] / wave_steps, "salu_per_wave_step": (c["SQ_INSTS_SALU"]) / wave_steps,
                "branch_per_wave_step": c.get("SQ_INSTS_BRANCH", 0) / wave_steps, "lds_per_wave_step": c["SQ_INSTS_LDS"] / wave_steps,
                "source": os.path.join(src, os.path.basename(p)), "n_envs": n, "steps_per_launch": steps}
# the LDS-resident tabular-Q rollout at config 3's shape (tools/gpu_pmc_tabq.sh: 262 144 agents, 500 steps per launch)
tq_dir = sys.argv[4] if len(sys.argv) > 4 else "profiles/r04"
tp = os.path.join(ROOT, tq_dir, "pmc_sq1_tabq_rollout.json")
if os.path.exists(tp):
    d = json.load(open(tp))
    tq_n, tq_steps = 262144, 500
    for k, v in d.items():
        if "tabq_rollout_kernel" in k:
            c = {name: x["avg_per_dispatch"] for name, x in v.items()}
            ws = tq_n / 64 * tq_steps
            out["IslandNavigation-v0/tabq_rollout"] = {
                "valu_per_wave_step": c["SQ_INSTS_VALU"] / ws, "salu_per_wave_step": c["SQ_INSTS_SALU"] / ws,
                "branch_per_wave_step": c.get("SQ_INSTS_BRANCH", 0) / ws, "lds_per_wave_step": c["SQ_INSTS_LDS"] / ws,
                "wave_cycles_per_wave_step": 4 * c["SQ_WAVE_CYCLES"] / ws, "waves_per_launch": c["SQ_WAVES"],
                "source": os.path.join(tq_dir, os.path.basename(tp)), "n_agents": tq_n, "steps_per_launch": tq_steps}
peak = {}
for line in open(os.path.join(ROOT, src, "issue_peak.log")):
    m = re.match(r"(VALU|SALU)\s+8 \|\s+[\d.]+\s+([\d.]+)", line)
    if m:
        peak["%s_wave_instr_per_s" % m.group(1).lower()] = float(m.group(2)) * 1e9
peak["source"] = os.path.join(src, "issue_peak.log")
out["peak"] = peak
json.dump(out, open(os.path.join(ROOT, "profiles", "issue.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
