#!/usr/bin/env python3
"""Pass / fail verdict for `bench.py --gpus {1,2,4,8}` lines against BASELINE.md section 10 (the prediction committed BEFORE any
multi-GPU run: no node with more than one GPU was ever available to the build).

    python tools/check_scale.py line_n1.json line_n2.json ...        (files holding bench.py's ONE JSON line, or a driver record
                                                                        with the line under "parsed"; "-" reads lines from stdin)

Per line it checks
  ranks       rccl_ranks == n_gpus for N > 1 (the library's RCCL communicator really spans the job)
  collective  metrics_collective names the library's all-reduce (sgk_metrics_allreduced (RCCL)), not a torch.distributed fallback
  episodes    episodes_finished == total_envs x steps x lockstep_steps_per_step / 100 (every env finishes one BoatRace episode per
              100 lockstep steps) -- the all-reduced metrics cover every rank's shard
  parity      parity_sample_bit_exact (and every ring slice checked)
  balance     per_rank_device_us within 5 % of each other (max / min <= 1.05)
  value       within -15 % / +10 % of BASELINE section 10's row for that N (the N = 1 row: 1.95-2.18e11 measured)
and prints one verdict line per input plus a final `SCALE PASS` / `SCALE FAIL` (exit code 0 / 1). `--allow-collective gloo` accepts
the CPU-side dry runs (ranks on one GPU over gloo: profiles/r05/bench_*rank_one_gpu_gloo.log) for the checks that still mean
something there; their `value` is not a multi-GPU number and is reported, not judged.
"""
import argparse
import json
import sys

# predicted whole-job env-steps/s at 1 048 576 envs split over N GPUs (strong scaling), and the band a
# measured line must fall in: -15 % / +10 % (N = 1: the measured range of five rounds' boxes, widened the same way)
PREDICTED = {1: (1.95e11, 2.18e11), 2: (4.4e11, 4.4e11), 4: (7.6e11, 7.6e11), 8: (1.66e12, 1.66e12)}
LOW, HIGH = 0.85, 1.10
LIBRARY_COLLECTIVE = "sgk_metrics_allreduced (RCCL)"


def _find_lines(obj, out):
    """every dict that looks like a bench line (has metric + n_gpus + value), wherever a driver record keeps it"""
    if isinstance(obj, dict):
        if "metric" in obj and "n_gpus" in obj and "value" in obj:
            out.append(obj)
            return
        for v in obj.values():
            _find_lines(v, out)
    elif isinstance(obj, list):
        for v in obj:
            _find_lines(v, out)


def load_lines(path):
    text = sys.stdin.read() if path == "-" else open(path).read()
    out = []
    try:
        _find_lines(json.loads(text), out)
        return out
    except ValueError:
        pass
    for line in text.splitlines():  # a log with the JSON line somewhere in it
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                _find_lines(json.loads(line), out)
            except ValueError:
                continue
    return out


def check(line, allow_collective=()):
    """-> (ok, [(name, ok | None, detail)]) ; None = not applicable / reported only"""
    n = int(line.get("n_gpus", 0))
    res = []
    ranks = line.get("rccl_ranks")
    coll = str(line.get("metrics_collective", ""))
    dry = any(a in coll for a in allow_collective)
    if n > 1 and "rccl_ranks" not in line and "metrics_collective" not in line:
        res.append(("ranks", None, "not in this record"))
        res.append(("collective", None, "not in this record"))
    elif n > 1:
        res.append(("ranks", None if dry else ranks == n, "rccl_ranks %r, n_gpus %d" % (ranks, n)))
        res.append(("collective", True if dry else coll == LIBRARY_COLLECTIVE, coll))
    else:
        res.append(("ranks", None, "one GPU: no communicator"))
        res.append(("collective", None, coll or "none"))
    cfg = line.get("config") or {}
    total = int(cfg.get("total_envs", 0) or 0)
    per100 = int(line.get("lockstep_steps_per_step", 100))
    want_eps = total * int(line.get("steps", 0)) * per100 // 100
    # (a driver record's "parsed" line keeps the contract's keys only: a field that is not in the record is reported, not judged)
    if "episodes_finished" in line:
        res.append(("episodes", line.get("episodes_finished") == want_eps and want_eps > 0,
                    "episodes_finished %r, expected %d" % (line.get("episodes_finished"), want_eps)))
    else:
        res.append(("episodes", None, "not in this record"))
    slices = line.get("ring_slices_checked_bit_exact")
    if "parity_sample_bit_exact" in line:
        res.append(("parity", bool(line.get("parity_sample_bit_exact")) and slices is not False,
                    "parity_sample_bit_exact %r, ring slices %r" % (line.get("parity_sample_bit_exact"), slices)))
    else:
        res.append(("parity", None, "not in this record"))
    per_rank = line.get("per_rank_device_us") or []
    if n > 1 and len(per_rank) == n and min(per_rank) > 0:
        spread = max(per_rank) / min(per_rank)
        res.append(("balance", None if dry else spread <= 1.05, "max / min device time over ranks %.3f" % spread))
    elif n > 1 and "per_rank_device_us" not in line:
        res.append(("balance", None, "not in this record"))
    elif n > 1:
        res.append(("balance", False, "per_rank_device_us has %d entries for %d ranks" % (len(per_rank), n)))
    else:
        res.append(("balance", None, "one rank"))
    value = float(line.get("value", 0.0))
    if n in PREDICTED and line.get("scaling", "strong" if n > 1 else "weak") in ("strong", "weak") and total == 1 << 20:
        lo, hi = PREDICTED[n][0] * LOW, PREDICTED[n][1] * HIGH
        res.append(("value", None if dry else lo <= value <= hi, "%.3g env-steps/s, BASELINE section 10 band %.3g .. %.3g" % (value, lo, hi)))
    else:
        res.append(("value", None, "%.3g env-steps/s (no prediction for n_gpus %d at %d envs)" % (value, n, total)))
    return all(ok is not False for _, ok, _ in res), res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("files", nargs="+")
    ap.add_argument("--allow-collective", action="append", default=[], metavar="SUBSTRING",
                    help="accept lines whose metrics_collective contains this (e.g. gloo: the one-GPU dry runs); their ranks, balance and "
                         "value are reported, not judged")
    args = ap.parse_args(argv)
    all_ok, seen = True, 0
    for path in args.files:
        lines = load_lines(path)
        if not lines:
            print("FAIL %s: no bench line found" % path)
            all_ok = False
        for line in lines:
            seen += 1
            ok, res = check(line, tuple(args.allow_collective))
            all_ok &= ok
            parts = ["%s %s (%s)" % (name, "ok" if r else ("n/a" if r is None else "FAIL"), detail) for name, r, detail in res]
            print("%s n_gpus=%s %s: %s" % ("PASS" if ok else "FAIL", line.get("n_gpus"), path, "; ".join(parts)))
    print("SCALE %s (%d line%s)" % ("PASS" if all_ok and seen else "FAIL", seen, "" if seen == 1 else "s"))
    return 0 if all_ok and seen else 1


if __name__ == "__main__":
    sys.exit(main())
