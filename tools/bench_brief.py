# one-line digest of a bench.py JSON line on stdin
import json, sys
for l in sys.stdin:
    if not l.startswith("{"): continue
    d = json.loads(l)
    r = d.get("roofline") or {}
    print("value %.1f (min %.1f max %.1f) ms/step %.4f loop_us/step %.1f setup_us %.1f solve_us/step %.1f pcg_it %d acc %d parity %s" % (
        d["value"], d["value_min"], d["value_max"], d["ms_per_step"], d["loop_seconds"] / max(d["steps_run"], 1) * 1e6, d["setup_seconds"] * 1e6,
        d["solve_seconds"] / max(d["steps_run"], 1) * 1e6, d["pcg_iterations"], d["accepted_steps"], d.get("parity_rel")))
    print("dominant %s %.2f us frac %.4f | " % (r.get("kernel"), r.get("avg_launch_us", 0), r.get("frac", 0)) + " ".join("%s %.1f" % (k, v["avg_us"]) for k, v in (r.get("kernels") or {}).items()))
    for a in d.get("also") or []:
        print("also:", a["workload"][:60], a["value"])
