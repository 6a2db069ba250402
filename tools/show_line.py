"""Prints the interesting numbers of a bench.py line: python tools/show_line.py <file with the JSON line>"""
import json
import sys

lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not lines:
    print("no JSON line in %s (the bench failed: see its stderr)" % sys.argv[1])
    sys.exit(0)
d = json.loads(lines[-1])


def show(name, r):
    rf = r.get("roofline", {})
    sp = r.get("step_ms_spread") or {}
    print("%-11s value %-10s %-12s ms/step %-8s spread %s/%s/%s  %s %s %s frac %s kernel_ms %s traffic %s checked %s" % (
        name, r.get("value"), r.get("unit"), r.get("ms_per_step"), sp.get("min"), sp.get("median"), sp.get("max"), rf.get("bound"),
        rf.get("achieved"), rf.get("unit"), rf.get("frac"), rf.get("kernel_ms"), rf.get("traffic"),
        {k: v for k, v in (r.get("checked") or {}).items() if k != "rule"}))


if "workloads" in d:
    for k, v in d["workloads"].items():
        show(k, v)
    print("by_workload", json.dumps(d["roofline"].get("by_workload")))
else:
    show(d.get("config", {}).get("workload", "?")[:10], d)
print("box", d.get("box"))
