"""Summarises the rocprofv3 csv outputs written by tools/prof.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


box = os.path.join(out, "box.json")
if os.path.isfile(box):
    print("== box " + open(box).read().strip())
tl = os.path.join(out, "trace.log")
if os.path.isfile(tl):                              # the bench line the traced run itself printed (its step time, same box)
    for line in open(tl):
        if line.startswith("{"):
            import json
            try:
                d = json.loads(line)
                rf = d.get("roofline", {})
                print("== traced run's own line: ms_per_step %s kernel_ms %s frac %s value %s %s" % (
                    d.get("ms_per_step"), rf.get("kernel_ms"), rf.get("frac"), d.get("value"), d.get("unit")))
            except ValueError:
                pass

for f in find("trace", "*kernel_stats.csv"):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("  %-60s calls %6s  avg %10.2f us  total %8.3f ms  %5s%%" % (
            r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3,
            float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))

for sub in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("== counters (%s), mean per dispatch" % os.path.relpath(f, out))
        for k, cs in acc.items():
            if "correlate" not in k and "fft_fwd" not in k and "frontend" not in k and "wf_" not in k \
                    and "fir_" not in k and "ddc_" not in k:
                continue
            print("  " + k[:90])
            for c, v in sorted(cs.items()):
                print("      %-28s %16.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))
