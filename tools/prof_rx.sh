#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/prof_rx; mkdir -p $out
timeout 300 python3 bench.py --full-line --no-cpu --no-live-traffic --workload receivers --steps 40 --warmup 4 2> $out/err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['step_ms_spread']
print('receivers step %.4f ms cadence of 3-step chunks (min %.4f med %.4f max %.4f) host enqueue %.1f us' % (d['ms_per_step'], s['min'], s['median'], s['max'], d['host_enqueue_us_per_step']))"
KIWIGPU_BENCH_PREROLL_S=0.12 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --full-line --no-cpu --no-live-traffic --workload receivers --steps 40 --warmup 4 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print("%-60s calls %5s avg %8.1f us min %8.1f max %8.1f %5.1f%%"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
