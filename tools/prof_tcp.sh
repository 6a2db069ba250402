#!/bin/bash
# L1 / TA / L2-path counters of the acquisition bench (GPU box).  usage: tools/prof_tcp.sh <tag> [bench args]
set -u
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
i=0
# at most two counters of a block per pass (more: "exceeds the capabilities of the hardware" and
# rocprofv3 hangs after the abort) and a timeout of its own around every pass
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_BUSY_sum" \
           "GRBM_GUI_ACTIVE TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 bench.py --full-line --no-cpu --no-live-traffic --workload acq --steps 30 --warmup 5 "$@" > $out/p$i.log 2>&1
  echo "pass $i ($set): rc $?"
done
python3 - $out <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "correlate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()):
        print("%-40s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
