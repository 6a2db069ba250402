#!/bin/bash
# L1 / TA / L2-path counters of the waterfall DDC kernels (GPU box).  usage: tools/prof_tcp_ddc.sh <tag> [bench args]
set -u
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCC_REQ_sum TCC_BUSY_sum" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 bench.py --full-line --no-cpu --no-live-traffic --workload ddc14 --steps 10 --warmup 3 "$@" > $out/p$i.log 2>&1
  echo "pass $i ($set): rc $?"
done
python3 - $out <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ddc_wf" not in k:
            continue
        k = k.replace("void ", "")[:26]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("    %-36s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
