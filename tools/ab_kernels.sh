#!/bin/bash
# Per-kernel A/B on the GPU box: tools/ab_kernels.sh "<bench args>" variant...   (rocprofv3 kernel trace of each)
args=$1; shift
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib=$PWD/flydog_sdr_gps_amd/libkiwigpu_$v.so; fi
  out=/tmp/abk_$v; rm -rf $out
  KIWIGPU_LIBRARY=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --full-line --no-cpu --no-live-traffic $args > /dev/null 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("  %-48s calls %5s avg %10.2f us" % (r["Name"][:48], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
