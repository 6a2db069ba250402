#!/bin/bash
# Round profile set (GPU box): kernel trace + PMC passes per workload, every rocprofv3 under its own timeout.
# usage: tools/prof_round.sh <tag>
tag=$1
export TMPDIR=/tmp
export KIWIGPU_BENCH_PREROLL_S=0.12     # fewer traced launches
for wl in acq acq59 acq10ms wf14; do
  extra="--steps 40"; [ $wl = acq10ms ] && extra="--steps 8 --warmup 2"; [ $wl = wf14 ] && extra="--steps 400 --warmup 40"
  tools/prof.sh ${tag}_$wl --workload $wl $extra > gpurun_out/${tag}_$wl.summary.txt 2>&1
  echo "== $wl"; grep -E "calls" gpurun_out/${tag}_$wl.summary.txt | head -6
done
for wl in ddc14 cfg2_chain receivers; do
  out=gpurun_out/${tag}_$wl; mkdir -p $out
  extra="--steps 40"; [ $wl = receivers ] && extra="--steps 20 --warmup 3"
  [ $wl = ddc14 ] && { tools/prof.sh ${tag}_$wl --workload $wl $extra > gpurun_out/${tag}_$wl.summary.txt 2>&1; echo "== $wl"; grep -E "calls" gpurun_out/${tag}_$wl.summary.txt | head -7; continue; }
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu --no-live-traffic --workload $wl $extra > $out/trace.log 2>&1
  echo "== $wl"; f=$(find $out/trace -name "*kernel_stats.csv" | head -1); head -8 $f | cut -d, -f1-5 | cut -c1-140
  find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
done
