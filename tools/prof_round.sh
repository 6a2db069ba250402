#!/bin/bash
# Round profile set (GPU box): kernel trace + PMC passes per workload, every rocprofv3 under its own timeout.
# usage: tools/prof_round.sh <tag>          (run in the SAME gpurun call as the default bench.py line: tools/gpu_round.sh)
tag=$1
export TMPDIR=/tmp
for wl in acq acq59 acq10ms wf14 ddc14; do
  extra="--steps 200"; [ $wl = acq10ms ] && extra="--steps 20 --warmup 3"; [ $wl = wf14 ] && extra="--steps 400 --warmup 40"
  tools/prof.sh ${tag}_$wl --workload $wl $extra > gpurun_out/${tag}_$wl.summary.txt 2>&1
  echo "== $wl"; grep -E "calls|traced run" gpurun_out/${tag}_$wl.summary.txt | head -8
done
for wl in cfg2_chain receivers receivers_light; do
  extra="--steps 200"; case $wl in receivers*) extra="--steps 60 --warmup 4";; esac
  PROF_PMC=0 tools/prof.sh ${tag}_$wl --workload $wl $extra > gpurun_out/${tag}_$wl.summary.txt 2>&1
  echo "== $wl"; grep -E "calls|traced run" gpurun_out/${tag}_$wl.summary.txt | head -10
done
