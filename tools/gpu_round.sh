#!/bin/bash
# One GPU-box session: the gpu tests, then the default bench line (all six workloads).  Outputs under gpurun_out/<tag>/.
tag=${1:-run}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q --maxfail=10 -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -15 $out/pytest.log
start=$(date +%s)
timeout 900 python3 bench.py > $out/bench_all.json 2> $out/bench_all.err; echo "bench rc=$? in $(( $(date +%s) - start )) s"
head -c 12000 $out/bench_all.json; tail -5 $out/bench_all.err
