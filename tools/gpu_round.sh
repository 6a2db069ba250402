#!/bin/bash
# One GPU-box session: the gpu tests, the default bench line, the configs[4] line.  Outputs under gpurun_out/<tag>/.
tag=${1:-run}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q --maxfail=10 -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -15 $out/pytest.log
timeout 600 python3 bench.py > $out/bench_all.json 2> $out/bench_all.err; echo "bench rc=$?"
cat $out/bench_all.json | head -c 6000; tail -3 $out/bench_all.err
timeout 600 python3 bench.py --workload acq10ms --steps 10 --warmup 2 > $out/bench_acq10ms.json 2> $out/bench_acq10ms.err; echo "bench10 rc=$?"
cat $out/bench_acq10ms.json | head -c 3000; tail -3 $out/bench_acq10ms.err
