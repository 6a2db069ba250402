#!/bin/bash
# One GPU-box session (one gpurun lease = one box): the gpu tests, the default bench line (all six workloads), then -- with
# PROF=1 -- the round's profile set of the same library on the same box.  Outputs under gpurun_out/<tag>/.
tag=${1:-run}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 -c "import json, bench; print(json.dumps(bench.box_identity()))" | tee $out/box.json
if [ "${TESTS:-1}" = 1 ]; then
timeout 1800 python3 -m pytest tests -m gpu -q --maxfail=10 -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -15 $out/pytest.log
fi
start=$(date +%s)
# the driver's own command: the compact line on stdout (<= 8 KB), the whole object in gpurun_out/bench_full.json
timeout 900 python3 bench.py > $out/bench_line.json 2> $out/bench_all.err; echo "bench rc=$? in $(( $(date +%s) - start )) s"
echo "stdout line: $(wc -c < $out/bench_line.json) bytes"; cat $out/bench_line.json
cp gpurun_out/bench_full.json $out/bench_all.json
python3 tools/show_line.py $out/bench_all.json; grep -v "^FULL " $out/bench_all.err | tail -12
if [ "${PROF:-0}" = 1 ]; then tools/prof_round.sh ${tag}p; fi
