#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes for bench.py.
# usage: tools/prof.sh <tag> [bench args...]
# The trace pass runs with bench.py's default pre-roll, so that its per-kernel averages are steady-state averages (the
# cold launches of a 0.12 s pre-roll were 10 % of a short trace and pulled its averages above the bench's step time);
# the counter passes, whose values do not depend on the clock, keep the short one.
set -u
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
args="$@"
python3 -c "import json, bench; print(json.dumps(bench.box_identity()))" > $out/box.json 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --full-line --no-cpu --no-live-traffic $args > $out/trace.log 2>&1
export KIWIGPU_BENCH_PREROLL_S=${KIWIGPU_BENCH_PREROLL_S:-0.12}     # fewer counted launches
if [ "${PROF_PMC:-1}" = 1 ]; then
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/pmc1 -- python3 bench.py --full-line --no-cpu --no-live-traffic $args > $out/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $out/pmc2 -- python3 bench.py --full-line --no-cpu --no-live-traffic $args > $out/pmc2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -- python3 bench.py --full-line --no-cpu --no-live-traffic $args > $out/pmc3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc4 -- python3 bench.py --full-line --no-cpu --no-live-traffic $args > $out/pmc4.log 2>&1
fi
python3 tools/prof_summary.py $out > $out/summary.txt 2>&1
# the raw per-dispatch files are tens of MB per pass (gpurun brings back 64 MiB at most): the summary and the stats stay
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
cat $out/summary.txt
