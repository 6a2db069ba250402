export TMPDIR=/tmp
tools/prof_round.sh r03p 2>&1 | tail -60
python3 tools/dbg_stamps_e1b.py 32 > gpurun_out/r03_e1b8_stamps.txt 2>&1
tail -3 gpurun_out/r03_e1b8_stamps.txt
