# GPU box, one lease: the one-GPU rehearsal of the N = 2 line, the default N = 1 line, the receivers trace
export TMPDIR=/tmp
echo "== rehearsal: 2 ranks on one GPU over gloo"
KIWIGPU_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --full-line --gpus 2 --steps 20 --warmup 3 > gpurun_out/rehearsal2.json 2> gpurun_out/rehearsal2.err; echo "rc $?"; tail -3 gpurun_out/rehearsal2.err
echo "== default line"
timeout 900 python3 bench.py --full-line > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "rc $?"; tail -2 gpurun_out/bench_default.err
echo "== receivers trace"
bash tools/prof_rx.sh 2>&1 | tail -30
