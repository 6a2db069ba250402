# GPU box: runs per channel of the two DDCs at 128 receivers (survey mix), time and the ADPCM kernels' new forms
export KIWIGPU_TUNING=1
for rx in 8192 4096 2048; do for wf in 0 4096 2048; do
  export KIWIGPU_RXDDC_RUNS=$rx; if [ $wf = 0 ]; then unset KIWIGPU_DDC_RUNS; else export KIWIGPU_DDC_RUNS=$wf; fi
  echo "== rx runs $rx, wf runs ${wf}"; python3 tools/time_rxbank.py survey 128 60 2>&1 | grep "wall" | tail -1
done; done
unset KIWIGPU_TUNING KIWIGPU_RXDDC_RUNS KIWIGPU_DDC_RUNS
python3 -m pytest tests/test_wire_gpu.py tests/test_receivers_gpu.py tests/test_post_gpu.py -x -q 2>&1 | tail -3
