"""Fills the results table of BASELINE.md section 2 from a default bench.py line (python tools/fill_baseline.py
profiles/<line>.json).  Every figure is that run's own: the GPU column is `value` of the workload on one MI355X, the
CPU columns the same run's cpu_baseline legs on the GPU box's host cores, HBM GB/s the live rocprofv3 --pmc traffic
divided by the kernel (or step) time.  2 / 4 / 8 GPUs: unmeasured (no multi-GPU node was available to this round)."""
import json
import re
import sys

line = json.load(open(sys.argv[1]))
w = line["workloads"]


def row(cfg, wl, what):
    x = w[wl]
    cb = x.get("cpu_baseline", {})
    pf = x.get("cpu_baseline_pocketfft", {})
    hbm = x.get("hbm", {})
    gbps = hbm.get("measured_GBps")
    if gbps is None and x["roofline"].get("traffic") and x["roofline"].get("kernel_ms"):
        gbps = round(x["roofline"]["traffic"] / (x["roofline"]["kernel_ms"] * 1e-3) / 1e9, 1)
    t1 = cb.get("single_thread_value")
    tall = "%s (%d threads)" % (cb.get("value"), cb.get("cores", 0)) if cb else "–"
    if pf.get("value"):
        tall += "; pocketfft %s" % pf["value"]
    return "| %s | %s | %s | %s | unmeasured | unmeasured | unmeasured | %s (%s) | %s | %s |" % (
        cfg, t1 if t1 is not None else "–", tall, "%.4g ms/step" % x["ms_per_step"], x["value"], what,
        gbps if gbps is not None else "–", ("%.2f %%" % (100.0 * gbps / 8000.0)) if gbps is not None else "–")


rows = [
    "| 0 | (smoke / `tests/test_acq_gpu.py::test_config0_prn1`: the reference's own CPU case, checked, not timed) | | n/a | | | | | | |",
    row("1", "acq", "IQ Msamples/s"),
    row("2 frames", "wf14", "IQ Msamples/s of DDC output"),
    row("2 DDC", "ddc14", "ADC Msamples/s"),
    row("2 end to end", "cfg2_chain", "ADC Msamples/s"),
    row("3 (128 receivers = one GPU's share; SURVEY's mix)", "receivers", "receiver x ADC Msamples/s"),
    row("3 (the same on rounds 2-4's lighter receiver set)", "receivers_light", "receiver x ADC Msamples/s"),
    row("4 (2 blocks per step)", "acq10ms", "IQ Msamples/s"),
]
table = ("| Config | CPU T₁ (Msamples/s) | CPU T_all (Msamples/s) | 1 GPU | 2 GPU | 4 GPU | 8 GPU | value (unit) | HBM GB/s (rocprof) | % of 8 TB/s |\n"
         "|---|---|---|---|---|---|---|---|---|---|\n" + "\n".join(rows))
src = open("BASELINE.md").read()
start = src.index("Results table")
out = src[:start] + ("Results table, filled from the default `python bench.py` line of round 6 (`%s`; one MI355X; CPU legs: the oracle port\n"
                     "and, where the work goes through an FFT, scipy.fft / pocketfft, on the GPU box's host cores in the same run):\n\n" % sys.argv[1]
                     ) + table + "\n" + NOTES if (NOTES := """
Notes recorded beside the configs:
* configs[4] uses the FFT bin as its Doppler step: 4.092 MHz / 65536 = **62.44 Hz**, 256 bins = −128..127
  (SURVEY §8(d) wrote "100 Hz": that is 1/T of a 10 ms integration; the circular-shift formulation steps in bins of
  the transform, DESIGN.md §2.4).  Its detection threshold is **30** (`synth.MIN_SIG_10MS`), not the reference's
  `MIN_SIG` = 16 (`gps/gps.h:60`), which is sized for 41 × 4092 trials per SV: 256 × 4092 (E1B: 16368) lags are 1.0 M
  (4.2 M) trials whose noise maximum alone reaches 14 … 19.
* configs[3]'s line is ONE GPU's share (128 of the 1024 receivers, `--receivers 128`); `bench.py --gpus 8 --workload
  receivers` runs the eight shares, one rank per GPU, no data-path collective.  Since round 5 the receivers are SURVEY.md
  8(d)'s -- receiver k at 100 kHz + k 29 kHz, zoom 8 + (k mod 4) -- stepped by ONE C-ABI call (`kg_rxbank_step`): zooms
  8..10 take the reference's non-overlapped frame (`CmdWFReset` + one-shot sampler), zoom 11 its overlapped / continuous
  sampler (`rx/rx_waterfall.cpp:962-1041`), and so does the CPU baseline.  Rounds 2-4 ran a lighter set (zoom 1 + k mod 10,
  a quarter of the waterfall DDC work): kept as `receivers_light`; neither is comparable with round 3's figure (continuous
  sampler over the whole block for every receiver).
* configs[2] DDC / end to end: since round 4 the steps walk nine distinct 32 MiB ADC blocks (past the Infinity Cache).
* `value` is the HBM-resident rate; the PCIe-inclusive rate of configs[1] is `ingest_pcie_Msps` in the same line.
""") else ""
open("BASELINE.md", "w").write(out)
print(table)
