"""Where the streaming (double-buffered ADC ring) loop of a receiver bank loses time: variants of the loop, ms per step."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flydog_sdr_gps_amd import synth
from flydog_sdr_gps_amd.rxbank import MIXES, RxBank
n, nrx = 1 << 22, 128
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
bank = RxBank(nrx, n)
bank.configure(MIXES["survey"](nrx, 0, n))
adc = synth.adc_stream(n, 0x5EED0004)
host = torch.from_numpy(adc).pin_memory()
ring = [torch.from_numpy(adc).to(dev) for _ in range(9)]
up = torch.cuda.Stream(device=dev)
evs = [torch.cuda.Event() for _ in range(9)]
def run(name, copy, done, ready, steps=60, nb=2):
    def one(k):
        if copy or done or ready:
            with torch.cuda.stream(up):
                if done: bank.adc_done(up.cuda_stream, min(nb, 8))
                if copy: ring[k % nb].copy_(host, non_blocking=True)
                evs[k % nb].record(up)
        bank.step(ring[k % nb].data_ptr(), adc_ready_event=evs[k % nb].cuda_event if ready else None)
    for k in range(8): one(k)
    bank.sync(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps): one(k)
    t1 = time.perf_counter()
    bank.sync(); torch.cuda.synchronize(dev)
    print("%-46s host loop %.3f ms, wall %.4f ms per step" % (name, (t1 - t0) / steps * 1e3, (time.perf_counter() - t0) / steps * 1e3), flush=True)
run("resident (no copy, no events)", False, False, False)
run("events only (no copy)", False, True, True)
run("copy, no dependencies (racy: timing only)", True, False, False)
run("copy + adc_ready only", True, False, True)
run("copy + adc_done only", True, True, False)
run("copy + both (the streaming loop)", True, True, True)
run("nine buffers: copy + adc_ready", True, False, True, nb=9)
run("nine buffers: copy, no dependencies", True, False, False, nb=9)
run("nine buffers: no copy, no events", False, False, False, nb=9)
run("four buffers: copy + both", True, True, True, nb=4)
run("eight buffers: copy + both", True, True, True, nb=8)
run("resident again", False, False, False)
bank.close()
