"""Host enqueue time and wall time per step of a receiver bank (kg_rxbank): `python tools/time_rxbank.py [mix] [nrx] [steps]`."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (the library's constructor does the same; torch is imported first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import synth                                   # noqa: E402
from flydog_sdr_gps_amd.rxbank import MIXES, RxBank                     # noqa: E402

if os.environ.get("KG_TOOL_TORCH") == "1":          # the bench's situation: torch owns the device before the bank's streams exist
    import torch
    torch.zeros(16, device="cuda").sum().item()
mix_name = sys.argv[1] if len(sys.argv) > 1 else "survey"
nrx = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
n = 1 << 22
bank = RxBank(nrx, n)
bank.configure(MIXES[mix_name](nrx, 0, n))
adc = synth.adc_stream(n, 0x5EED0004)
d_adc = bank.ctx.alloc(adc.nbytes)
bank.ctx.upload(d_adc, adc)
for _ in range(12):
    info = bank.step(d_adc)
bank.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps):
        bank.step_fast(d_adc)
    t1 = time.perf_counter()
    bank.sync()
    t2 = time.perf_counter()
    print("%s mix, %d receivers: host enqueue %.1f us per step, wall %.4f ms per step (table %d bytes, frames %d)"
          % (mix_name, nrx, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e3, info.table_bytes, info.nframes), flush=True)
# the host's share with the GPU idle between steps: enqueue, then drain, per step
bank.host_profile()
t_enq = 0.0
for _ in range(30):
    t0 = time.perf_counter()
    bank.step_fast(d_adc)
    t_enq += time.perf_counter() - t0
    bank.sync()
print("%s mix: host enqueue with an idle GPU %.1f us per step" % (mix_name, t_enq / 30 * 1e6))
print(bank.host_profile())
bank.close()
