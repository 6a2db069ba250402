#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X acquisition hot path.

Workload (BASELINE.json configs[1]): 32 GPS L1 C/A SVs x 41 Doppler bins, 4 ms
coherent FFT correlation on synthetic int16 IQ (65536 samples @ 16.368 MS/s),
the sample block already resident in HBM.  One step = Sample() front end
(mix, 2x half-band /2, 16384-pt FFT) + Correlate() for all 32 SVs x 41 bins +
best-bin selection, for --blocks independent 4 ms sample blocks (default 32: a
128 ms / 8 MiB batch of the IQ stream per step; --blocks 1 is the single-block latency case).

Metric: IQ Msamples/s ingested = blocks * 65536 * steps * n_gpus / seconds.
Multi-GPU (weak scaling): every rank searches its own resident block(s); there
is no data-path collective; results are all-gathered over RCCL
after the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks B] [--no-cpu]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NSAMPLES = 65536
FFT_LEN = 16384
NSV = 32
NDOP = 41
# algorithmic bytes of one (SV, Doppler) cell = data spectrum +
# code spectrum read once (2 * N * 8 B) + a 16-byte result.
BYTES_PER_CELL = 2 * FFT_LEN * 8 + 16
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FLOPS_PER_CELL = 4 * 5 * 4096 * 12 + 6 * FFT_LEN + 3 * 4096 * 10


def usable_cores():
    """Host threads this process can really run at once: the affinity mask, capped by the
    cgroup CPU quota (os.cpu_count() reports the machine, not the container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:      # cgroup v1
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(iq, chips_list, budget_s=12.0):
    """The CPU oracle (kind "port": fp32 FFT) timed on the same configs[1] workload: every
    host thread runs whole single-threaded steps (Sample + 32 SV x 41 bins) on its own,
    back to back for about budget_s seconds -- independent blocks, like the GPU's batch,
    and no per-call thread start-up in the timed loop (ctypes drops the GIL in the calls)."""
    import threading
    from oracle import kiwi_oracle as ko
    ko.lib()
    cores = usable_cores()
    codes = np.stack([ko.code_fft(c, prec=0) for c in chips_list])
    limits = [4092] * len(chips_list)

    def one_step():
        data = ko.sample_iq16(iq, prec=0)
        ko.correlate_many(codes, data, limits, prec=0, nthreads=1, want_cells=False)

    t1_0 = time.perf_counter()
    one_step()                                                    # warm-up + single-thread figure
    t1 = time.perf_counter() - t1_0
    done = [0] * cores
    deadline = time.perf_counter() + budget_s

    def worker(k):
        while time.perf_counter() < deadline:
            one_step()
            done[k] += 1

    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    el = time.perf_counter() - t0
    reps = sum(done)
    return {
        "value": round(reps * NSAMPLES / el / 1e6, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "machine_cpus": os.cpu_count(),
        "kind": "port",
        "sample": "%d x the full configs[1] block (Sample + 32 SV x 41 bins), oracle fp32 FFT, "
                  "%d threads each running whole blocks, %.1f s" % (reps, cores, el),
        "single_thread_value": round(NSAMPLES / t1 / 1e6, 4),
    }


def bench_waterfall(args):
    """Secondary workload (not the headline): waterfall frames, BASELINE configs[2]'s
    14-channel zoom set, DDC output buffers (8192 int16 IQ per frame) resident in HBM.
        python bench.py --workload waterfall [--frames F]"""
    import torch
    from flydog_sdr_gps_amd import Context, Waterfall, WfParams, synth, wf
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = Context(0, torch.cuda.current_stream(dev).cuda_stream)
    zooms = [0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14]
    w = Waterfall(ctx, nchan=len(zooms))
    w.set_tables()
    for ch, z in enumerate(zooms):
        w.set_channel(ch, WfParams.for_zoom(z, 1.0e6 * ch), interp=wf.WF_CMA, cic_comp=True)
    F = args.frames
    nfr = F * len(zooms)
    base = np.stack([synth.wf_iq_frame(seed=i) for i in range(32)])
    iq = torch.from_numpy(base[np.arange(nfr) % 32].copy()).to(dev)
    out = torch.empty((nfr, 1024), dtype=torch.uint8, device=dev)
    chan_of = np.arange(nfr, dtype=np.int32) % len(zooms)
    for _ in range(args.warmup):
        w.frames_dev(chan_of, iq.data_ptr(), out.data_ptr())
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w.frames_dev(chan_of, iq.data_ptr(), out.data_ptr())
    kernel_ms = ctx.timer_stop() / args.steps
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    bytes_frame = 8192 * 4 + 1024          # int16 IQ in + u8 row out; tables are L2-resident
    achieved = nfr * bytes_frame / (kernel_ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "waterfall IQ Msamples/s (window + 8192-pt FFT + pixel reduce + dB + u8)",
        "value": round(nfr * 8192 * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "waterfall frames, 14 channels (zooms %s) x %d frames per step" % (zooms, F),
                   "frames_per_step": nfr},
        "roofline": {"bound": "hbm", "kernel": "wf_frame_kernel", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": measured_traffic("waterfall", nfr), "kernel_ms": round(kernel_ms, 5),
                     "algorithmic_bytes_per_launch": nfr * bytes_frame},
    }), flush=True)
    w.close()
    ctx.close()


def bench_ddc(args):
    """Secondary workload: BASELINE configs[2] -- 14 waterfall channels, NCO mix + CIC
    decimate (+ 8192-pt FFT frames) on a synthetic 16-bit ADC stream resident in HBM.
        python bench.py --workload ddc [--log2n 24]"""
    import torch
    from flydog_sdr_gps_amd import Context, Ddc, Waterfall, WfParams, wf
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = Context(0, torch.cuda.current_stream(dev).cuda_stream)
    zooms = [0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14]
    n = 1 << args.log2n
    rng = np.random.Generator(np.random.PCG64(0x5EED0003))
    t = np.arange(n, dtype=np.float64)
    x = rng.normal(0, 10.0, n)
    for f, a in ((0.0123, 3000.0), (0.071, 300.0), (0.2003, 30.0), (0.31, 3.0)):
        x += a * np.cos(2 * np.pi * f * t)
    adc = torch.from_numpy(np.clip(np.rint(x), -32768, 32767).astype(np.int16)).to(dev)
    d = Ddc(ctx, nchan=len(zooms), max_samples=n)
    chans = list(range(len(zooms)))
    for ch, z in enumerate(zooms):
        p = WfParams.for_zoom(z, 1.0e6 * ch, adc_clock=66.6666e6, ui_srate=30.0e6)
        d.set_wf(ch, p.i_offset, p.decim)
    stride = n + 1
    out = torch.zeros((len(zooms), stride, 2), dtype=torch.int16, device=dev)
    for _ in range(args.warmup):
        d.push_dev(adc.data_ptr(), n, chans, out.data_ptr(), stride)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        d.push_dev(adc.data_ptr(), n, chans, out.data_ptr(), stride)
    ms = ctx.timer_stop() / args.steps
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    print(json.dumps({
        "metric": "ADC Msamples/s ingested by the 14-channel waterfall DDC (NCO mix + CIC decimate)",
        "value": round(n * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "dtype": "int128/int32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2] DDC: %d ADC samples, 14 channels, zooms %s" % (n, zooms)},
        "gpu_ms_per_step": round(ms, 4),
        "x_realtime_at_66.6MSps": round(n / (ms * 1e-3) / 66.6666e6, 1),
    }), flush=True)
    d.close()
    ctx.close()


def bench_receivers(args):
    """BASELINE configs[3]: a batch of virtual receivers per GPU (weak scaling over ranks), each
    with a waterfall and an audio path, fed from one ADC block resident in HBM per step:
      waterfall: NCO mix + CIC decimate -> 8192-sample frame -> u8 row -> wf_pkt_t (ADPCM)
      audio:     NCO mix + CIC/CIC/CICF decimate -> rx_iq_t -> unpack -> CFastFIR -> S-meter +
                 CAgc (mono16) -> IMA ADPCM
        python bench.py --workload receivers [--receivers 128] [--log2n 22] [--gpus N via torch.distributed.run]"""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    from flydog_sdr_gps_amd import Adpcm, Context, Ddc, FastFir, Post, RxDdc, Waterfall, WfParams, post, snd, wf, wire
    from flydog_sdr_gps_amd.ddc import RX_DECIM, rx_phase_inc
    # the waterfall chain and the audio chain of a receiver are independent: one context (= one
    # stream) each, so the short latency-bound kernels of one hide under the DDC passes of the other
    two = os.environ.get("KIWIGPU_BENCH_ONE_STREAM") != "1"
    ctx = Context(local_rank, torch.cuda.current_stream(dev).cuda_stream)
    side = torch.cuda.Stream(device=dev) if two else None
    ctx_au = Context(local_rank, side.cuda_stream) if two else ctx
    NR, n = args.receivers, 1 << args.log2n
    assert n >= 512 * 8192, "--log2n >= 22: every step must complete a waterfall frame at zoom 10"
    adc_clock, ui_srate = 66.6666e6, 30.0e6
    rng = np.random.Generator(np.random.PCG64(0x5EED0004 + rank))
    t = np.arange(n, dtype=np.float64)
    x = rng.normal(0, 10.0, n)
    for f, a in ((0.0123, 3000.0), (0.071, 300.0), (0.2003, 30.0), (0.31, 3.0)):
        x += a * np.cos(2 * np.pi * f * t)
    adc = torch.from_numpy(np.clip(np.rint(x), -32768, 32767).astype(np.int16)).to(dev)
    chans = list(range(NR))

    d = Ddc(ctx, nchan=NR, max_samples=n)
    W = Waterfall(ctx, nchan=NR)
    W.set_tables()
    hz_per_start = ui_srate / (1024 << 14)
    params = []
    for ch in range(NR):
        z = 1 + ch % 10
        p = WfParams.for_zoom(z, (1.0e6 + 0.2e6 * (ch % 97)) / hz_per_start, adc_clock=adc_clock, ui_srate=ui_srate)
        params.append(p)
        d.set_wf(ch, p.i_offset, p.decim)
        W.set_channel(ch, p, interp=wf.WF_MAX, window_func=wf.WINF_HANNING, cic_comp=True)
    rx = RxDdc(ctx_au, nchan=NR, max_samples=n)
    nrec_max = n // RX_DECIM + 2
    fir = FastFir(ctx_au, nchan=NR, max_in=nrec_max)
    P = Post(ctx_au, nchan=NR)
    A = Adpcm(ctx_au, nchan=NR)
    fs = adc_clock / RX_DECIM
    for ch in range(NR):
        rx.set_freq(ch, rx_phase_inc(0.0123 * adc_clock - 1000.0 - 10.0 * ch, adc_clock))
        fir.setup(ch, 300.0, 2700.0, 0.0, fs)
        P.set_agc(ch, True, False, -100, 50, 6, 1000, fs)
        P.set_smeter(ch, fs); P.set_mode(ch, post.MODE_SSB); P.reset(ch)

    wf_stride = n + 1
    wf_iq = torch.zeros((NR, wf_stride, 2), dtype=torch.int16, device=dev)
    frames = torch.zeros((NR, 8192, 2), dtype=torch.int16, device=dev)
    rows = torch.zeros((NR, 1024), dtype=torch.uint8, device=dev)
    pkts = torch.zeros((NR, wire.WF_PKT_MAX), dtype=torch.uint8, device=dev)
    raw = torch.zeros((NR, nrec_max * 6), dtype=torch.uint8, device=dev)
    xin = torch.zeros((NR, nrec_max, 2), dtype=torch.float32, device=dev)
    firo = torch.zeros((NR, 1024, 2), dtype=torch.float32, device=dev)
    s16 = torch.zeros((NR, 512), dtype=torch.int16, device=dev)
    pay = torch.zeros((NR, 256), dtype=torch.uint8, device=dev)
    infos = [(int(params[ch].start), params[ch].zoom, 0, True) for ch in range(NR)]
    counts = {"frames": 0, "audio_blocks": 0}
    torch.cuda.synchronize(dev)                              # buffers exist before the side stream touches them

    def step():
        # audio chain first: its calls only enqueue on the side stream
        audio()
        nw = d.push_dev(adc.data_ptr(), n, chans, wf_iq.data_ptr(), wf_stride)
        assert int(nw.min()) >= 8192
        frames.copy_(wf_iq[:, :8192])                       # the frame each receiver's waterfall takes this step
        W.frames_dev(chans, frames.data_ptr(), rows.data_ptr())
        wire.wf_packets_dev(ctx, rows.data_ptr(), 1024, infos, pkts.data_ptr())
        counts["frames"] += NR

    def audio():
        nr = rx.push_dev(adc.data_ptr(), n, chans, raw.data_ptr(), nrec_max)
        nrec = int(nr.min())
        assert nrec == int(nr.max())
        snd.unpack_rows_dev(ctx_au, raw.data_ptr(), nrec_max, nrec, NR, xin.data_ptr(), nrec_max)
        nout = fir.process_dev(chans, xin.data_ptr(), nrec_max, nrec, firo.data_ptr(), 1024)
        if int(nout[0]) == 512:
            P.process_dev(chans, firo.data_ptr(), 1024, 512, s16.data_ptr(), 0, 0, 512)
            A.encode_dev(chans, s16.data_ptr(), 512, 512, pay.data_ptr(), 256)
            counts["audio_blocks"] += NR

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)                         # all streams of the device

    for _ in range(args.warmup):
        step()
    barrier()
    counts["frames"] = counts["audio_blocks"] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # sanity: the strongest carrier is in the band of every waterfall row and audio came out
    assert int(rows.max()) > 100 and counts["audio_blocks"] > 0 and int(pay.to(torch.int32).abs().sum()) > 0
    if rank == 0:
        step_s = elapsed / args.steps
        print(json.dumps({
            "metric": "receiver x ADC Msamples/s ingested (waterfall + audio chain per virtual receiver)",
            "value": round(n * NR * world / step_s / 1e6, 1), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int128/int64/f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: %d virtual receivers per GPU x %d GPU(s), %d-sample 16-bit ADC "
                                   "block @66.67 MS/s resident in HBM per step; per receiver a waterfall channel "
                                   "(zooms 1..10) and an SSB audio channel" % (NR, world, n),
                       "receivers_per_gpu": NR, "adc_samples_per_step": n,
                       "parallelism": "receivers sharded over ranks, no data-path collective"},
            "adc_ms_per_step": round(n / adc_clock * 1e3, 3),
            "x_realtime_all_receivers": round(n / adc_clock / step_s, 2),
            "waterfall_frames_per_s": round(counts["frames"] * world / elapsed, 1),
            "audio_blocks_per_s": round(counts["audio_blocks"] * world / elapsed, 1),
        }), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(workload, units):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/hbm_traffic.json, written from tools/prof.sh output: FETCH_SIZE x2 on gfx950
    + WRITE_SIZE, separate --pmc runs of this same command).  A bench run cannot read
    counters itself; None when no measurement exists for this configuration."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            tab = json.load(f)
        return tab[workload][str(units)]["bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=None, help="ddc / receivers: log2 of the ADC samples per step (24 / 22)")
    ap.add_argument("--receivers", type=int, default=128, help="receivers: virtual receivers per GPU")
    ap.add_argument("--workload", default="acq", choices=["acq", "waterfall", "ddc", "receivers"])
    ap.add_argument("--frames", type=int, default=512, help="waterfall: frames per channel per step")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=32, help="independent 4 ms blocks per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()
    if args.log2n is None:
        args.log2n = 22 if args.workload == "receivers" else 24
    if args.workload == "waterfall":
        return bench_waterfall(args)
    if args.workload == "ddc":
        return bench_ddc(args)
    if args.workload == "receivers":
        return bench_receivers(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # KIWIGPU_BENCH_FORCE_DIST=1 takes the process-group path with one rank as well (used to
    # exercise init / barrier / all_reduce / all_gather over RCCL on a single-GPU box)
    distributed = world > 1 or os.environ.get("KIWIGPU_BENCH_FORCE_DIST") == "1"
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    from flydog_sdr_gps_amd import Context, Searcher, prn, sats, synth
    from flydog_sdr_gps_amd import shard

    # run on torch's current stream so torch.cuda.Event / synchronize see the work
    stream = torch.cuda.current_stream(dev).cuda_stream
    ctx = Context(local_rank, stream)
    B = args.blocks
    s = Searcher(ctx, max_blocks=2 * B)      # two sets of blocks, used alternately
    svs = list(range(NSV))
    chips_list = []
    for sat in svs:
        _, t1, t2, _ = sats.SATS[sat]
        chips_list.append(prn.cacode(t1, t2))
        s.set_code(sat, chips_list[-1])

    # each rank gets its own seeded blocks ("receivers"), resident in HBM
    blocks = shard.block_ids(rank, world, B)
    iq_host = [synth.config1_iq16(seed=0x5EED0002 + b) for b in blocks]
    iq_dev = torch.from_numpy(np.stack(iq_host)).to(dev)      # [B][2*65536] int16, resident
    iq_ptr = int(iq_dev.data_ptr())

    parity = [0]

    def step():
        # Sample() then Correlate() of this step's blocks, in order on one stream (two sets
        # of blocks alternate so that consecutive steps never touch the same spectra); all
        # of it is inside the timed region.
        first = parity[0] * B
        parity[0] ^= 1
        s.sample_iq16_batch(iq_ptr, B, first_block=first)
        s.correlate_async(svs, nblocks=B, first_block=first)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enqueued = time.perf_counter() - t0      # host side only: diagnostic, not the metric
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # dominant kernel alone: Correlate() launches back to back, HIP events on its stream
    kreps = max(20, min(args.steps, 200))
    for _ in range(3):
        s.correlate_async(svs, nblocks=B)
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for _ in range(kreps):
        s.correlate_async(svs, nblocks=B)
    kernel_ms = ctx.timer_stop() / kreps

    res, _ = s.fetch(want_cells=False)
    found = sorted(int(sv) + 1 for sv in svs if res[0, sv]["snr"] >= 16)
    if distributed:
        gathered = shard.gather_results(res, dev)       # RCCL all_gather of the tiny result arrays
        assert gathered.shape[0] == world * B
    if rank == 0:
        expect = sorted(p for p, *_ in synth.CONFIG1_PRESENT)
        assert found == expect, "acquisition result wrong: %s != %s" % (found, expect)

    if rank == 0:
        total_samples = float(B) * NSAMPLES * args.steps * world
        cells = B * NSV * NDOP
        achieved = cells * BYTES_PER_CELL / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "IQ Msamples/s ingested (GPS acq: Sample + 32 SV x 41 Doppler Correlate)",
            "value": round(total_samples / elapsed / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: 32 GPS L1 C/A SVs x 41 Doppler bins, 4 ms "
                            "coherent FFT correlate, synthetic int16 IQ @16.368 MS/s resident in HBM",
                "blocks_per_step_per_gpu": B,
                "samples_per_block": NSAMPLES,
                "cells_per_step_per_gpu": cells,
                "parallelism": "replicated codes, sample blocks sharded over %d GPU(s), "
                               "no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "acq_correlate_kernel<1>",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic("acq", B),
                "kernel_ms": round(kernel_ms, 5),
                "algorithmic_bytes_per_launch": cells * BYTES_PER_CELL,
                "note": "algorithmic bytes = 262160 B per (SV,Doppler) cell (SURVEY 8d); the "
                        "spectra are shared between cells and served from L2/Infinity Cache, so "
                        "this can exceed the HBM peak; the kernel is fp32-VALU/LDS bound",
            },
            # What actually bounds the kernel: fp32 vector arithmetic.  Nominal
            # FFT arithmetic of one cell = four 4096-point sub-transforms (5 N log2 N) + the
            # 16384 conj-multiplies (6) + three twiddled accumulations of 4096 points (8 + 2);
            # peak = 157.3 TFLOP/s fp32 vector (MI355X_MICROARCH.md).  Supplementary to `roofline`.
            "valu": {
                "flops_per_cell": FLOPS_PER_CELL,
                "achieved_tflops": round(cells * FLOPS_PER_CELL / (kernel_ms * 1e-3) / 1e12, 2),
                "peak_tflops": 157.3,
                "frac": round(cells * FLOPS_PER_CELL / (kernel_ms * 1e-3) / 1e12 / 157.3, 4),
            },
            "found_prns": found,
            "host_enqueue_ms_per_step": round(t_enqueued / args.steps * 1e3, 5),
        }
        if not args.no_cpu and world == 1:                  # the CPU leg runs at N = 1 only
            out["cpu_baseline"] = cpu_baseline(iq_host[0], chips_list)
            out["speedup_vs_cpu_all_cores"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    s.close()
    ctx.close()


if __name__ == "__main__":
    main()
