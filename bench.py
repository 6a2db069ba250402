#!/usr/bin/env python3
"""bench.py -- the headline benchmark of the MI355X DSP hot path (BASELINE.json's metric:
"IQ Msamples/s ingested (waterfall + GPS acq)").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload W] [--no-cpu]

With --gpus N > 1 (and no WORLD_SIZE in the environment) this process only LAUNCHES: it starts N
rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one per GPU), relays
rank 0's JSON line and exits with the first non-zero return code.  It never touches the GPU or
imports torch itself.  Started by `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`, it is a rank; --gpus must then equal WORLD_SIZE.

Workloads (all inputs synthetic, seeded, resident in HBM before the timed region):
  acq        BASELINE configs[1]: 32 GPS L1 C/A SVs x 41 Doppler bins, 4 ms coherent FFT correlate
             on int16 IQ; a step = Sample() front end + Correlate() + best-bin selection for
             --blocks independent 4 ms blocks (default 32).           <- `value` of the line
  wf14       BASELINE configs[2], frame half: 14 waterfall channels, window + 8192-point FFT +
             power + pixel reduce + dB + u8 on DDC output buffers; the input is sized past the
             256 MiB Infinity Cache so that the HBM figure is an HBM figure.
  ddc14      BASELINE configs[2], DDC half: NCO mix + 5-stage pruned CIC for the 14 channels on a
             16-bit ADC stream.
  all        (default) the three above; ONE JSON line whose top level is `acq` (metric, value,
             roofline, cpu_baseline) and whose "workloads" object holds every workload's own
             value, step times, roofline and cpu_baseline.
  acq59      the 4 ms shape with the reference's whole SV list (36 C/A + QZSS, 23 E1B): profiles the
             four-accumulator (16368-sample window) correlator beside the C/A one.
  acq10ms    BASELINE configs[4]: joint L1 C/A + QZSS + Galileo E1B, 10 ms coherent (65536-point
             transforms), 256 Doppler bins, all 59 SVs.
  receivers  BASELINE configs[3]: --receivers virtual receivers per GPU (waterfall + audio chain).
  waterfall / ddc   the old names of wf14 / ddc14 as single lines.

Multi-GPU (weak scaling): every rank works on its own resident units (sample blocks / frames /
receivers); no data-path collective; the tiny acquisition results are all-gathered
over RCCL after the timed region.  `value` = units of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NSAMPLES = 65536
FFT_LEN = 16384
NSV = 32
NDOP = 41
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (MI355X_MICROARCH.md)
INT_PEAK_TOPS = 78.6         # 256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz: one 32-bit integer op per lane per clock
ZOOMS14 = [0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14]      # BASELINE configs[2]


# ------------------------------------------------------------------------------------------------
# Launcher: `python bench.py --gpus N` starts N ranks.  Runs before anything imports torch.
# ------------------------------------------------------------------------------------------------
def launch_ranks(n, argv):
    import socket
    with socket.socket() as s:                     # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else None, text=True))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:                                    # a rank failed: stop the others (exact PIDs)
            rc = bad[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    out0 = procs[0].stdout.read() or ""            # one JSON line: far below the pipe buffer
    sys.stdout.write(out0)
    sys.stdout.flush()
    if rc != 0:
        sys.stderr.write("bench.py: a rank exited with status %d\n" % rc)
    return rc


class Dist:
    """The process-group side of a rank (RCCL on GPUs; gloo for the CPU launcher test)."""

    def __init__(self, backend="nccl"):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # KIWIGPU_BENCH_FORCE_DIST=1 takes the process-group path with one rank as well
        self.on = self.world > 1 or os.environ.get("KIWIGPU_BENCH_FORCE_DIST") == "1"
        self.backend = backend
        self.dev = None
        if backend == "nccl":
            import torch
            ndev = torch.cuda.device_count()       # does not initialise the GPU
            if self.local_rank >= ndev:
                sys.stderr.write("bench.py: rank %d needs GPU %d but this node has %d\n"
                                 % (self.rank, self.local_rank, ndev))
                sys.exit(3)
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        if self.on:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group("gloo")

    def barrier(self):
        if self.on:
            import torch.distributed as dist
            dist.barrier()
        if self.dev is not None:
            import torch
            torch.cuda.synchronize(self.dev)       # every stream of the device

    def max_over_ranks(self, seconds):
        if not self.on:
            return seconds
        import torch
        import torch.distributed as dist
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev if self.dev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.on:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


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


def cpu_threads(one_unit, budget_s):
    """Every usable host thread runs whole single-threaded units back to back for about budget_s
    seconds (independent units, like the GPU's batch; ctypes drops the GIL inside the oracle).
    -> (units done, seconds, cores, seconds of one unit on one thread)"""
    import threading
    cores = usable_cores()
    t0 = time.perf_counter()
    one_unit()                                                    # warm-up + single-thread figure
    t1 = time.perf_counter() - t0
    done = [0] * cores
    deadline = time.perf_counter() + budget_s

    def worker(k):
        while time.perf_counter() < deadline:
            one_unit()
            done[k] += 1

    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    return sum(done), time.perf_counter() - t0, cores, t1


LIVE_TRAFFIC = {}            # workload -> (bytes per launch, source), filled by live_traffic_passes()
TRAFFIC_KERNELS = {"acq": ("acq_correlate_kernel<4, 1,",), "acq59": ("acq_correlate_kernel<4, 1,", "acq_correlate_kernel<4, 4,"),
                   "acq10ms": ("acq_correlate_kernel<16, 1,", "acq_correlate_kernel<16, 4,"), "wf14": ("wf_frame_kernel",)}


def under_profiler():
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def live_traffic_passes(args):
    """HBM bytes per launch of each workload's dominant kernel(s), measured NOW: before this process
    touches the GPU it runs this same file as a child under `rocprofv3 --pmc` -- one pass for
    FETCH_SIZE, one for WRITE_SIZE (separate passes, nothing but the counter: MI355X_MICROARCH.md's
    HBM section) -- on the same per-launch configuration, a few steps each.  bytes = (2 x FETCH_SIZE
    + WRITE_SIZE) KB: gfx950 tallies its 128-byte read requests as 64.  Any failure (no rocprofv3,
    already under a profiler, a pass timing out) leaves LIVE_TRAFFIC without the entry and the line
    falls back to the committed figure, saying so."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None or under_profiler():
        return
    wls = ["acq", "wf14"] if args.workload == "all" else [args.workload]
    for wl in wls:
        keys = TRAFFIC_KERNELS.get(wl)
        if keys is None:
            continue
        child = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", "4", "--warmup", "1",
                 "--no-cpu", "--no-live-traffic", "--frames", str(args.frames)]
        if args.blocks is not None:
            child += ["--blocks", str(args.blocks)]
        mean = {}
        tmp = tempfile.mkdtemp(prefix="kiwigpu_pmc_")
        try:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, counter)
                cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child
                env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
                p = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                     start_new_session=True)     # its own process group: a stuck pass is ended whole
                try:
                    rc = p.wait(timeout=240)
                except subprocess.TimeoutExpired:
                    os.killpg(p.pid, 9)
                    p.wait()
                    raise RuntimeError("pass timed out")
                if rc != 0:
                    raise RuntimeError("pass failed, status %d" % rc)
                per = {k: [] for k in keys}
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    with open(f) as fh:
                        for row in csv.DictReader(fh):
                            if row["Counter_Name"] != counter:
                                continue
                            for k in keys:
                                if k in row["Kernel_Name"]:
                                    per[k].append(float(row["Counter_Value"]))
                if any(not v for v in per.values()):
                    raise RuntimeError("kernel not seen")
                mean[counter] = sum(sum(v) / len(v) for v in per.values())       # KB per launch, all parts
            LIVE_TRAFFIC[wl] = (int((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024),
                                "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run "
                                "(2 x %.1f KB + %.1f KB per launch)" % (mean["FETCH_SIZE"], mean["WRITE_SIZE"]))
        except Exception as e:                                    # noqa: BLE001 -- fall back, and say so
            sys.stderr.write("bench.py: live PMC pass for %s not available (%s)\n" % (wl, e))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)


def measured_traffic(workload, units):
    """HBM bytes per launch of the workload's dominant kernel: the live PMC child passes of this
    run (live_traffic_passes) when they ran, else the committed passes (profiles/hbm_traffic.json,
    written from tools/prof.sh output the same way).  -> (bytes or None, where the number comes from)"""
    if workload in LIVE_TRAFFIC:
        return LIVE_TRAFFIC[workload]
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            tab = json.load(f)
        e = tab[workload][str(units)]
        return e["bytes_per_launch"], "committed profile: %s" % e.get("source", tab.get("_source", "profiles/hbm_traffic.json"))
    except (OSError, KeyError, ValueError):
        return None, "no committed PMC pass for this configuration"


PREROLL_S = 0.12


def preroll(step, warmup):
    """The W untimed warm-up steps, then more untimed ones until about PREROLL_S seconds of work have run
    (at least 64 steps): an MI355X that has been idle reaches its steady clock only after tens of
    milliseconds of load (measured on the correlator: 0.96 ms per launch at the start, 0.84 after 20
    launches, 0.80 after 60), and the first time two streams of a process depend on each other (the DDC's
    side stream) the HIP runtime stalls the GPU side for 30-50 ms, once per process, some thirty calls
    in.  A short timed region would otherwise hold both.  Never part of the timed region."""
    import torch
    for _ in range(max(1, warmup)):
        step()
    torch.cuda.synchronize()                         # one-time costs (code load, lazy allocations) are behind us
    probe = max(4, min(warmup, 16))
    t0 = time.perf_counter()
    for _ in range(probe):
        step()
    torch.cuda.synchronize()
    est = max((time.perf_counter() - t0) / probe, 1e-5)
    for _ in range(min(4000, max(64, int(PREROLL_S / est)))):
        step()


def timed_steps(dist, step, steps, warmup):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides, max
    over ranks; then the same K steps once more with a device event between steps for the spread
    (not part of the headline time).  -> (seconds, host enqueue seconds, spread dict)"""
    import torch
    preroll(step, warmup)
    dist.barrier()                                   # barrier, then synchronize
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()                         # this rank's K steps are done ...
    local = time.perf_counter() - t0
    dist.barrier()                                   # ... every rank's are: the slowest rank's time counts
    elapsed = dist.max_over_ranks(local)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    dts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    spread = {"min": round(dts[0], 5), "median": round(dts[len(dts) // 2], 5), "max": round(dts[-1], 5),
              "how": "device events between the steps of a second, untimed pass of K steps"}
    return elapsed, t_enq, spread


# ------------------------------------------------------------------------------------------------
# GPS acquisition: configs[1] (4 ms) and configs[4] (10 ms)
# ------------------------------------------------------------------------------------------------
def acq_flops_per_cell(P, limit_quarters):
    """Nominal FFT arithmetic of one (SV, Doppler) cell: P 4096-point sub-transforms at 5 N log2 N,
    the N conj-multiplies (6 flops), P - 1 twiddled accumulations of 4096 points per output quarter
    (8 + 2 flops)."""
    return P * 5 * 4096 * 12 + 6 * P * 4096 + (P - 1) * 4096 * 10 * limit_quarters


def run_acq(args, dist, ten_ms=False, all_svs=False):
    import numpy as np
    import torch
    from flydog_sdr_gps_amd import Context, Searcher, acq, prn, sats, shard, synth
    dev = dist.dev
    ctx = Context(dist.local_rank, torch.cuda.current_stream(dev).cuda_stream)
    if ten_ms:
        B = args.blocks or 2
        nsamples, fft_len, dop_lo, dop_hi = acq.NSAMPLES_10MS, acq.FFT_LEN_10MS, -128, 127
        codes = synth.all_sv_codes()
        svs = list(range(len(codes)))
        make_block = lambda b: synth.config4_iq16(seed=0x5EED0005 + b, codes=codes)     # noqa: E731
    else:
        B = args.blocks or 32
        nsamples, fft_len, dop_lo, dop_hi = NSAMPLES, FFT_LEN, -20, 20
        # all_svs: the reference's whole Sats[] list (36 C/A + QZSS rows and the 23 E1B rows, whose
        # 16368-sample window takes the four-accumulator kernel) on the configs[1] blocks
        codes = synth.all_sv_codes() if all_svs else \
            [(prn.cacode(sats.SATS[s][1], sats.SATS[s][2]), False) for s in range(NSV)]
        svs = list(range(len(codes)))
        make_block = lambda b: synth.config1_iq16(seed=0x5EED0002 + b)                  # noqa: E731
    ndop, P = dop_hi - dop_lo + 1, fft_len // 4096
    s = Searcher(ctx, dop_lo=dop_lo, dop_hi=dop_hi, max_blocks=2 * B, nsamples=nsamples, fft_len=fft_len)
    for sat, (chips, boc) in enumerate(codes):
        s.set_code(sat, chips, boc=boc)
    # each rank gets its own seeded blocks ("receivers"), resident in HBM
    blocks = shard.block_ids(dist.rank, dist.world, B)
    iq_host = [make_block(b) for b in blocks]
    iq_dev = torch.from_numpy(np.stack(iq_host)).to(dev)          # [B][2*nsamples] int16, resident
    iq_ptr = int(iq_dev.data_ptr())
    parity = [0]

    def step():
        # Sample() then Correlate() of this step's blocks, in order on one stream (two sets of
        # blocks alternate so that consecutive steps never touch the same spectra)
        first = parity[0] * B
        parity[0] ^= 1
        s.sample_iq16_batch(iq_ptr, B, first_block=first)
        s.correlate_async(svs, nblocks=B, first_block=first)

    elapsed, t_enq, spread = timed_steps(dist, step, args.steps, args.warmup)

    # dominant kernel alone: Correlate() launches back to back, HIP events on its own stream
    kreps = max(10, min(args.steps, 200))
    for _ in range(3):
        s.correlate_async(svs, nblocks=B)
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for _ in range(kreps):
        s.correlate_async(svs, nblocks=B)
    kernel_ms = ctx.timer_stop() / kreps

    res, _ = s.fetch(want_cells=False)
    # (E1B rows: 41 x 16368 trials per SV put the noise maximum close to the reference's MIN_SIG = 16)
    min_sig = synth.MIN_SIG_10MS if ten_ms else (24.0 if all_svs else acq.MIN_SIG)
    found = sorted(int(sv) for sv in svs if res[0, sv]["snr"] >= min_sig)
    if dist.on and dist.backend == "nccl":
        gathered = shard.gather_results(res, dev)       # RCCL all_gather of the tiny result arrays
        assert gathered.shape[0] == dist.world * B
    expect = sorted(p[0] for p in synth.CONFIG4_PRESENT) if ten_ms else \
        sorted(p - 1 for p, *_ in synth.CONFIG1_PRESENT)
    # every injected SV must be found on every rank; an absent SV above the threshold is a noise
    # false alarm of that rank's own seeded block (41 x 4092 trials at MIN_SIG = 16: about 2 % per SV)
    assert set(expect) <= set(found), "acquisition result wrong: %s lacks %s" % (found, sorted(set(expect) - set(found)))
    if dist.rank == 0 and not all_svs:
        assert found == expect, "acquisition result wrong: %s != %s" % (found, expect)

    n1 = sum(1 for _, boc in codes if not boc)
    n4 = len(codes) - n1
    flops_launch = B * ndop * (n1 * acq_flops_per_cell(P, 1) + n4 * acq_flops_per_cell(P, 4))
    cells = B * len(svs) * ndop
    bytes_cell = 2 * fft_len * 8 + 16                  # both spectra once + the result
    traffic, source = measured_traffic("acq10ms" if ten_ms else "acq", B)
    tfl = flops_launch / (kernel_ms * 1e-3) / 1e12
    out = {
        "metric": "IQ Msamples/s ingested (GPS acq: Sample + %d SV x %d Doppler Correlate)" % (len(svs), ndop),
        "value": round(float(B) * nsamples * args.steps * dist.world / elapsed / 1e6, 3),
        "unit": "Msamples/s",
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "step_ms_spread": spread,
        "dtype": "f32",
        "config": {
            "workload": ("BASELINE configs[4]: GPS L1 C/A + QZSS + Galileo E1B joint acquisition, 10 ms coherent "
                         "(163680 samples @16.368 MS/s -> 65536-point transforms), 256 Doppler bins of 62.44 Hz, "
                         "59 SVs (36 C/A, 23 E1B), synthetic int16 IQ resident in HBM") if ten_ms else
                        (("the reference's whole SV list on BASELINE configs[1]'s blocks: 36 C/A + QZSS and 23 Galileo E1B SVs "
                          "x 41 Doppler bins, 4 ms coherent, synthetic int16 IQ @16.368 MS/s resident in HBM") if all_svs else
                         ("BASELINE configs[1]: 32 GPS L1 C/A SVs x 41 Doppler bins, 4 ms coherent FFT correlate, "
                          "synthetic int16 IQ @16.368 MS/s resident in HBM")),
            "blocks_per_step_per_gpu": B, "samples_per_block": nsamples, "cells_per_step_per_gpu": cells,
            "parallelism": "replicated codes, sample blocks sharded over %d GPU(s), no data-path collective" % dist.world,
        },
        # What bounds the correlator: fp32 vector arithmetic -- every operand is
        # served from L2, the kernel cannot be HBM-bound.  achieved = nominal FFT flops / HIP-event time.
        "roofline": {
            "bound": "valu", "kernel": "acq_correlate_kernel<%d,1>%s" % (P, " + <%d,4>" % P if n4 else ""),
            "achieved": round(tfl, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tfl / VALU_PEAK_TFLOPS, 4),
            "traffic": traffic, "traffic_source": source,
            "kernel_ms": round(kernel_ms, 5), "flops_per_launch": flops_launch,
        },
        # Secondary: SURVEY 8(d)'s per-cell byte model and, where a PMC pass exists, measured HBM bytes.
        "hbm": {
            "algorithmic_bytes_per_launch": cells * bytes_cell,
            "algorithmic_GBps": round(cells * bytes_cell / (kernel_ms * 1e-3) / 1e9, 1),
            "measured_GBps": None if traffic is None else round(traffic / (kernel_ms * 1e-3) / 1e9, 1),
            "peak": HBM_PEAK_GBS,
            "frac_measured": None if traffic is None else round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "note": "the spectra are shared between cells and served from L2 / Infinity Cache: the algorithmic "
                    "figure is not HBM traffic and may exceed the HBM peak",
        },
        "found_svs": found,
        "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 5),
    }
    if not args.no_cpu and dist.world == 1 and dist.rank == 0:
        out["cpu_baseline"] = cpu_acq(iq_host[0], codes, nsamples, fft_len, dop_lo, dop_hi, args.cpu_seconds)
        out["speedup_vs_cpu_all_cores"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    s.close()
    ctx.close()
    return out


def cpu_acq(iq, codes, nsamples, fft_len, dop_lo, dop_hi, budget_s):
    """The CPU oracle (kind "port": scalar fp32 radix-4 FFT in place of FFTW3f, which is absent from
    this image) on the same workload: Sample() + Correlate() for every SV and bin of one block."""
    import numpy as np
    from oracle import kiwi_oracle as ko
    ko.lib()
    spectra = np.stack([ko.code_fft(c, boc=b, prec=0, fft_len=fft_len) for c, b in codes])
    limits = [ko.E1B_LIMIT if b else ko.L1_LIMIT for _, b in codes]
    ndop = dop_hi - dop_lo + 1
    if fft_len == FFT_LEN:
        def unit():
            data = ko.sample_iq16(iq, prec=0)
            ko.correlate_many(spectra, data, limits, prec=0, nthreads=1, want_cells=False)
        samples_per_unit, what = nsamples, "the full configs[1] block (Sample + 32 SV x 41 bins)"
    else:
        # a bounded slice of the 59 x 256 cells: Sample() + 2 SVs (one C/A, one E1B) x 256 bins
        sub = [0, len(codes) - 1]
        frac = len(sub) / len(codes)

        def unit():
            data = ko.sample_iq16(iq, prec=0, nsamples=nsamples, fft_len=fft_len)
            ko.correlate_many(spectra[sub], data, [limits[i] for i in sub], dop_lo=dop_lo, dop_hi=dop_hi,
                              prec=0, nthreads=1, want_cells=False)
        samples_per_unit = nsamples * frac
        what = "Sample + 2 of the 59 SVs x %d bins of one configs[4] block (scaled by 2/59)" % ndop
    reps, el, cores, t1 = cpu_threads(unit, budget_s)
    return {
        "value": round(reps * samples_per_unit / el / 1e6, 5), "unit": "Msamples/s", "cores": cores,
        "machine_cpus": os.cpu_count(), "kind": "port",
        "sample": "%d x %s, oracle fp32 FFT, %d threads each running whole units, %.1f s" % (reps, what, cores, el),
        "single_thread_value": round(samples_per_unit / t1 / 1e6, 5),
        "port_vs_reference": "FFTW3f is unavailable here; a tuned FFTW build is estimated 2-4x faster than this "
                             "scalar port, so divide the speed-up by up to 4 for a fair reference build",
    }


# ------------------------------------------------------------------------------------------------
# Waterfall frames (configs[2], frame half)
# ------------------------------------------------------------------------------------------------
def run_wf14(args, dist):
    import numpy as np
    import torch
    from flydog_sdr_gps_amd import Context, Waterfall, WfParams, synth, wf
    dev = dist.dev
    ctx = Context(dist.local_rank, torch.cuda.current_stream(dev).cuda_stream)
    zooms = ZOOMS14
    w = Waterfall(ctx, nchan=len(zooms))
    tables = (wf.window_functions(), wf.cic_comp_table())
    w.set_tables(*tables)
    params = []
    for ch, z in enumerate(zooms):
        p = WfParams.for_zoom(z, 1.0e6 * ch)
        params.append(p)
        w.set_channel(ch, p, interp=wf.WF_CMA, cic_comp=True)
    F = args.frames                                   # frames per channel per step
    nfr = F * len(zooms)
    base = np.stack([synth.wf_iq_frame(seed=i + 1000 * dist.rank) for i in range(32)])
    # distinct frames all the way (a rolled copy per repetition): 8192 x 4 B each, the whole input
    # must not fit the 256 MiB Infinity Cache or FETCH_SIZE would count MALL hits
    iq = torch.from_numpy(base).to(dev)
    reps = (nfr + 31) // 32
    iq = torch.cat([torch.roll(iq, shifts=7 * r + 1, dims=1) for r in range(reps)])[:nfr].contiguous()
    out = torch.empty((nfr, 1024), dtype=torch.uint8, device=dev)
    chan_of = np.arange(nfr, dtype=np.int32) % len(zooms)

    def step():
        w.frames_dev(chan_of, iq.data_ptr(), out.data_ptr())

    steps = max(5, args.steps)                      # (a quarter of them left the post-synchronize clock ramp, ~1 ms, in the average)
    elapsed, t_enq, spread = timed_steps(dist, step, steps, max(2, args.warmup))
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for _ in range(steps):
        step()
    kernel_ms = ctx.timer_stop() / steps
    assert int(out.max()) > 100
    bytes_frame = 8192 * 4 + 1024          # int16 IQ in + u8 row out; window/maps are L2-resident
    achieved = nfr * bytes_frame / (kernel_ms * 1e-3) / 1e9
    traffic, source = measured_traffic("wf14", nfr)
    res = {
        "metric": "waterfall IQ Msamples/s (window + 8192-pt FFT + power + pixel reduce + dB + u8)",
        "value": round(nfr * 8192.0 * steps * dist.world / elapsed / 1e6, 1), "unit": "Msamples/s",
        "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 5), "step_ms_spread": spread, "dtype": "f32",
        "config": {"workload": "BASELINE configs[2] frames: 14 channels (zooms %s) x %d frames per step, "
                               "%.0f MiB of int16 IQ per step (past the 256 MiB Infinity Cache)"
                               % (zooms, F, nfr * 32768 / 2 ** 20), "frames_per_step": nfr},
        "roofline": {"bound": "hbm", "kernel": "wf_frame_kernel", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": source, "kernel_ms": round(kernel_ms, 5),
                     "algorithmic_bytes_per_launch": nfr * bytes_frame},
        # the other roof, for scale: nominal flops of the 8192-point transform (5 N log2 N) per frame against
        # the fp32 vector peak; the kernel issues 1239 vector instructions per wave and frame, 600 of them
        # the two 4096-point transforms, so its vector floor (0.25 ms) is above its HBM floor
        "valu": {"nominal_flops_per_launch": nfr * 5 * 8192 * 13,
                 "achieved_TFLOPs": round(nfr * 5 * 8192 * 13 / (kernel_ms * 1e-3) / 1e12, 2), "peak": VALU_PEAK_TFLOPS,
                 "frac": round(nfr * 5 * 8192 * 13 / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4)},
    }
    if not args.no_cpu and dist.world == 1 and dist.rank == 0:
        from oracle import kiwi_oracle as ko
        ko.lib()
        maps = [wf.build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, False) for p in params]
        scales = [np.full(1024, p.fft_scale, np.float32) for p in params]
        k = [0]

        def unit():                                   # one frame of each of the 14 channels
            for ch, p in enumerate(params):
                f = base[(k[0] + ch) % 32]
                samps = ko.wf_window_iq(f, tables[0][wf.WINF_HANNING])
                ko.wf_compute_frame(samps, p.zoom, wf.WINF_HANNING, wf.WF_CMA, True, False, p.fft_used,
                                    p.plot_width, p.plot_width_clamped, maps[ch][0], maps[ch][1], scales[ch],
                                    (scales[ch] / np.float32(2)).astype(np.float32), p.fft_offset, tables[1], prec=0)
            k[0] += 1
        reps_done, el, cores, t1 = cpu_threads(unit, args.cpu_seconds)
        res["cpu_baseline"] = {
            "value": round(reps_done * 14 * 8192 / el / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d x (one frame of each of the 14 channels: sample_wf window + compute_frame), oracle fp32 "
                      "FFT, %d threads, %.1f s" % (reps_done, cores, el),
            "single_thread_value": round(14 * 8192 / t1 / 1e6, 4)}
        res["speedup_vs_cpu_all_cores"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    w.close()
    ctx.close()
    return res


# ------------------------------------------------------------------------------------------------
# Waterfall DDC (configs[2], DDC half)
# ------------------------------------------------------------------------------------------------
def adc_block(n, seed):
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n, dtype=np.float64)
    x = rng.normal(0, 10.0, n)
    for f, a in ((0.0123, 3000.0), (0.071, 300.0), (0.2003, 30.0), (0.31, 3.0)):
        x += a * np.cos(2 * np.pi * f * t)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def run_ddc14(args, dist):
    import numpy as np
    import torch
    from flydog_sdr_gps_amd import Context, Ddc, WfParams
    dev = dist.dev
    ctx = Context(dist.local_rank, torch.cuda.current_stream(dev).cuda_stream)
    zooms = ZOOMS14
    n = 1 << args.log2n
    adc_host = adc_block(n, 0x5EED0003)               # the same stream on every rank (configs[2]/[3])
    adc = torch.from_numpy(adc_host).to(dev)
    d = Ddc(ctx, nchan=len(zooms), max_samples=n)
    chans = list(range(len(zooms)))
    prm = []
    for ch, z in enumerate(zooms):
        p = WfParams.for_zoom(z, 1.0e6 * ch, adc_clock=66.6666e6, ui_srate=30.0e6)
        prm.append(p)
        d.set_wf(ch, p.i_offset, p.decim)
    stride = n + 1
    out = torch.zeros((len(zooms), stride, 2), dtype=torch.int16, device=dev)

    def step():
        d.push_dev(adc.data_ptr(), n, chans, out.data_ptr(), stride)

    steps = max(5, args.steps)                      # (a quarter of them left the post-synchronize clock ramp, ~1 ms, in the average)
    elapsed, t_enq, spread = timed_steps(dist, step, steps, max(2, args.warmup))
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for _ in range(steps):
        step()
    gpu_ms = ctx.timer_stop() / steps
    # Integer work per ADC sample and channel, counted in 32-bit operations on the algorithm (not on
    # the kernels): NCO phase add (48 bit: 2) + 2 table reads + 2 multiplies + 2 roundings (4) = 10;
    # five integrators on I and Q, the first four 89 bits wide (3 words) and the fifth 28 (1): 2 x 13 = 26;
    # the combs and the output rounding run at 1/R and are left out.
    ops_sample_chan = 36
    tops = n * len(zooms) * ops_sample_chan / (gpu_ms * 1e-3) / 1e12
    out_bytes = sum((n // p.decim) * 4 for p in prm)
    res = {
        "metric": "ADC Msamples/s ingested by the 14-channel waterfall DDC (NCO mix + pruned 5-stage CIC)",
        "value": round(float(n) * steps * dist.world / elapsed / 1e6, 1), "unit": "Msamples/s",
        "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 5), "step_ms_spread": spread,
        "dtype": "int128/int64/int32",
        "config": {"workload": "BASELINE configs[2] DDC: %d ADC samples per step, 14 channels, zooms %s" % (n, zooms),
                   "adc_samples_per_step": n},
        "x_realtime_at_66.6MSps": round(n / (gpu_ms * 1e-3) / 66.6666e6, 1),
        # integer-VALU bound by arithmetic intensity (SURVEY 8d's caveat): 2 bytes in per ADC sample for
        # 14 x 36 integer operations
        "roofline": {"bound": "valu", "kernel": "ddc_wf passes A + scan + B + comb (whole step)",
                     "achieved": round(tops, 3), "peak": INT_PEAK_TOPS, "unit": "Tiop/s",
                     "frac": round(tops / INT_PEAK_TOPS, 4), "traffic": None,
                     "traffic_source": "not profiled: HBM traffic is the 2 B/sample ADC block + the decimated outputs",
                     "kernel_ms": round(gpu_ms, 5), "int_ops_per_sample_per_channel": ops_sample_chan},
        "hbm": {"algorithmic_bytes_per_step": 2 * n + out_bytes,
                "algorithmic_GBps": round((2 * n + out_bytes) / (gpu_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS},
    }
    if not args.no_cpu and dist.world == 1 and dist.rank == 0:
        from oracle import kiwi_oracle as ko
        ko.lib()
        m = 1 << 18                                      # a bounded piece of the same stream
        piece = adc_host[:m]
        log2r = [int(np.log2(p.decim)) for p in prm]

        def unit():                                      # all 14 channels over the piece
            for ch, p in enumerate(prm):
                ko.ddc_wf(piece, p.i_offset, log2r[ch])
        reps_done, el, cores, t1 = cpu_threads(unit, args.cpu_seconds)
        res["cpu_baseline"] = {
            "value": round(reps_done * m / el / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d x (2^18 ADC samples through the 14 channels), the oracle's sequential Verilog-structured "
                      "model (the reference has no CPU DDC: it is FPGA fabric), %d threads, %.1f s" % (reps_done, cores, el),
            "single_thread_value": round(m / t1 / 1e6, 4)}
        res["speedup_vs_cpu_all_cores"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    d.close()
    ctx.close()
    return res


# ------------------------------------------------------------------------------------------------
# configs[3]: virtual receivers
# ------------------------------------------------------------------------------------------------
class ReceiverBank:
    """NR virtual receivers on one GPU, each with a waterfall and an audio path, fed from one ADC
    block resident in HBM per step (receiver k of the whole job = first_rx + local index):
      waterfall: NCO mix + CIC decimate -> 8192-sample frame -> u8 row -> wf_pkt_t (ADPCM)
      audio:     NCO mix + CIC/CIC/CICF decimate -> rx_iq_t -> unpack -> CFastFIR -> S-meter +
                 CAgc (mono16) -> IMA ADPCM
    tests/test_receivers_gpu.py steps the same object and checks every stage of every receiver
    against the oracle."""
    ADC_CLOCK, UI_SRATE = 66.6666e6, 30.0e6

    def __init__(self, local_rank, dev, NR, n, first_rx, two_streams=True):
        import torch
        from flydog_sdr_gps_amd import Adpcm, Context, Ddc, FastFir, Post, RxDdc, Waterfall, WfParams, post, wf, wire
        from flydog_sdr_gps_amd.ddc import RX_DECIM, rx_phase_inc
        assert n >= 512 * 8192, "--log2n >= 22: every step must complete a waterfall frame at zoom 10"
        self.dev, self.NR, self.n, self.first_rx = dev, NR, n, first_rx
        # the waterfall chain and the audio chain of a receiver are independent: one context (= one
        # stream) each, so the short latency-bound kernels of one hide under the DDC passes of the other
        self.ctx = ctx = Context(local_rank, torch.cuda.current_stream(dev).cuda_stream)
        self.side = torch.cuda.Stream(device=dev) if two_streams else None
        self.ctx_au = ctx_au = Context(local_rank, self.side.cuda_stream) if two_streams else ctx
        # the tails of both chains are sequential per-channel recurrences (ADPCM of the waterfall row; CAgc and
        # ADPCM of the audio block: one lane per channel, 100-140 us each whatever the load): on streams of
        # their own they no longer stand between a chain's DDC of this step and of the next one
        self.tails = two_streams and os.environ.get("KIWIGPU_BENCH_TAIL_STREAMS") != "0"
        self.s_pk = torch.cuda.Stream(device=dev) if self.tails else None
        self.s_tail = torch.cuda.Stream(device=dev) if self.tails else None
        self.ctx_pk = Context(local_rank, self.s_pk.cuda_stream) if self.tails else ctx
        self.ctx_tail = Context(local_rank, self.s_tail.cuda_stream) if self.tails else ctx_au
        self.main = torch.cuda.current_stream(dev)
        self.ev_frames, self.ev_pk, self.ev_fir, self.ev_tail = (torch.cuda.Event() for _ in range(4))
        self.pk_pending = self.tail_pending = False
        self.adc_host = adc_block(n, 0x5EED0004)                      # every GPU sees the SAME stream (configs[3])
        self.adc = torch.from_numpy(self.adc_host).to(dev)
        self.chans = list(range(NR))

        self.d = d = Ddc(ctx, nchan=NR, max_samples=n)
        self.W = W = Waterfall(ctx, nchan=NR)
        W.set_tables()
        hz_per_start = self.UI_SRATE / (1024 << 14)
        self.params = []
        for ch in range(NR):
            k = first_rx + ch
            z = 1 + k % 10
            p = WfParams.for_zoom(z, (1.0e6 + 0.2e6 * (k % 97)) / hz_per_start, adc_clock=self.ADC_CLOCK, ui_srate=self.UI_SRATE)
            self.params.append(p)
            d.set_wf(ch, p.i_offset, p.decim)
            W.set_channel(ch, p, interp=wf.WF_MAX, window_func=wf.WINF_HANNING, cic_comp=True)
        self.rx = RxDdc(ctx_au, nchan=NR, max_samples=n)
        self.nrec_max = nrec_max = n // RX_DECIM + 2
        self.fir = FastFir(ctx_au, nchan=NR, max_in=nrec_max)
        self.P = Post(self.ctx_tail, nchan=NR)
        self.A = Adpcm(self.ctx_tail, nchan=NR)
        self.fs = fs = self.ADC_CLOCK / RX_DECIM
        self.rx_inc = []
        for ch in range(NR):
            self.rx_inc.append(rx_phase_inc(0.0123 * self.ADC_CLOCK - 1000.0 - 10.0 * (first_rx + ch), self.ADC_CLOCK))
            self.rx.set_freq(ch, self.rx_inc[ch])
            self.fir.setup(ch, 300.0, 2700.0, 0.0, fs)
            self.P.set_agc(ch, True, False, -100, 50, 6, 1000, fs)
            self.P.set_smeter(ch, fs); self.P.set_mode(ch, post.MODE_SSB); self.P.reset(ch)

        self.wf_stride = n + 1
        self.wf_iq = torch.zeros((NR, self.wf_stride, 2), dtype=torch.int16, device=dev)
        self.frames = torch.zeros((NR, 8192, 2), dtype=torch.int16, device=dev)
        self.rows = torch.zeros((NR, 1024), dtype=torch.uint8, device=dev)
        self.pkts = torch.zeros((NR, wire.WF_PKT_MAX), dtype=torch.uint8, device=dev)
        self.raw = torch.zeros((NR, nrec_max * 6), dtype=torch.uint8, device=dev)
        self.xin = torch.zeros((NR, nrec_max, 2), dtype=torch.float32, device=dev)
        self.firo = torch.zeros((NR, 1024, 2), dtype=torch.float32, device=dev)
        self.s16 = torch.zeros((NR, 512), dtype=torch.int16, device=dev)
        self.pay = torch.zeros((NR, 256), dtype=torch.uint8, device=dev)
        self.infos = [(int(self.params[ch].start), self.params[ch].zoom, 0, True) for ch in range(NR)]
        self.counts = {"frames": 0, "audio_blocks": 0}
        self.last = {}                                                # what the last step produced (counts per stage)
        torch.cuda.synchronize(dev)                                   # buffers exist before the side stream touches them

    def audio(self):
        from flydog_sdr_gps_amd import snd
        NR, chans, nrec_max = self.NR, self.chans, self.nrec_max
        nr = self.rx.push_dev(self.adc.data_ptr(), self.n, chans, self.raw.data_ptr(), nrec_max)
        nrec = int(nr.min())
        assert nrec == int(nr.max())
        snd.unpack_rows_dev(self.ctx_au, self.raw.data_ptr(), nrec_max, nrec, NR, self.xin.data_ptr(), nrec_max)
        if self.tails and self.tail_pending:
            self.side.wait_event(self.ev_tail)               # the tail of the step before has read firo
            self.tail_pending = False
        nout = self.fir.process_dev(chans, self.xin.data_ptr(), nrec_max, nrec, self.firo.data_ptr(), 1024)
        self.last.update(nrec=nrec, nout=int(nout[0]))
        assert int(nout.min()) == int(nout.max())
        if int(nout[0]) == 512:
            if self.tails:
                self.ev_fir.record(self.side)
                self.s_tail.wait_event(self.ev_fir)
            self.P.process_dev(chans, self.firo.data_ptr(), 1024, 512, self.s16.data_ptr(), 0, 0, 512)
            self.A.encode_dev(chans, self.s16.data_ptr(), 512, 512, self.pay.data_ptr(), 256)
            if self.tails:
                self.ev_tail.record(self.s_tail)
                self.tail_pending = True
            self.counts["audio_blocks"] += NR

    def step(self):
        from flydog_sdr_gps_amd import wire
        self.audio()                                         # only enqueues, on the side stream
        nw = self.d.push_dev(self.adc.data_ptr(), self.n, self.chans, self.wf_iq.data_ptr(), self.wf_stride)
        assert int(nw.min()) >= 8192
        self.last["nw"] = nw
        self.frames.copy_(self.wf_iq[:, :8192])             # the frame each receiver's waterfall takes this step
        if self.tails and self.pk_pending:
            self.main.wait_event(self.ev_pk)                 # the packets of the step before have read rows
            self.pk_pending = False
        self.W.frames_dev(self.chans, self.frames.data_ptr(), self.rows.data_ptr())
        if self.tails:
            self.ev_frames.record(self.main)
            self.s_pk.wait_event(self.ev_frames)
        wire.wf_packets_dev(self.ctx_pk, self.rows.data_ptr(), 1024, self.infos, self.pkts.data_ptr())
        if self.tails:
            self.ev_pk.record(self.s_pk)
            self.pk_pending = True
        self.counts["frames"] += self.NR

    def close(self):
        for o in (self.d, self.W, self.rx, self.fir, self.P, self.A):
            o.close()


def run_receivers(args, dist):
    """BASELINE configs[3]: a ReceiverBank per GPU (weak scaling over ranks)."""
    import torch
    dev = dist.dev
    two = os.environ.get("KIWIGPU_BENCH_ONE_STREAM") != "1"
    NR, n = args.receivers, 1 << args.log2n
    adc_clock = ReceiverBank.ADC_CLOCK
    bank = ReceiverBank(dist.local_rank, dev, NR, n, dist.rank * NR, two)   # this rank's slice of the receiver set
    step, counts, rows, pay = bank.step, bank.counts, bank.rows, bank.pay

    preroll(step, args.warmup)
    dist.barrier()
    counts["frames"] = counts["audio_blocks"] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)                      # both streams of this rank
    local = time.perf_counter() - t0
    dist.barrier()
    elapsed = dist.max_over_ranks(local)
    # sanity: the strongest carrier is in the band of every waterfall row and audio came out
    assert int(rows.max()) > 100 and counts["audio_blocks"] > 0 and int(pay.to(torch.int32).abs().sum()) > 0
    step_s = elapsed / args.steps
    world = dist.world
    return {
        "metric": "receiver x ADC Msamples/s ingested (waterfall + audio chain per virtual receiver)",
        "value": round(n * NR * world / step_s / 1e6, 1), "unit": "Msamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int128/int64/f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: %d virtual receivers per GPU x %d GPU(s), one %d-sample 16-bit ADC "
                               "block @66.67 MS/s (the same stream on every GPU) resident in HBM per step; per receiver a "
                               "waterfall channel (zooms 1..10) and an SSB audio channel" % (NR, world, n),
                   "receivers_per_gpu": NR, "adc_samples_per_step": n,
                   "parallelism": "receivers sharded over ranks, no data-path collective"},
        "adc_ms_per_step": round(n / adc_clock * 1e3, 3),
        "x_realtime_all_receivers": round(n / adc_clock / step_s, 2),
        "waterfall_frames_per_s": round(counts["frames"] * world / elapsed, 1),
        "audio_blocks_per_s": round(counts["audio_blocks"] * world / elapsed, 1),
    }


def run_stub(args, dist):
    """Launcher / rendezvous self-test without a GPU (tests/test_host_cpu.py): gloo process group,
    barrier, max-over-ranks, one line from rank 0."""
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (dist.rank + 1))
    dist.barrier()
    el = dist.max_over_ranks(time.perf_counter() - t0)
    return {"metric": "launcher self-test", "value": round(dist.world / el, 3), "unit": "ranks/s",
            "config": {"workload": "stub (no GPU)"}, "dtype": "none"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="all",
                    choices=["all", "acq", "acq59", "acq10ms", "wf14", "ddc14", "waterfall", "ddc", "receivers", "stub"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=None, help="acq: independent sample blocks per step (32; acq10ms: 2)")
    ap.add_argument("--frames", type=int, default=2048, help="wf14: frames per channel per step (2048 = 896 MiB of input)")
    ap.add_argument("--log2n", type=int, default=None, help="ddc14 / receivers: log2 of the ADC samples per step (24 / 22)")
    ap.add_argument("--receivers", type=int, default=128, help="receivers: virtual receivers per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the rocprofv3 --pmc child passes; roofline.traffic then comes from profiles/hbm_traffic.json")
    args = ap.parse_args()
    if args.log2n is None:
        args.log2n = 22 if args.workload == "receivers" else 24
    args.workload = {"waterfall": "wf14", "ddc": "ddc14"}.get(args.workload, args.workload)

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))      # the parent never touches the GPU
    if world_env is not None and int(world_env) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s\n" % (args.gpus, world_env))
        sys.exit(2)

    if world_env is None and args.gpus == 1 and not args.no_live_traffic:
        live_traffic_passes(args)                            # children; this process has not touched the GPU yet
    dist = Dist("gloo" if args.workload == "stub" else "nccl")
    common = {"n_gpus": dist.world, "steps": args.steps, "warmup": args.warmup, "preroll_s": PREROLL_S, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "data": "synthetic"}
    if args.workload == "stub":
        line = dict(run_stub(args, dist), **common)
    elif args.workload == "receivers":
        line = run_receivers(args, dist)
    elif args.workload in ("acq", "acq59", "acq10ms", "wf14", "ddc14"):
        fn = {"acq": lambda: run_acq(args, dist), "acq10ms": lambda: run_acq(args, dist, ten_ms=True),
              "acq59": lambda: run_acq(args, dist, all_svs=True),
              "wf14": lambda: run_wf14(args, dist), "ddc14": lambda: run_ddc14(args, dist)}[args.workload]
        r = fn()
        line = dict(r, **common)
        line["steps"] = r.get("steps", args.steps)
    else:
        import torch
        a = run_acq(args, dist)
        torch.cuda.empty_cache()
        w = run_wf14(args, dist)
        torch.cuda.empty_cache()
        d = run_ddc14(args, dist)
        line = dict(a, **common)
        line["metric"] = "IQ Msamples/s ingested (waterfall + GPS acq); value = GPS acq, BASELINE configs[1]"
        line["workloads"] = {"acq": {k: a[k] for k in a if k not in ("metric", "config")},
                             "wf14": w, "ddc14": d}
        line["config"]["also_in_this_line"] = "workloads.wf14 and workloads.ddc14: BASELINE configs[2]"
    if dist.rank == 0:
        print(json.dumps(line), flush=True)
    dist.close()


if __name__ == "__main__":
    main()
