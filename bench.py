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
  cfg2_chain BASELINE configs[2] end to end: 2^24 ADC samples -> the 14 DDC channels -> every
             8192-sample frame those complete (~8190, read where the DDC left them) -> rows, one step
             (the loop of rx/rx_waterfall.cpp:930-939 at 100 % duty).
  receivers  BASELINE configs[3]: --receivers virtual receivers per GPU (waterfall + audio chain).
  acq10ms    BASELINE configs[4]: joint L1 C/A + QZSS + Galileo E1B, 10 ms coherent (65536-point
             transforms), 256 Doppler bins, all 59 SVs.  --shard sv: ONE block, the SV list split
             over the ranks (strong scaling), results all-gathered and merged.
  all        (default) the six above; ONE JSON line whose top level is `acq` (metric, value,
             roofline, cpu_baseline) and whose "workloads" object holds every workload's own
             value, step times, roofline (traffic measured live) and cpu_baseline.
  acq59      the 4 ms shape with the reference's whole SV list (36 C/A + QZSS, 23 E1B): profiles the
             four-accumulator (16368-sample window) correlator beside the C/A one.
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
# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; streams
# beyond that SHARE a queue and run in order with whoever they share it with.  The workloads here use up to six streams
# (the library's side streams included) and RCCL brings its own: with four queues, which streams serialise depended on
# what else had created a stream first -- `receivers` ran 1.12 ms per step alone and 1.40 ms with a process group up,
# `ddc14` 0.48 and 0.58.  Eight queues make the stream topology the one the code states (measured: tools/ab_rxenv.sh).
# Read by the runtime when it initialises, i.e. it must be set before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# knock-out builds of a kernel (tools/ko_*.sh: "what is the time sensitive to") produce wrong results by construction: the
# result checks are skipped and the line says so
TIMING_EXPERIMENT = os.environ.get("KIWIGPU_BENCH_TIMING_EXPERIMENT", "0") == "1"

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
        # KIWIGPU_BENCH_SHARE_GPU=1: a REHEARSAL of the N > 1 code path on a box with one GPU -- every rank on device 0, the
        # process group over gloo (RCCL refuses two ranks on one device); the line is tagged invalid_for_scaling
        self.share_gpu = backend == "nccl" and self.world > 1 and os.environ.get("KIWIGPU_BENCH_SHARE_GPU") == "1"
        if self.share_gpu:
            self.local_rank = 0
            self.backend = "gloo"
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
            os.environ.setdefault("RANK", "0")                   # (KIWIGPU_BENCH_FORCE_DIST=1 without a launcher)
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.backend == "nccl":
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
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev if (self.dev is not None and self.backend == "nccl") else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather_bytes(self, out_dev, mine_dev):
        """every rank's `mine_dev` (a uint8 device tensor) into out_dev, rank-major: one RCCL all-gather on the device; in the
        one-GPU rehearsal (gloo) through the host"""
        import torch
        import torch.distributed as dist
        if self.backend == "nccl":
            dist.all_gather_into_tensor(out_dev, mine_dev)
            return
        raw = mine_dev.cpu()
        outs = [torch.empty_like(raw) for _ in range(self.world)]
        dist.all_gather(outs, raw)
        out_dev.copy_(torch.cat(outs))

    def close(self):
        if self.on:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


_T0 = time.perf_counter()


def sha16(path):
    import hashlib
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def box_identity(local_rank=0):
    """Which machine, which GPU, which build produced this line (the profiles/ files of the same gpurun lease carry the
    same record, tools/prof.sh): boxes differ by +-4 %, so a trace and a bench line are comparable only when these agree.
    Never initialises the GPU in this process (rocminfo / rocm-smi run as children)."""
    import re
    import shutil
    import socket
    # the library that is MEASURED: KIWIGPU_LIBRARY names another build for A/B runs (flydog_sdr_gps_amd/_lib.py, library_path)
    lib_path = os.environ.get("KIWIGPU_LIBRARY") or os.path.join(ROOT, "flydog_sdr_gps_amd", "libkiwigpu.so")
    box = {"host": socket.gethostname(), "lib_sha16": sha16(lib_path), "bench_sha16": sha16(os.path.abspath(__file__)),
           "gpu_uuid": None, "gpu_name": None}
    if os.environ.get("KIWIGPU_LIBRARY"):
        box["lib_path"] = lib_path
    try:
        if shutil.which("rocminfo"):
            txt = subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=60).stdout
            gpus = re.findall(r"Uuid:\s+(GPU-\S+)", txt)
            names = re.findall(r"Marketing Name:\s+(.*\S)", txt)
            if gpus:
                box["gpu_uuid"] = gpus[min(local_rank, len(gpus) - 1)]
            gnames = [n for n in names if "Instinct" in n or "MI3" in n]
            if gnames:
                box["gpu_name"] = gnames[0]
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        with open("/etc/machine-id") as f:
            box["machine_id8"] = f.read().strip()[:8]
    except OSError:
        pass
    return box


def log(msg):
    """progress on stderr (stdout carries the ONE JSON line)"""
    if os.environ.get("RANK", "0") == "0":
        sys.stderr.write("bench.py [%6.1f s] %s\n" % (time.perf_counter() - _T0, msg))
        sys.stderr.flush()


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


_CPU_UNIT = {}               # what the forked workers of a CPU leg run (inherited through fork, never pickled)


def _cpu_worker(budget):
    fn = _CPU_UNIT["unit"]
    fn()                                              # warm-up in this process (page faults, lazy imports)
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget:
            return n, el


def cpu_threads(one_unit, budget_s):
    """Every usable host core runs whole single-threaded units back to back for about budget_s seconds
    (independent units, like the GPU's batch), one forked worker PROCESS per core: threads of one interpreter
    serialise on its lock wherever a unit is many short calls (round 2's thread version gave the 14-channel
    frame leg 1.5x on 16 threads).  Forks: so this is only ever called BEFORE the process initialises the GPU
    (cpu_legs).  -> (units done, seconds, cores, seconds of one unit on one core)"""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    cores = usable_cores()
    one_unit()                                                    # warm-up, then the single-core figure
    t0 = time.perf_counter()
    one_unit()
    t1 = time.perf_counter() - t0
    _CPU_UNIT["unit"] = one_unit
    with ProcessPoolExecutor(cores, mp_context=mp.get_context("fork")) as pool:
        got = list(pool.map(_cpu_worker, [budget_s] * cores))
    el = max(e for _, e in got)
    return sum(n for n, _ in got), el, cores, t1


CPU_LEGS = {}                # workload -> cpu_baseline object, filled by cpu_legs() BEFORE the GPU is touched


# ------------------------------------------------------------------------------------------------
# HBM traffic, measured by the run that prints it
# ------------------------------------------------------------------------------------------------
ALL_WORKLOADS = ["acq", "wf14", "ddc14", "cfg2_chain", "receivers", "receivers_light", "acq10ms"]     # the default line, in run order
if os.environ.get("KIWIGPU_BENCH_ORDER"):            # a diagnostic: another run order / a subset (the line then lacks workloads)
    ALL_WORKLOADS = [w for w in os.environ["KIWIGPU_BENCH_ORDER"].split(",") if w in ALL_WORKLOADS]
    if "acq" not in ALL_WORKLOADS:
        ALL_WORKLOADS.insert(0, "acq")
PMC_WORKLOADS = ALL_WORKLOADS + ["acq59"]
PMC_STEPS = 4                # steps inside a workload's marked window of a counter pass
LIVE_TRAFFIC = {}            # workload -> (bytes, source), filled by live_traffic_passes()
# What `traffic` is per workload: the named (dominant) kernels per launch -- the same unit as roofline.achieved --
# or, for the workloads whose roofline is the whole step, every kernel dispatched in a step.
TRAFFIC_KERNELS = {"acq": ("acq_correlate_kernel<4, 1,",),
                   "acq59": ("acq_correlate_kernel<4, 1,", "acq_correlate8_kernel<4"),
                   "acq10ms": ("acq_correlate_kernel<16, 1,", "acq_correlate8_kernel<16"),
                   "wf14": ("wf_frame_kernel",),
                   "ddc14": None, "cfg2_chain": None, "receivers": None, "receivers_light": None}


def pmc_tag(workload):
    """Grid size (in workgroups) of the kg_ctx_mark() kernel that opens a workload's window; + 1 closes it."""
    return 100 + 2 * PMC_WORKLOADS.index(workload)


def pmc_window(ctx, workload, step, sync):
    """Counter-pass mode (--pmc-child): two untimed steps, then PMC_STEPS steps between two marker kernels.
    `rocprofv3 --pmc` rows carry neither timestamps nor user markers, so the parent attributes rows to
    this window by dispatch order (kg_ctx_mark, include/kiwigpu.h)."""
    for _ in range(2):
        step()
    sync()
    ctx.mark(pmc_tag(workload))
    for _ in range(PMC_STEPS):
        step()
    sync()                                           # every stream of the step has been enqueued AND has run
    ctx.mark(pmc_tag(workload) + 1)
    sync()
    return {"pmc_child": workload}


def under_profiler():
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def parse_counter_rows(files, counter):
    """counter_collection.csv files -> rows [(dispatch id, kernel name, workgroups, value)] in dispatch order."""
    import csv
    rows = []
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                wg = max(1, int(float(row.get("Workgroup_Size", "64") or 64)))
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], int(float(row["Grid_Size"])) // wg,
                             float(row["Counter_Value"])))
    rows.sort()
    return rows


def window_traffic(rows, workload):
    """KB of the workload's marked window: -> (per launch of the dominant kernels | per step of all kernels,
    {kernel: KB per step} of the largest contributors)."""
    tag, inside, win = pmc_tag(workload), False, []
    for _, name, groups, val in rows:
        if "kg_mark_kernel" in name:
            if groups == tag:
                inside, win = True, []
            elif groups == tag + 1 and inside:
                inside = False
                break
            continue
        if inside:
            win.append((name, val))
    if inside or not win:
        raise RuntimeError("window of %s not found" % workload)
    per = {}
    for name, val in win:
        per.setdefault(name, []).append(val)
    keys = TRAFFIC_KERNELS[workload]
    by_step = {k: sum(v) / PMC_STEPS for k, v in per.items()}
    top = dict(sorted(by_step.items(), key=lambda kv: -kv[1])[:4])
    if keys is None:
        return sum(by_step.values()), top
    tot = 0.0
    for k in keys:
        vals = [v for name, vs in per.items() if k in name for v in vs]
        if not vals:
            raise RuntimeError("kernel %s not seen" % k)
        tot += sum(vals) / len(vals)
    return tot, top


def live_traffic_passes(args):
    """HBM bytes of each workload, measured NOW: before this process touches the GPU it runs this same file
    as a child under `rocprofv3 --pmc` -- ONE pass for FETCH_SIZE, one for WRITE_SIZE (separate passes,
    nothing but the counter: MI355X_MICROARCH.md's HBM section), each running EVERY workload of this run on
    its own per-launch configuration, PMC_STEPS steps between two marker kernels.  bytes = (2 x FETCH_SIZE +
    WRITE_SIZE) KB: gfx950 tallies its 128-byte read requests as 64.  Any failure (no rocprofv3, already
    under a profiler, a pass timing out) leaves LIVE_TRAFFIC without the entry and the line falls back to
    the committed figure, saying so."""
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None or under_profiler():
        return
    wls = [w for w in (ALL_WORKLOADS if args.workload == "all" else [args.workload]) if w in TRAFFIC_KERNELS]
    if not wls:
        return
    child = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--pmc-child",
             "--no-cpu", "--no-live-traffic", "--frames", str(args.frames), "--receivers", str(args.receivers)]
    if args.blocks is not None:
        child += ["--blocks", str(args.blocks)]
    if args.log2n_given:
        child += ["--log2n", str(args.log2n)]
    kb = {}
    tmp = tempfile.mkdtemp(prefix="kiwigpu_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child
            # (a child of a rank started by torch.distributed.run must not think it is one)
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                     "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            env["TMPDIR"] = os.environ.get("TMPDIR", "/tmp")
            p = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                 start_new_session=True)     # its own process group: a stuck pass is ended whole
            try:
                rc = p.wait(timeout=300)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)
                p.wait()
                raise RuntimeError("pass timed out")
            if rc != 0:
                raise RuntimeError("pass failed, status %d" % rc)
            rows = parse_counter_rows(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), counter)
            log("counter pass %s: %d rows" % (counter, len(rows)))
            for wl in wls:
                try:
                    kb.setdefault(wl, {})[counter] = window_traffic(rows, wl)
                except RuntimeError as e:
                    sys.stderr.write("bench.py: live PMC figure for %s not available (%s)\n" % (wl, e))
        for wl in wls:
            if len(kb.get(wl, {})) != 2:
                continue
            (fk, ftop), (wk, wtop) = kb[wl]["FETCH_SIZE"], kb[wl]["WRITE_SIZE"]
            what = "per launch of the dominant kernel(s)" if TRAFFIC_KERNELS[wl] else "per step, every kernel of the step"
            LIVE_TRAFFIC[wl] = (int((2 * fk + wk) * 1024),
                                "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run, %s "
                                "(2 x %.1f KB + %.1f KB)" % (what, fk, wk),
                                {k: int((2 * ftop.get(k, 0.0) + wtop.get(k, 0.0)) * 1024)
                                 for k in list(ftop)[:3]})
    except Exception as e:                                    # noqa: BLE001 -- fall back, and say so
        sys.stderr.write("bench.py: live PMC passes not available (%s)\n" % e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measured_traffic(workload, units):
    """HBM bytes of the workload (per launch of its dominant kernels, or per step: TRAFFIC_KERNELS): the live
    PMC child passes of this run (live_traffic_passes) when they ran, else the committed passes
    (profiles/hbm_traffic.json, written the same way).  -> (bytes or None, where the number comes from,
    {kernel: bytes per step} of the largest contributors or None)"""
    if workload in LIVE_TRAFFIC:
        return LIVE_TRAFFIC[workload]
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            tab = json.load(f)
        e = tab[workload][str(units)]
        return (e["bytes_per_launch"], "committed profile: %s" % e.get("source", tab.get("_source", "profiles/hbm_traffic.json")),
                None)
    except (OSError, KeyError, ValueError):
        return None, "no live PMC pass and no committed one for this configuration", None


def event_spread(launch, reps):
    """Per-launch duration of `launch` (one enqueue on torch's current stream, which the contexts of the bench
    share): a device event between back-to-back launches.  -> (min, median) in ms"""
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        launch()
        ev[i + 1].record()
    torch.cuda.synchronize()
    dts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return round(dts[0], 5), round(dts[len(dts) // 2], 5)


PREROLL_S = float(os.environ.get("KIWIGPU_BENCH_PREROLL_S", "0.5"))    # 0.12 s left the first timed region of a workload 5-30 % slow on some boxes


def dev_sync():
    """torch.cuda.synchronize() where there is a device (the launcher self-test runs timed_steps without one)"""
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def preroll(step, warmup, dist=None):
    """The W untimed warm-up steps, then more untimed ones until about PREROLL_S seconds of work have run
    (at least 64 steps): an MI355X that has been idle reaches its steady clock only after tens of
    milliseconds of load (measured on the correlator: 0.96 ms per launch at the start, 0.84 after 20
    launches, 0.80 after 60), and the first time two streams of a process depend on each other (the DDC's
    side stream) the HIP runtime stalls the GPU side for 30-50 ms, once per process, some thirty calls
    in.  A short timed region would otherwise hold both.  Never part of the timed region.
    The NUMBER of pre-roll steps is agreed between the ranks (the largest estimate counts): a step may hold a collective --
    `--shard sv` gathers the winners every step -- and ranks that each sized their pre-roll from their own clock ran different
    numbers of steps: the ranks with more of them waited for ever (found by the one-GPU rehearsal of the N > 1 line, round 5)."""
    for _ in range(max(1, warmup)):
        step()
    dev_sync()                                       # one-time costs (code load, lazy allocations) are behind us
    probe = max(4, min(warmup, 16))
    t0 = time.perf_counter()
    for _ in range(probe):
        step()
    dev_sync()
    est = max((time.perf_counter() - t0) / probe, 1e-5)
    count = min(4000, max(64, int(PREROLL_S / est)))
    if dist is not None and dist.on:
        count = int(dist.max_over_ranks(float(count)))
    for _ in range(count):
        step()


def timed_steps(dist, step, steps, warmup):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides, max
    over ranks; then the same K steps once more with a device event between steps for the spread
    (not part of the headline time).  -> (seconds, host enqueue seconds, spread dict)"""
    import torch
    preroll(step, warmup, dist)
    dist.barrier()                                   # barrier, then synchronize
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_enq = time.perf_counter() - t0
    dev_sync()                                       # this rank's K steps are done ...
    local = time.perf_counter() - t0
    dist.barrier()                                   # ... every rank's are: the slowest rank's time counts
    elapsed = dist.max_over_ranks(local)
    if not torch.cuda.is_available():                # (the launcher self-test: the same K steps once more, host clock)
        dts = []
        for i in range(steps):
            t1 = time.perf_counter()
            step()
            dts.append((time.perf_counter() - t1) * 1e3)
        dts.sort()
        return elapsed, t_enq, {"min": round(dts[0], 5), "median": round(dts[len(dts) // 2], 5), "max": round(dts[-1], 5), "how": "host clock (no device)"}
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
def acq_flops_per_cell(fft_len, window):
    """SURVEY.md 8(d)'s figure for one (SV, Doppler) cell: 5 N log2 N for the inverse transform, 6 N for the N
    conjugate products, 3 W for power, maximum and sum over the W lags searched (W = 4092 for C/A, 16368 for E1B) --
    1 257 460 (C/A) and 1 294 288 (E1B) at N = 16384.  (Rounds 2-4 counted the operations of the kernels' own decomposition:
    fewer for the C/A cells -- the pruned transform skips work --, more for the E1B cells, where it charged 40 flops per
    point for the four output quarters the kernel forms with 14: VERDICT r4, weak 5.)"""
    log2n = fft_len.bit_length() - 1
    assert 1 << log2n == fft_len
    return 5 * fft_len * log2n + 6 * fft_len + 3 * window


class DevBytes:
    """A device buffer as torch sees it (zero copy): torch.as_tensor(DevBytes(ptr, n), device=...)."""

    def __init__(self, dptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(dptr), False), "version": 2}


def run_acq(args, dist, ten_ms=False, all_svs=False):
    import numpy as np
    import torch
    from flydog_sdr_gps_amd import Context, Searcher, acq, prn, sats, shard, synth
    from flydog_sdr_gps_amd._lib import result_dtype
    from tests.fixtures import e1b_chips
    dev = dist.dev
    ctx = Context(dist.local_rank, torch.cuda.current_stream(dev).cuda_stream)
    wl = "acq10ms" if ten_ms else ("acq59" if all_svs else "acq")
    shard_sv = ten_ms and args.shard == "sv"
    if ten_ms:
        B = args.blocks or (1 if shard_sv else 2)
        nsamples, fft_len, dop_lo, dop_hi = acq.NSAMPLES_10MS, acq.FFT_LEN_10MS, -128, 127
        codes = synth.all_sv_codes(e1b_chips())
        make_block = lambda b: synth.config4_iq16(codes, seed=0x5EED0005 + b)            # noqa: E731
    else:
        B = args.blocks or 32
        nsamples, fft_len, dop_lo, dop_hi = NSAMPLES, FFT_LEN, -20, 20
        # all_svs: the reference's whole Sats[] list (36 C/A + QZSS rows and the 23 E1B rows, whose
        # 16368-sample window takes the four-accumulator kernel) on the configs[1] blocks
        codes = synth.all_sv_codes(e1b_chips()) if all_svs else \
            [(prn.cacode(sats.SATS[s][1], sats.SATS[s][2]), False) for s in range(NSV)]
        make_block = lambda b: synth.config1_iq16(seed=0x5EED0002 + b)                  # noqa: E731
    nsv_all = len(codes)
    # --shard sv (first bullet; gps/search.cpp:512-604 is the loop being partitioned): every rank
    # holds the SAME block(s) and searches its share of the SV list -- cost-balanced (shard.split_units_weighted: an
    # E1B SV costs 1.6 C/A ones and gps/sats.cpp:25-142 puts the 23 E1B rows last; a contiguous split gives one rank of
    # eight 124 % of the mean)
    sv_weights = shard.sv_weights([b for _, b in codes])
    sv_shares = shard.split_units_weighted(sv_weights, dist.world) if shard_sv else [list(range(nsv_all))]
    svs = sv_shares[dist.rank] if shard_sv else list(range(nsv_all))
    ndop, P = dop_hi - dop_lo + 1, fft_len // 4096
    s = Searcher(ctx, dop_lo=dop_lo, dop_hi=dop_hi, max_blocks=2 * B, nsamples=nsamples, fft_len=fft_len)
    for sat in svs:
        s.set_code(sat, codes[sat][0], boc=codes[sat][1])
    # each rank gets its own seeded blocks ("receivers"), resident in HBM; --shard sv: every rank the same ones
    blocks = list(range(B)) if shard_sv else shard.block_ids(dist.rank, dist.world, B)
    iq_host = [make_block(b) for b in blocks]
    iq_dev = torch.from_numpy(np.stack(iq_host)).to(dev)          # [B][2*nsamples] int16, resident
    iq_ptr = int(iq_dev.data_ptr())
    parity = [0]
    nsv_max = max(len(sh) for sh in sv_shares) if shard_sv else len(svs)
    rbytes = B * nsv_max * result_dtype.itemsize
    if shard_sv and dist.on and dist.dev is not None:
        res_dev = torch.as_tensor(DevBytes(s.results_dev(), rbytes), device=dev)     # the library's result array, zero copy
        gathered_dev = torch.empty(dist.world * rbytes, dtype=torch.uint8, device=dev)

    def step():
        # Sample() then Correlate() of this step's blocks, in order on one stream (two sets of
        # blocks alternate so that consecutive steps never touch the same spectra)
        first = parity[0] * B
        parity[0] ^= 1
        s.sample_iq16_batch(iq_ptr, B, first_block=first)
        s.correlate_async(svs, nblocks=B, first_block=first)
        if shard_sv and dist.on and dist.dev is not None:
            # the exchange step of the SV-sharded search: every rank's winners to every rank (RCCL, 16 B per
            # (block, SV)); the merge is shard.merge_sv_shards on the host after the timed region
            dist.all_gather_bytes(gathered_dev, res_dev)

    if args.pmc_child:
        out = pmc_window(ctx, wl, step, lambda: torch.cuda.synchronize(dev))
        s.close(); ctx.close()
        return out
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
    k_min, k_med = event_spread(lambda: s.correlate_async(svs, nblocks=B), kreps)

    res, _ = s.fetch(want_cells=False)
    if shard_sv:
        if dist.on and dist.dev is not None:
            parts = gathered_dev.cpu().numpy().view(result_dtype).reshape(dist.world, -1)
        else:
            parts = res.reshape(1, -1)
        res = shard.merge_sv_shards(parts, B, sv_shares)                                    # [B][59]
        assert res.shape == (B, nsv_all)
    # (E1B rows: 41 x 16368 trials per SV put the noise maximum close to the reference's MIN_SIG = 16)
    min_sig = synth.MIN_SIG_10MS if ten_ms else (24.0 if all_svs else acq.MIN_SIG)
    res_svs = list(range(nsv_all)) if shard_sv else svs
    found = sorted(int(sv) for i, sv in enumerate(res_svs) if res[0, i]["snr"] >= min_sig)
    if dist.on and dist.dev is not None and not shard_sv:
        gathered = shard.gather_results(res, dev if dist.backend == "nccl" else None)       # RCCL all_gather of the tiny result arrays
        assert gathered.shape[0] == dist.world * B
    expect = sorted(p[0] for p in synth.CONFIG4_PRESENT) if ten_ms else \
        sorted(p - 1 for p, *_ in synth.CONFIG1_PRESENT)
    # every injected SV must be found on every rank; an absent SV above the threshold is a noise
    # false alarm of that rank's own seeded block (41 x 4092 trials at MIN_SIG = 16: about 2 % per SV)
    assert TIMING_EXPERIMENT or set(expect) <= set(found), "acquisition result wrong: %s lacks %s" % (found, sorted(set(expect) - set(found)))
    if dist.rank == 0 and not all_svs:
        assert TIMING_EXPERIMENT or found == expect, "acquisition result wrong: %s != %s" % (found, expect)

    # After the timed region: block 0's winners against the oracle's chain run end to end from the int16 samples (rank 0 of a
    # one-GPU run, the 4 ms shape, C/A rows: 0.2 s on one host core) -- Doppler bin, code phase and the valid flag EQUAL, snr to
    # north_star's 1e-5.  tests/test_acq_gpu.py holds all 1312 cells of this configuration; this is the same statement about
    # the very data the timed launches correlated.
    oracle_check = None
    if not TIMING_EXPERIMENT and dist.world == 1 and dist.rank == 0 and not shard_sv and not args.pmc_child:
        from oracle import kiwi_oracle as ko
        from tests.fixtures import oracle_next_rows
        ca = [i for i, sat in enumerate(svs) if not codes[sat][1]]
        if ten_ms:                                      # configs[4]: 256 bins of 65536 points a cell -- four rows (two of them present), ~1 s
            ca = ca[:2] + [i for i in ca[2:] if svs[i] in (6, 13)][:2]
        data = ko.sample_iq16(iq_host[0], nsamples=nsamples, fft_len=fft_len)
        c_fft = np.stack([ko.code_fft(codes[svs[i]][0], boc=False, fft_len=fft_len) for i in ca])
        want, _ = ko.correlate_many(c_fft, data, [sats.L1_LIMIT] * len(ca), dop_lo=dop_lo, dop_hi=dop_hi, nthreads=8,
                                    nexts=oracle_next_rows(ko, s, [svs[i] for i in ca]))
        got = res[0, ca]
        same = bool(np.array_equal(got["dop"], want["dop"]) and np.array_equal(got["idx"], want["idx"]) and np.array_equal(got["valid"], want["valid"]))
        snr_err = float(np.max(np.abs(got["snr"] - want["snr"]) / np.maximum(np.abs(want["snr"]), 1e-30)))
        assert same and snr_err <= 1e-5, "block 0's winners differ from the oracle's (bins/phases equal: %s, snr rel. error %.2e)" % (same, snr_err)
        oracle_check = {"block0_svs_vs_oracle": len(ca), "dop_idx_valid": "equal", "snr_rel_err_max": float("%.2e" % snr_err)}

    n1 = sum(1 for sat in svs if not codes[sat][1])
    n4 = len(svs) - n1
    flops_launch = B * ndop * (n1 * acq_flops_per_cell(fft_len, sats.L1_LIMIT) + n4 * acq_flops_per_cell(fft_len, sats.E1B_LIMIT))
    cells = B * len(svs) * ndop
    bytes_cell = 2 * fft_len * 8 + 16                  # both spectra once + the result
    traffic, source, _ = measured_traffic(wl, B)
    tfl = flops_launch / (kernel_ms * 1e-3) / 1e12
    out = {
        "metric": "IQ Msamples/s ingested (GPS acq: Sample + %d SV x %d Doppler Correlate)" % (nsv_all, ndop),
        # weak scaling: every rank its own blocks; --shard sv (strong): the same B blocks, searched once by all ranks
        "value": round(float(B) * nsamples * args.steps * (1 if shard_sv else dist.world) / elapsed / 1e6, 3),
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
            "parallelism": ("--shard sv: the same %d block(s) on every GPU, the %d SVs split over %d GPU(s) by cost (E1B = %.1f C/A; "
                            "rank loads %s of mean %.2f), winners all-gathered over RCCL every step"
                            % (B, nsv_all, dist.world, shard.SV_WEIGHT_E1B, [round(sum(sv_weights[i] for i in sh), 1) for sh in sv_shares],
                               sum(sv_weights) / dist.world)) if shard_sv else
                           "replicated codes, sample blocks sharded over %d GPU(s), no data-path collective" % dist.world,
        },
        # What bounds the correlator: fp32 vector arithmetic -- every operand is
        # served from L2, the kernel cannot be HBM-bound.  achieved = nominal FFT flops / HIP-event time.
        "roofline": {
            "bound": "valu", "kernel": "acq_correlate_kernel<%d,1>%s" % (P, " + acq_correlate8_kernel<%d>" % P if n4 else ""),
            "achieved": round(tfl, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tfl / VALU_PEAK_TFLOPS, 4),
            "traffic": traffic, "traffic_source": source,
            "kernel_ms": round(kernel_ms, 5), "kernel_ms_min": k_min, "kernel_ms_median": k_med,
            "flops_per_launch": flops_launch, "flops_model": "SURVEY 8(d): 5 N log2 N + 6 N + 3 W per cell",
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
    if oracle_check:
        out["checked"] = dict(oracle_check, found_svs=found)
    if shard_sv:
        out["scaling"] = "strong"
    if dist.world == 1 and dist.rank == 0 and not shard_sv:
        # SURVEY 8(d)'s "ingested" variant: the blocks start in HOST memory (one pinned transfer per batch on the copy
        # stream) and the winners end there; two batches in flight.  Never `value`.
        host = np.ascontiguousarray(np.stack(iq_host))
        reps = max(6, min(40, args.steps))
        par = 0
        t0 = time.perf_counter()                             # untimed, 0.3 s: a process's first host-to-device copies are slow (the link /
        while time.perf_counter() - t0 < 0.3:                # DMA engine leaving its idle state: see run_receivers)
            for _ in range(4):
                s.sample_iq16_host_batch(host, par * B); s.correlate_async(svs, nblocks=B, first_block=par * B); par ^= 1
            s.fetch(want_cells=False)
        t0 = time.perf_counter()
        for _ in range(reps):
            s.sample_iq16_host_batch(host, par * B)
            s.correlate_async(svs, nblocks=B, first_block=par * B)
            par ^= 1
        s.fetch(want_cells=False)                            # waits for the stream, copies the last batch's winners out
        t_host = (time.perf_counter() - t0) / reps
        out["ingest_pcie_Msps"] = round(B * nsamples / t_host / 1e6, 1)
        out["ingest_pcie_note"] = ("host int16 IQ in (%d blocks per pinned batch transfer, two batches in flight), winners "
                                   "fetched to the host: %.1f us per block" % (B, t_host / B * 1e6))
    if wl in CPU_LEGS:
        out["cpu_baseline"] = CPU_LEGS[wl]
        out["speedup_vs_cpu_all_cores"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        if wl in POCKETFFT:
            out["cpu_baseline_pocketfft"] = POCKETFFT[wl]
            if POCKETFFT[wl].get("value"):
                out["speedup_vs_cpu_pocketfft"] = round(out["value"] / POCKETFFT[wl]["value"], 1)
    s.close()
    ctx.close()
    return out


def cpu_rate(one_unit, budget_s):
    """units per second of `one_unit` on all usable host threads (cpu_threads) -> (rate, cores, single-thread seconds)"""
    reps, el, cores, t1 = cpu_threads(one_unit, budget_s)
    return max(reps, 1) / el if reps else 1.0 / t1, cores, t1


def cpu_acq(iq, codes, nsamples, fft_len, dop_lo, dop_hi, budget_s):
    """The CPU oracle (kind "port": scalar fp32 radix-4 FFT in place of FFTW3f, which is absent from
    this image; built for this machine's CPU when gcc is present, native_oracle()) on the same workload:
    Sample() + Correlate() for every SV and bin of one block."""
    import numpy as np
    from oracle import kiwi_oracle as ko
    ko.lib()
    spectra = np.stack([ko.code_fft(c, boc=b, prec=0, fft_len=fft_len) for c, b in codes])
    limits = [ko.E1B_LIMIT if b else ko.L1_LIMIT for _, b in codes]
    ndop = dop_hi - dop_lo + 1
    if fft_len == FFT_LEN and not any(b for _, b in codes):
        def unit():
            data = ko.sample_iq16(iq, prec=0)
            ko.correlate_many(spectra, data, limits, prec=0, nthreads=1, want_cells=False)
        reps, el, cores, t1 = cpu_threads(unit, budget_s)
        value, single = reps * nsamples / el / 1e6, nsamples / t1 / 1e6
        what = "%d x the full configs[1] block (Sample + %d SV x %d bins)" % (reps, len(codes), ndop)
    else:
        # The real mix, leg by leg: Sample() once per block, then one Correlate() per SV -- n_ca C/A SVs (4092-lag
        # peak search) and n_e1b E1B SVs (16368 lags).  Each leg is timed on all cores; a block on all cores takes
        # 1 / r_sample + n_ca / r_ca + n_e1b / r_e1b.
        ca = next(i for i, (_, b) in enumerate(codes) if not b)
        e1b = next(i for i, (_, b) in enumerate(codes) if b)
        n_e1b = sum(1 for _, b in codes if b)
        n_ca = len(codes) - n_e1b
        data0 = ko.sample_iq16(iq, prec=0, nsamples=nsamples, fft_len=fft_len)

        def corr(i):
            return lambda: ko.correlate_many(spectra[[i]], data0, [limits[i]], dop_lo=dop_lo, dop_hi=dop_hi,
                                             prec=0, nthreads=1, want_cells=False)
        r_s, cores, t_s = cpu_rate(lambda: ko.sample_iq16(iq, prec=0, nsamples=nsamples, fft_len=fft_len), budget_s / 6)
        r_ca, _, t_ca = cpu_rate(corr(ca), budget_s / 3)
        r_e1b, _, t_e1b = cpu_rate(corr(e1b), budget_s / 2)
        value = nsamples / (1.0 / r_s + n_ca / r_ca + n_e1b / r_e1b) / 1e6
        single = nsamples / (t_s + n_ca * t_ca + n_e1b * t_e1b) / 1e6
        what = ("legs timed separately and combined as t_sample + %d t_ca + %d t_e1b (one SV x %d bins each: C/A %.3f s, "
                "E1B %.3f s, Sample %.4f s on one thread)" % (n_ca, n_e1b, ndop, t_ca, t_e1b, t_s))
        el = budget_s
    return {
        "value": round(value, 5), "unit": "Msamples/s", "cores": cores,
        "machine_cpus": os.cpu_count(), "kind": "port",
        "sample": "%s, oracle fp32 FFT (%s), %d worker processes each running whole units, ~%.0f s" % (what, native_oracle_note(), cores, el),
        "single_thread_value": round(single, 5),
        "port_vs_reference": "FFTW3f is unavailable here; cpu_baseline_pocketfft times the same work with the tuned FFT "
                             "that IS in the image (scipy.fft / pocketfft)",
    }


POCKETFFT = {}               # workload -> cpu_baseline_pocketfft object, filled by pocketfft_legs() BEFORE the GPU is touched
_PF = {}                     # what the forked workers of a leg run (inherited through fork, never pickled)


def _pf_worker(budget):
    fn = _PF["unit"]
    fn()                                              # warm-up in this process (plans, page faults)
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget:
            return n, el


def pf_rate(unit, budget_s):
    """units per second of `unit` with one forked worker PROCESS per usable core (many small numpy calls: threads
    would serialise on the interpreter lock).  Only ever called before this process has initialised the GPU."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    cores = usable_cores()
    _PF["unit"] = unit
    with ProcessPoolExecutor(cores, mp_context=mp.get_context("fork")) as pool:
        got = list(pool.map(_pf_worker, [budget_s] * cores))
    return sum(n / el for n, el in got), cores


def cpu_acq_pocketfft(iq, codes, nsamples, fft_len, dop_lo, dop_hi, budget_s):
    """The same Sample() FFT + Correlate() work with an independent tuned FFT (scipy.fft = pocketfft, complex64,
    eight Doppler bins per batched call so that the working set stays in L2, one SV per host process): what the FFT
    library that IS in this image does with this loop on these cores.  Not a parity oracle; its winner is compared
    with the oracle's as a sanity flag only."""
    import numpy as np
    import scipy.fft as sf
    from numpy.lib.stride_tricks import as_strided
    from oracle import kiwi_oracle as ko
    ko.lib()
    N, ndop = fft_len, dop_hi - dop_lo + 1
    _, td = ko.sample_iq16(iq, prec=0, want_td=True, nsamples=nsamples, fft_len=fft_len)   # mix + 2 half-bands: not FFT work
    td = np.ascontiguousarray(td, np.complex64)
    i_ca = next(i for i, (_, b) in enumerate(codes) if not b)
    i_e1b = next((i for i, (_, b) in enumerate(codes) if b), i_ca)
    code_td = [np.ascontiguousarray(ko.code_replica(codes[i][0], boc=codes[i][1], fft_len=fft_len)[0], np.complex64)
               for i in (i_ca, i_e1b)]
    chunk = 8

    def one_sv(which, limit):
        dconj = np.conj(sf.fft(td))                                  # Sample()'s transform
        cspec = sf.fft(code_td[which])                               # (SearchInit's; cheap beside the 41 / 256 inverse ones)
        c3 = np.concatenate([cspec, cspec, cspec])                   # code[(k - dop) mod N] = c3[N + k - dop]: a strided view, no copy
        best = (0.0, 0, 0)
        for d0 in range(dop_lo, dop_hi + 1, chunk):
            nd = min(chunk, dop_hi + 1 - d0)
            v = as_strided(c3[N - d0:], shape=(nd, N), strides=(-c3.strides[0], c3.strides[0]), writeable=False)
            y = sf.ifft(dconj * v, axis=1, norm="forward")[:, :limit]                   # unnormalised backward transform
            pw = y.real * y.real + y.imag * y.imag
            i = pw.argmax(axis=1)
            mx = pw[np.arange(nd), i]
            snr = mx / (pw.sum(axis=1) / limit)
            j = int(snr.argmax())
            if snr[j] > best[0]:
                best = (float(snr[j]), d0 + j, int(i[j]))
        return best

    n_e1b = sum(1 for _, b in codes if b)
    n_ca = len(codes) - n_e1b
    legs = [(0, ko.L1_LIMIT, n_ca)] + ([(1, ko.E1B_LIMIT, n_e1b)] if n_e1b else [])
    t_block, notes, cores = 0.0, [], 1
    for which, limit, count in legs:
        rate, cores = pf_rate(lambda: one_sv(which, limit), budget_s / len(legs))
        t_block += count / rate
        notes.append("%s: %.4f s per SV x %d bins on all cores" % ("E1B" if which else "C/A", 1.0 / rate, ndop))
    got = one_sv(0, ko.L1_LIMIT)
    # (this leg indexes the code spectrum modulo N; the reference reads the NEXT row behind a satellite's own for a negative bin --
    # the oracle is told to put the same spectrum there, which IS modulo N: the flag compares like with like)
    c_ca = ko.code_fft(codes[i_ca][0], prec=0, fft_len=fft_len)
    want, _ = ko.correlate(c_ca, ko.sample_iq16(iq, prec=0, nsamples=nsamples, fft_len=fft_len), dop_lo=dop_lo, dop_hi=dop_hi, prec=0,
                           code_next=c_ca)
    return {"value": round(nsamples / t_block / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "independent tuned FFT",
            "sample": "scipy.fft (pocketfft) complex64 + numpy, one SV per process, %d processes, 8 Doppler bins per batched call; "
                      "%s; block time = sum of count x per-SV time; Sample()'s mix + decimation left out (0.2 %% of the work)"
                      % (cores, "; ".join(notes)),
            "agrees_with_oracle": bool(got[1] == want["dop"] and got[2] == want["idx"])}


def pocketfft_legs(args):
    """cpu_baseline_pocketfft of the workloads of this run that go through an FFT, measured with forked worker
    processes -- hence BEFORE this process initialises the GPU (a fork of a process that holds a HIP context is
    not something to rely on).  Inputs are the workloads' own (same seeds)."""
    import numpy as np
    from flydog_sdr_gps_amd import WfParams, acq, prn, sats, synth, wf
    from tests.fixtures import e1b_chips
    wls = ALL_WORKLOADS if args.workload == "all" else [args.workload]
    budget = args.cpu_seconds / 2
    try:
        import scipy.fft  # noqa: F401
    except ImportError as e:
        for wl in wls:
            POCKETFFT[wl] = {"value": None, "note": "scipy.fft unavailable: %s" % e}
        return
    if "acq" in wls:
        codes = [(prn.cacode(sats.SATS[s][1], sats.SATS[s][2]), False) for s in range(NSV)]
        POCKETFFT["acq"] = cpu_acq_pocketfft(synth.config1_iq16(seed=0x5EED0002), codes, NSAMPLES, FFT_LEN, -20, 20, budget)
    if "acq10ms" in wls:
        codes = synth.all_sv_codes(e1b_chips())
        POCKETFFT["acq10ms"] = cpu_acq_pocketfft(synth.config4_iq16(codes, seed=0x5EED0005), codes, acq.NSAMPLES_10MS,
                                                  acq.FFT_LEN_10MS, -128, 127, 2 * budget)
    if "wf14" in wls:
        params = [WfParams.for_zoom(z, 1.0e6 * ch) for ch, z in enumerate(ZOOMS14)]
        maps = [wf.build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, False) for p in params]
        base = np.stack([synth.wf_iq_frame(seed=i) for i in range(32)])
        POCKETFFT["wf14"] = cpu_wf_pocketfft(base, params, (wf.window_functions(), wf.cic_comp_table()), maps, budget)


_NATIVE = {}


def native_oracle():
    """The oracle built for THIS machine's CPU (-O3 -march=native) when gcc is here: the shipped
    libkiwi_oracle.so is built for the x86-64-v3 baseline because it travels between machines.  Selected
    through KIWI_ORACLE_LIBRARY before the oracle is first loaded; any failure keeps the shipped build."""
    if _NATIVE:
        return
    _NATIVE["note"] = "shipped -march=x86-64-v3 build"
    if os.environ.get("KIWI_ORACLE_LIBRARY"):
        _NATIVE["note"] = "KIWI_ORACLE_LIBRARY=%s" % os.environ["KIWI_ORACLE_LIBRARY"]
        return
    import glob
    import shutil
    import tempfile
    if shutil.which("gcc") is None:
        return
    d = tempfile.mkdtemp(prefix="kiwi_oracle_native_")
    out = os.path.join(d, "libkiwi_oracle_native.so")
    srcs = [os.path.join(ROOT, "oracle", "kiwi_oracle.c")] + sorted(glob.glob(os.path.join(ROOT, "oracle", "kiwi_oracle_*.c")))
    cmd = ["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fPIC", "-std=gnu11", "-shared", "-o", out] + srcs + ["-lm", "-lpthread"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        os.environ["KIWI_ORACLE_LIBRARY"] = out
        _NATIVE["note"] = "built here with -O3 -march=native"
    except (OSError, subprocess.SubprocessError):
        pass


def native_oracle_note():
    return _NATIVE.get("note", "shipped -march=x86-64-v3 build")


# ------------------------------------------------------------------------------------------------
# Result checks of the timed workloads: the same objects, right after their timed region, against the oracle
# ------------------------------------------------------------------------------------------------
def check_wf_rows(cases, interp, window_func=None):
    """cases: [(WfParams, int16 frame [8192][2], the GPU's u8 row)] -> rows checked.  The oracle's sample_wf window +
    compute_frame (rx/rx_waterfall.cpp:1049-1066, 1275-1575) on the same frame; bytes must agree under the rule of
    tests/test_wf_gpu.py: identical but for one-LSB flips where the oracle's dB lies within the power bound of an (int)
    edge.  The power bound is north_star's: 1e-5 of the SPECTRUM's maximum -- all 8192 bins, not only the displayed
    ones: a zoomed channel shows 2048 of them, and a DDC output whose strongest line lies outside the plot (or in the
    zeroed DC bins) carries that line's fp32 rounding noise in every displayed bin, as the reference's FFTW-float
    transform does.  (The tests' synthetic frames have their maximum inside the plot, where both bounds coincide.)
    Raises AssertionError on a mismatch: the bench fails."""
    import numpy as np
    if TIMING_EXPERIMENT:
        return 0, 0
    from flydog_sdr_gps_amd import wf
    from oracle import kiwi_oracle as ko
    from tests.test_wf_gpu import DB_EDGE, RTOL, oracle_frame
    ko.lib()
    tables = (wf.window_functions(), wf.cic_comp_table())
    wfun = wf.WINF_HANNING if window_func is None else window_func
    flips = 0
    for p, iq, got in cases:
        w_out, _, w_pwr_out, w_dB = oracle_frame(ko, tables, iq, p, interp, wfun, True, False, False)
        x = (iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)) * tables[0][wfun].astype(np.float64)
        full = np.abs(np.fft.fft(x)) ** 2
        if p.zoom > 1:
            full = full * tables[1].astype(np.float64) ** 2
        dp = RTOL * float(full.max())
        tol = 10.0 * np.log10(1.0 + dp / np.maximum(w_pwr_out.astype(np.float64), 1e-300)) + 1e-4
        diff = got.astype(int) - w_out.astype(int)
        clamped = np.clip(w_dB.astype(np.float64), -200.0, 0.0)
        for i in np.nonzero(diff)[0]:
            assert abs(diff[i]) <= 1 + int(tol[i]), ("row differs from the oracle", p.zoom, int(i), int(got[i]), int(w_out[i]), float(tol[i]))
            assert abs(clamped[i] - np.rint(clamped[i])) < tol[i] + DB_EDGE, \
                ("row differs from the oracle away from an (int) edge", p.zoom, int(i), float(w_dB[i]), float(tol[i]))
            flips += 1
    return len(cases), flips


def adc_rotation(adc_host, dev, nblk, world_rank=0):
    """nblk distinct copies of the ADC block in HBM (block r = the base rotated by 4099 r samples; block 0 = the base):
    the steps walk them in turn so that consecutive pushes never read the same lines -- 9 x 32 MiB is past the
    256 MiB Infinity Cache, as wf14's input is.  -> (list of tensors, host function giving block r)"""
    import numpy as np
    import torch
    base = torch.from_numpy(adc_host).to(dev)
    blocks = [base] + [torch.roll(base, 4099 * r) for r in range(1, nblk)]
    return blocks, (lambda r: adc_host if r == 0 else np.roll(adc_host, 4099 * r))


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

    if args.pmc_child:
        res = pmc_window(ctx, "wf14", step, lambda: torch.cuda.synchronize(dev))
        w.close(); ctx.close()
        return res
    steps = max(5, args.steps)                      # (a quarter of them left the post-synchronize clock ramp, ~1 ms, in the average)
    elapsed, t_enq, spread = timed_steps(dist, step, steps, max(2, args.warmup))
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for _ in range(steps):
        step()
    kernel_ms = ctx.timer_stop() / steps
    k_min, k_med = event_spread(step, steps)
    assert TIMING_EXPERIMENT or int(out.max()) > 100
    # result check at the bench's own shape: 42 frames spread over the launch (every channel three times, the first and
    # the last frame among them), rows as the timed steps left them, against the oracle on the same frames
    picks = sorted(set([0, nfr - 1] + [int(x) for x in np.linspace(0, nfr - 1, 3 * len(zooms)).round()]))
    picks = sorted(set(picks + [f + c for f in (0, nfr // 2 // len(zooms) * len(zooms)) for c in range(len(zooms)) if f + c < nfr]))
    t_chk = time.perf_counter()
    sel = torch.as_tensor(picks, device=dev)
    iq_sel, out_sel = iq[sel].cpu().numpy(), out[sel].cpu().numpy()
    checked, flips = check_wf_rows([(params[f % len(zooms)], iq_sel[i], out_sel[i]) for i, f in enumerate(picks)], wf.WF_CMA)
    log("wf14: %d rows of the timed launch checked against the oracle (%d edge flips) in %.2f s" % (checked, flips, time.perf_counter() - t_chk))
    bytes_frame = 8192 * 4 + 1024          # int16 IQ in + u8 row out; window/maps are L2-resident
    achieved = nfr * bytes_frame / (kernel_ms * 1e-3) / 1e9
    traffic, source, _ = measured_traffic("wf14", nfr)
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
                     "kernel_ms_min": k_min, "kernel_ms_median": k_med,
                     "algorithmic_bytes_per_launch": nfr * bytes_frame},
        "checked": {"rows_vs_oracle": checked, "edge_flips": flips, "channels": len(zooms),
                    "rule": "u8 rows of the timed launch; identical but for 1 LSB at (int) edges inside the 1e-5-of-spectrum-maximum power bound"},
        # the other roof, for scale: nominal flops of the 8192-point transform (5 N log2 N) per frame against
        # the fp32 vector peak; the kernel issues 1239 vector instructions per wave and frame, 600 of them
        # the two 4096-point transforms, so its vector floor (0.25 ms) is above its HBM floor
        "valu": {"nominal_flops_per_launch": nfr * 5 * 8192 * 13,
                 "achieved_TFLOPs": round(nfr * 5 * 8192 * 13 / (kernel_ms * 1e-3) / 1e12, 2), "peak": VALU_PEAK_TFLOPS,
                 "frac": round(nfr * 5 * 8192 * 13 / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4)},
    }
    if "wf14" in CPU_LEGS:
        res["cpu_baseline"] = CPU_LEGS["wf14"]
        res["speedup_vs_cpu_all_cores"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        if "wf14" in POCKETFFT:
            res["cpu_baseline_pocketfft"] = POCKETFFT["wf14"]
            if POCKETFFT["wf14"].get("value"):
                res["speedup_vs_cpu_pocketfft"] = round(res["value"] / POCKETFFT["wf14"]["value"], 1)
    w.close()
    ctx.close()
    return res


def wf_rows_numpy(iq, p, window, cic, fmap, comp_on):
    """compute_frame() for a batch of frames of ONE channel, vectorised (WF_CMA, the bench's mode): window,
    scipy.fft, CIC compensation, power, per-pixel mean over the mapped bins, dB, u8.  iq: int16 [F][8192][2]."""
    import numpy as np
    import scipy.fft as sf
    x = iq[..., 0].astype(np.float32) * window + 1j * (iq[..., 1].astype(np.float32) * window)
    X = sf.fft(x.astype(np.complex64), axis=1)[:, :p.fft_used]
    if comp_on:
        X = X * cic[:p.fft_used]
    pw = X.real * X.real + X.imag * X.imag
    pw[:, :2] = 0.0
    m = np.asarray(fmap[:p.fft_used], np.int64)
    limit = int(np.argmax(m >= 1024)) if (m >= 1024).any() else p.fft_used
    starts = np.flatnonzero(np.r_[True, m[1:limit] != m[:limit - 1]])
    sums = np.add.reduceat(pw[:, :limit], starts, axis=1)
    counts = np.diff(np.r_[starts, limit]).astype(np.float32)
    rows = np.zeros((iq.shape[0], 1024), np.float32)
    rows[:, m[starts]] = sums / counts
    with np.errstate(divide="ignore"):
        dB = 10.0 * np.log10(rows * np.float32(p.fft_scale) + np.float32(1e-30)) + p.fft_offset
    return (np.clip(dB, -200.0, 0.0) - 1.0).astype(np.int32).astype(np.uint8)


def cpu_wf_pocketfft(base, params, tables, maps, budget_s):
    """The waterfall frame work with scipy.fft (pocketfft) and numpy, one chunk of 32 frames per call, one worker
    process per core (pf_rate: before the GPU is touched)."""
    from flydog_sdr_gps_amd import wf
    win, cic = tables[0][wf.WINF_HANNING], tables[1]
    k = [0]

    def chunk():                                     # 32 frames of the next channel in turn
        i = k[0] % len(params)
        k[0] += 1
        p = params[i]
        return wf_rows_numpy(base, p, win, cic, maps[i][0], p.zoom > 1)
    rate, cores = pf_rate(chunk, budget_s)
    return {"value": round(rate * base.shape[0] * 8192 / 1e6, 3), "unit": "Msamples/s", "cores": cores,
            "kind": "independent tuned FFT",
            "sample": "chunks of %d frames (channels in turn), scipy.fft complex64 + numpy (window, power, CIC compensation, "
                      "per-pixel mean, dB, u8), one chunk per call, %d worker processes, %.1f s" % (base.shape[0], cores, budget_s)}


# ------------------------------------------------------------------------------------------------
# Waterfall DDC (configs[2], DDC half)
# ------------------------------------------------------------------------------------------------
_ADC_CACHE = {}


def adc_block(n, seed):
    """n samples of the synthetic 16-bit ADC stream (the same call returns the same array: ddc14 and
    cfg2_chain share theirs)."""
    if (n, seed) not in _ADC_CACHE:
        _ADC_CACHE.clear()
        _ADC_CACHE[(n, seed)] = _adc_block(n, seed)
    return _ADC_CACHE[(n, seed)]


def _adc_block(n, seed):
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n, dtype=np.float64)
    x = rng.normal(0, 10.0, n)
    for f, a in ((0.0123, 3000.0), (0.071, 300.0), (0.2003, 30.0), (0.31, 3.0)):
        x += a * np.cos(2 * np.pi * f * t)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


ADC_BLOCKS = int(os.environ.get("KIWIGPU_BENCH_ADC_BLOCKS", "9"))     # x 32 MiB at the default 2^24 samples: past the 256 MiB Infinity Cache
CHECK_SAMPLES = 1 << 20                                             # ADC samples of a step the oracle re-computes after the timed region


def check_ddc_prefix(d, prm, push_block0, out, adc_host, n):
    """After the timed region: every channel re-armed (CmdSetWFFreq / CmdSetWFDecim: phase 0, CIC registers and
    decimation counter cleared), ONE more push of block 0 through the same object and buffers, and the first
    CHECK_SAMPLES / R output pairs of every channel compared BIT FOR BIT with the oracle's sequential Verilog-structured
    model of verilog/rx/waterfall_1cic.v on the same samples (an output depends only on the samples before it).
    -> output pairs checked"""
    import numpy as np
    import torch
    from oracle import kiwi_oracle as ko
    ko.lib()
    for ch, p in enumerate(prm):
        d.set_wf(ch, p.i_offset, p.decim)
    push_block0()
    torch.cuda.synchronize()
    m = min(CHECK_SAMPLES, n)
    total = 0
    for ch, p in enumerate(prm):
        want, _ = ko.ddc_wf(adc_host[:m], p.i_offset, int(np.log2(p.decim)))
        got = out[ch, :want.shape[0]].cpu().numpy()
        assert TIMING_EXPERIMENT or (want.shape[0] == m // p.decim and np.array_equal(got, want)), \
            "waterfall DDC channel %d (R = %d) differs from the oracle in the bench's own run" % (ch, p.decim)
        total += want.shape[0]
    return total


def run_ddc14(args, dist):
    import numpy as np
    import torch
    from flydog_sdr_gps_amd import Context, Ddc, WfParams
    dev = dist.dev
    ctx = Context(dist.local_rank, torch.cuda.current_stream(dev).cuda_stream)
    zooms = ZOOMS14
    n = 1 << args.log2n
    adc_host = adc_block(n, 0x5EED0003)               # the same stream on every rank (configs[2]/[3])
    adc_blocks, _ = adc_rotation(adc_host, dev, ADC_BLOCKS)
    d = Ddc(ctx, nchan=len(zooms), max_samples=n)
    chans = list(range(len(zooms)))
    prm = []
    for ch, z in enumerate(zooms):
        p = WfParams.for_zoom(z, 1.0e6 * ch, adc_clock=66.6666e6, ui_srate=30.0e6)
        prm.append(p)
        d.set_wf(ch, p.i_offset, p.decim)
    stride = n + 1
    out = torch.zeros((len(zooms), stride, 2), dtype=torch.int16, device=dev)
    kstep = [0]
    # the streaming form: a push's output stage (bypass, run-total prefix, combs) on the object's own stream, the next
    # push's run passes already running beside it (kg_ddc_wf_set_deferred; KIWIGPU_BENCH_DDC_DEFERRED=0: everything in line)
    deferred = os.environ.get("KIWIGPU_BENCH_DDC_DEFERRED", "0") != "0"
    d.set_deferred(deferred)

    def step():
        a = adc_blocks[kstep[0] % ADC_BLOCKS]          # a different 32 MiB of ADC samples every step
        kstep[0] += 1
        d.push_dev(a.data_ptr(), n, chans, out.data_ptr(), stride)

    def drain():                                       # the context's (= torch's current) stream waits for the last output stage
        if deferred:
            d.join()

    if args.pmc_child:
        res = pmc_window(ctx, "ddc14", step, lambda: torch.cuda.synchronize(dev))
        d.close(); ctx.close()
        return res
    steps = max(5, args.steps)                      # (a quarter of them left the post-synchronize clock ramp, ~1 ms, in the average)
    elapsed, t_enq, spread = timed_steps(dist, step, steps, max(2, args.warmup))      # (synchronize = every stream of the device)
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for _ in range(steps):
        step()
    drain()
    gpu_ms = ctx.timer_stop() / steps
    k_min, k_med = event_spread(step, steps)
    drain()
    t_chk = time.perf_counter()
    checked = check_ddc_prefix(d, prm, lambda: (d.push_dev(adc_blocks[0].data_ptr(), n, chans, out.data_ptr(), stride), drain()), out, adc_host, n)
    log("ddc14: %d output pairs of 14 channels bit-exact against the oracle in %.2f s" % (checked, time.perf_counter() - t_chk))
    traffic, source, top = measured_traffic("ddc14", n)
    # Integer work per ADC sample and channel, counted in 32-bit operations on the algorithm (not on
    # the kernels): NCO phase add (48 bit: 2) + 2 table reads + 2 multiplies + 2 roundings (4) = 10;
    # five integrators on I and Q, the first four 89 bits wide (3 words) and the fifth 28 (1): 2 x 13 = 26;
    # the combs and the output rounding run at 1/R and are left out.
    # An R = 1 channel is the bypass (cic_prune_var.v:289-297): NCO + mixer only, 10 operations.
    ops_sample_chan = 36
    ops_step = n * sum(10 if p.decim == 1 else ops_sample_chan for p in prm)
    tops = ops_step / (gpu_ms * 1e-3) / 1e12
    out_bytes = sum((n // p.decim) * 4 for p in prm)
    res = {
        "metric": "ADC Msamples/s ingested by the 14-channel waterfall DDC (NCO mix + pruned 5-stage CIC)",
        "value": round(float(n) * steps * dist.world / elapsed / 1e6, 1), "unit": "Msamples/s",
        "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 5), "step_ms_spread": spread,
        "dtype": "int128/int64/int32",
        "config": {"workload": "BASELINE configs[2] DDC: %d ADC samples per step, 14 channels, zooms %s; the steps walk %d distinct "
                               "ADC blocks (%.0f MiB, past the 256 MiB Infinity Cache)" % (n, zooms, ADC_BLOCKS, ADC_BLOCKS * n * 2 / 2 ** 20),
                   "adc_samples_per_step": n, "adc_blocks": ADC_BLOCKS},
        "checked": {"ddc_pairs_bit_exact_vs_oracle": checked,
                    "rule": "channels re-armed after the timed region, one push of block 0, first 2^20 / R pairs of all 14 channels bit for bit"},
        "deferred_output_stage": deferred,
        "x_realtime_at_66.6MSps": round(n / (gpu_ms * 1e-3) / 66.6666e6, 1),
        # integer-VALU bound by arithmetic intensity (SURVEY 8d's caveat): 2 bytes in per ADC sample for
        # 14 x 36 integer operations
        "roofline": {"bound": "valu", "kernel": "ddc_wf passes A + scan + B + comb (whole step)",
                     "achieved": round(tops, 3), "peak": INT_PEAK_TOPS, "unit": "Tiop/s",
                     "frac": round(tops / INT_PEAK_TOPS, 4), "traffic": traffic, "traffic_source": source,
                     "traffic_top_kernels": top,
                     "kernel_ms": round(gpu_ms, 5), "kernel_ms_min": k_min, "kernel_ms_median": k_med,
                     "int_ops_per_sample_per_channel": ops_sample_chan, "int_ops_per_sample_bypass_channel": 10,
                     "int_ops_per_sample_all_channels": ops_step // n},
        "hbm": {"algorithmic_bytes_per_step": 2 * n + out_bytes,
                "algorithmic_GBps": round((2 * n + out_bytes) / (gpu_ms * 1e-3) / 1e9, 1),
                "measured_GBps": None if traffic is None else round(traffic / (gpu_ms * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS},
    }
    if "ddc14" in CPU_LEGS:
        res["cpu_baseline"] = CPU_LEGS["ddc14"]
        res["speedup_vs_cpu_all_cores"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    d.close()
    ctx.close()
    return res


# ------------------------------------------------------------------------------------------------
# configs[2] end to end: ADC stream -> 14 DDC channels -> the frames those complete -> rows
# ------------------------------------------------------------------------------------------------
def run_cfg2_chain(args, dist):
    """One step = 2^log2n ADC samples through kg_ddc_wf_push_dev on the 14 zooms, then every 8192-sample frame the
    channels have completed (read in place from the DDC's per-channel rows, kg_wf_frames_at_dev) through
    kg_wf_frames: what rx/rx_waterfall.cpp:930-939 loops over (sample_wf + compute_frame), at 100 % duty.  The
    zoom-14 channel (R = 8192) yields 2048 samples per 2^24-sample step: its row fills over four steps and its
    frame is taken on the fourth."""
    import numpy as np
    import torch
    from flydog_sdr_gps_amd import Context, Ddc, Waterfall, WfParams, wf
    dev = dist.dev
    main = torch.cuda.current_stream(dev)
    ctx = Context(dist.local_rank, main.cuda_stream)
    # The streaming form (KIWIGPU_BENCH_DDC_DEFERRED=0: everything in line on one stream, round 3's shape): the DDC's output
    # stage runs on the object's own stream (kg_ddc_wf_set_deferred) and the frames of step k on a second stream of the
    # caller, behind that stage (kg_ddc_wf_join) -- both under the run passes of step k + 1; the next step's writers of the
    # rows wait for the frames that still read them (kg_ddc_wf_tail_after).
    pipelined = os.environ.get("KIWIGPU_BENCH_DDC_DEFERRED", "0") != "0"
    s2 = torch.cuda.Stream(device=dev) if pipelined else None
    ctx_fr = Context(dist.local_rank, s2.cuda_stream) if pipelined else ctx
    ev_fr = [torch.cuda.Event(), torch.cuda.Event()]
    zooms = ZOOMS14
    n = 1 << args.log2n
    adc_host = adc_block(n, 0x5EED0003)
    adc_blocks, adc_host_of = adc_rotation(adc_host, dev, ADC_BLOCKS)
    C14 = len(zooms)
    d = Ddc(ctx, nchan=C14, max_samples=n)
    d.set_deferred(pipelined)
    w = Waterfall(ctx_fr, nchan=C14)
    tables = (wf.window_functions(), wf.cic_comp_table())
    w.set_tables(*tables)
    chans = list(range(C14))
    prm = []
    for ch, z in enumerate(zooms):
        p = WfParams.for_zoom(z, 1.0e6 * ch, adc_clock=66.6666e6, ui_srate=30.0e6)
        prm.append(p)
        d.set_wf(ch, p.i_offset, p.decim)
        w.set_channel(ch, p, interp=wf.WF_CMA, cic_comp=True)
    nout = [n // p.decim for p in prm]
    assert all(n % p.decim == 0 for p in prm)
    frac = sorted(set(x for x in nout if x < 8192))
    assert len(frac) <= 1 and all(8192 % x == 0 for x in frac), "one fractional class of channels per step"
    cyc = 8192 // frac[0] if frac else 1                   # steps per frame of the slowest channel
    q = frac[0] if frac else 0
    stride = n + 8192                                      # pairs between channel rows: room for the (k % cyc) q shift
    wf_iq = torch.zeros((C14, stride, 2), dtype=torch.int16, device=dev)
    # frame tables of the cyc kinds of step: in step k every channel's new samples start q (k % cyc) pairs into its row
    tabs = []
    for k in range(cyc):
        co, fo = [], []
        for ch in range(C14):
            if nout[ch] >= 8192:
                for f in range(nout[ch] // 8192):
                    co.append(ch); fo.append(ch * stride + q * k + 8192 * f)
            elif k == cyc - 1:
                co.append(ch); fo.append(ch * stride)
        tabs.append((np.asarray(co, np.int32), np.asarray(fo, np.uint64)))
    rows = torch.zeros((max(len(t[0]) for t in tabs), 1024), dtype=torch.uint8, device=dev)
    kstep = [0]
    base_ptr = wf_iq.data_ptr()

    def ddc_part(k, blk=0):
        d.push_dev(adc_blocks[blk].data_ptr(), n, chans, base_ptr + 4 * q * k, stride)

    def frames_part(k):
        w.frames_dev(tabs[k][0], base_ptr, rows.data_ptr(), frame_off=tabs[k][1], iq_len=C14 * stride)

    fr_pending = [False]

    def step():
        k = kstep[0] % cyc
        blk = kstep[0] % ADC_BLOCKS                        # a different 32 MiB of ADC samples every step
        if pipelined and fr_pending[0]:
            d.tail_after(ev_fr[(kstep[0] + 1) & 1].cuda_event)     # this push's writers of the rows: after the frames of the step before
        ddc_part(k, blk)
        if pipelined:
            d.join(s2.cuda_stream)                         # the frames' stream waits for the rows
        frames_part(k)
        if pipelined:
            ev_fr[kstep[0] & 1].record(s2)
            fr_pending[0] = True
        kstep[0] += 1

    def drain():                                           # torch's current stream behind everything the steps enqueued
        if pipelined:
            d.join()
            main.wait_stream(s2)

    if args.pmc_child:
        res = pmc_window(ctx, "cfg2_chain", step, lambda: torch.cuda.synchronize(dev))
        d.close(); w.close(); ctx.close()
        if pipelined:
            ctx_fr.close()
        return res
    steps = max(2 * cyc, (args.steps + cyc - 1) // cyc * cyc)          # whole cycles
    elapsed, t_enq, spread = timed_steps(dist, step, steps, max(cyc, args.warmup // cyc * cyc))
    log("cfg2_chain: host enqueue %.4f ms per step, wall %.4f ms per step" % (t_enq / steps * 1e3, elapsed / steps * 1e3))
    if os.environ.get("KIWIGPU_BENCH_HOST_SPLIT") == "1":           # where the host spends a step (a diagnostic, not in the line)
        for rep in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            t1 = time.perf_counter()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            log("cfg2_chain repeat %d: host loop %.1f us, wall %.1f us per step" % (rep, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
        torch.cuda.synchronize(dev)
        ta = tb = 0.0
        t0 = time.perf_counter()
        for i in range(steps):
            a = time.perf_counter(); ddc_part(i % cyc); b = time.perf_counter(); frames_part(i % cyc); c = time.perf_counter()
            ta += b - a; tb += c - b
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        log("cfg2_chain host split: push_dev %.1f us, frames_dev %.1f us per step; loop %.1f us, with the drain %.1f us per step"
            % (ta / steps * 1e6, tb / steps * 1e6, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for _ in range(steps):
        step()
    drain()
    gpu_ms = ctx.timer_stop() / steps
    k_min, k_med = event_spread(step, steps)
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for i in range(steps):
        ddc_part(i % cyc, i % ADC_BLOCKS)
    drain()
    ddc_ms = ctx.timer_stop() / steps
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for i in range(steps):
        frames_part(i % cyc)
    drain()
    frames_ms = ctx.timer_stop() / steps
    torch.cuda.synchronize(dev)
    fr_pending[0] = False
    assert int(rows.max()) > 100
    # ---- result check at the bench's own shape (after the timed region, the same objects and buffers): channels
    # re-armed, one cycle of steps over ADC blocks 0 .. cyc-1; per channel the DDC output that holds its first frames
    # bit for bit against the oracle, and the u8 rows of up to six of those frames under the tests' rule
    from oracle import kiwi_oracle as ko
    ko.lib()
    t_chk = time.perf_counter()
    for ch, p in enumerate(prm):
        d.set_wf(ch, p.i_offset, p.decim)
    kstep[0] = 0
    snap = {}
    for k in range(cyc):
        step()
        torch.cuda.synchronize(dev)
        if k in (0, cyc - 1):
            snap[k] = (wf_iq.cpu().numpy(), rows.cpu().numpy().copy())
    adc_cat = np.concatenate([adc_host_of(k % ADC_BLOCKS) for k in range(cyc)]) if cyc > 1 else adc_host
    pairs_checked, row_cases = 0, []
    for ch, p in enumerate(prm):
        R = p.decim
        slow = nout[ch] < 8192                                    # its frame is taken on the last step of the cycle
        need = min(max(CHECK_SAMPLES, 8192 * R), adc_cat.size if slow else n)
        want, _ = ko.ddc_wf(adc_cat[:need], p.i_offset, int(np.log2(R)))
        g_iq, g_rows = snap[cyc - 1 if slow else 0]
        got = g_iq[ch, :want.shape[0]]
        assert want.shape[0] == need // R and np.array_equal(got, want), "chain: DDC channel %d (R = %d) differs from the oracle" % (ch, R)
        pairs_checked += want.shape[0]
        k_tab = cyc - 1 if slow else 0
        mine = [i for i, c in enumerate(tabs[k_tab][0]) if c == ch]       # this channel's frames in that step's table, in order
        nfull = want.shape[0] // 8192
        for f in sorted(set(list(range(min(3, nfull))) + list(range(max(0, nfull - 3), nfull)))):
            row_cases.append((p, want[8192 * f:8192 * (f + 1)], g_rows[mine[f]]))
    rows_checked, flips = check_wf_rows(row_cases, wf.WF_CMA)
    assert len(set(id(c[0]) for c in row_cases)) == C14, "every channel's rows must be among the checked ones"
    log("cfg2_chain: %d DDC pairs bit-exact and %d rows of all 14 channels checked against the oracle in %.2f s"
        % (pairs_checked, rows_checked, time.perf_counter() - t_chk))
    frames_per_step = sum(len(t[0]) for t in tabs) / cyc
    ops_sample_chan = 36                                   # run_ddc14's count: NCO + mixer + five integrators per sample and channel; R = 1 channels: 10
    tops = n * sum(10 if p.decim == 1 else ops_sample_chan for p in prm) / (gpu_ms * 1e-3) / 1e12
    out_bytes = sum(4 * x for x in nout)
    alg_bytes = 2 * n + out_bytes + frames_per_step * (8192 * 4 + 1024)   # ADC in, DDC rows out, frames back in, u8 rows out
    traffic, source, top = measured_traffic("cfg2_chain", n)
    res = {
        "metric": "ADC Msamples/s ingested by the 14-channel waterfall end to end (DDC -> frames -> rows)",
        "value": round(float(n) * steps * dist.world / elapsed / 1e6, 1), "unit": "Msamples/s",
        "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 5), "step_ms_spread": spread,
        "dtype": "int128/int64/int32 + f32",
        "config": {"workload": "BASELINE configs[2] end to end: %d ADC samples per step -> 14 DDC channels (zooms %s) -> "
                               "the %.2f frames per step those complete -> window + 8192-pt FFT + power + pixels + dB + u8 rows"
                               % (n, zooms, frames_per_step),
                   "adc_samples_per_step": n, "frames_per_step": frames_per_step, "adc_blocks": ADC_BLOCKS},
        "checked": {"ddc_pairs_bit_exact_vs_oracle": pairs_checked, "rows_vs_oracle": rows_checked, "edge_flips": flips, "channels": C14,
                    "rule": "after the timed region: channels re-armed, one cycle of steps; DDC prefix bit for bit, u8 rows under tests/test_wf_gpu.py's rule"},
        "x_realtime_at_66.6MSps": round(n / (gpu_ms * 1e-3) / 66.6666e6, 1),
        "frames_per_s": round(frames_per_step * steps * dist.world / elapsed, 1),
        # the step is the DDC's integer work (ddc14's roof) plus a quarter as much frame time; the numerator counts the
        # DDC's operations only, so `frac` is a lower bound on the vector units' use over the step
        "roofline": {"bound": "valu", "kernel": "whole step: ddc_wf passes + wf_frame_kernel",
                     "achieved": round(tops, 3), "peak": INT_PEAK_TOPS, "unit": "Tiop/s", "frac": round(tops / INT_PEAK_TOPS, 4),
                     "traffic": traffic, "traffic_source": source, "traffic_top_kernels": top,
                     "kernel_ms": round(gpu_ms, 5), "kernel_ms_min": k_min, "kernel_ms_median": k_med,
                     "ddc_ms_alone": round(ddc_ms, 5), "frames_ms_alone": round(frames_ms, 5),
                     "int_ops_per_sample_per_channel": ops_sample_chan,
                     "frames_nominal_TFLOPs": round(frames_per_step * 5 * 8192 * 13 / (gpu_ms * 1e-3) / 1e12, 3)},
        "hbm": {"algorithmic_bytes_per_step": int(alg_bytes),
                "algorithmic_GBps": round(alg_bytes / (gpu_ms * 1e-3) / 1e9, 1),
                "measured_GBps": None if traffic is None else round(traffic / (gpu_ms * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS},
    }
    res["pipelined"] = pipelined
    if "cfg2_chain" in CPU_LEGS:
        res["cpu_baseline"] = CPU_LEGS["cfg2_chain"]
        res["speedup_vs_cpu_all_cores"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    d.close()
    w.close()
    ctx.close()
    if pipelined:
        ctx_fr.close()
    return res


# ------------------------------------------------------------------------------------------------
# configs[3]: virtual receivers -- the C ABI's receiver bank (kg_rxbank, flydog_sdr_gps_amd/rxbank.py)
# ------------------------------------------------------------------------------------------------
RX_MIX = {"receivers": "survey", "receivers_light": "light"}


def run_receivers(args, dist, wl="receivers"):
    """BASELINE configs[3]: a bank of virtual receivers per GPU (weak scaling over ranks), stepped with ONE C call per step
    (kg_rxbank_step).  `receivers` = SURVEY.md 8(d)'s mix (f_k = 100 kHz + k 29 kHz, zoom 8 + (k mod 4): zooms 8..10 the
    reference's non-overlapped frame, zoom 11 its overlapped / continuous sampler); `receivers_light` = rounds 2-4's mix
    (zooms 1..10, every frame one-shot), kept as a second named workload."""
    import numpy as np
    import torch
    from flydog_sdr_gps_amd.rxbank import ADC_CLOCK, MIXES, RxBank
    dev = dist.dev
    NR, n = args.receivers, 1 << args.log2n
    mix_name = RX_MIX[wl]
    mix = MIXES[mix_name](NR, dist.rank * NR, n)                 # this rank's slice of the 1024
    adc_host = adc_block(n, 0x5EED0004)                          # every GPU sees the SAME stream (configs[3])
    adc = torch.from_numpy(adc_host).to(dev)
    d_adc = adc.data_ptr()
    torch.cuda.synchronize(dev)

    def make_bank():
        b = RxBank(NR, n, device=dist.local_rank)
        b.configure(mix)
        return b
    bank = make_bank()

    def step():
        bank.step_fast(d_adc)

    def sync():
        bank.sync()
    if args.pmc_child:
        res = pmc_window(bank.ctx, wl, step, sync)
        bank.close()
        return res
    preroll(step, args.warmup, dist)
    sync()
    dist.barrier()
    bank.host_profile()                              # (reset: the next call reports the timed loop alone)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_loop = time.perf_counter() - t0
    sync()                                           # every stream of the bank
    log("%s: host phases of the timed loop: %s" % (wl, bank.host_profile()))
    torch.cuda.synchronize(dev)
    local = time.perf_counter() - t0
    dist.barrier()
    elapsed = dist.max_over_ranks(local)
    step_s = elapsed / args.steps
    # The host's share of a step.  In the loop above the host runs BANK_SLOTS steps ahead and then waits for the GPU (the
    # step-table slot it wants is still in use): its time there is the GPU's.  What a step costs the host is measured with
    # the GPU idle: enqueue one step, time it, drain.
    t_enq = []
    for _ in range(24):
        t1 = time.perf_counter()
        step()
        t_enq.append(time.perf_counter() - t1)
        sync()
    t_enq.sort()
    enq_us = t_enq[len(t_enq) // 2] * 1e6
    log("%s: host enqueue %.1f us per step (median of 24, idle GPU; min %.1f), loop %.4f ms, wall %.4f ms per step"
        % (wl, enq_us, t_enq[0] * 1e6, t_loop / args.steps * 1e3, local / args.steps * 1e3))
    # cadence: chunks of three steps (one audio cycle: 402.7 records a step, a 512-sample block on two steps of three)
    chunks = []
    for _ in range(max(8, args.steps // 3)):
        sync()
        t1 = time.perf_counter()
        for _ in range(3):
            step()
        sync()
        chunks.append((time.perf_counter() - t1) / 3 * 1e3)
    chunks.sort()
    # SURVEY 8(d)'s "ingested" variant: every step's ADC block starts in (pinned) HOST memory and is copied into the next of
    # NINE device buffers on the caller's own stream -- adc_ready_event orders the step behind its block's copy; the copy needs
    # no ordering behind the buffer's last reader, step k - 9: kg_rxbank_step returns only when step k - 8 has completed
    # (kiwigpu.h, KG_RXBANK_SLOTS).  No host synchronisation inside the loop.  Never `value`.
    pcie_ms = None
    if dist.world == 1:
        host_adc = torch.from_numpy(adc_host).pin_memory()
        NB = 9
        ring = [torch.empty_like(adc) for _ in range(NB)]
        up = torch.cuda.Stream(device=dev)
        evs = [torch.cuda.Event() for _ in range(NB)]
        nst = max(2 * NB, args.steps)

        def ring_step(k):
            with torch.cuda.stream(up):
                ring[k % NB].copy_(host_adc, non_blocking=True)
                evs[k % NB].record(up)
            bank.step(ring[k % NB].data_ptr(), adc_ready_event=evs[k % NB].cuda_event)
        k0 = 0                                       # untimed, 0.3 s: a process's first host-to-device copies are slow (measured 1.40 ms
        t1 = time.perf_counter()                     # per step over the first ~100 ms of streaming, 1.25 from then on: the link / DMA
        while time.perf_counter() - t1 < 0.3:        # engine leaving its idle state, not the bank)
            for _ in range(NB):
                ring_step(k0); k0 += 1
        sync(); torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for k in range(k0, k0 + nst):
            ring_step(k)
        sync(); torch.cuda.synchronize(dev)
        pcie_ms = (time.perf_counter() - t1) / nst * 1e3
        log("%s: with every block copied from pinned host memory (a ring of nine device buffers, no host synchronisation): %.4f ms per step" % (wl, pcie_ms))
    info = bank.step(d_adc)
    sync()
    rows = bank.fetch("rows", range(info.nframes))
    pay = bank.fetch("pay", range(NR))
    assert info.nframes == NR and int(rows.max()) > 100 and int(np.abs(pay.astype(np.int32)).sum()) > 0
    frames_per_step = info.nframes
    bank.close()
    if os.environ.get("KIWIGPU_BENCH_RX_REPEAT"):    # a diagnostic: is the step time a property of the bank INSTANCE (its streams)?
        for rep in range(int(os.environ["KIWIGPU_BENCH_RX_REPEAT"])):
            bk = make_bank()
            for _ in range(12):
                bk.step_fast(d_adc)
            bk.sync()
            t1 = time.perf_counter()
            for _ in range(30):
                bk.step_fast(d_adc)
            bk.sync()
            log("%s: bank instance %d: %.4f ms per step" % (wl, rep, (time.perf_counter() - t1) / 30 * 1e3))
            bk.close()
    # Integer work per ADC sample and receiver, counted on the algorithm: the waterfall DDC's 36 (run_ddc14; R = 1 bypass
    # channels: the NCO / mixer's 10) + the audio DDC's NCO / mixer (10) and rx1's three integrators on I and Q (55, 55 and
    # 26 bits: 2 + 2 + 1 words, x 2 = 10); everything behind the first decimation runs at <= 1 / 1543 of the rate.  A one-shot
    # waterfall channel consumes 8192 R samples of the block, an overlapped one all of it.
    wf_ops = sum((10.0 if p.decim == 1 else 36.0) * (n if ov else min(n, 8192 * p.decim)) for p, ov, _ in mix)
    wf_samples = sum(n if ov else min(n, 8192 * p.decim) for p, ov, _ in mix)
    ops = round((wf_ops + 20.0 * n * NR) / (n * NR), 2)       # per receiver and ADC sample of the block
    tops = (wf_ops + 20.0 * n * NR) / step_s / 1e12
    traffic, source, top = measured_traffic(wl, NR)
    # result check at the bench's own shape: a fresh bank of the same receivers (the timed one's state is 100s of steps old),
    # three steps, every stage of a sample of the receivers (every zoom at least once) against the oracle
    t_chk = time.perf_counter()
    from tests.rxbank_check import check_bank
    pick = sorted(set([0, NR - 1] + [int(x) for x in np.linspace(0, NR - 1, min(NR, 12)).round()]))
    if TIMING_EXPERIMENT:
        checked = {"skipped": "timing experiment"}
    else:
        bank2 = make_bank()
        try:
            checked = check_bank(bank2, lambda k: adc_host, lambda k: d_adc, pick)
        finally:
            bank2.close()
    checked["rule"] = "fresh bank, 3 steps, every stage of %d of the %d receivers vs the oracle (tests/test_receivers_gpu.py checks all)" % (len(pick), NR)
    log("%s: %s in %.2f s" % (wl, checked, time.perf_counter() - t_chk))
    # SURVEY 8(d) bytes of a step: the ADC block once, every waterfall channel's DDC outputs, its frame back in and its u8
    # row out (33 792 B), the audio chain's rx_iq_t records out and in + unpacked samples (6 + 6 + 8 B), a CFastFIR block
    # (16 384 B) every 512 records
    nrec_step = n / 10416.0
    alg_bytes = 2 * n + sum(4 * ((n // p.decim) if ov else 8192) for p, ov, _ in mix) + NR * (8192 * 4 + 1024) + NR * nrec_step * (20 + 16384 / 512.0)
    zooms = sorted(set(p.zoom for p, _, _ in mix))
    n_ov = sum(1 for _, ov, _ in mix if ov)
    world = dist.world
    res = {
        "metric": "receiver x ADC Msamples/s ingested (waterfall + audio chain per virtual receiver)",
        "value": round(n * NR * world / step_s / 1e6, 1), "unit": "Msamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4),
        "step_ms_spread": {"min": round(chunks[0], 5), "median": round(chunks[len(chunks) // 2], 5), "max": round(chunks[-1], 5),
                           "how": "a second, untimed pass: chunks of three steps (one audio cycle) enqueued and drained, wall time / 3"},
        "host_enqueue_us_per_step": round(enq_us, 1),
        "ingest_pcie_ms_per_step": None if pcie_ms is None else round(pcie_ms, 4),
        "ingest_pcie_note": "each step's %d-byte ADC block copied from pinned host memory into the next of nine device buffers on the "
                            "caller's stream (adc_ready_event), no host synchronisation in the loop" % (2 * n),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int128/int64/f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]%s: %d virtual receivers per GPU x %d GPU(s), one %d-sample 16-bit ADC "
                               "block @66.67 MS/s (the same stream on every GPU) resident in HBM per step; receiver mix '%s': %s; "
                               "per receiver a waterfall channel (one frame per step: %d one-shot, rx/rx_waterfall.cpp:1005-1041, "
                               "%d overlapped / continuous sampler, :967-991) and an SSB audio channel (continuous); the whole "
                               "step is ONE C-ABI call (kg_rxbank_step)"
                               % (" on rounds 2-4's lighter receiver set" if wl != "receivers" else "", NR, world, n, mix_name,
                                  "f_k = 100 kHz + k 29 kHz, zoom 8 + (k mod 4) (SURVEY.md 8d)" if mix_name == "survey"
                                  else "zoom 1 + (k mod 10)", NR - n_ov, n_ov),
                   "receivers_per_gpu": NR, "adc_samples_per_step": n, "zooms": zooms, "overlapped": n_ov,
                   "parallelism": "receivers sharded over ranks, no data-path collective"},
        "adc_ms_per_step": round(n / ADC_CLOCK * 1e3, 3),
        "x_realtime_all_receivers": round(n / ADC_CLOCK / step_s, 2),
        "waterfall_frames_per_s": round(frames_per_step * world / step_s, 1),
        "audio_blocks_per_s": round(NR * world * (n / 10416.0 / 512.0) / step_s, 1),
        "roofline": {"bound": "valu", "kernel": "whole step (both DDCs' run passes dominate; three streams: the two chains and one for both tails)",
                     "achieved": round(tops, 3), "peak": INT_PEAK_TOPS, "unit": "Tiop/s", "frac": round(tops / INT_PEAK_TOPS, 4),
                     "traffic": traffic, "traffic_source": source, "traffic_top_kernels": top,
                     "kernel_ms": round(step_s * 1e3, 5), "kernel_ms_min": round(chunks[0], 5),
                     "kernel_ms_median": round(chunks[len(chunks) // 2], 5), "int_ops_per_sample_per_receiver": ops,
                     "waterfall_ddc_samples_per_step": wf_samples, "audio_ddc_samples_per_step": n * NR},
        "hbm": {"algorithmic_bytes_per_step": int(alg_bytes),
                "algorithmic_GBps": round(alg_bytes / step_s / 1e9, 1),
                "measured_GBps": None if traffic is None else round(traffic / step_s / 1e9, 1), "peak": HBM_PEAK_GBS},
        "checked": checked,
    }
    if wl in CPU_LEGS:
        res["cpu_baseline"] = CPU_LEGS[wl]
        res["speedup_vs_cpu_all_cores"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    return res


def cpu_receivers(NR, n, budget_s, mix_name="survey"):
    """The oracle on whole receivers of the bench's bank: one unit = one receiver's complete step (waterfall DDC -> frame ->
    row -> wf_pkt_t; audio DDC -> rx_iq_t -> unpack -> CFastFIR -> CAgc mono16 -> ADPCM) on the bank's ADC block."""
    import numpy as np
    from flydog_sdr_gps_amd import wf
    from flydog_sdr_gps_amd.ddc import RX_DECIM
    from flydog_sdr_gps_amd.rxbank import ADC_CLOCK, MIXES
    from oracle import kiwi_oracle as ko
    ko.lib()
    adc = adc_block(n, 0x5EED0004)
    mix = MIXES[mix_name](NR, 0, n)
    fs = ADC_CLOCK / RX_DECIM
    tables = (wf.window_functions(), wf.cic_comp_table())
    coef = ko.fir_design(300.0, 2700.0, 0.0, fs, prec=0)[1]
    k = [os.getpid() % NR]                            # (workers are forked: each walks the receivers from its own start)

    def unit():
        ch = k[0] % NR
        k[0] += 1
        p, ov, rx_inc = mix[ch]
        # one-shot: the 8192 R samples that fill the sampler; overlapped: the whole block through the continuous sampler
        # (its n / R outputs are half of the frame the step then takes: repeated here to make one)
        iq, _ = ko.ddc_wf(adc if ov else adc[:8192 * p.decim], p.i_offset, int(np.log2(p.decim)))
        if ov:
            iq = np.concatenate([iq] * (8192 // iq.shape[0]))
        fmap, drop = wf.build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, False)
        scale = np.full(1024, p.fft_scale, np.float32)
        samps = ko.wf_window_iq(iq[:8192], tables[0][wf.WINF_HANNING])
        row = ko.wf_compute_frame(samps, p.zoom, wf.WINF_HANNING, wf.WF_MAX, True, bool(ov), p.fft_used, p.plot_width,
                                  p.plot_width_clamped, fmap, drop, scale, (scale / np.float32(2)).astype(np.float32),
                                  p.fft_offset, tables[1], prec=0)[0]
        ko.wf_packet(row, int(p.start), p.zoom, 0, True)
        raw, _ = ko.ddc_rx(adc, rx_inc)
        nrec = raw.size // 6
        x = ko.dpump_unpack(raw, nrec, 1)[0]
        y, _ = ko.fir_process(ko.fir_new_state(), coef, x, prec=0)
        if y.size:
            agc = ko.Agc()
            agc.set_parameters(True, False, -100, 50, 6, 1000, fs)
            ko.adpcm_encode_i16(agc.process_s16(y[:512]))
    reps, el, cores, t1 = cpu_threads(unit, budget_s)
    return {"value": round(reps * n / el / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d x (one virtual receiver of the '%s' mix over the step's %d ADC samples: both chains, every stage; the DDCs are the "
                      "oracle's sequential Verilog-structured models -- in the reference they are FPGA fabric), %d worker processes, %.1f s"
                      % (reps, mix_name, n, cores, el),
            "single_thread_value": round(n / t1 / 1e6, 4)}


def cpu_wf14(budget_s):
    import numpy as np
    from flydog_sdr_gps_amd import WfParams, synth, wf
    from oracle import kiwi_oracle as ko
    ko.lib()
    params = [WfParams.for_zoom(z, 1.0e6 * ch) for ch, z in enumerate(ZOOMS14)]
    tables = (wf.window_functions(), wf.cic_comp_table())
    base = np.stack([synth.wf_iq_frame(seed=i) for i in range(32)])
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
    reps_done, el, cores, t1 = cpu_threads(unit, budget_s)
    return {"value": round(reps_done * 14 * 8192 / el / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d x (one frame of each of the 14 channels: sample_wf window + compute_frame), oracle fp32 "
                      "FFT, %d worker processes, %.1f s" % (reps_done, cores, el),
            "single_thread_value": round(14 * 8192 / t1 / 1e6, 4)}


def ddc14_params(adc_clock=66.6666e6, ui_srate=30.0e6):
    from flydog_sdr_gps_amd import WfParams
    return [WfParams.for_zoom(z, 1.0e6 * ch, adc_clock=adc_clock, ui_srate=ui_srate) for ch, z in enumerate(ZOOMS14)]


def cpu_ddc14(n, budget_s):
    import numpy as np
    from oracle import kiwi_oracle as ko
    ko.lib()
    prm = ddc14_params()
    m = 1 << 18                                      # a bounded piece of the same stream
    piece = adc_block(n, 0x5EED0003)[:m]
    log2r = [int(np.log2(p.decim)) for p in prm]

    def unit():                                      # all 14 channels over the piece
        for ch, p in enumerate(prm):
            ko.ddc_wf(piece, p.i_offset, log2r[ch])
    reps_done, el, cores, t1 = cpu_threads(unit, budget_s)
    return {"value": round(reps_done * m / el / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d x (2^18 ADC samples through the 14 channels), the oracle's sequential Verilog-structured "
                      "model (the reference has no CPU DDC: it is FPGA fabric), %d worker processes, %.1f s" % (reps_done, cores, el),
            "single_thread_value": round(m / t1 / 1e6, 4)}


def cpu_cfg2_chain(n, budget_s):
    import numpy as np
    from flydog_sdr_gps_amd import wf
    from oracle import kiwi_oracle as ko
    ko.lib()
    prm = ddc14_params()
    tables = (wf.window_functions(), wf.cic_comp_table())
    m = 1 << 20                                      # a bounded piece of the same stream
    piece = adc_block(n, 0x5EED0003)[:m]
    log2r = [int(np.log2(p.decim)) for p in prm]
    maps = [wf.build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, False) for p in prm]
    scales = [np.full(1024, p.fft_scale, np.float32) for p in prm]
    done_frames = [0]

    def unit():                                      # 2^20 ADC samples: 14 DDC channels, then every complete frame
        nf = 0
        for ch, p in enumerate(prm):
            iq, _ = ko.ddc_wf(piece, p.i_offset, log2r[ch])
            for f in range(iq.shape[0] // 8192):
                samps = ko.wf_window_iq(iq[8192 * f:8192 * (f + 1)], tables[0][wf.WINF_HANNING])
                ko.wf_compute_frame(samps, p.zoom, wf.WINF_HANNING, wf.WF_CMA, True, False, p.fft_used,
                                    p.plot_width, p.plot_width_clamped, maps[ch][0], maps[ch][1], scales[ch],
                                    (scales[ch] / np.float32(2)).astype(np.float32), p.fft_offset, tables[1], prec=0)
                nf += 1
        done_frames[0] = nf
    reps_done, el, cores, t1 = cpu_threads(unit, budget_s)
    return {"value": round(reps_done * m / el / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d x (2^20 ADC samples through the oracle's 14 DDC channels, then sample_wf window + compute_frame of the "
                      "%d complete frames they hold), %d worker processes, %.1f s" % (reps_done, done_frames[0], cores, el),
            "single_thread_value": round(m / t1 / 1e6, 4)}


def cpu_legs(args):
    """cpu_baseline of every workload of this run: the oracle ("port") on the box's usable cores, one forked worker process
    per core -- hence BEFORE this process initialises the GPU.  Inputs are the workloads' own (same seeds)."""
    from flydog_sdr_gps_amd import acq, prn, sats, synth
    from tests.fixtures import e1b_chips
    wls = ALL_WORKLOADS if args.workload == "all" else [args.workload]
    T = args.cpu_seconds

    def log2n_for(wl):
        return args.log2n if args.log2n_given else (22 if wl in RX_MIX else 24)
    skip_legs = os.environ.get("KIWIGPU_BENCH_SKIP_LEGS", "").split(",")       # a diagnostic
    for wl in wls:
        if wl in skip_legs:
            continue
        if wl == "acq":
            codes = [(prn.cacode(sats.SATS[s][1], sats.SATS[s][2]), False) for s in range(NSV)]
            CPU_LEGS[wl] = cpu_acq(synth.config1_iq16(seed=0x5EED0002), codes, NSAMPLES, FFT_LEN, -20, 20, T)
        elif wl == "acq59":
            CPU_LEGS[wl] = cpu_acq(synth.config1_iq16(seed=0x5EED0002), synth.all_sv_codes(e1b_chips()), NSAMPLES, FFT_LEN, -20, 20, T)
        elif wl == "acq10ms":
            codes = synth.all_sv_codes(e1b_chips())
            CPU_LEGS[wl] = cpu_acq(synth.config4_iq16(codes, seed=0x5EED0005), codes, acq.NSAMPLES_10MS, acq.FFT_LEN_10MS, -128, 127, T)
        elif wl == "wf14":
            CPU_LEGS[wl] = cpu_wf14(T)
        elif wl == "ddc14":
            CPU_LEGS[wl] = cpu_ddc14(1 << log2n_for(wl), T)
        elif wl == "cfg2_chain":
            CPU_LEGS[wl] = cpu_cfg2_chain(1 << log2n_for(wl), T)
        elif wl in RX_MIX:
            CPU_LEGS[wl] = cpu_receivers(args.receivers, 1 << log2n_for(wl), T, RX_MIX[wl])
        log("cpu_baseline %s: %s %s on %d cores" % (wl, CPU_LEGS[wl]["value"], CPU_LEGS[wl]["unit"], CPU_LEGS[wl]["cores"]))


def run_stub(args, dist):
    """Launcher / rendezvous self-test without a GPU (tests/test_host_cpu.py): gloo process group,
    barrier, max-over-ranks, one line from rank 0."""
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (dist.rank + 1))
    dist.barrier()
    el = dist.max_over_ranks(time.perf_counter() - t0)
    out = {"metric": "launcher self-test", "value": round(dist.world / el, 3), "unit": "ranks/s", "ms_per_step": round(el * 1e3, 3),
           "config": {"workload": "stub (no GPU)"}, "dtype": "none",
           # the SHAPE of a real line, so that the CPU suite holds the stdout line's size and top-level fields (compact_line):
           # placeholders, not measurements
           "roofline": {"bound": "none", "kernel": "stub", "achieved": 0.0, "peak": 1.0, "unit": "GB/s", "frac": 0.0, "traffic": None,
                        "traffic_source": "x" * 300, "by_workload": {"pad": ["y" * 200] * 8}},
           "cpu_baseline": {"value": 0.0, "unit": "ranks/s", "cores": 1, "kind": "port", "sample": "none (stub) " + "z" * 400},
           "workloads": {"w%d" % i: {"ms_per_step": 1.0, "value": 2.0, "unit": "u", "roofline": {"bound": "hbm", "frac": 0.5, "kernel_ms": 0.9,
                                                                                                   "traffic": 10, "algorithmic_bytes_per_launch": 5},
                                     "cpu_baseline": {"value": 3.0, "kind": "port"}, "checked": {"n": 1}, "notes": "n" * 3000}
                         for i in range(8)}}
    if args.shard == "sv":
        # the --shard sv exchange of run_acq without a GPU: every rank fabricates the winners of ITS share of the 59 SVs
        # (a pure function of block and SV), all ranks gather (gloo), the merge must give the unsharded table
        import numpy as np
        import torch
        import torch.distributed as tdist
        from flydog_sdr_gps_amd import shard
        from flydog_sdr_gps_amd._lib import result_dtype
        B, nsv = 2, 59
        weights = shard.sv_weights([sv >= 36 for sv in range(nsv)])     # gps/sats.cpp:25-142: the 23 E1B rows come last
        shares = shard.split_units_weighted(weights, dist.world)

        def fake(b, sv):
            return (16.0 + sv + 0.25 * b, sv - 30, (97 * sv + b) % 4092, 1)
        full = np.array([[fake(b, sv) for sv in range(nsv)] for b in range(B)], result_dtype)
        my = shares[dist.rank]
        nmax = max(len(sh) for sh in shares)
        mine = np.zeros(B * nmax, result_dtype)                 # the library's result array: [B][this rank's SVs], then slack
        mine[:B * len(my)] = full[:, my].reshape(-1)
        raw = torch.from_numpy(mine.view(np.uint8).copy())
        if dist.on:
            outs = [torch.empty_like(raw) for _ in range(dist.world)]
            tdist.all_gather(outs, raw)
            parts = np.stack([o.numpy().view(result_dtype) for o in outs])
        else:
            parts = mine.reshape(1, -1)
        merged = shard.merge_sv_shards(parts, B, shares)
        assert merged.shape == (B, nsv) and np.array_equal(merged, full), "sv-sharded merge differs from the unsharded table"
        # the timed loop of run_acq's --shard sv leg, shape for shape: a collective inside every step, and ranks whose steps take
        # DIFFERENT times (so that their own pre-roll estimates differ: the ranks must still run the same number of steps)
        ncalls = [0]

        def step():
            time.sleep(0.0002 * (1 + 3 * dist.rank))
            if dist.on:
                tdist.all_gather([torch.empty_like(raw) for _ in range(dist.world)], raw)
            ncalls[0] += 1
        el_s, _, spread = timed_steps(dist, step, 6, 2)
        counts = [ncalls[0]]
        if dist.on:
            t = torch.tensor([ncalls[0]], dtype=torch.int64)
            got = [torch.empty_like(t) for _ in range(dist.world)]
            tdist.all_gather(got, t)
            counts = [int(g.item()) for g in got]
        assert len(set(counts)) == 1, "the ranks ran different numbers of steps: %s" % counts
        out["shard_sv_timed_steps"] = {"steps_per_rank": counts, "seconds": round(el_s, 4)}
        out["shard_sv"] = {"world": dist.world, "shares": shares, "load": [round(sum(weights[i] for i in sh), 3) for sh in shares],
                           "merged_equals_unsharded": True}
        out["scaling"] = "strong"
    return out


def by_workload_table(rs):
    tab = {"_cols": ["ms_per_step", "frac", "bound", "hbm_frac_algorithmic", "traffic_over_algorithmic"]}
    for wl, r in rs.items():
        rf, hb = r.get("roofline", {}), r.get("hbm", {})
        alg = hb.get("algorithmic_bytes_per_launch") or hb.get("algorithmic_bytes_per_step") or rf.get("algorithmic_bytes_per_launch")
        ms = rf.get("kernel_ms") or r.get("ms_per_step")
        tr = rf.get("traffic")
        tab[wl] = [r.get("ms_per_step"), rf.get("frac"), rf.get("bound"),
                   None if not (alg and ms) else round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   None if not (alg and tr) else round(tr / alg, 3)]
    return tab


COMPACT_MAX = 6000                     # bytes of the ONE stdout line (the driver's record keeps the last 8 000 characters)
_KEEP_TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "invalid", "invalid_for_scaling")
_KEEP_ROOF = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "flops_per_launch",
              "algorithmic_bytes_per_launch", "int_ops_per_launch")
_KEEP_CPU = ("value", "unit", "cores", "kind", "sample", "note")


def _short(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "~"


def compact_line(line):
    """The ONE stdout line: the headline's contract fields, its `roofline` and `cpu_baseline` at top level, and one row per
    other workload -- [ms_per_step, value, unit, bound, frac, kernel_ms, traffic / algorithmic bytes, cpu value, cpu kind,
    result check ran].  Everything else (full workload objects, host phases, notes) goes to stderr and to
    gpurun_out/bench_full.json (publish_full_line).  Kept well under COMPACT_MAX bytes; tests/test_host_cpu.py holds it there."""
    out = {k: line[k] for k in _KEEP_TOP if k in line}
    cfg = dict(line.get("config") or {})
    cfg.pop("also_in_this_line", None)
    out["config"] = {k: _short(v, 160) for k, v in cfg.items()}
    if "roofline" in line:
        out["roofline"] = {k: line["roofline"][k] for k in _KEEP_ROOF if k in line["roofline"]}
    if "cpu_baseline" in line:
        out["cpu_baseline"] = {k: _short(line["cpu_baseline"][k], 200) for k in _KEEP_CPU if k in line["cpu_baseline"]}
    pf = line.get("cpu_baseline_pocketfft")
    if pf:
        out["cpu_baseline_tuned_fft"] = {"value": pf.get("value"), "unit": pf.get("unit"), "cores": pf.get("cores"), "kind": "scipy.fft (pocketfft)"}
    if line.get("checked"):
        out["checked"] = line["checked"]
    elif line.get("found_svs") is not None:
        out["checked"] = {"found_svs": line["found_svs"]}
    elif line.get("checked") is not None:
        out["checked"] = True
    wls = line.get("workloads")
    if wls:
        tab = {"_cols": ["ms_per_step", "value", "unit", "bound", "frac", "kernel_ms", "traffic_over_algorithmic", "cpu_value", "cpu_kind", "checked"]}
        for wl, r in wls.items():
            rf, hb, cb = r.get("roofline", {}), r.get("hbm", {}), r.get("cpu_baseline", {})
            alg = hb.get("algorithmic_bytes_per_launch") or hb.get("algorithmic_bytes_per_step") or rf.get("algorithmic_bytes_per_launch")
            tr = rf.get("traffic")
            tab[wl] = [r.get("ms_per_step"), r.get("value"), r.get("unit"), rf.get("bound"), rf.get("frac"), rf.get("kernel_ms"),
                       None if not (alg and tr) else round(tr / alg, 3), cb.get("value"), cb.get("kind"),
                       bool(r.get("checked") or r.get("found_svs"))]
        out["workloads"] = tab
    if "box" in line:
        out["box"] = {k: line["box"].get(k) for k in ("lib_sha16", "bench_sha16", "gpu_uuid")}
    out["full_line"] = "stderr `FULL {...}` and gpurun_out/bench_full.json"
    n = len(json.dumps(out, separators=(",", ":")))
    if n > COMPACT_MAX:                                    # never near the record's limit: drop the optional parts, largest first
        for k in ("workloads", "cpu_baseline_tuned_fft", "box", "checked"):
            out.pop(k, None)
            if len(json.dumps(out, separators=(",", ":"))) <= COMPACT_MAX:
                break
    return out


def publish_full_line(line):
    """The whole object: one `FULL {...}` line on stderr and, when the directory can be written, gpurun_out/bench_full.json."""
    txt = json.dumps(line)
    sys.stderr.write("FULL " + txt + "\n")
    sys.stderr.flush()
    try:
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_full.json"), "w") as f:
            f.write(txt + "\n")
    except OSError as e:
        log("bench_full.json not written: %s" % e)


def summary_line(line):
    """<= 400 characters: `SUMMARY wl=ms_per_step/frac+bound/traffic:algorithmic/check ...` for every workload of the line."""
    wls = dict(line.get("workloads") or {})
    if not wls:
        wls = {line.get("config", {}).get("workload", "?")[:8] if "workloads" in line else "run": line}
    out = []
    for wl, r in wls.items():
        rf, hb = r.get("roofline", {}), r.get("hbm", {})
        alg = hb.get("algorithmic_bytes_per_launch") or hb.get("algorithmic_bytes_per_step")
        tr = rf.get("traffic")
        ok = "ok" if (r.get("checked") or r.get("found_svs")) else "-"
        out.append("%s=%.3f/%s%s/%s/%s" % (wl.replace("receivers", "rx").replace("cfg2_chain", "chain"), r.get("ms_per_step") or 0.0,
                                          ("%.3f" % rf["frac"]) if rf.get("frac") is not None else "-", (rf.get("bound") or "-")[0],
                                          ("%.1fx" % (tr / alg)) if (tr and alg) else "-", ok))
    return ("SUMMARY ms/frac+roof/traffic:alg/check " + " ".join(out))[:400]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="all",
                    choices=["all", "acq", "acq59", "acq10ms", "wf14", "ddc14", "cfg2_chain", "waterfall", "ddc", "receivers", "receivers_light", "stub"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=None, help="acq: independent sample blocks per step (32; acq10ms: 2)")
    ap.add_argument("--frames", type=int, default=2048, help="wf14: frames per channel per step (2048 = 896 MiB of input)")
    ap.add_argument("--log2n", type=int, default=None, help="ddc14 / cfg2_chain / receivers: log2 of the ADC samples per step (24 / 24 / 22)")
    ap.add_argument("--receivers", type=int, default=128, help="receivers: virtual receivers per GPU")
    ap.add_argument("--shard", choices=["blocks", "sv"], default="blocks",
                    help="acq10ms: what the ranks share out -- their own blocks (weak scaling, default) or the SVs of the "
                         "same block (strong scaling, results all-gathered over RCCL every step)")
    ap.add_argument("--cpu-seconds", type=float, default=None, help="budget of each cpu_baseline leg (10; `all`: 4)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the rocprofv3 --pmc child passes; roofline.traffic then comes from profiles/hbm_traffic.json")
    ap.add_argument("--full-line", action="store_true",
                    help="print the whole object on stdout instead of the compact line (tools/*.sh; the default stdout line stays "
                         "under COMPACT_MAX bytes, the whole object goes to stderr as `FULL {...}` and to gpurun_out/bench_full.json)")
    ap.add_argument("--pmc-child", action="store_true",
                    help="(internal) counter-pass mode: a few steps of every workload between marker kernels, no timing")
    args = ap.parse_args()
    args.log2n_given = args.log2n is not None
    if args.cpu_seconds is None:
        args.cpu_seconds = 4.0 if args.workload == "all" else 10.0
    args.workload = {"waterfall": "wf14", "ddc": "ddc14"}.get(args.workload, args.workload)

    world_env = os.environ.get("WORLD_SIZE")
    if os.environ.get("KIWIGPU_BENCH_WATCHDOG_S") and not (world_env is None and args.gpus > 1):
        import faulthandler                                   # a diagnostic: a RANK that hangs says where (every thread's stack) and exits
        faulthandler.dump_traceback_later(float(os.environ["KIWIGPU_BENCH_WATCHDOG_S"]), exit=True)
    if world_env is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))      # the parent never touches the GPU
    if world_env is not None and int(world_env) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s\n" % (args.gpus, world_env))
        sys.exit(2)
    # stdout carries exactly ONE line, the JSON: whatever a native library writes to file descriptor 1 (RCCL prints its
    # version banner there when a process group comes up) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    single = args.gpus == 1 and (world_env is None or int(world_env) == 1)
    if single and not args.no_live_traffic and not args.pmc_child and args.workload != "stub":
        live_traffic_passes(args)                            # children; this process has not touched the GPU yet
    if not args.no_cpu and not args.pmc_child and args.workload != "stub" and (world_env is None or int(world_env) == 1):
        skip = os.environ.get("KIWIGPU_BENCH_SKIP", "").split(",")          # a diagnostic: leave parts of the CPU side out
        if "native" not in skip:
            native_oracle()                                  # before anything loads the oracle
        if "cpu" not in skip:
            cpu_legs(args)                                   # forked workers: before this process touches the GPU
        if "pocketfft" not in skip:
            pocketfft_legs(args)
        log("pocketfft legs done: %s" % {k: v.get("value") for k, v in POCKETFFT.items()})
    dist = Dist("gloo" if args.workload == "stub" else "nccl")
    common = {"n_gpus": dist.world, "steps": args.steps, "warmup": args.warmup, "preroll_s": PREROLL_S, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "data": "synthetic"}

    def log2n_for(wl):
        return args.log2n if args.log2n_given else (22 if wl in RX_MIX else 24)

    def run(wl):
        import copy
        a = copy.copy(args)
        a.log2n = log2n_for(wl)
        if args.workload == "all" and wl == "acq10ms":
            a.steps, a.warmup = min(args.steps, 20), min(args.warmup, 3)       # 3.4 ms steps; the default K = 200 is sized for acq
        if args.workload == "all" and wl in RX_MIX:
            a.steps, a.warmup = min(args.steps, 40), min(args.warmup, 4)
        fn = {"acq": lambda: run_acq(a, dist), "acq10ms": lambda: run_acq(a, dist, ten_ms=True),
              "acq59": lambda: run_acq(a, dist, all_svs=True), "wf14": lambda: run_wf14(a, dist),
              "ddc14": lambda: run_ddc14(a, dist), "cfg2_chain": lambda: run_cfg2_chain(a, dist),
              "receivers": lambda: run_receivers(a, dist), "receivers_light": lambda: run_receivers(a, dist, "receivers_light")}[wl]
        log("workload %s ..." % wl)
        r = fn()
        if dist.dev is not None:
            import torch
            torch.cuda.synchronize(dist.dev)
            torch.cuda.empty_cache()
        if not args.pmc_child:
            r.setdefault("steps", a.steps)
            r.setdefault("warmup", a.warmup)
        return r

    multi_note = {"value": None, "note": "measured on rank 0 at N = 1 only (the bench contract): see the --gpus 1 line of the same build"}

    def same_fields(r):
        """An N > 1 line carries the same fields as the N = 1 line: roofline.traffic falls back to the committed counter
        passes (measured_traffic says so), cpu_baseline is a note."""
        if isinstance(r, dict) and "roofline" in r:
            r.setdefault("cpu_baseline", dict(multi_note))
        return r

    if args.workload == "stub":
        line = dict(common, **run_stub(args, dist))
    elif args.workload == "all":
        rs = {wl: run(wl) for wl in ALL_WORKLOADS}
        if (dist.world > 1 or os.environ.get("KIWIGPU_BENCH_FORCE_SV") == "1") and not args.pmc_child:     # (FORCE_SV: the N > 1 shape of the line on one GPU, for testing)
            # strong scaling of ONE configs[4] acquisition (first bullet): the same block on every GPU, the 59
            # SVs dealt over the ranks by cost, winners all-gathered over RCCL every step -- so that a driver SCALE run
            # (N = 1, 2, 4, 8 of this command) records it beside the weak-scaling workloads
            args.shard = "sv"
            rs["acq10ms_sv"] = run("acq10ms")
            args.shard = "blocks"
        if args.pmc_child:
            line = {"pmc_child": list(rs)}
        else:
            a = rs["acq"]
            line = dict(a, **common)
            line["metric"] = "IQ Msamples/s ingested (waterfall + GPS acq); value = GPS acq, BASELINE configs[1]"
            line["workloads"] = dict({"acq": {k: a[k] for k in a if k not in ("metric", "config")}},
                                     **{wl: rs[wl] for wl in rs if wl != "acq"})
            line["config"]["also_in_this_line"] = ("workloads.wf14 / ddc14 / cfg2_chain: BASELINE configs[2] (frames, DDC, end to "
                                                   "end); workloads.receivers: configs[3] per-GPU share; workloads.acq10ms: configs[4]"
                                                   + ("; workloads.acq10ms_sv: configs[4] with the SVs split over the GPUs (strong)"
                                                      if "acq10ms_sv" in rs else ""))
            if dist.world > 1 or os.environ.get("KIWIGPU_BENCH_FORCE_SV") == "1":
                same_fields(line)
                for r in line["workloads"].values():
                    same_fields(r)
            # every workload's step time, roofline fraction and traffic ratio in ONE compact object inside `roofline`, so that
            # a record that keeps only the top-level keys still holds all of them: [ms_per_step, frac of the roof that
            # binds it (roofline.bound of the workload), algorithmic HBM bytes / time as a fraction of 8 TB/s, counter
            # traffic / algorithmic bytes]
            line["roofline"] = dict(line["roofline"], by_workload=by_workload_table(rs))
    else:
        r = run(args.workload)
        line = dict(common, **r)
        if dist.world > 1:
            same_fields(line)
    if args.workload != "stub" and not args.pmc_child:
        line["box"] = box_identity(dist.local_rank)
        if dist.share_gpu:
            line["invalid_for_scaling"] = ("KIWIGPU_BENCH_SHARE_GPU=1: %d ranks on ONE GPU, process group over gloo -- a rehearsal of the "
                                           "N > 1 code path (sharding, gathers, merge, line fields), not a scaling measurement" % dist.world)
        if TIMING_EXPERIMENT:
            line["invalid"] = "KIWIGPU_BENCH_TIMING_EXPERIMENT: a knock-out build whose rows are wrong by construction; not a measurement of the path"
    if dist.rank == 0:
        if args.pmc_child or args.full_line:
            out = line
        elif args.workload == "stub":
            out = compact_line(line)
        else:
            publish_full_line(line)
            out = compact_line(line)
        os.write(json_fd, (json.dumps(out, separators=(",", ":")) + "\n").encode())
    dist.close()
    if dist.rank == 0 and args.workload != "stub" and not args.pmc_child:
        # The LAST line on stderr: every workload's step time, roofline fraction + the roof that binds it (v: vector
        # arithmetic, h: HBM), counter traffic / algorithmic bytes, and whether its post-timed-region result check ran --
        # a record that keeps only the tail of stderr still holds all of them.
        sys.stderr.write(summary_line(line) + "\n")
        sys.stderr.flush()


if __name__ == "__main__":
    main()
