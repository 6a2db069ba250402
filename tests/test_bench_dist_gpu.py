"""The bench's N > 1 code path on ONE GPU, over RCCL itself: KIWIGPU_BENCH_FORCE_DIST=1 takes the process-group path with a
single rank (backend nccl = RCCL: init with the device, barrier, the all-reduce of max_over_ranks, the agreed pre-roll count),
KIWIGPU_BENCH_FORCE_SV=1 adds the strong-scaling leg of configs[4], whose every step ends in all_gather_into_tensor of the
library's result array.  (Two ranks on one device are refused by RCCL: that shape runs over gloo,
KIWIGPU_BENCH_SHARE_GPU=1; two GPUs are not to be had on the test box.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_process_group_path_over_rccl_with_one_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(KIWIGPU_BENCH_FORCE_DIST="1", KIWIGPU_BENCH_FORCE_SV="1", KIWIGPU_BENCH_PREROLL_S="0.05", KIWIGPU_BENCH_WATCHDOG_S="240",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "acq10ms", "--shard", "sv", "--steps", "6", "--warmup", "2",
                          "--no-cpu", "--no-live-traffic", "--full-line"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["steps"] == 6
    assert line["value"] > 0 and "roofline" in line and "configs[4]" in line["config"]["workload"]
    assert "all-gathered over RCCL every step" in line["config"]["parallelism"]
    # every injected SV found after the gather and the merge of the (single) shard: asserted inside run_acq, rc 0 says so


def _gpu_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs on one node (the test box has one): unmeasured on hardware until the driver's SCALE run")
def test_two_ranks_over_rccl_strong_and_weak():
    """`bench.py --gpus 2` as the driver launches it (the parent starts two ranks, one per GPU, RCCL over xGMI):
    (a) configs[4] with the 59 SVs dealt over the ranks (--shard sv): every step ends in an all-gather of the result arrays; the
        merged winners must be the unsharded run's (asserted inside run_acq: every injected SV found) -- strong scaling;
    (b) the receivers line with each rank's own 128 receivers (no data-path collective) -- weak scaling: value = both ranks'
        samples over the slower rank's time.
    Skipped where fewer than two GPUs are visible: the first N > 1 run is then a test, not a discovery."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(KIWIGPU_BENCH_PREROLL_S="0.05", KIWIGPU_BENCH_WATCHDOG_S="240")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "acq10ms", "--shard", "sv", "--steps", "6",
                          "--warmup", "2", "--no-cpu", "--no-live-traffic", "--full-line"], env=env, capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 6 and line["value"] > 0
    assert "all-gathered over RCCL every step" in line["config"]["parallelism"] and "invalid_for_scaling" not in line
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "acq10ms", "--steps", "6", "--warmup", "2",
                             "--no-cpu", "--no-live-traffic", "--full-line"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert single.returncode == 0, single.stderr[-4000:]
    one = json.loads([l for l in single.stdout.splitlines() if l.startswith("{")][-1])
    assert sorted(line["found_svs"]) == sorted(one["found_svs"]), (line["found_svs"], one["found_svs"])     # merged winners = the unsharded run's
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "receivers", "--steps", "8", "--warmup", "2",
                          "--no-cpu", "--no-live-traffic"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 8192                       # ONE compact line from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and "roofline" in line and "cpu_baseline" in line
