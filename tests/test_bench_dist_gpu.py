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
