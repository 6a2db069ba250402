"""Multi-GPU sharding of the path (one process per GPU, torch.distributed).

The units of this path are independent: sample blocks ("receivers") for
acquisition, receiver channels for the waterfall / audio chain.  They are dealt
to ranks with no data-path collective; only the tiny per-unit results are
gathered (RCCL all_gather on GPUs, gloo in the CPU tests).  SURVEY.md 8(e).
"""
import numpy as np


def block_ids(rank, world, per_rank):
    """Global ids of the sample blocks rank `rank` owns: contiguous runs of
    per_rank blocks, rank-major (weak scaling: per-rank work is fixed)."""
    if not (0 <= rank < world) or per_rank < 1:
        raise ValueError("bad rank/world/per_rank: %r %r %r" % (rank, world, per_rank))
    return list(range(rank * per_rank, (rank + 1) * per_rank))


def split_units(n_units, world):
    """Strong-scaling split of n_units independent units (SVs, receiver channels)
    into `world` contiguous near-equal ranges; returns [(start, stop)] per rank."""
    if n_units < 0 or world < 1:
        raise ValueError("bad n_units/world")
    base, extra = divmod(n_units, world)
    out, start = [], 0
    for r in range(world):
        stop = start + base + (1 if r < extra else 0)
        out.append((start, stop))
        start = stop
    return out


def gather_results(local, device=None):
    """all_gather a structured numpy array of per-unit results (equal shape on every
    rank) along axis 0.  Works on any initialised process group: nccl (= RCCL)
    with device given, gloo with device None."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    local = np.ascontiguousarray(local)
    raw = torch.from_numpy(local.view(np.uint8).reshape(-1).copy())
    if device is not None:
        raw = raw.to(device)
    outs = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(outs, raw)
    parts = [o.cpu().numpy().view(local.dtype).reshape(local.shape) for o in outs]
    return np.concatenate(parts, axis=0)


def merge_sv_shards(parts, nblocks, ranges):
    """Strong-scaling acquisition (first bullet): rank r searched the SVs ranges[r] = (lo, hi) of the
    SAME nblocks sample blocks and holds results [nblocks][hi - lo] (block-major, the layout of kg_acq_fetch) at the
    start of its gathered row parts[r] (rows are padded to the largest share).  -> [nblocks][total SVs]."""
    total = ranges[-1][1]
    out = np.zeros((nblocks, total), parts.dtype)
    for r, (lo, hi) in enumerate(ranges):
        if hi > lo:
            out[:, lo:hi] = np.asarray(parts[r])[:nblocks * (hi - lo)].reshape(nblocks, hi - lo)
    return out


def best_of(results):
    """Merge per-rank kg_acq_result rows for the SAME (block, SV) searched over
    disjoint Doppler ranges: keep the higher snr, ties to the lower Doppler bin,
    which is what the serial strict-> scan of gps/search.cpp:495 yields."""
    best = results[0].copy()
    for r in results[1:]:
        take = (r["valid"] == 1) & ((best["valid"] == 0) | (r["snr"] > best["snr"])
                                    | ((r["snr"] == best["snr"]) & (r["dop"] < best["dop"])))
        best[take] = r[take]
    return best
