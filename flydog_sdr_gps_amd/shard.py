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


# Cost of one SV's Correlate() relative to a C/A SV, from the kernel traces (profiles/r03_acq59_kernel_stats.csv: the
# 16368-lag correlator 997.95 us / 23 SVs against the C/A one 957.81 us / 36 -> 1.63; profiles/r03_acq10ms_kernel_stats.csv:
# 1518.8 / 23 against 1489.9 / 36 -> 1.60).  gps/sats.cpp:25-142 puts the 23 E1B rows LAST, so a contiguous split
# hands one rank seven of them at world 8 (124 % of the mean).
SV_WEIGHT_CA, SV_WEIGHT_E1B = 1.0, 1.6


def sv_weights(is_e1b):
    """Per-SV cost weights for split_units_weighted: is_e1b[i] true for a 16368-lag (Galileo E1B) row of Sats[]."""
    return [SV_WEIGHT_E1B if e else SV_WEIGHT_CA for e in is_e1b]


def split_units_weighted(weights, world):
    """Cost-balanced strong-scaling split: unit i costs weights[i]; -> one ascending index list per rank, every unit in
    exactly one.  Longest-processing-time-first, then moves / swaps out of the heaviest rank while they lower
    (max load, sum of squares) -- exact enough for the few weight classes of an SV list: the worst rank of the
    reference's 36 C/A + 23 E1B list is within 105 % of the mean at world 2, 4 and 8 (tests/test_host_cpu.py)."""
    if world < 1 or any(w <= 0 for w in weights):
        raise ValueError("bad weights/world")
    n = len(weights)
    shares = [[] for _ in range(world)]
    load = [0.0] * world
    for i in sorted(range(n), key=lambda i: (-weights[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        shares[r].append(i)
        load[r] += weights[i]

    def score(ld):
        return (round(max(ld), 9), round(sum(x * x for x in ld), 9))
    improved = True
    while improved:
        improved = False
        hi = max(range(world), key=lambda r: load[r])
        best = None
        for r in range(world):
            if r == hi:
                continue
            for a in shares[hi]:
                ld = list(load)
                ld[hi] -= weights[a]; ld[r] += weights[a]
                if score(ld) < score(load) and (best is None or score(ld) < best[0]):
                    best = (score(ld), a, None, r)
                for b in shares[r]:
                    ld = list(load)
                    d = weights[a] - weights[b]
                    ld[hi] -= d; ld[r] += d
                    if score(ld) < score(load) and (best is None or score(ld) < best[0]):
                        best = (score(ld), a, b, r)
        if best is not None:
            _, a, b, r = best
            shares[hi].remove(a); shares[r].append(a)
            load[hi] -= weights[a]; load[r] += weights[a]
            if b is not None:
                shares[r].remove(b); shares[hi].append(b)
                load[r] -= weights[b]; load[hi] += weights[b]
            improved = True
    return [sorted(sh) for sh in shares]


def shares_of(ranges):
    """split_units' (start, stop) ranges as index lists (the form merge_sv_shards takes)."""
    return [list(range(lo, hi)) for lo, hi in ranges]


def gather_results(local, device=None):
    """all_gather a structured numpy array of per-unit results (equal shape on every
    rank) along axis 0.  Works on any initialised process group: nccl (= RCCL)
    with device given -- ONE all_gather_into_tensor on the device, one copy back --
    gloo with device None."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    local = np.ascontiguousarray(local)
    raw = torch.from_numpy(local.view(np.uint8).reshape(-1).copy())
    if device is not None:
        raw = raw.to(device)
        out = torch.empty(world * raw.numel(), dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(out, raw)
        flat = out.cpu().numpy()
    else:
        outs = [torch.empty_like(raw) for _ in range(world)]
        dist.all_gather(outs, raw)
        flat = np.concatenate([o.numpy() for o in outs])
    return flat.view(local.dtype).reshape((world * local.shape[0],) + local.shape[1:])


def merge_sv_shards(parts, nblocks, shares):
    """Strong-scaling acquisition (first bullet): rank r searched the SVs shares[r] (an index list, in the
    order it passed them to Correlate(); split_units_weighted, or shares_of(split_units)) of the SAME nblocks sample
    blocks and holds results [nblocks][len(shares[r])] (block-major, the layout of kg_acq_fetch) at the start of its
    gathered row parts[r] (rows are padded to the largest share).  -> [nblocks][total SVs]."""
    shares = [list(range(sh[0], sh[1])) if isinstance(sh, tuple) else list(sh) for sh in shares]
    total = sum(len(sh) for sh in shares)
    assert sorted(i for sh in shares for i in sh) == list(range(total)), "shares must partition the SV list"
    out = np.zeros((nblocks, total), parts.dtype)
    for r, sh in enumerate(shares):
        if sh:
            out[:, sh] = np.asarray(parts[r])[:nblocks * len(sh)].reshape(nblocks, len(sh))
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
