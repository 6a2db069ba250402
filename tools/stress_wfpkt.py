"""Stress: waterfall packets against the oracle, many rounds (debugging aid)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, wire   # noqa: E402
from oracle import kiwi_oracle as ko           # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(8)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    nrows = int(rng.integers(1, 40))
    rows = rng.integers(0, 256, (nrows, 1024)).astype(np.uint8)
    infos = [(int(rng.integers(0, 2 ** 31)), z % 15, 1000 + z, bool(rng.integers(0, 2))) for z in range(nrows)]
    pk = wire.wf_packets(ctx, rows, infos)
    for r in range(nrows):
        want = ko.wf_packet(rows[r], *infos[r])
        if not np.array_equal(pk[r], want):
            d = np.nonzero(pk[r] != want)[0]
            print("iter", it, "row", r, "of", nrows, "compress", infos[r][3], "first diffs at", d[:8], "count", d.size,
                  "got", pk[r][d[:4]], "want", want[d[:4]])
            bad += 1
print("bad", bad)
