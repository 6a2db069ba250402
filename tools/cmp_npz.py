"""python tools/cmp_npz.py a.npz b.npz -- are two golden files the same data (used after regenerating a reference-made vector)."""
import sys

import numpy as np

a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = [k for k in a.files if k not in b.files or a[k].shape != b[k].shape
       or not np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind in "fc")]
print(sys.argv[1], "keys", len(a.files), "differing", bad[:10] + [k for k in b.files if k not in a.files][:10])
