"""Input data shared by the tests and bench.py (never imported by the product package)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def e1b_chips(path=None):
    """{prn: chips uint8[4092]} from tests/golden/e1b_ref.npz: the 50 Galileo E1-B memory codes as the
    reference's own gps/e1bcode.h produced them (tools/make_ref_golden.py).  In a deployment the caller
    hands over its own table."""
    g = np.load(path or os.path.join(GOLDEN, "e1b_ref.npz"))
    chips = np.unpackbits(g["chips_packed"], axis=1)[:, :4092]
    return {i + 1: chips[i].copy() for i in range(chips.shape[0])}
