"""PRN chip generators used by the host side when it builds the code tables
(the reference does this in SearchInit(), gps/search.cpp:243-346, with
gps/cacode.h and gps/e1bcode.h).  Chips are {0,1} uint8, chip 0 first."""
import numpy as np

from .sats import E1B_CODELEN, L1_CODELEN


def cacode(t1, t2):
    """GPS/QZSS C/A code, 1023 chips (gps/cacode.h:23-53).

    G1 = x^10 + x^3 + 1, G2 = x^10 + x^9 + x^8 + x^6 + x^3 + x^2 + 1.  With both
    taps <= 10 the chip is g1[10] ^ g2[t1] ^ g2[t2] and G2 starts all-ones;
    otherwise t2 is G2's initial state (bit i-1 -> stage i) and the chip is
    g1[10] ^ g2[10]."""
    g1 = [1] * 11
    use_taps = not (t1 > 10 or t2 > 10)
    if use_taps:
        g2 = [1] * 11
    else:
        g2 = [0] + [(t2 >> i) & 1 for i in range(10)]
    out = np.empty(L1_CODELEN, np.uint8)
    for n in range(L1_CODELEN):
        out[n] = (g1[10] ^ g2[t1] ^ g2[t2]) if use_taps else (g1[10] ^ g2[10])
        f1 = g1[3] ^ g1[10]
        f2 = g2[2] ^ g2[3] ^ g2[6] ^ g2[8] ^ g2[9] ^ g2[10]
        g1 = [0, f1] + g1[1:10]
        g2 = [0, f2] + g2[1:10]
    return out


def e1b_from_hex(hexstr):
    """Galileo E1-B memory code: 1023 hex digits -> 4092 chips, MSB first per
    digit (gps/e1bcode.h:70-76)."""
    if len(hexstr) < E1B_CODELEN // 4:
        raise ValueError("E1B memory code needs %d hex digits" % (E1B_CODELEN // 4))
    nib = np.array([int(c, 16) for c in hexstr[:E1B_CODELEN // 4]], np.uint8)
    return ((nib[:, None] >> np.array([3, 2, 1, 0], np.uint8)) & 1).reshape(-1).astype(np.uint8)
