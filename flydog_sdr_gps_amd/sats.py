"""Satellite table: the host-side mirror of the reference's Sats[] (gps/sats.cpp:25-142).

Rows are (prn, t1, t2, kind).  For Navstar t1,t2 are the G2 tap pair of
IS-GPS-200; for QZSS t1 is the G2 delay (documentation only) and t2 the 10-bit
G2 initial state given in octal by the QZSS ICD (selected because t1 > 10,
gps/cacode.h:28-33); E1B rows carry only the PRN (memory codes).
"""
NAVSTAR, QZSS, E1B = "Navstar", "QZSS", "E1B"

_NAVSTAR_TAPS = [
    (2, 6), (3, 7), (4, 8), (5, 9), (1, 9), (2, 10), (1, 8), (2, 9), (3, 10), (2, 3), (3, 4),
    (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (1, 4), (2, 5), (3, 6), (4, 7), (5, 8), (6, 9),
    (1, 3), (4, 6), (5, 7), (6, 8), (7, 9), (8, 10), (1, 6), (2, 7), (3, 8), (4, 9),
]
_QZSS = [(194, 208, 0o1607), (195, 711, 0o1747), (196, 189, 0o1305), (199, 663, 0o727)]
_E1B_PRNS = [2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 15, 19, 21, 24, 25, 26, 27, 30, 31, 33, 34, 36]

SATS = ([(i + 1, a, b, NAVSTAR) for i, (a, b) in enumerate(_NAVSTAR_TAPS)]
        + [(p, d, g, QZSS) for p, d, g in _QZSS]
        + [(p, 0, 0, E1B) for p in _E1B_PRNS])

MAX_SATS = 64            # gps/gps.h:123
L1_CODELEN = 1023        # kiwi.config:266
E1B_CODELEN = 4092       # kiwi.config:270
L1_LIMIT = 4092          # SAMPLE_RATE/1000 * L1_CODE_PERIOD  (gps/search.cpp:486)
E1B_LIMIT = 16368        # SAMPLE_RATE/1000 * E1B_CODE_PERIOD


def navstar_index(prn):
    """Index into SATS (the reference's `sat`) of Navstar PRN prn."""
    assert 1 <= prn <= 32
    return prn - 1
