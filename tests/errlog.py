"""Achieved maxima of the float comparisons the parity tests make: every call appends one line to
gpurun_out/float_errors.txt (when that directory exists: on the GPU box), so that the tolerance written in a test can be
read beside what was reached.  Test infrastructure."""
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "float_errors.txt")


def rel_max(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))


def record(name, got, want, tol):
    """-> the maximum relative error of got against want; logged with the tolerance the caller is about to apply"""
    e = rel_max(got, want)
    try:
        if os.path.isdir(os.path.dirname(_PATH)):
            with open(_PATH, "a") as f:
                f.write("%-58s max rel err %.3e  (tolerance %.1e, n = %d)\n" % (name, e, tol, np.asarray(want).size))
    except OSError:
        pass
    return e
