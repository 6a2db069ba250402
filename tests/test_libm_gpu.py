"""The device's log10f (csrc/kg_libm.h: the GNU C Library 2.35 algorithm restated) against the log10f of the image's libm -- the
one the reference's S-meter and CAgc call and CAgc branches on -- through the C ABI (kg_math_log10f_dev): BIT-EXACT, NaNs as
NaNs.  Every mantissa of two binades either side of 1 (the only place the mantissa enters), every exponent strided, the
subnormals, the special values; tools/check_log10f.py --exhaustive walks all 2^31 + patterns (profiles/r06_log10f_exhaustive.txt)."""
import numpy as np
import pytest

from flydog_sdr_gps_amd import post

pytestmark = pytest.mark.gpu


def same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def test_every_mantissa_either_side_of_one(gpu_ctx, oracle):
    for first in (0x3f000000, 0x3f800000, 0x00800000, 0x7f000000):          # [0.5, 1), [1, 2), the lowest and highest binade
        n = 1 << 23
        got, want = post.log10f(gpu_ctx, first_bits=first, n=n), oracle.libm_log10f_bits(first, n)
        bad = np.flatnonzero(~same(got, want))
        assert bad.size == 0, (hex(first), bad[:4], got[bad[:4]], want[bad[:4]])


def test_every_exponent_strided_subnormals_and_specials(gpu_ctx, oracle):
    rng = np.random.default_rng(10)
    bits = np.concatenate([np.arange(0, 0x7f800000, 251, dtype=np.uint32),                       # 8.5 M patterns over every exponent
                           np.arange(0, 0x00800000, 7, dtype=np.uint32),                          # subnormals
                           rng.integers(0, 1 << 32, 1 << 20, dtype=np.uint32),                    # anything, negatives and NaNs included
                           np.array([0, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0xffc00000, 0x7f800001, 1, 0x007fffff,
                                     0x00800000, 0x3f800000, 0x3f7fffff, 0x3f800001, 0x7f7fffff, 0x80000001, 0xbf800000], np.uint32)])
    x = bits.view(np.float32)
    got, want = post.log10f(gpu_ctx, x), oracle.libm_log10f(x)
    bad = np.flatnonzero(~same(got, want))
    assert bad.size == 0, ([hex(int(b)) for b in bits[bad[:6]]], got[bad[:6]], want[bad[:6]])
    assert got[-16] == -np.inf and got[-15] == -np.inf and got[-14] == np.inf and np.isnan(got[-13]) and got[-6] == 0.0


def test_the_values_the_receivers_take_it_of(gpu_ctx, oracle):
    """The S-meter's and CAgc's arguments for real sample powers: pwr / max + 1e-30 and mag / MAX^2 + 1e-16 in double, to float."""
    rng = np.random.default_rng(11)
    amp = np.concatenate([10.0 ** rng.uniform(-6, 4.5, 1 << 20), np.zeros(64)]).astype(np.float32)
    pwr = amp * amp
    smax = np.float32(((1 << 13) - 1) ** 2)
    for arg in ((pwr / smax).astype(np.float64) + 1e-30, pwr.astype(np.float64) / (32767.0 * 32767.0) + 1e-16):
        x = arg.astype(np.float32)
        assert np.all(same(post.log10f(gpu_ctx, x), oracle.libm_log10f(x)))
