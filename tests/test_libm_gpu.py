"""The device's log10f, powf and expf (csrc/kg_libm.h: the GNU C Library 2.35 algorithms restated) against the image's libm -- what
the reference's S-meter, CAgc and aperture_auto() call, and CAgc branches on -- through the C ABI (kg_math_dev): BIT-EXACT, NaNs as
NaNs.  log10f: every mantissa of two binades either side of 1 (the only place the mantissa enters), every exponent strided, the
subnormals, the special values; powf(10, y) and expf: strided over all 2^32 arguments, dense where the receivers use them, the
overflow / underflow edges.  tools/check_libm.py --exhaustive walks every argument (profiles/r06_libm_exhaustive.txt)."""
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


def test_powf_and_expf_strided_over_every_argument(gpu_ctx, oracle):
    bits = np.concatenate([np.arange(0, 1 << 32, 509, dtype=np.uint64).astype(np.uint32),                 # 8.4 M patterns, both signs, NaNs
                           np.array([0, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0x4202422f, 0xc27c65d9, 0x42b17218, 0x42b17219,
                                     0xc2cff1b4, 0xc2cff1b5, 0x421a209a, 0x421a209b, 0xc23369f4, 0x3f800000, 0xbf800000, 1, 0x80000001], np.uint32)])
    fma_host = "fma" in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split()
    if not fma_host:        # glibc's baseline expf evaluates its residual unfused: 0x4202422f and 0xc27c65d9 then differ from the FMA build's (and the device's)
        bits = bits[~np.isin(bits, np.array([0x4202422f, 0xc27c65d9], np.uint32))]
    x = bits.view(np.float32)
    for name, got, want in (("powf", post.math_dev(gpu_ctx, post.MATH_POWF, x, base=10.0), oracle.libm_powf(10.0, x)),
                            ("expf", post.math_dev(gpu_ctx, post.MATH_EXPF, x), oracle.libm_expf(x)),
                            ("powf 2.5", post.math_dev(gpu_ctx, post.MATH_POWF, x, base=2.5), oracle.libm_powf(2.5, x)),
                            ("powf 0.3", post.math_dev(gpu_ctx, post.MATH_POWF, x, base=0.3), oracle.libm_powf(0.3, x))):
        bad = np.flatnonzero(~same(got, want))
        assert bad.size == 0, (name, [hex(int(b)) for b in bits[bad[:6]]], got[bad[:6]], want[bad[:6]])


def test_powf_and_expf_where_the_receivers_use_them(gpu_ctx, oracle):
    """CAgc's exponent mag * (slope - 1) for mag in (-8, 0] and slopes of 0 .. 10 dB (agc.cpp:250-253): all floats of [-8, -2^-20]; the
    aperture IIR's -param * pwr / 255 for pwr in -213 .. 0 dBm."""
    for first, n in ((0xb5800000, 0xc1000000 - 0xb5800000),):                  # -2^-20 .. -8: every float
        for lo in range(0, n, 1 << 25):
            m = min(1 << 25, n - lo)
            got, want = post.math_dev(gpu_ctx, post.MATH_POWF, first_bits=first + lo, n=m, base=10.0), oracle.libm_powf_bits(10.0, first + lo, m)
            assert np.all(same(got, want)), hex(first + lo)
    rng = np.random.default_rng(12)
    arg = (-rng.uniform(0.01, 20.0, 1 << 20).astype(np.float32) * rng.integers(-213, 1, 1 << 20).astype(np.float32) / 255.0).astype(np.float32)
    assert np.all(same(post.math_dev(gpu_ctx, post.MATH_EXPF, arg), oracle.libm_expf(arg)))
    assert np.all(same(post.math_dev(gpu_ctx, post.MATH_EXPF, -arg), oracle.libm_expf(-arg)))
    with pytest.raises(Exception):
        post.math_dev(gpu_ctx, post.MATH_POWF, arg[:4], base=-2.0)
    with pytest.raises(Exception):
        post.math_dev(gpu_ctx, 7, arg[:4])
