"""numpy model of the device FFT decomposition (index/twiddle math only).

4096 = 16*16*16 Stockham, 256 threads x 16 points; 16384 = 4 x 4096 with the
residue-major ("k2-major") layout.  Validates against numpy.fft.
"""
import numpy as np

M, R, T = 4096, 16, 256
N = 4 * M

def dft16(x, sign):
    # x: [16, ...] -> Y[m] = sum_j x[j] W16^{sign*j*m}
    j = np.arange(16)
    W = np.exp(sign * 2j * np.pi * np.outer(j, j) / 16)
    return np.tensordot(W, x, axes=(1, 0))

def subfft4096(xin, sign):
    """xin: [4096] -> y[n] natural; emulate thread mapping."""
    t = np.arange(T)
    lds = np.zeros(M, complex)
    # pass 0
    x = np.stack([xin[t + 256 * j] for j in range(16)])         # [16, T]
    y = dft16(x, sign)
    for m in range(16):
        lds[16 * t + m] = y[m]
    # pass 1, Ns = 16
    x = np.stack([lds[t + 256 * j] for j in range(16)])
    for j in range(16):
        x[j] = x[j] * np.exp(sign * 2j * np.pi * j * (t & 15) / 256)
    y = dft16(x, sign)
    lds2 = np.zeros(M, complex)
    for m in range(16):
        lds2[(t >> 4) * 256 + (t & 15) + 16 * m] = y[m]
    # pass 2, Ns = 256
    x = np.stack([lds2[t + 256 * j] for j in range(16)])
    for j in range(16):
        x[j] = x[j] * np.exp(sign * 2j * np.pi * j * t / 4096)
    y = dft16(x, sign)                                           # y[m] <-> n = t + 256 m
    out = np.zeros(M, complex)
    for m in range(16):
        out[t + 256 * m] = y[m]
    return out

rng = np.random.default_rng(1)
x = rng.standard_normal(M) + 1j * rng.standard_normal(M)
for sign in (+1, -1):
    ref = np.fft.ifft(x) * M if sign > 0 else np.fft.fft(x)
    got = subfft4096(x, sign)
    print("sub4096 sign", sign, np.abs(got - ref).max() / np.abs(ref).max())

# pruned backward 16384: y[n], n<4096 (NQ=1) and full (NQ=4)
X = rng.standard_normal(N) + 1j * rng.standard_normal(N)
ref = np.fft.ifft(X) * N
Xp = np.stack([X[k2::4] for k2 in range(4)])     # residue-major Xp[k2][k1] = X[4k1+k2]
acc = np.zeros((4, M), complex)
n1 = np.arange(M)
for k2 in range(4):
    sub = subfft4096(Xp[k2], +1)
    for q in range(4):
        acc[q] += sub * np.exp(2j * np.pi * n1 * k2 / N) * (1j) ** (q * k2)
got = acc.reshape(-1)    # n = n1 + 4096 q
print("bwd16384", np.abs(got - ref).max() / np.abs(ref).max())

# forward 16384 DIT from natural time input; output in residue-major layout
x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
ref = np.fft.fft(x)
acc = np.zeros((4, M), complex)
for n2 in range(4):
    sub = subfft4096(x[n2::4], -1)
    for q in range(4):
        acc[q] += sub * np.exp(-2j * np.pi * n1 * n2 / N) * (-1j) ** (q * n2)
got = acc.reshape(-1)
print("fwd16384", np.abs(got - ref).max() / np.abs(ref).max())

# doppler shift in residue-major layout
C = rng.standard_normal(N) + 1j * rng.standard_normal(N)
Cp = np.stack([C[r::4] for r in range(4)])
for dop in (-20, -3, 0, 1, 7, 20):
    for k2 in range(4):
        s = k2 - dop
        r, q = s & 3, s >> 2
        k1 = np.arange(M)
        got = Cp[r][(k1 + q) & (M - 1)]
        ref = C[(4 * k1 + k2 - dop) % N]
        assert np.array_equal(got, ref), (dop, k2)
print("doppler layout ok")

# swizzle conflict check: P(e) = e ^ ((e>>4)&15)
def P(e): return e ^ ((e >> 4) & 15)
t = np.arange(256)
def wr_ok(e):   # ds_write_b64: 16-lane groups distinct mod 16
    return all(len(set(P(e[g:g+16]) % 16)) == 16 for g in range(0, 256, 16))
def rd_ok(e):   # ds_read_b64: 32-lane groups distinct mod 32
    return all(len(set(P(e[g:g+32]) % 32)) == 32 for g in range(0, 256, 32))
print("p0 write", all(wr_ok(16 * t + m) for m in range(16)))
print("p1 read ", all(rd_ok(t + 256 * j) for j in range(16)))
print("p1 write", all(wr_ok((t >> 4) * 256 + (t & 15) + 16 * m) for m in range(16)))
assert len(set(P(np.arange(M)))) == M

# ---- 1024 = 16 * 16 * 4 by one wave (64 threads x 16 points), kg_snd.hip -------------
def fft1024_wave(xin, sign):
    N, t = 1024, np.arange(64)
    A = np.zeros(N, complex); B = np.zeros(N, complex)
    x = np.stack([xin[t + 64 * j] for j in range(16)])
    y = dft16(x, sign)
    for m in range(16): A[16 * t + m] = y[m]
    x = np.stack([A[t + 64 * j] for j in range(16)])
    for j in range(16): x[j] = x[j] * np.exp(sign * 2j * np.pi * j * (t & 15) / 256)
    y = dft16(x, sign)
    for m in range(16): B[(t >> 4) * 256 + (t & 15) + 16 * m] = y[m]
    out = np.zeros(N, complex)
    W4 = np.exp(sign * 2j * np.pi * np.outer(np.arange(4), np.arange(4)) / 4)
    held = {}
    for u in range(4):
        b = t + 64 * u
        x = np.stack([B[b + 256 * j] * np.exp(sign * 2j * np.pi * j * b / 1024) for j in range(4)])
        y = np.tensordot(W4, x, axes=(1, 0))
        for m in range(4):
            out[b + 256 * m] = y[m]
            held[(u, m)] = b + 256 * m
    # the positions a thread ends with are exactly the t + 64 j it starts the next transform with
    for (u, m), pos in held.items():
        assert np.array_equal(pos, t + 64 * (u + 4 * m))
    return out

x = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
for sign in (+1, -1):
    ref = np.fft.ifft(x) * 1024 if sign > 0 else np.fft.fft(x)
    print("fft1024 sign", sign, np.abs(fft1024_wave(x, sign) - ref).max() / np.abs(ref).max())
tt = np.arange(64)
def rd_ok64(e): return all(len(set(P(e[g:g+32]) % 32)) == 32 for g in (0, 32))
def wr_ok64(e): return all(len(set(P(e[g:g+16]) % 16)) == 16 for g in range(0, 64, 16))
print("1024 p0 write", all(wr_ok64(16 * tt + m) for m in range(16)),
      "p1 read", all(rd_ok64(tt + 64 * j) for j in range(16)),
      "p1 write", all(wr_ok64((tt >> 4) * 256 + (tt & 15) + 16 * m) for m in range(16)),
      "p2 read", all(rd_ok64(tt + 64 * u + 256 * j) for u in range(4) for j in range(4)))
