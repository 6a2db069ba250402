"""numpy model of the WAVE-LOCAL form of the 512-thread radix-8 4096-point transform (kg_subfft4096_r8w, kg_fft.h):
thread i = 64 T2 + 8 T1 + T0 holds X[pin(i) + 512 j], pin(i) = 64 T1 + 8 T2 + T0; every exchange swaps the register digit
with ONE digit of the thread index -- exchange 1 with T2 (the wave number: the only exchange that crosses waves, one workgroup
barrier), exchanges 0 and 2 with T1 and T0 (inside a wave: LDS in program order, no barrier).  The output of thread i,
register m is the transform at n = rev(i) + 512 m, rev(i) = T1 + 8 T2 + 64 T0; the twiddles are the Stockham form's with
rev(i) in i's seat.
Checks the transform against numpy.fft, the E1B combine on it, and the LDS rules (proto_fft8.py) for every instruction."""
import numpy as np

M, T = 4096, 512
S1, S2 = 72, 65            # row strides of the wave-private tiles (elements of 8 bytes)


def dft8(x, sign):
    j = np.arange(8)
    return np.tensordot(np.exp(sign * 2j * np.pi * np.outer(j, j) / 8), x, axes=(1, 0))


def rev(i):         # the lane's place in the output: k0 = T1, k1 = T2, k2 = T0
    return ((i >> 3) & 7) + 8 * (i >> 6) + 64 * (i & 7)


def pin(i):         # the lane's place in the input: n2 = T1, n1 = T2, n0 = T0
    return 64 * ((i >> 3) & 7) + 8 * (i >> 6) + (i & 7)


# address (8-byte elements) of the store of register m / the load of register j, as the device code forms them
# exchange 0: register <-> T1 (wave-private), exchange 1: register <-> T2 (the wave number: across waves), exchange 2: register <-> T0
def w0(i, m): return (i >> 6) * 8 * S1 + (i & 63) + S1 * m
def r0(i, j): return (i >> 6) * 8 * S1 + ((i >> 3) & 7) * S1 + (i & 7) + 8 * j
def w1(i, m): return 512 * m + i
def r1(i, j): return (i >> 6) * 512 + (i & 63) + 64 * j
def w2(i, m): return (i >> 6) * 8 * S2 + (i & 63) + S2 * m
def r2(i, j): return (i >> 6) * 8 * S2 + (i & 7) * S2 + ((i >> 3) & 7) * 8 + j


def subfft4096_r8w(xin, sign):
    i = np.arange(T)
    ri = rev(i)
    x = np.stack([xin[pin(i) + 512 * j] for j in range(8)])
    for p, (w, r, size) in enumerate(((w0, r0, 8 * 8 * S1), (w1, r1, M), (w2, r2, 8 * 8 * S2), (None, None, 0))):
        ns = 8 ** p
        if p > 0:
            for j in range(8):
                x[j] = x[j] * np.exp(sign * 2j * np.pi * j * (ri % ns) / (8 * ns))
        y = dft8(x, sign)
        if p == 3:
            out = np.zeros(M, complex)
            for m in range(8):
                out[ri + 512 * m] = y[m]
            return out
        lds = np.full(size, np.nan, complex)
        for m in range(8):
            lds[w(i, m)] = y[m]
        x = np.stack([lds[r(i, j)] for j in range(8)])
        assert not np.isnan(x).any()


def conflicts_and_locality():
    worst = 0
    i = np.arange(T)
    for p, (w, r) in enumerate(((w0, r0), (w1, r1), (w2, r2))):
        owner = {}
        for m in range(8):
            a = w(i, m)
            for t, e in zip(i.tolist(), a.tolist()):
                assert e not in owner
                owner[e] = t
            for g in range(T // 16):
                worst = max(worst, 16 - len(set((a[16 * g:16 * g + 16] % 16).tolist())))
        for j in range(8):
            a = r(i, j)
            for g in range(T // 32):
                worst = max(worst, 32 - len(set((a[32 * g:32 * g + 32] % 32).tolist())))
            if p != 1:                                       # wave-local: the reader's wave wrote the element
                assert all(owner[e] >> 6 == t >> 6 for t, e in zip(i.tolist(), a.tolist()))
    return worst


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    x = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    for sign in (+1, -1):
        ref = np.fft.ifft(x) * M if sign > 0 else np.fft.fft(x)
        print("sub4096 radix-8 wave-local, sign", sign, np.abs(subfft4096_r8w(x, sign) - ref).max() / np.abs(ref).max())
    print("bank conflicts (extra lanes on a busy bank, worst instruction):", conflicts_and_locality())
    for Pn in (4, 16):
        N = Pn * M
        X = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        ref = np.fft.ifft(X) * N
        acc = np.zeros((4, M), complex)
        n1 = np.arange(M)
        for k2 in range(Pn):
            sub = subfft4096_r8w(X[k2::Pn], +1)
            tw = np.exp(2j * np.pi * (n1 % 512) * k2 / N) * np.exp(2j * np.pi * (n1 // 512) * k2 / (N // 512))
            for q in range(4):
                acc[q] += sub * tw * np.exp(2j * np.pi * q * k2 / Pn)
        print("bwd N=%d first 16384 lags" % N, np.abs(acc.reshape(-1) - ref[:4 * M]).max() / np.abs(ref).max())
