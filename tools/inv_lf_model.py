"""Design check of round 6's inverse transform (csrc/fft512.hpp: inv_q1 / inv_q2 / inv_q3, numpy only):  python tools/inv_lf_model.py

The inverse of the folded negacyclic transform, z_j = psi^-j sum_k A_k W^(-jk)  (psi = e^(i pi/1024), W = psi^4, M = 512; the 1/M lives in
the key spectrum), decimated over the INPUT index k = k0 + 8 k1 + 64 k2 so that every inter-pass twiddle sits on the INPUT side of a
pass and rides inside Linzer-Feig butterflies (a +- W b), exactly as in the forward transform:
    -j (4k + 1) = -256 j k2 - 32 j k1 - 4 j k0 - j,        j = 64 j2 + 8 j1 + j0 = 64 j2 + L
    pass 1 (arrangement F, lane'' = (k0, k1), register k2 -> j0):  psi^(-256 j k2) = w^(-j0 k2)              plain inverse DFT8
    pass 2 (lane' = (k0, j0), register k1 -> j1):                  psi^(-32 j k1) = (zeta2 w^-j1)^k1,  zeta2 = psi^(-32 j0)
    pass 3 (lane = L, register k0 -> j2):                          psi^(-4 j k0)  = (zeta3 w^-j2)^k0,  zeta3 = psi^(-4 L)
    output twist psi^(-(L + 64 j2)) = c (1 - i t): merged with the rounding — p = x + t y (1 FMA), fma(c, p, 1.5 * 2^52) (1 FMA)
The tangent form W = c (1 + i t) is singular at odd multiples of pi/2, and zeta2^4 = -i for j0 = 4, zeta3^4 = -i for L = 32.  Scaling the
INPUT of the inverse by a unit phase per frequency is free (it goes into the key spectrum at init), and an input-side factor phi^-k1
turns zeta2 into zeta2 phi: with gamma_k = psi^(16 k1 + 2 k0) the passes run with zeta2' = psi^(-(32 j0 + 16)), zeta3' = psi^(-(4 L + 2)),
whose powers stay clear of +-pi/2 (closest: 0.70 degrees, |t| <= 81.5 — the forward transform has the same).
Checks: (1) the algebra against the O(n^2) definition; (2) the twisted DFT8 of fft512.hpp (natural-order X[k] = sum_m x[m] (zeta w^k)^m)
used with output k -> -k; (3) the angles' distance from the singularities; (4) the instruction budget."""
import numpy as np

M = 512
psi = np.exp(1j * np.pi / 1024)
w = np.exp(1j * np.pi / 4)
rng = np.random.default_rng(6)


def tdft8(x, zeta):
    """fft512.hpp tdft8: X[k] = sum_m x[m] (zeta w^k)^m, natural order"""
    return np.array([sum(x[m] * (zeta * w**k) ** m for m in range(8)) for k in range(8)])


def inverse_new(A):
    gamma = np.array([psi ** (16 * ((k >> 3) & 7) + 2 * (k & 7)) for k in range(M)])
    A = A * gamma                                   # at init, inside the key spectrum
    a = A.reshape(8, 8, 8)                          # [k2][k1][k0]
    B = np.zeros((8, 8, 8), complex)                # [k0][k1][j0]
    for k0 in range(8):
        for k1 in range(8):
            x = a[:, k1, k0]
            B[k0, k1] = [sum(x[k2] * w ** (-j0 * k2) for k2 in range(8)) for j0 in range(8)]
    C = np.zeros((8, 8, 8), complex)                # [k0][j1][j0]
    for k0 in range(8):
        for j0 in range(8):
            X = tdft8(B[k0, :, j0], psi ** (-(32 * j0 + 16)))
            C[k0, :, j0] = [X[(8 - j1) % 8] for j1 in range(8)]      # inverse sign = output index negated
    z = np.zeros(M, complex)
    for L in range(64):
        j1, j0 = L >> 3, L & 7
        X = tdft8(C[:, j1, j0], psi ** (-(4 * L + 2)))
        for j2 in range(8):
            z[L + 64 * j2] = psi ** (-(L + 64 * j2)) * X[(8 - j2) % 8]
    return z


def inverse_def(A):
    j = np.arange(M)
    return np.array([psi ** (-jj) * np.sum(A * psi ** (-4.0 * jj * np.arange(M))) for jj in j])


A = rng.normal(size=M) + 1j * rng.normal(size=M)
err = np.abs(inverse_new(A) - inverse_def(A)).max()
print(f"(1)+(2) max |new - definition| = {err:.2e}")
assert err < 1e-9

worst = 90.0
tmax = 0.0
for name, angles in (("pass 2", [-(32 * j0 + 16) for j0 in range(8)]), ("pass 3", [-(4 * L + 2) for L in range(64)])):
    for num in angles:
        for mult, plus in ((4, 0), (2, 0), (1, 0), (1, 256)):
            deg = (mult * num + plus) * 180.0 / 1024
            d = min(abs(((deg - 90) + 180) % 360 - 180), abs(((deg + 90) + 180) % 360 - 180))
            worst = min(worst, d)
            tmax = max(tmax, abs(np.tan(np.deg2rad(deg))))
print(f"(3) butterfly angles: closest to +-90 degrees = {worst:.2f} degrees, largest |tan| = {tmax:.1f};"
      f" output twist: largest tan = {np.tan(np.pi * 511 / 1024):.0f} (c |1 + i t| = 1: the bound does not depend on it)")
assert worst > 0.5   # the forward transform lives with the same 0.70 degrees (t = -81.5): Lemma 1' does not depend on t

old = 56 * 3 + 28 + 32 + 28 + 16      # three DFT8, conj T2 (7), conj T1 (8), conj twist (6 + rotation), 16 rounding additions
new = 56 + 72 + 72 + 32                # plain DFT8, two twisted DFT8 of 6-instruction butterflies, twist + rounding as 2 FMAs per word
print(f"(4) arithmetic per inverse transform incl. rounding: {old} -> {new}; per CMUX step (4 inverses): {4 * (old - new)} fewer")
