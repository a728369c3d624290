"""A THIRD restatement of the gate bootstrap, written from the formulas of SURVEY.md section 8 (a-ext) in plain numpy —
no C, no shared code with oracle/*.c or the HIP kernels.  TEST INFRASTRUCTURE ONLY (like everything under oracle/): it
pins the two C restatements at formula level — mod-switch, gadget decomposition (offset + rounding), CMUX by
(X^abar - 1), row order r = c*l + j, sample extraction, identity key-switch digit extraction — on the rows
oracle_lib.adversarial_rows builds and on fresh encryptions (tests/test_numpy_restatement.py).

What each function follows (call sites in the reference; the arithmetic itself lives in the un-vendored TFHEpp):
  gate linear step         TFHEpp::HomNAND & co as called at /root/reference/src/iyokan_tfhepp.hpp:131-144
  bootstrap()              GateBootstrappingTLWE2TLWEFFT: mod-switch, BlindRotate (CMUXFFTwithPolynomialMulByXaiMinusOne),
                           SampleExtractIndex(0); then IdentityKeySwitch<lvl10param>
Exactness: the negacyclic products are integer convolutions of 6..10-bit digits with 32-bit words.  They are computed by
float64 FFTs of length 2N on the 16-bit halves of the key words; every partial sum is below 2^35, five orders of
magnitude inside the range where the rounded FFT result is the exact integer (checked against a direct O(N^2) product
in the test).
"""
import numpy as np

U32 = np.uint32
MASK32 = (1 << 32) - 1


def _log2(x):
    return int(x).bit_length() - 1


def modswitch(lin, N):
    """(abar[0..n), bbar): abar_i = round(a_i * 2N / 2^32) mod 2N (add half an output step, drop the low bits, in uint32
    arithmetic: the sum wraps), bbar = 2N - trunc(b * 2N / 2^32)  (TFHEpp rounds a, truncates b)."""
    sh = 32 - 1 - _log2(N)
    a = np.asarray(lin[:-1], dtype=np.uint64)
    abar = (((a + (1 << (sh - 1))) & MASK32) >> sh).astype(np.int64)
    bbar = (2 * N - (int(lin[-1]) >> sh)) & (2 * N - 1)
    return abar, bbar


def mul_by_xai(poly, a, N):
    """X^a * poly mod (X^N + 1) on uint32 coefficients, 0 <= a < 2N."""
    a = int(a) % (2 * N)
    p = poly if a < N else (-poly.astype(np.int64)).astype(np.uint64) & MASK32
    s = a % N
    out = np.empty(N, dtype=np.uint64)
    out[s:] = p[: N - s]
    out[:s] = (-p[N - s:].astype(np.int64)) & MASK32
    return out.astype(np.uint64) & MASK32


def decompose(poly, l, Bgbit):
    """Signed gadget digits d_j in [-Bg/2, Bg/2), j = 1..l, of every coefficient (uint32 in, int64 [l][N] out):
    offset = sum_j (Bg/2) 2^(32 - j Bgbit), round = 2^(32 - l Bgbit - 1); d_j = ((p + offset + round) >> (32 - j Bgbit)) mod Bg - Bg/2."""
    Bg = 1 << Bgbit
    offset = sum((Bg // 2) << (32 - j * Bgbit) for j in range(1, l + 1))
    rnd = 1 << (32 - l * Bgbit - 1)
    t = (poly.astype(np.uint64) + offset + rnd) & MASK32
    return np.stack([((t >> (32 - j * Bgbit)) & (Bg - 1)).astype(np.int64) - Bg // 2 for j in range(1, l + 1)])


class KeyFFT:
    """FFTs of the 16-bit halves of every BK polynomial, computed on first use per CMUX step."""

    def __init__(self, bk, p):
        self.p = p
        self.bk = bk.reshape(p.n, (p.k + 1) * p.l, p.k + 1, p.N)
        self.cache = {}

    def step(self, i):
        if i not in self.cache:
            w = self.bk[i].astype(np.uint64)
            lo = np.fft.rfft((w & 0xFFFF).astype(np.float64), 2 * self.p.N, axis=-1)
            hi = np.fft.rfft((w >> 16).astype(np.float64), 2 * self.p.N, axis=-1)
            self.cache[i] = (lo, hi)
        return self.cache[i]


def external_product(digits, keyfft_i, N):
    """res_c = sum_r d_r (*) BK_i[r][c] mod (X^N + 1, 2^32); digits int64 [rows][N]; returns uint64 [k+1][N]."""
    lo, hi = keyfft_i
    D = np.fft.rfft(digits.astype(np.float64), 2 * N, axis=-1)               # [rows][N+1]
    out = []
    for part in (lo, hi):
        lin = np.fft.irfft(np.einsum("rf,rcf->cf", D, part), 2 * N, axis=-1)  # linear convolution, length 2N
        out.append(np.rint(lin[:, :N] - lin[:, N:]).astype(np.int64))        # folded: X^N = -1
    return ((out[0] + (out[1] << 16)) & MASK32).astype(np.uint64)


def negacyclic_direct(d, w, N):
    """O(N^2) reference for external_product's FFT: d int64 [N], w uint32 [N] -> uint64 [N] mod 2^32 (Python integers)."""
    res = [0] * N
    for i in range(N):
        di = int(d[i])
        if di == 0:
            continue
        for j in range(N):
            k = i + j
            v = di * int(w[j])
            if k < N:
                res[k] += v
            else:
                res[k - N] -= v
    return np.array([r & MASK32 for r in res], dtype=np.uint64)


def blind_rotate(lin, p, keyfft):
    """acc = (0, X^bbar * mu (1 + X + ... + X^(N-1))); for i < n with abar_i != 0: acc += BK_i [.] ((X^abar_i - 1) acc)."""
    N, l = p.N, p.l
    abar, bbar = modswitch(lin, N)
    tv = np.full(N, p.mu, dtype=np.uint64)
    acc = [np.zeros(N, dtype=np.uint64), mul_by_xai(tv, bbar, N)]
    for i in range(p.n):
        a = int(abar[i])
        if a == 0:
            continue
        diff = [(mul_by_xai(acc[c], a, N) - acc[c]) & MASK32 for c in range(2)]
        digits = np.concatenate([decompose(diff[c], l, p.Bgbit) for c in range(2)])   # row r = c * l + (j - 1)
        res = external_product(digits, keyfft.step(i), N)
        acc = [(acc[c] + res[c]) & MASK32 for c in range(2)]
    return acc


def sample_extract0(acc, N):
    """TLWE lvl1 (a'[0..N), b'): a'[0] = a[0], a'[j] = -a[N - j], b' = b[0]."""
    a, b = acc
    out = np.empty(N + 1, dtype=np.uint64)
    out[0] = a[0]
    out[1:N] = (-a[N - 1:0:-1].astype(np.int64)) & MASK32
    out[N] = b[0]
    return out.astype(U32)


def keyswitch(t1, ksk, p):
    """(0, ..., 0, b') - sum_i sum_j KSK[i][j][v_ij - 1], v_ij = j-th basebit-wide digit (from the top) of a'_i + prec,
    prec = 2^(32 - (1 + basebit t)); digit 0 subtracts nothing."""
    N, t, bb, n = p.N, p.t, p.basebit, p.n
    base = 1 << bb
    rows = ksk.reshape(N, t, base - 1, n + 1)
    prec = 1 << (32 - (1 + bb * t))
    at = (t1[:N].astype(np.uint64) + prec) & MASK32
    out = np.zeros(n + 1, dtype=np.uint64)
    out[n] = int(t1[N])
    for j in range(t):
        v = ((at >> (32 - (j + 1) * bb)) & (base - 1)).astype(np.int64)
        idx = np.nonzero(v)[0]
        if len(idx):
            out = (out - rows[idx, j, v[idx] - 1].astype(np.uint64).sum(axis=0)) & MASK32
    return out.astype(U32)


def keyswitch_pair_table(ksk, p):
    """The table of iyokan_amd/csrc/kernels.hpp: ks_lut_build_kernel, restated (round 6).  The t 2-bit digits are taken in PAIRS from
    the most significant one; stage s of coefficient i holds the 16 sums
        table[i][16 s + ((v_h << 2) | v_l)] = (v_h ? KSK[i][2s][v_h-1] : 0) + (v_l ? KSK[i][2s+1][v_l-1] : 0)      (mod 2^32)
    and an odd t ends in a stage of 4 rows for its last digit: 16 (t // 2) + 4 (t % 2) rows of n + 1 words per coefficient."""
    N, t, bb, n = p.N, p.t, p.basebit, p.n
    assert bb == 2
    rows = ksk.reshape(N, t, 3, n + 1).astype(np.uint64)
    zero = np.zeros((N, 1, n + 1), dtype=np.uint64)
    cand = [np.concatenate([zero, rows[:, j]], axis=1) for j in range(t)]          # [N][4][n + 1]: digit value -> row (0 -> zeros)
    out = []
    for s in range(t // 2):
        hi, lo = cand[2 * s], cand[2 * s + 1]
        out.append(((hi[:, :, None, :] + lo[:, None, :, :]) & MASK32).reshape(N, 16, n + 1))   # index (v_h << 2) | v_l
    if t % 2:
        out.append(cand[t - 1])
    return np.concatenate(out, axis=1).astype(U32)


def keyswitch_by_table(t1, table, p):
    """keyswitch() through keyswitch_pair_table(): one row per digit PAIR, selected by its 4-bit value — what
    keyswitch_lut_kernel computes.  Equal to keyswitch() word for word (integer sums commute mod 2^32)."""
    N, t, bb, n = p.N, p.t, p.basebit, p.n
    dbits = bb * t
    prec = 1 << (32 - (1 + dbits))
    d = ((t1[:N].astype(np.uint64) + prec) & MASK32) >> (32 - dbits)                # all t digits of a'_i + prec
    out = np.zeros(n + 1, dtype=np.uint64)
    out[n] = int(t1[N])
    idx = np.arange(N)
    for s in range(t // 2):
        v = ((d >> (dbits - 4 * (s + 1))) & 15).astype(np.int64)
        out = (out - table[idx, 16 * s + v].astype(np.uint64).sum(axis=0)) & MASK32
    if t % 2:
        v = (d & 3).astype(np.int64)
        out = (out - table[idx, 16 * (t // 2) + v].astype(np.uint64).sum(axis=0)) & MASK32
    return out.astype(U32)


GATE_COEFFS = {  # (sa, sb, offset in units of mu): out = sa * ca + sb * cb + (0, ..., 0, off)
    "NAND": (-1, -1, +1), "AND": (+1, +1, -1), "OR": (+1, +1, +1), "NOR": (-1, -1, -1),
    "XOR": (+2, +2, +2), "XNOR": (-2, -2, -2), "ANDNOT": (+1, -1, -1), "ORNOT": (+1, -1, +1),
}


def linear(kind, ca, cb, p):
    sa, sb, off = GATE_COEFFS[kind]
    out = (sa * ca.astype(np.int64) + sb * cb.astype(np.int64)) & MASK32
    out[-1] = (int(out[-1]) + off * p.mu) & MASK32
    return out.astype(U32)


def bootstrap(lin, p, keyfft, ksk):
    """lvl0 TLWE (after the gate's linear step) -> bootstrapped lvl0 TLWE; also returns the lvl1 TLWE in between."""
    t1 = sample_extract0(blind_rotate(lin, p, keyfft), p.N)
    return keyswitch(t1, ksk, p), t1


def phase0(ct, s0):
    """b - <a, s0> mod 2^32 as a signed fraction of the torus in [-1/2, 1/2)."""
    ph = (int(ct[-1]) - int((ct[:-1].astype(np.uint64) * s0.astype(np.uint64)).sum() & MASK32)) & MASK32
    return (ph - (1 << 32) if ph >= (1 << 31) else ph) / 2.0 ** 32


def ksk_row_noise(keys):
    """Phase noise e[i][j][v] of every key-switching row of THIS key (needs the secret keys; torus units)."""
    p = keys.params
    nb = (1 << p.basebit) - 1
    rows = keys.ksk.reshape(-1, p.n + 1)
    inner = (rows[:, :-1] * keys.s0.astype(U32)[None, :]).sum(axis=1, dtype=U32)
    ph = (rows[:, -1] - inner).astype(U32).reshape(p.N, p.t, nb)
    msg = np.zeros((p.N, p.t, nb), dtype=np.uint64)
    for j in range(p.t):
        for v in range(nb):
            msg[:, j, v] = (keys.s1.astype(np.uint64) * (v + 1)) << (32 - (j + 1) * p.basebit)
    return ((ph.astype(np.uint64) - msg) & MASK32).astype(U32).astype(np.int32).astype(np.float64) / 2.0 ** 32


def predicted_output_stats(keys):
    """Mean and variance (over ciphertexts, for THIS key) of the phase error of a bootstrapped output, torus units.

    A key's noise terms are drawn once, so over many ciphertexts they are constants that the data-dependent digits select:
      key switch   cell (i, j) subtracts nothing (digit 0) or row v in {1 .. 2^basebit - 1}, each with probability
                   2^-basebit: contribution -X_ij with X uniform on {0, e_ij1, ..}: the MEAN -sum_ij avg(X) is a fixed,
                   key-dependent bias (~ sqrt(3 N t / 16) alpha0, about one output sigma), the variance is
                   sum_ij Var(X) (9/16 N t alpha0^2 on average, not the 3/4 N t alpha0^2 of a fresh-noise model);
                   + hw(s1) 2^(-2 t basebit) / 12 for rounding a'_i to t digits;
      blind rotate n (k+1) l N Var(d) alpha1^2 with d uniform on [-Bg/2, Bg/2) (its mean -1/2 times the sum of the
                   key's row noises is a bias ~ 10^-2 sigma: ignored), + hw(s0) (1 + hw(s1)) 2^(-2 l Bgbit) / 12 for the
                   gadget rounding at the steps whose key bit is 1.
    (Shapes of Chillotti-Gama-Georgieva-Izabachene, J. Cryptology 2020, Thm 4.3 / Lemma 4.6, specialised to a fixed key.)"""
    p = keys.params
    e = ksk_row_noise(keys)
    q = 2.0 ** -p.basebit
    cell_mean = e.sum(axis=2) * q
    cell_var = (e ** 2).sum(axis=2) * q - cell_mean ** 2
    hw0, hw1 = int(np.count_nonzero(keys.s0)), int(np.count_nonzero(keys.s1))
    ks_mean = -float(cell_mean.sum())
    ks_var = float(cell_var.sum()) + hw1 * 2.0 ** (-2 * p.t * p.basebit) / 12
    Bg = 1 << p.Bgbit
    var_d = sum(d * d for d in range(-Bg // 2, Bg // 2)) / Bg - 0.25
    br_var = p.n * (p.k + 1) * p.l * p.N * var_d * p.alpha1 ** 2 + hw0 * (1 + hw1) * 2.0 ** (-2 * p.l * p.Bgbit) / 12
    return ks_mean, br_var + ks_var, br_var, ks_var


def phase_errors(cts, bits, s0, mu):
    """Signed phase error of every ciphertext (rows of n + 1 words) encrypting bits[i] as +-mu, in torus units."""
    cts = np.ascontiguousarray(cts, dtype=U32)
    inner = (cts[:, :-1] * s0.astype(U32)[None, :]).sum(axis=1, dtype=U32)                 # wraps mod 2^32
    target = np.where(np.asarray(bits) != 0, U32(mu), U32((1 << 32) - mu)).astype(U32)
    err = (cts[:, -1] - inner - target).astype(U32)                                         # mod 2^32
    return err.astype(np.int32).astype(np.float64) / 2.0 ** 32


def check_noise_against_cggi(keys, cts, bits, rel_tol):
    """Noise KAT: the outputs' phase errors must have the mean and the variance that CGGI's analysis
    predicts for these parameters and THIS key (predicted_output_stats) — |mean - prediction| < 4 sigma / sqrt(n), variance
    within rel_tol.  This is the one in-container check that sees a wrong rounding offset (a bias of many sigma), a
    missing `prec` in the key switch (a mean shift of hw(s1) 2^-(t basebit + 1), more than one output sigma where the
    tolerance is sigma / 10 or tighter) or a mis-scaled gadget row that still decrypts."""
    p = keys.params
    e = phase_errors(cts, bits, keys.s0, p.mu)
    n = len(e)
    pmean, pvar, br, ks = predicted_output_stats(keys)
    mean, var = float(e.mean()), float(e.var())
    assert abs(mean - pmean) < 4.0 * np.sqrt(pvar / n) + 0.02 * np.sqrt(pvar), (mean, pmean, np.sqrt(pvar / n))
    assert abs(var / pvar - 1.0) < rel_tol, (var, pvar, br, ks)
    assert np.abs(e - pmean).max() < 8.0 * np.sqrt(pvar), (np.abs(e - pmean).max(), np.sqrt(pvar))   # no outlier
    return mean, var, pmean, pvar
