"""The oracle against INDEPENDENT float64 definitions on random shapes (CPU only, seeded, a few seconds).

The fixed cases of test_oracle_dsp.py / test_oracle_kiss.py pin the oracle at chosen sizes; the randomised device-against-oracle run
(tests/fuzz_parity.py) cannot see an error both sides share.  Here numpy's float64 FFT and correlation define the answers, at shapes drawn
at random: any transform size up to 6000 (primes included), any tap count / decimation, overlap-save against the direct correlation,
the channelizer against its polyphase definition, the chain against FIR-then-FFT.  Tolerances are the f32 rounding bounds, written out."""
import numpy as np
import pytest

EPS = 2.0 ** -24


def largest_prime_factor(n):
    p, m = 1, n
    f = 2
    while f * f <= m:
        while m % f == 0:
            p, m = f, m // f
        f += 1
    return max(p, m) if m > 1 else p


def test_fft_any_size_against_float64(oracle):
    rng = np.random.default_rng(20260401)
    sizes = [int(v) for v in rng.integers(1, 6000, 40)] + [2, 3, 5, 7, 11, 13, 4096, 3125, 2187, 5999, 4999, 30 * 49, 17 * 19]
    for n in sizes:
        x = oracle.synth_iq(n, 0, 2 * n)
        for inv in (False, True):
            got = oracle.fft(x, n, inverse=inv).astype(np.complex128).reshape(2, n)
            xs = x.astype(np.complex128).reshape(2, n)
            want = np.fft.ifft(xs, axis=1) * n if inv else np.fft.fft(xs, axis=1)
            # radix 2/3/4/5 stages: error grows with log n; a generic-radix stage of prime p sums p terms per output
            tol = 2e-6 * max(1.0, largest_prime_factor(n) / 8.0)
            err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
            assert err <= tol, (n, inv, err, tol)


def test_fir_any_shape_against_float64(oracle):
    rng = np.random.default_rng(20260402)
    for _ in range(60):
        k = int(rng.choice([1, 2, 3, 63, 127, int(rng.integers(1, 600))])); d = int(rng.integers(1, 14))
        n = k - 1 + int(rng.integers(0, 5000)); cplx = bool(rng.integers(0, 2))
        taps = oracle.synth_f32(int(rng.integers(1, 1 << 30)), 0, k)
        x = (oracle.synth_iq if cplx else oracle.synth_f32)(int(rng.integers(1, 1 << 30)), 0, n)
        want = np.correlate(x.astype(np.complex128 if cplx else np.float64), taps.astype(np.float64), "valid")[::d] if n >= k else np.zeros(0)
        bound = k * EPS * np.correlate(np.abs(x).astype(np.float64), np.abs(taps).astype(np.float64), "valid")[::d] if n >= k else np.zeros(0)
        for fused in (False, True):
            got = oracle.fir(x, taps, d, fused=fused)
            assert got.shape == want.shape, (k, d, n, cplx)
            assert np.all(np.abs(got - want) <= 2.0 * bound + 1e-30), (k, d, n, cplx, fused)   # K eps sum |u v| (x 2 for the complex magnitude)


def test_overlap_save_against_direct_correlation(oracle):
    rng = np.random.default_rng(20260403)
    for _ in range(25):
        nfft = int(rng.choice([64, 256, 1024, 4096, 1000, 30, int(rng.integers(2, 3000))])); k = int(rng.integers(1, nfft + 1)); hop = nfft - k + 1
        n = nfft + int(rng.integers(0, 5)) * hop + int(rng.integers(0, hop))
        h = oracle.synth_f32(int(rng.integers(1, 1 << 30)), 0, k)
        x = oracle.synth_iq(int(rng.integers(1, 1 << 30)), 0, n)
        got = oracle.overlap_save(x, h, nfft)
        direct = np.correlate(x.astype(np.complex128), h.astype(np.float64), "valid")
        assert len(got) == ((n - nfft) // hop + 1) * hop <= len(direct)
        scale = np.abs(x).max() * np.abs(h).sum()
        # two transforms of nfft points and a product: a few eps log2(nfft) of the largest possible output (generic-radix sizes more)
        tol = 4e-6 * max(1.0, largest_prime_factor(nfft) / 8.0) * scale
        assert np.abs(got - direct[: len(got)]).max() <= tol, (nfft, k, n)


def test_channelizer_against_its_polyphase_definition(oracle):
    rng = np.random.default_rng(20260404)
    for _ in range(25):
        M = int(rng.choice([64, 32, 7, 100, int(rng.integers(1, 200))])); P = int(rng.choice([4, 8, 16, int(rng.integers(1, 12))])); rows = int(rng.integers(1, 60))
        h = oracle.synth_f32(int(rng.integers(1, 1 << 30)), 0, M * P)
        x = oracle.synth_iq(int(rng.integers(1, 1 << 30)), 0, M * (P - 1 + rows) + int(rng.integers(0, M)))
        X = x[: M * (P - 1 + rows)].astype(np.complex128).reshape(P - 1 + rows, M)
        H = h.astype(np.float64).reshape(P, M)
        branch = np.stack([(X[r:r + P] * H).sum(axis=0) for r in range(rows)])     # y[r][m] = sum_p x[(r + p) M + m] h[p M + m]
        want = np.fft.fft(branch, axis=1)                                           # kissfft forward across the branches
        for fused in (False, True):
            got = oracle.pfb_channelizer(x, h, M, P, fused)
            assert got.shape == (rows, M)
            tol = 4e-6 * max(1.0, largest_prime_factor(M) / 8.0) * np.abs(x).max() * np.abs(h).sum()
            assert np.abs(got - want).max() <= tol, (M, P, rows, fused)


def test_chain_is_fir_then_fft(oracle):
    rng = np.random.default_rng(20260405)
    for _ in range(12):
        k = int(rng.integers(1, 200)); d = int(rng.integers(1, 9)); nfft = int(rng.choice([1024, 64, 256, 100, int(rng.integers(1, 700))])); nb = int(rng.integers(1, 5))
        taps = oracle.synth_f32(int(rng.integers(1, 1 << 30)), 0, k)
        x = oracle.synth_iq(int(rng.integers(1, 1 << 30)), 0, nb * nfft * d + (k - d) + int(rng.integers(0, nfft * d)))
        for fused in (False, True):
            y = oracle.fir(x, taps, d, fused=fused)
            want = oracle.fft(y[: (len(y) // nfft) * nfft], nfft).reshape(-1, nfft)
            got = oracle.chain_fir_fft(x, taps, d, nfft, fused=fused)
            assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, d, nfft, nb, fused)


@pytest.mark.parametrize("conv", [0, 1, 2])
def test_resampler_reconstructs_a_tone(oracle, conv):
    """samplerate::resample is a band-limited interpolator: output j is the input signal at time j / ratio.  A tone well inside the pass
    band comes back as the same tone on the new grid -- amplitude, frequency AND phase (no hidden delay: the converter's latency shows as
    outputs withheld at the end of a call, SURVEY.md 8a A6, not as a shift) -- to 5e-5 once the left wing is full of signal."""
    tab, half, inc = oracle.src_table(conv)
    for ratio in (0.02, 0.3, 0.5, 1.0, 48000 / 44100, 2.0, 3.7):
        f0 = 0.11 * min(1.0, ratio)
        reach = half / inc / min(1.0, ratio)
        n = int(max(20000, 3 * reach + 10000))
        x = np.sin(2 * np.pi * f0 * np.arange(n)).astype(np.float32)
        err, y, used = oracle.Resampler(conv).process(x, ratio, int(ratio * n + 1))
        assert err == 0 and used == n and len(y) > ratio * (n - 2 * reach) - 2
        j0 = int(ratio * (reach + 5)) + 2
        want = np.sin(2 * np.pi * f0 * np.arange(len(y)) / ratio)
        assert np.abs(y[j0:] - want[j0:]).max() <= 5e-5, (conv, ratio)
