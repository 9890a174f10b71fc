"""Signed zeros, subnormals, infinities, NaNs and magnitudes whose products overflow or underflow, through every floating-point operator.

The kernels keep the reference's operations in the reference's order, so IEEE-754 decides these cases the same way on both sides:
a skipped multiplication by a unit twiddle, an accumulator initialised with the first product instead of 0 + product, a flushed
subnormal or a fused operation where the reference rounds twice would all pass the random-data parity tests and fail here.
Bar: identical bits wherever the oracle's value is not a NaN; a NaN exactly where the oracle has one (payloads are not compared:
x86 and gfx950 propagate different payloads, and kissfft / dsputils define none)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPECIALS = np.array([0.0, -0.0, 1e-40, -1e-40, 1.4e-45, 1e-30, -1e-30, 1e30, -1e30, 3e38, -3e38, np.inf, -np.inf, np.nan,
                     1.0, -1.0, 0.5, 2.0 ** -126, -(2.0 ** -126), 2.0 ** 127], np.float32)


def same_special(got, want):
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    if got.shape != want.shape:
        return False
    g, w = got.view(np.float32).reshape(-1), want.view(np.float32).reshape(-1)
    wn = np.isnan(w)
    return np.array_equal(np.isnan(g), wn) and np.array_equal(g.view(np.uint32)[~wn], w.view(np.uint32)[~wn])


def where_differs(got, want, limit=6):
    g, w = np.ascontiguousarray(got).view(np.float32).reshape(-1), np.ascontiguousarray(want).view(np.float32).reshape(-1)
    bad = np.flatnonzero((np.isnan(g) != np.isnan(w)) | (~np.isnan(w) & (g.view(np.uint32) != w.view(np.uint32))))
    return [(int(i), float(g[i]), float(w[i])) for i in bad[:limit]], len(bad)


def sprinkle(oracle, seed, n, cplx, density, scale=1.0):
    """the hash stream with special values written over a fraction `density` of its words"""
    rng = np.random.default_rng(seed)
    x = (oracle.synth_iq(seed, 0, n) if cplx else oracle.synth_f32(seed, 0, n)) * np.float32(scale)
    w = x.view(np.float32).reshape(-1)
    k = max(1, int(density * len(w)))
    w[rng.integers(0, len(w), k)] = SPECIALS[rng.integers(0, len(SPECIALS), k)]
    return x


FINITE = SPECIALS[np.isfinite(SPECIALS)]


def sprinkle_finite(oracle, seed, n, cplx, density):
    rng = np.random.default_rng(seed)
    x = oracle.synth_iq(seed, 0, n) if cplx else oracle.synth_f32(seed, 0, n)
    w = x.view(np.float32).reshape(-1)
    k = max(1, int(density * len(w)))
    w[rng.integers(0, len(w), k)] = FINITE[rng.integers(0, len(FINITE), k)]
    return x


@pytest.mark.parametrize("k,d", [(127, 5), (63, 1), (63, 5), (127, 1), (100, 3), (7, 13), (3, 1), (1, 1), (1000, 2)])
@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("fused", [False, True])
def test_fir_special_values(gpu, redio, oracle, k, d, cplx, fused):
    n = 40000
    taps = oracle.synth_f32(77 + k, 0, k)
    step = max(1, k // 5)
    taps[::step] = np.resize(np.array([0.0, -0.0, 1e-30, 1e30, -1.0, 1e-40], np.float32), len(taps[::step]))
    for seed, maker, dens in ((1, sprinkle_finite, 0.002), (2, sprinkle_finite, 0.2), (3, sprinkle, 0.0005), (4, sprinkle, 0.05)):
        x = maker(oracle, seed * 1000 + k, n, cplx, dens)
        got = redio.Fir(taps, d, complex_input=cplx, fused=fused)(gpu.from_numpy(x).cuda()).cpu().numpy()
        want = oracle.fir(x, taps, d, fused=fused)
        assert same_special(got, want), (k, d, cplx, fused, seed, where_differs(got, want))


@pytest.mark.parametrize("nfft", [4, 16, 64, 256, 1024, 4096, 16384, 65536, 2, 8, 32, 128, 512, 2048, 8192, 32768, 131072, 1 << 18, 1 << 19, 1 << 20,
                                  3, 5, 15, 100, 1000, 1536, 6144, 7, 49, 20000])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft_special_values(gpu, redio, oracle, nfft, inverse):
    """kissfft multiplies by every twiddle, the unit ones too: 0 x inf = NaN and (-0) - (+0) = -0 are part of its results"""
    nb = max(3, 8192 // nfft)
    plan = redio.Fft(nfft, inverse=inverse)
    for seed, maker, dens in ((1, sprinkle_finite, 0.01), (2, sprinkle_finite, 0.3), (3, sprinkle, 2.0 / (nfft * nb)), (4, sprinkle, 0.02)):
        x = maker(oracle, seed * 77 + nfft, nfft * nb, True, dens)
        x[:nfft] = 0  # a block of +0 and a block of -0: every product and sum of signed zeros
        x[nfft:2 * nfft] = np.complex64(complex(-0.0, -0.0))
        got = plan(gpu.from_numpy(x).cuda()).cpu().numpy()
        want = oracle.fft(x, nfft, inverse=inverse)
        assert same_special(got, want), (nfft, inverse, seed, where_differs(got, want))


@pytest.mark.parametrize("k,d", [(127, 5), (63, 5), (127, 3), (127, 1), (63, 1), (31, 2)])
@pytest.mark.parametrize("fused", [False, True])
def test_chain_special_values(gpu, redio, oracle, k, d, fused):
    taps = oracle.lpf_corrected(k, 0.08)
    n = 1024 * d * 9 + k + 11
    for seed, maker, dens in ((1, sprinkle_finite, 0.01), (2, sprinkle_finite, 0.3), (3, sprinkle, 0.0002), (4, sprinkle, 0.02)):
        x = maker(oracle, seed * 31 + k, n, True, dens)
        got = redio.Chain(taps, d, 1024, fused=fused)(gpu.from_numpy(x).cuda()).cpu().numpy()
        want = oracle.chain_fir_fft(x, taps, d, 1024, fused=fused)
        assert same_special(got, want), (k, d, fused, seed, where_differs(got, want))


@pytest.mark.parametrize("M,P", [(64, 16), (64, 4), (32, 8), (256, 16), (1024, 4), (48, 5)])
def test_channelizer_special_values(gpu, redio, oracle, M, P):
    h = oracle.lpf_corrected(M * P, 0.45 / M)
    n = M * (P - 1 + 300) + 5
    for seed, maker, dens in ((1, sprinkle_finite, 0.01), (2, sprinkle_finite, 0.3), (3, sprinkle, 0.0005), (4, sprinkle, 0.02)):
        x = maker(oracle, seed * 13 + M, n, True, dens)
        got = redio.Channelizer(h, M, P)(gpu.from_numpy(x).cuda()).cpu().numpy()
        want = oracle.pfb_channelizer(x, h, M, P, fused=True)
        assert same_special(got, want), (M, P, seed, where_differs(got, want))


@pytest.mark.parametrize("nfft,k", [(1024, 127), (4096, 1025), (16384, 127), (65536, 8193), (32768, 127), (8192, 127), (2048, 513), (131072, 127), (1000, 101)])
def test_overlap_save_special_values(gpu, redio, oracle, nfft, k):
    h = oracle.lpf_corrected(k, 0.08)
    hop = nfft - k + 1
    n = nfft + 4 * hop + 17
    for seed, maker, dens in ((1, sprinkle_finite, 0.01), (2, sprinkle_finite, 0.3), (3, sprinkle, 1.0 / n), (4, sprinkle, 0.02)):
        x = maker(oracle, seed * 7 + nfft, n, True, dens)
        got = redio.OverlapSave(h, nfft)(gpu.from_numpy(x).cuda()).cpu().numpy()
        want = oracle.overlap_save(x, h, nfft)
        assert same_special(got, want), (nfft, k, seed, where_differs(got, want))


@pytest.mark.parametrize("ratio,conv", [(0.02, 1), (0.5, 1), (2.0, 1), (48000 / 44100, 1), (0.0213, 2), (1.5, 0), (0.3, 3), (1.7, 4)])
def test_resampler_special_values(gpu, redio, oracle, ratio, conv):
    """libsamplerate multiplies in double and narrows once per output: inf and NaN spread over one filter length, subnormal inputs count"""
    n = 60000
    for seed, maker, dens in ((1, sprinkle_finite, 0.01), (2, sprinkle_finite, 0.3), (3, sprinkle, 3.0 / n), (4, sprinkle, 0.01)):
        x = maker(oracle, seed * 5 + conv, n, False, dens)
        src, ref = redio.Src(1, conv), oracle.Resampler(conv)
        for lo, hi in ((0, 25001), (25001, n)):
            cap = int(ratio * (hi - lo) + 1.0)
            a, ua = src.process(gpu.from_numpy(x[None, lo:hi].copy()).cuda(), ratio, output_frames=cap)
            err, want, wused = ref.process(x[lo:hi], ratio, cap)
            assert err == 0 and wused == ua
            assert same_special(a.cpu().numpy()[0], want), (ratio, conv, seed, lo, where_differs(a.cpu().numpy()[0], want))


def test_norm_block_sums_discretize_special_values(gpu, redio, oracle):
    x = sprinkle(oracle, 99, 512 * 300, True, 0.01)
    want = oracle.norm(x)
    got = redio.bitfount.norm(gpu.from_numpy(x).cuda()).cpu().numpy()
    assert same_special(got, want), where_differs(got, want)
    m = np.abs(sprinkle(oracle, 98, 512 * 300, False, 0.01))
    sums = redio.bitfount.block_sums(gpu.from_numpy(m).cuda(), 512).cpu().numpy()
    wsum = np.array([oracle.block_sum(m[b * 512:(b + 1) * 512]) for b in range(300)], np.float32)
    assert same_special(sums, wsum), where_differs(sums, wsum)
    for mm in (m, np.where(np.isinf(m), np.float32(5.0), m)):  # f32::max ignores NaN; with an infinite maximum max / 2 is infinite
        assert np.array_equal(redio.bitfount.discretize(gpu.from_numpy(mm).cuda()).cpu().numpy(), oracle.discretize(mm).astype(np.uint8))
