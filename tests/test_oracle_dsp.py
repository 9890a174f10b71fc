"""The oracle against independent float64 math and known answers ("parity unpinned": the reference
has no vectors, so these pins are ours -- SURVEY.md 8c)."""
import numpy as np
import pytest


def test_convolve_is_valid_mode_correlation(oracle):
    rng = np.random.default_rng(0)
    for nu, nv in ((10, 3), (500, 63), (127, 127), (2000, 127)):
        u = rng.standard_normal(nu).astype(np.float32)
        v = rng.standard_normal(nv).astype(np.float32)
        y = oracle.convolve(u, v)
        ref = np.correlate(u.astype(np.float64), v.astype(np.float64), "valid")
        assert len(y) == nu - nv + 1
        bound = nv * 2.0 ** -24 * np.convolve(np.abs(u), np.abs(v)[::-1], "valid")
        assert (np.abs(y - ref) <= bound).all()


def test_convolve_taps_not_reversed_and_known_answer(oracle):
    assert oracle.convolve([1, 2, 3, 4], [1, 10]).tolist() == [21.0, 32.0, 43.0]
    assert len(oracle.convolve([1, 2], [1, 2, 3])) == 0  # windows() yields nothing
    with pytest.raises(ValueError):
        oracle.convolve([1, 2], [])                      # windows(0) panics


def test_convolve_f64_generic(oracle):
    u = np.arange(10, dtype=np.float64)
    assert np.array_equal(oracle.convolve(u, np.array([1.0, -1.0])), -np.ones(9))


def test_fold_order_is_sequential(oracle):
    # ((0 + 1e8) + 1) - 1e8 in f32 loses the 1; any other association would not
    y = oracle.convolve(np.array([1e8, 1.0, -1e8], np.float32), np.ones(3, np.float32))
    assert y[0] == 0.0


def test_window_quirks(oracle):
    # values measured in SURVEY.md 8a A2
    w4 = oracle.window(4)
    assert len(w4) == 5 and np.isnan(w4[1])
    assert np.allclose(w4[[0, 2, 3, 4]], [-1.4739, -1.4739, 1.9390, 1.3503], atol=2e-4)
    w63 = oracle.window(63)
    assert len(w63) == 64 and np.isnan(w63[1]) and np.isnan(w63).sum() == 1
    assert np.allclose(w63[[0, 2, 3, 4, 5]], [-23.377, -23.377, 31.355, 12.775, 19.559], atol=2e-3)


def test_sinc_support_and_assert(oracle):
    s = oracle.sinc(64, 0.1)
    assert s[32] == np.float32(0.2)  # n == 0 tap only for even m
    n = np.arange(63) - 31.5
    assert np.allclose(oracle.sinc(63, 0.1), np.sin(2 * np.pi * 0.1 * n) / (np.pi * n), atol=1e-6)
    with pytest.raises(ValueError):
        oracle.sinc(8, 0.5)


def test_lpf_hpf_bsf_bpf_relations(oracle):
    m = 63
    l, h = oracle.lpf(m, 0.1), oracle.hpf(m, 0.1)
    assert np.isnan(l[1]) and np.isnan(l).sum() == 1
    e = np.zeros(m, np.float32); e[m // 2 - 1] = 1.0
    ok = ~np.isnan(l)
    assert np.array_equal((h - e)[ok], (-l)[ok])
    b = oracle.bsf(m, 0.1, 0.2)
    assert np.array_equal(oracle.bpf(m, 0.1, 0.2)[ok], (-b)[ok])
    with pytest.raises(ValueError):
        oracle.hpf(1, 0.1)


def test_lpf_corrected_is_a_lowpass(oracle):
    h = oracle.lpf_corrected(127, 0.08)
    assert np.allclose(h, h[::-1]) and abs(h.sum() - 1) < 1e-4
    H = np.abs(np.fft.rfft(h, 8192))
    f = np.arange(len(H)) / 8192
    assert H[f < 0.05].min() > 0.99 and H[f > 0.12].max() < 1e-4


def test_synth_range_and_determinism(oracle):
    x = oracle.synth_iq(0x5EED0002, 0, 100000)
    assert x.real.min() >= -1 and x.real.max() < 1 and abs(x.real.mean()) < 0.01
    assert np.array_equal(oracle.synth_iq(0x5EED0002, 50, 10), x[50:60])
    assert oracle.lib().orc_hash32(0, 0) == oracle.lib().orc_hash32(0, 0)


def test_chain_matches_float64(oracle):
    taps = oracle.lpf_corrected(127, 0.08)
    x = oracle.synth_iq(1, 0, 3 * 5120 + 126 + 77)
    s = oracle.chain_fir_fft(x, taps, 5, 1024)
    assert s.shape == (3, 1024)
    y = np.correlate(x.astype(np.complex128), taps.astype(np.float64), "valid")[::5][: 3 * 1024].reshape(3, 1024)
    ref = np.fft.fft(y, axis=1)
    assert np.linalg.norm(s - ref) / np.linalg.norm(ref) < 2e-6


def test_resampler_ratio_decrease_reads_silence_in_front_of_the_buffer(oracle):
    """libsamplerate 0.1.8 reads in front of its buffer when the ratio falls between two calls (the filter widens beyond the retained
    history: negative data_index in calc_output_single); the oracle DEFINES those samples as +0.0f.  The sequence the randomised run found
    (round 4) must be finite, bounded by the input range times the filter gain, and reproducible from a fresh state."""
    msgs = [(1086, 708690820, '0x1.47ae147ae147bp-6', 22), (7842, 861348262, '0x1.e43d5e17e519ap-7', 116), (1103, 139141258, '0x1.e43d5e17e519ap-7', 17),
            (2719, 811523618, '0x1.ee08c42c7828dp-7', 41)]
    runs = []
    for _ in range(2):
        ref, outs = oracle.Resampler(0, 1), []
        for m, seed, rh, cap in msgs:
            err, y, used = ref.process(oracle.synth_f32(seed, 0, m), float.fromhex(rh), cap, False)
            assert err == 0 and used == m
            outs.append(y)
        y = np.concatenate(outs)
        assert len(y) == 48 and np.all(np.isfinite(y)) and np.all(np.abs(y) < 4.0)
        runs.append(y)
    assert np.array_equal(runs[0].view(np.uint32), runs[1].view(np.uint32))
