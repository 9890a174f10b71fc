"""oracle_kiss.c (kissfft restatement) against numpy float64 and known answers."""
import numpy as np
import pytest

SIZES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 25, 30, 64, 100, 128, 243, 1000, 1024, 2048, 65536, 11, 221, 1009]


def test_factor_order(oracle):
    assert oracle.kiss_factors(1024) == [(4, 256), (4, 64), (4, 16), (4, 4), (4, 1)]
    assert oracle.kiss_factors(2048)[-1] == (2, 1)
    assert oracle.kiss_factors(64) == [(4, 16), (4, 4), (4, 1)]
    assert oracle.kiss_factors(30) == [(2, 15), (3, 5), (5, 1)]
    assert oracle.kiss_factors(7) == [(7, 1)]
    assert len(oracle.kiss_factors(65536)) == 8


@pytest.mark.parametrize("n", SIZES)
def test_fft_matches_numpy(oracle, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    X = np.fft.fft(x.astype(np.complex128))
    tol = 2e-6  # relative L2, SURVEY.md 8c
    assert np.linalg.norm(oracle.fft(x) - X) <= tol * max(np.linalg.norm(X), 1e-30)
    Xi = np.fft.ifft(x.astype(np.complex128)) * n  # unnormalised inverse
    assert np.linalg.norm(oracle.fft(x, inverse=True) - Xi) <= tol * max(np.linalg.norm(Xi), 1e-30)


def test_known_answers(oracle):
    n = 1024
    imp = np.zeros(n, np.complex64); imp[0] = 1
    assert np.array_equal(oracle.fft(imp), np.ones(n, np.complex64))
    dc = np.ones(n, np.complex64)
    X = oracle.fft(dc)
    assert X[0] == n and np.abs(X[1:]).max() < 1e-3
    k = 37
    tone = np.exp(2j * np.pi * k * np.arange(n) / n).astype(np.complex64)
    X = oracle.fft(tone)
    assert abs(X[k] - n) < 1e-2 and np.abs(np.delete(X, k)).max() < 2e-2


def test_parseval_and_roundtrip(oracle):
    x = oracle.synth_iq(3, 0, 4096)
    X = oracle.fft(x)
    assert abs((np.abs(X) ** 2).sum() / 4096 - (np.abs(x) ** 2).sum()) < 1e-3 * (np.abs(x) ** 2).sum()
    assert np.abs(oracle.fft(X, inverse=True) / 4096 - x).max() < 1e-5


def test_block_contract(oracle):
    x = oracle.synth_iq(5, 0, 64 * 3)
    y = oracle.fft(x, 64)
    for b in range(3):
        assert np.array_equal(y[64 * b: 64 * b + 64], oracle.fft(x[64 * b: 64 * b + 64]))
    with pytest.raises(AssertionError):
        oracle.fft(x[:100], 64)  # assert!(din.len() == block_size), kissfft.rs:24
