"""The segmentation arithmetic of the carried-history streams (libredio_amd/csrc/stream_split.h), run on the CPU:
a Python model of stream_carry.hip's enqueue (same staging buffer, same head / body / tail rule) with the ORACLE as
the stateless kernel must reproduce the one-shot oracle result for any cut of the stream."""
import ctypes as C

import numpy as np
import pytest


def split(emu, hist, W, H, n):
    out = (C.c_size_t * 5)()
    emu.emu_stream_split.argtypes = [C.c_size_t] * 4 + [C.POINTER(C.c_size_t)]
    emu.emu_stream_split(hist, W, H, n, out)
    return tuple(out)


class Model:
    def __init__(self, emu, W, H, kernel):
        self.emu, self.W, self.H, self.kernel, self.tail, self.skip = emu, W, H, kernel, None, 0

    def feed(self, new):
        drop = min(self.skip, len(new))     # H > W: samples between two windows are never read
        new, self.skip = new[drop:], self.skip - drop
        if len(new) == 0:
            return []
        hist = 0 if self.tail is None else len(self.tail)
        assert hist < self.W
        nh, head_in, nb, off, body_in = split(self.emu, hist, self.W, self.H, len(new))
        m = min(len(new), self.W - 1)
        stage = new[:m] if hist == 0 else np.concatenate([self.tail, new[:m]])
        outs = []
        if nh:
            assert head_in <= len(stage) and (nh - 1) * self.H < hist
            outs.append(self.kernel(stage[:head_in]))
        if nb:
            assert off + body_in <= len(new)
            outs.append(self.kernel(new[off:off + body_in]))
        consumed = (nh + nb) * self.H
        whole = new if hist == 0 else np.concatenate([self.tail, new])
        if consumed >= len(whole):
            self.skip, self.tail = consumed - len(whole), None
            return outs
        if consumed < hist:
            assert len(new) < self.W - 1        # then the staging buffer holds all of [tail | new]
        self.tail = whole[consumed:].copy()
        return outs


@pytest.mark.parametrize("k,d", [(127, 5), (63, 1), (1, 1), (17, 4), (5, 9), (2, 2)])
def test_fir_stream_model_any_segmentation(emu, oracle, k, d):
    rng = np.random.default_rng(k * 7 + d)
    taps = oracle.synth_f32(5, 0, k)
    n = 20000
    x = oracle.synth_iq(77, 0, n)
    want = oracle.fir(x, taps, d, False)
    for trial in range(6):
        m = Model(emu, k, d, lambda v: oracle.fir(v, taps, d, False))
        pos, outs = 0, []
        while pos < n:
            step = int(rng.choice([0, 1, 2, 3, k - 1, k, k + 1, 2 * k, 500, 4097, int(rng.integers(0, 3000))]))
            step = min(max(step, 0), n - pos)
            outs += m.feed(x[pos:pos + step])
            pos += step
        got = np.concatenate(outs) if outs else np.zeros(0, np.complex64)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, d, trial)
        left = 0 if m.tail is None else len(m.tail)
        assert left - m.skip == n - len(want) * d and left < k


def test_chain_stream_model_any_segmentation(emu, oracle):
    k, d, nfft = 31, 4, 64
    rng = np.random.default_rng(3)
    taps = oracle.synth_f32(6, 0, k)
    n = 9 * nfft * d + k + 100
    x = oracle.synth_iq(78, 0, n)
    want = oracle.chain_fir_fft(x, taps, d, nfft)
    W, H = (nfft - 1) * d + k, nfft * d
    for trial in range(6):
        m = Model(emu, W, H, lambda v: oracle.chain_fir_fft(v, taps, d, nfft).reshape(-1))
        pos, outs = 0, []
        while pos < n:
            step = min(int(rng.choice([1, 50, H - 1, H, H + 1, W - 1, W, W + 1, 3 * H + 5, 1000])), n - pos)
            outs += m.feed(x[pos:pos + step])
            pos += step
        got = np.concatenate(outs).reshape(-1, nfft)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), trial
