"""Distinct handles from distinct threads (the reference runs one OS thread per block, src/ratpak.rs:60-185): the
library keeps no hidden global state, so concurrent plans must not disturb each other.  ctypes releases the GIL
during the calls, so these threads really overlap inside libredio."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_plans_on_their_own_streams(gpu, redio, oracle):
    taps = oracle.lpf_corrected(127, 0.08)
    errors = []

    def worker(kind, seed):
        try:
            s = gpu.cuda.Stream()
            with gpu.cuda.stream(s):
                for it in range(12):
                    if kind == "chain":
                        x = oracle.synth_iq(seed + it, 0, 6 * 5120 + 126)
                        got = redio.Chain(taps, 5, 1024, fused=False)(gpu.from_numpy(x).cuda()).cpu().numpy()
                        want = oracle.chain_fir_fft(x, taps, 5, 1024, fused=False)
                    elif kind == "fft":
                        n = [64, 1000, 2048, 4096][it % 4]
                        x = oracle.synth_iq(seed + it, 0, n * 9)
                        got = redio.Fft(n)(gpu.from_numpy(x).cuda()).cpu().numpy()
                        want = oracle.fft(x, n)
                    elif kind == "fir":
                        k, d = [(64, 1), (127, 5), (33, 2), (255, 10)][it % 4]
                        h = oracle.synth_f32(seed, 0, k)
                        x = oracle.synth_iq(seed + it, 0, 20000)
                        got = redio.Fir(h, d)(gpu.from_numpy(x).cuda()).cpu().numpy()
                        want = oracle.fir(x, h, d, False)
                    elif kind == "conv":   # the host-buffer drop-in with its per-thread cache
                        u = oracle.synth_f32(seed + it, 0, 5000 + it)
                        got = redio.dsputils.convolve(u, taps)
                        want = oracle.convolve(u, taps)
                    else:                  # resampler state per thread
                        x = oracle.synth_f32(seed + it, 0, 30000)
                        got = redio.samplerate.State(1, 1).block(x, 0.02)
                        want = oracle.Resampler(1).block(x, 0.02)
                    if not (got.shape == want.shape and np.array_equal(np.ascontiguousarray(got).view(np.uint32), np.ascontiguousarray(want).view(np.uint32))):
                        errors.append((kind, it))
        except Exception as e:  # noqa: BLE001
            errors.append((kind, repr(e)))

    kinds = ["chain", "fft", "fir", "conv", "src", "chain", "fft", "conv"]
    threads = [threading.Thread(target=worker, args=(k, 1000 * (i + 1))) for i, k in enumerate(kinds)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
