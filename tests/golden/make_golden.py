#!/usr/bin/env python3
"""Generates the committed golden vectors in tests/golden/ (small .npz files).

The reference ships no vectors and cannot be run here, so these are produced by the
CPU oracle (oracle/) and, where an independent definition exists, cross-checked against float64
numpy at generation time.  Inputs are the hash-generated streams of SURVEY.md 8d, stored alongside
the expected outputs so a fixture is self-contained data.  Re-running this script must reproduce the
files bit for bit (tests/test_golden.py::test_generator_is_reproducible).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402


def fir_cases():
    out = {}
    for name, k, d, cplx in (("k3", 3, 1, False), ("k63", 63, 1, False), ("k127", 127, 1, True), ("k127d5", 127, 5, True)):
        taps = O.lpf_corrected(k, 0.08) if k > 3 else np.array([0.25, 0.5, 0.25], np.float32)
        x = O.synth_iq(0x5EED0000 + k, 0, 4096) if cplx else O.synth_f32(0x5EED0000 + k, 0, 4096)
        y = O.fir(x, taps, d, fused=False)
        yf = O.fir(x, taps, d, fused=True)
        ref = np.correlate(x.astype(np.complex128 if cplx else np.float64), taps.astype(np.float64), "valid")[::d]
        assert np.abs(y - ref).max() <= k * 2.0 ** -24 * np.abs(taps).sum()
        out.update({f"{name}_x": x, f"{name}_taps": taps, f"{name}_y": y, f"{name}_y_fused": yf, f"{name}_decim": np.int64(d)})
    # quirk generators: NaN position recorded
    for m in (4, 63, 127):
        out[f"quirk_lpf_{m}"] = O.lpf(m, 0.1)
        out[f"quirk_window_{m}"] = O.window(m)
    out["corrected_lpf_63_0p1"] = O.lpf_corrected(63, 0.1)
    out["corrected_lpf_127_0p08"] = O.lpf_corrected(127, 0.08)
    return out


def fft_cases():
    out = {}
    for n in (4, 16, 64, 1024, 30, 7):
        for kind in ("impulse", "tone", "random"):
            if kind == "impulse":
                x = np.zeros(n, np.complex64); x[1 % n] = 1
            elif kind == "tone":
                x = np.exp(2j * np.pi * 3 * np.arange(n) / n).astype(np.complex64)
            else:
                x = O.synth_iq(0x5EED0100 + n, 0, n)
            for inv in (0, 1):
                X = O.fft(x, inverse=bool(inv))
                ref = np.fft.fft(x.astype(np.complex128)) if not inv else np.fft.ifft(x.astype(np.complex128)) * n
                assert np.linalg.norm(X - ref) <= 2e-6 * max(np.linalg.norm(ref), 1e-30)
                out[f"n{n}_{kind}_inv{inv}_x"] = x
                out[f"n{n}_{kind}_inv{inv}_X"] = X
    # 65536: hash only -- stored as the uint32 bit pattern checksum
    x = O.synth_iq(0x5EED0165, 0, 65536)
    X = O.fft(x)
    out["n65536_random_checksum"] = np.array([int(X.view(np.uint32).astype(np.uint64).sum())], np.uint64)
    return out


def chain_cases():
    taps = O.lpf_corrected(127, 0.08)
    x = O.synth_iq(0x5EED0002, 0, 2 * 5120 + 126)
    return {"x": x, "taps": taps, "spectra": O.chain_fir_fft(x, taps, 5, 1024, fused=False),
            "spectra_fused": O.chain_fir_fft(x, taps, 5, 1024, fused=True)}


def resample_cases():
    out = {}
    x = O.synth_f32(0x5EED0003, 0, 12000)
    out["x"] = x
    for name, ratio in (("r0p02", 0.02), ("r0p5", 0.5), ("r2", 2.0), ("r48_44p1", 48000 / 44100)):
        for sname, seg in (("one", [12000]), ("three", [5000, 1, 6999]), ("many", [1000] * 12)):
            r = O.Resampler(1)
            ys, off = [], 0
            for m in seg:
                ys.append(r.block(x[off:off + m], ratio)); off += m
            out[f"{name}_{sname}_y"] = np.concatenate(ys)
            out[f"{name}_{sname}_counts"] = np.array([len(y) for y in ys], np.int64)
        out[f"{name}_ratio"] = np.float64(ratio)
    return out


def bit_cases():
    out = {}
    bytes_all = np.arange(256, dtype=np.uint8)
    out["d2s_bytes"] = np.repeat(bytes_all, 2)
    out["d2s_samples"] = O.data_to_samples(out["d2s_bytes"])
    x = O.synth_f32(0x5EED0009, 0, 512) ** 2
    out["disc_x"] = x
    out["disc_bits"] = O.discretize(x).astype(np.uint8)
    rng = np.random.default_rng(0x5EED)
    bits36 = rng.integers(0, 2, 36).astype(np.uint64)
    out["eat_bits36"] = bits36
    out["eat_w_4_8_4_12_8"] = np.array(O.eat(bits36, [4, 8, 4, 12, 8]), np.uint64)   # ratpak.rs:115
    out["eat_w_4_8_2_10_12"] = np.array(O.eat(bits36, [4, 8, 2, 10, 12]), np.uint64)  # ratpak.rs:119
    out["b2d_101"] = np.array([O.b2d([1, 0, 1])], np.uint64)
    return out


def pfb_cases():
    h = O.lpf_corrected(1024, 0.45 / 64)
    x = O.synth_iq(0x5EED0004, 0, 64 * 20)
    return {"x": x, "proto": h, "y": O.pfb_channelizer(x, h, 64, 16, False), "y_fused": O.pfb_channelizer(x, h, 64, 16, True)}


def ovsave_cases():
    out = {}
    for nfft, k in ((256, 33), (1024, 127), (4096, 63)):
        taps = O.lpf_corrected(k, 0.1)
        x = O.synth_iq(0x5EED0005 + nfft, 0, nfft + 2 * (nfft - k + 1) + 5)
        y = O.overlap_save(x, taps, nfft)
        direct = O.fir(x, taps, 1, fused=False)[: len(y)]
        assert np.abs(y - direct).max() <= 2e-6 * np.abs(taps).sum() * np.sqrt(np.log2(nfft)) + 1e-7
        out.update({f"n{nfft}_x": x, f"n{nfft}_taps": taps, f"n{nfft}_y": y})
    return out


def front_end_cases():
    """The shipped graph's front end on one burst: bytes -> samples -> |x| -> 512-sample block sums -> slicer -> runs."""
    out = {}
    rng = np.random.default_rng(0x5EED0A)
    n = 512 * 9
    env = np.where((np.arange(n) // 300) % 2 == 0, 100.0, 8.0)                 # on/off keyed carrier
    i = np.clip(127.5 + env * np.cos(0.3 * np.arange(n)) + rng.normal(0, 2, n), 0, 255)
    q = np.clip(127.5 + env * np.sin(0.3 * np.arange(n)) + rng.normal(0, 2, n), 0, 255)
    raw = np.stack([i, q], axis=1).astype(np.uint8).reshape(-1)
    xs = O.data_to_samples(raw)
    mag = O.norm(xs)
    bitsv = O.discretize(mag).astype(np.uint8)
    runs = O.Rle().feed(bitsv)
    out["raw"] = raw
    out["mag"] = mag
    out["block_sums"] = np.array([O.block_sum(mag[b * 512:(b + 1) * 512]) for b in range(n // 512)], np.float32)
    out["bits"] = bitsv
    out["run_values"] = np.array([r[0] for r in runs], np.uint8)
    out["run_counts"] = np.array([r[1] for r in runs], np.int64)
    return out


def pfb_generic_cases():
    out = {}
    for M, P in ((32, 4), (100, 3)):
        h = O.synth_f32(0x5EED0B, 0, M * P)
        x = O.synth_iq(0x5EED0004 + M, 0, M * (P + 6))
        out.update({f"m{M}_x": x, f"m{M}_proto": h, f"m{M}_y": O.pfb_channelizer(x, h, M, P, False)})
    return out


def converter_cases():
    """Converters 3 / 4 (zero-order hold, linear) and interleaved channels (samplerate.rs:26-30 declares them; the reference
    only uses converter 1, mono).  Cross-checked at generation time against closed forms: a hold of an integer ramp at ratio
    1/4 returns every fourth sample, a linear interpolation of it at ratio 4 returns exact quarter steps."""
    out = {}
    ramp = np.arange(1000, dtype=np.float32)
    e, z, u = O.Resampler(3).process(ramp, 0.25, 300)
    assert e == 0 and np.array_equal(z[1:9], np.float32([3, 7, 11, 15, 19, 23, 27, 31]))
    e, l, u = O.Resampler(4).process(ramp, 4.0, 4100)
    assert e == 0 and np.array_equal(l[4:44], (np.arange(40) * 0.25).astype(np.float32))
    x2 = np.stack([O.synth_f32(0x5EED0C, 0, 3000), O.synth_f32(0x5EED0D, 0, 3000)], 1).reshape(-1)   # two interleaved channels
    out["x2"] = x2
    for conv in (1, 3, 4):
        for name, ratio in (("r0p3", 0.3), ("r1p5", 1.5)):
            r = O.Resampler(conv, 2)
            ys, counts, used = [], [], []
            for lo, hi in ((0, 1000), (1000, 1001), (1001, 3000)):
                cap = int(ratio * (hi - lo) + 1.0) + 2
                e, y, u = r.process(x2[2 * lo:2 * hi], ratio, cap, False)
                assert e == 0
                ys.append(y); counts.append(len(y) // 2); used.append(u)
            out[f"c{conv}_{name}_y"] = np.concatenate(ys)
            out[f"c{conv}_{name}_counts"] = np.array(counts, np.int64)
            out[f"c{conv}_{name}_used"] = np.array(used, np.int64)
    # channels are independent mono streams
    m = O.Resampler(1, 1)
    ym = np.concatenate([m.process(np.ascontiguousarray(x2[0::2][lo:hi]), 0.3, int(0.3 * (hi - lo) + 1.0) + 2, False)[1] for lo, hi in ((0, 1000), (1000, 1001), (1001, 3000))])
    assert np.array_equal(out["c1_r0p3_y"][0::2], ym)
    return out


SPECIALS = np.array([0.0, -0.0, 1e-40, -1e-40, 1.4e-45, 1e-30, -1e-30, 1e30, -1e30, 3e38, -3e38, np.inf, -np.inf, np.nan, 1.0, -1.0, 2.0 ** -126, 2.0 ** 127], np.float32)


def _canon(a):
    """every NaN as the canonical quiet NaN: payloads are what x86 and gfx950 disagree on, and nothing defines them"""
    a = np.array(a, copy=True)
    w = a.view(np.float32)
    w[np.isnan(w)] = np.float32(np.nan)
    return a


def _sprinkle(x, seed, count, finite):
    w = x.view(np.float32).reshape(-1)
    rng = np.random.default_rng(seed)
    pool = SPECIALS[np.isfinite(SPECIALS)] if finite else SPECIALS
    w[rng.integers(0, len(w), count)] = pool[rng.integers(0, len(pool), count)]
    if not finite:  # one of each for certain, in the last quarter of the stream
        w[len(w) - len(w) // 8 + 1], w[len(w) - len(w) // 5], w[len(w) - len(w) // 7] = np.float32(np.inf), np.float32(-np.inf), np.float32(np.nan)
    return x


def special_cases():
    """signed zeros, subnormals, overflowing / underflowing products (`fin`), and infinities / NaNs (`any`) through the arithmetic paths:
    IEEE-754 decides them the same way on both sides only if no operation is skipped, merged or reordered"""
    out = {}
    taps = O.lpf_corrected(127, 0.08)
    taps[[0, 31, 64, 126]] = np.array([-0.0, 1e30, 1e-40, 0.0], np.float32)
    out["fir_taps"] = taps
    for tag, finite, count in (("fin", True, 400), ("any", False, 3)):
        x = _sprinkle(O.synth_iq(0x5EED0900, 0, 4096), 1, count, finite)
        out[f"fir_{tag}_x"] = _canon(x)
        out[f"fir_{tag}_y"] = _canon(O.fir(x, taps, 5, fused=False))
        out[f"fir_{tag}_y_fused"] = _canon(O.fir(x, taps, 5, fused=True))
        for n in (16, 64, 1024, 30):
            xb = _sprinkle(O.synth_iq(0x5EED0901 + n, 0, 3 * n), 2 + n, max(2, (n // 8 if finite else 2)), finite)
            xb[:n] = 0
            xb[n:2 * n] = np.complex64(complex(-0.0, -0.0))
            out[f"fft{n}_{tag}_x"] = _canon(xb)
            for inv in (0, 1):
                out[f"fft{n}_{tag}_X{inv}"] = _canon(O.fft(xb, n, inverse=bool(inv)))
        xc = _sprinkle(O.synth_iq(0x5EED0902, 0, 5120 + 126), 3, 100 if finite else 2, finite)
        out[f"chain_{tag}_x"] = _canon(xc)
        out[f"chain_{tag}_spectra"] = _canon(O.chain_fir_fft(xc, O.lpf_corrected(127, 0.08), 5, 1024, fused=True))
        xr = _sprinkle(O.synth_f32(0x5EED0903, 0, 8000), 4, 80 if finite else 2, finite)
        err, yr, used = O.Resampler(1).process(xr, 0.02, 161)
        assert err == 0 and used == 8000
        out[f"src_{tag}_x"] = _canon(xr)
        out[f"src_{tag}_y"] = _canon(yr)
    return out


CASES = {"special": special_cases, "converters": converter_cases, "ovsave": ovsave_cases, "front_end": front_end_cases, "pfb_generic": pfb_generic_cases, "fir": fir_cases, "fft": fft_cases, "chain": chain_cases, "resample": resample_cases, "bits": bit_cases, "pfb": pfb_cases}


def generate(outdir=HERE):
    for name, fn in CASES.items():
        np.savez(os.path.join(outdir, f"{name}.npz"), **fn())


if __name__ == "__main__":
    generate()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
