"""The kernels' lane programs (libredio_amd/csrc/*_core.h) emulated on the CPU, bit for bit against
the oracle.  This is how index maps are validated in the build container, which has no GPU."""
import ctypes as C

import numpy as np
import pytest

c64 = np.ctypeslib.ndpointer(np.complex64, flags="C")
f32 = np.ctypeslib.ndpointer(np.float32, flags="C")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("inv", [0, 1])
def test_fft1k_wave_program(emu, oracle, inv):
    emu.emu_fft1k.argtypes = [c64, c64, C.c_int]
    x = oracle.synth_iq(21 + inv, 0, 1024)
    y = np.empty_like(x)
    emu.emu_fft1k(x, y, inv)
    assert np.array_equal(bits(y), bits(oracle.fft(x, inverse=bool(inv))))


def test_lds_images_are_bank_conflict_free(emu):
    assert emu.emu_fft1k_bank_conflicts() == 1
    assert emu.emu_fir_bank_conflicts() == 1


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8, 15, 16, 30, 64, 100, 210, 1024, 2048, 4096, 11, 221, 1009])
def test_generic_stage_program(emu, oracle, n):
    emu.emu_fft_generic.argtypes = [C.c_int, C.c_int, c64, c64]
    x = oracle.synth_iq(n, 0, n)
    for inv in (0, 1):
        y = np.empty_like(x)
        assert emu.emu_fft_generic(n, inv, x, y) == len(oracle.kiss_factors(n))
        assert np.array_equal(bits(y), bits(oracle.fft(x, inverse=bool(inv)))), (n, inv)


@pytest.mark.parametrize("k,d", [(127, 5), (127, 1), (63, 1), (63, 5)])
def test_fir_tile_program(emu, oracle, k, d):
    emu.emu_fir_c32.argtypes = [c64, C.c_long, f32, C.c_int, C.c_int, C.c_int, c64]
    emu.emu_fir_c32.restype = C.c_long
    taps = oracle.lpf_corrected(k, 0.08)
    for n in (k - 1, k, k + 3, 6000, 12345):
        x = oracle.synth_iq(7, 0, max(n, 1))[:n]
        x = np.ascontiguousarray(x)
        for fused in (0, 1):
            want = oracle.fir(x, taps, d, bool(fused)) if n >= k else np.empty(0, np.complex64)
            y = np.zeros(len(want) + 8, np.complex64)
            no = emu.emu_fir_c32(x if n else np.zeros(1, np.complex64), n, taps, k, d, fused, y)
            assert no == len(want)
            assert np.array_equal(bits(y[:no]), bits(want))


def test_fir_tile_program_real(emu, oracle):
    emu.emu_fir_f32.argtypes = [f32, C.c_long, f32, C.c_int, C.c_int, C.c_int, f32]
    emu.emu_fir_f32.restype = C.c_long
    for k, d in ((127, 5), (63, 1)):
        taps = oracle.lpf_corrected(k, 0.1)
        x = oracle.synth_f32(9, 0, 9000)
        for fused in (0, 1):
            want = oracle.fir(x, taps, d, bool(fused))
            y = np.zeros(len(want) + 8, np.float32)
            assert emu.emu_fir_f32(x, len(x), taps, k, d, fused, y) == len(want)
            assert np.array_equal(bits(y[: len(want)]), bits(want))


def test_resampler_position_recurrence_short_form(emu):
    """src_position.h: for step < 1 (every upsampling ratio) the library's per-output recurrence (libsamplerate 0.1.8,
    sinc_mono_vari_process; samplerate.rs:61 drives it) reduces to add / compare / subtract.  Same doubles, same advances
    as the literal fmod_one form: random operands, the rounding edges around 0.5, 1 and 1.5, and long chains at the ratios
    the resampler tests use."""
    emu.emu_src_advance_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long]
    emu.emu_src_advance_mismatches.restype = C.c_long
    rng = np.random.default_rng(7)
    eps = 2.0 ** -53
    xs, steps = [], []
    # random pairs over the whole range of ratios (1/256 .. 256)
    xs += list(rng.random(200000)); steps += list(1.0 / rng.uniform(1.0 / 256, 256.0, 200000))
    # edges: sums that land on or next to 0.5, 1.0, 1.5 and just below 2.0
    for target in (0.5, 1.0, 1.5, 2.0 - 2 * eps):
        for d in (-3, -2, -1, 0, 1, 2, 3):
            for x in (0.0, 0.25, 0.5 - eps, 0.5, 0.75, 1.0 - eps, rng.random()):
                st = target - x + d * eps
                if 0.0 < st:
                    xs.append(x); steps.append(st)
    x = np.array(xs, dtype=np.float64); st = np.array(steps, dtype=np.float64)
    assert emu.emu_src_advance_mismatches(x.ctypes.data, st.ctypes.data, len(x), 64) == 0
    # long chains at real ratios
    ratios = np.array([48000 / 44100, 2.0, 1.5, 4 / 3, 1.0884, 2 ** 0.5, 3.7, 256.0, 1.0, 0.3, 0.02, 0.0213], dtype=np.float64)
    x0 = np.zeros(len(ratios))
    assert emu.emu_src_advance_mismatches(x0.ctypes.data, (1.0 / ratios).ctypes.data, len(ratios), 2000000) == 0
