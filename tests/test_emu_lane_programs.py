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
