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


def test_fir_run_program_real(emu, oracle):
    """The wave-private run form of the real-sample FIR (fir_run_core.h / fir_run.hip, round 6), lane by lane on the CPU: runs of 1, 3 and 8
    sub-tiles, a shorter last run, inputs that end exactly where the last sub-tile's 16-byte loads end and a few samples beyond;
    the NaN-initialised image proves every window sample was written (head, parked loads or the carried halo)."""
    emu.emu_fir_run_f32.argtypes = [f32, C.c_long, f32, C.c_int, C.c_int, C.c_int, f32, C.c_long]
    emu.emu_fir_run_f32.restype = C.c_long
    k, d = 63, 1
    taps = oracle.synth_f32(77, 0, k)
    for n in (511, 512 + 63, 512 + 64, 512 + 66, 5 * 512 + 64, 11 * 512 + 64 + 300, 17 * 512 + 64 + 1):
        x = oracle.synth_f32(9, 0, n)
        for fused in (0, 1):
            want = oracle.fir(x, taps, d, bool(fused)) if n >= k else np.empty(0, np.float32)
            for spw in (1, 3, 8):
                y = np.full(len(want) + 8, np.nan, np.float32)
                done = emu.emu_fir_run_f32(x, n, taps, k, d, fused, y, spw)
                assert done == max(0, (n - 64) // 512) * 512 and done <= len(want), (n, done)
                assert np.array_equal(bits(y[:done]), bits(want[:done])), (n, fused, spw)
                assert np.isnan(y[done:]).all()


def test_fir_pair_image_program_real(emu, oracle):
    """The pair-image tile of the real-sample FIR (fir_core.h fir_lane_pairs, round 6): two copies of the tile (even-start and odd-start
    sample pairs), R / 2 packed accumulators per lane; every output's fold is still the reference's (dsputils.rs:31), bit for bit, for both
    roundings; tiles that end inside the input, exactly at its end, and ragged lengths; NaN-initialised images."""
    emu.emu_fir_pairs_f32.argtypes = [f32, C.c_long, f32, C.c_int, C.c_int, C.c_int, C.c_int, f32]
    emu.emu_fir_pairs_f32.restype = C.c_long
    assert emu.emu_fir_pairs_bank_conflicts() == 1
    taps = oracle.synth_f32(78, 0, 63)
    for r, nt in ((16, 128), (8, 256)):
        for n in (62, 63, 64, 100, 2048 + 62, 2048 + 63, 2 * 2048 + 62, 5000, 3 * 2048 + 700):
            x = oracle.synth_f32(11, 0, n)
            for fused in (0, 1):
                want = oracle.fir(x, taps, 1, bool(fused)) if n >= 63 else np.empty(0, np.float32)
                y = np.full(len(want) + 8, np.nan, np.float32)
                assert emu.emu_fir_pairs_f32(x, n, taps, 63, r, nt, fused, y) == len(want)
                assert np.array_equal(bits(y[:len(want)]), bits(want)), (r, nt, n, fused)
                assert np.isnan(y[len(want):]).all()


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


@pytest.mark.parametrize("R,U", [(2, 8), (2, 4), (4, 4), (4, 8), (8, 4)])
@pytest.mark.parametrize("S,ncl,ncr", [(50, 2285, 2284), (2, 93, 92), (4, 184, 183), (8, 367, 366), (16, 733, 732), (3, 139, 138), (1, 47, 46),
                                      (6, 97, 91), (50, 970, 969), (10, 458, 458)])
def test_resampler_register_blocked_wings(emu, R, U, S, ncl, ncr):
    """src_core.h: a lane owns one wing of R consecutive outputs and walks the union of their windows once; every
    accumulator must still receive exactly its own products in the library's order (libsamplerate 0.1.8
    calc_output_single, behind samplerate.rs:59-87).  Compared bit for bit with the plain per-output sums, with NaN in
    every slack float and guard entry, and with Inf / NaN samples in the stream: an output whose window does not
    contain them must not see them."""
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
    f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
    emu.emu_src_rb.argtypes = [f32p, f64p, C.c_int, f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, f32p, f32p]
    emu.emu_src_rb.restype = C.c_int
    if (R * S) % 4 or (R - 1) * S + U + 8 > min(ncl, ncr):
        pytest.skip("shape not served by the register-blocked kernel")
    NO = 64 * R if S <= 16 else 8 * R
    rng = np.random.default_rng(S * 131 + R * 7 + U)
    span = (NO - 1) * S + ncl + ncr
    L = rng.standard_normal(ncl)
    Rt = rng.standard_normal(ncr)
    for poison in (False, True):
        x = rng.standard_normal(span).astype(np.float32)
        if poison:
            x[rng.integers(0, S)] = np.inf          # inside the window of the first output only
            x[span - 1 - rng.integers(0, S)] = np.nan  # inside the window of the last output only
        out = np.zeros(NO, np.float32)
        ref = np.zeros(NO, np.float32)
        assert emu.emu_src_rb(x, L, ncl, Rt, ncr, S, R, U, NO, 0.02, out, ref) == 0
        assert np.array_equal(bits(out), bits(ref)), (R, U, S, poison, np.flatnonzero(bits(out) != bits(ref))[:8])
        if poison:
            assert not np.isfinite(ref[0]) and not np.isfinite(ref[-1]) and np.isfinite(ref[1:-1]).all()


# ---- the two-column ("pair") tile program of the four-stage passes (fft_big_core.h, fft_pair.h) --------------------------
@pytest.mark.parametrize("inv", [0, 1])
def test_pair_tile_program_fft_65536(emu, oracle, inv):
    """Gather pass + in-place pass of the 65536-point transform, every tile through the pair program's lane maps, LDS images and
    twiddle batches: the oracle's kiss_fft, bit for bit."""
    emu.emu_pair_fft64k.argtypes = [c64, c64, C.c_int]
    x = oracle.synth_iq(77 + inv, 0, 65536)
    y = np.empty_like(x)
    emu.emu_pair_fft64k(x, y, inv)
    assert np.array_equal(bits(y), bits(oracle.fft(x, inverse=bool(inv))))


@pytest.mark.parametrize("k", [127, 8193, 8192])
def test_pair_tile_program_overlap_save_block(emu, oracle, k):
    """One 65536-point overlap-save block through the three pair-program passes (gather; forward pass 1 x conj H x inverse pass 0 on
    one tile; inverse pass 1 with the masked, scaled store): orc_overlap_save on the same block, bit for bit (odd and even hop)."""
    emu.emu_pair_ovsave64k.argtypes = [c64, c64, c64, C.c_long]
    n = 65536
    x = oracle.synth_iq(5, 0, n)
    h = oracle.lpf_corrected(k, 0.1)
    hp = np.zeros(n, np.complex64); hp[:k] = h
    Hc = np.conj(oracle.fft(hp)).astype(np.complex64)
    hop = n - k + 1
    out = np.zeros(hop, np.complex64)
    emu.emu_pair_ovsave64k(x, np.ascontiguousarray(Hc), out, hop)
    assert np.array_equal(bits(out), bits(oracle.overlap_save(x, h, n)))


def test_pair_lds_images_are_bank_conflict_free(emu):
    assert emu.emu_pair_bank_conflicts() == 1


@pytest.mark.parametrize("lgn", [15, 17, 19])
@pytest.mark.parametrize("inv", [0, 1])
def test_pair_g128_gather_pass(emu, oracle, lgn, inv):
    """The four-stage gather pass of N = 2 * 4^L' points (radix-2 stage + three radix-4 stages on 128 rows x 32 columns per wavefront)
    through its lane maps and LDS image, followed by the plan's in-place pass (2^15) or the remaining generic stages (2^17): kiss_fft."""
    emu.emu_pair_g128_fft.argtypes = [C.c_int, c64, c64, C.c_int]
    n = 1 << lgn
    x = oracle.synth_iq(300 + lgn + inv, 0, n)
    y = np.empty_like(x)
    assert emu.emu_pair_g128_fft(lgn, x, y, inv) == len(oracle.kiss_factors(n))
    assert np.array_equal(bits(y), bits(oracle.fft(x, inverse=bool(inv))))


@pytest.mark.parametrize("k", [127, 4097, 4096])
def test_pair_overlap_save_32768_block(emu, oracle, k):
    """One 32768-point overlap-save block through the three passes (G128 forward; forward in-place pass x conj H x inverse G128 on one
    tile; inverse in-place pass with the masked, scaled store): orc_overlap_save, bit for bit."""
    emu.emu_pair_ovsave32k.argtypes = [c64, c64, c64, C.c_long]
    n = 32768
    x = oracle.synth_iq(6, 0, n)
    h = oracle.lpf_corrected(k, 0.1)
    hp = np.zeros(n, np.complex64); hp[:k] = h
    Hc = np.conj(oracle.fft(hp)).astype(np.complex64)
    hop = n - k + 1
    out = np.zeros(hop, np.complex64)
    emu.emu_pair_ovsave32k(x, np.ascontiguousarray(Hc), out, hop)
    assert np.array_equal(bits(out), bits(oracle.overlap_save(x, h, n)))


def test_pair_g128_image_is_bank_conflict_free(emu):
    assert emu.emu_pair_g128_bank_conflicts() == 1


@pytest.mark.parametrize("lgn", [18, 20])
@pytest.mark.parametrize("inv", [0, 1])
def test_pair_five_stage_passes(emu, oracle, lgn, inv):
    """The five-stage passes in the pair layout (four wavefronts per 1024-row tile, the fifth stage across them through the workgroup
    image): 2^18 = four-stage gather pass + five-stage in-place pass, 2^20 = five-stage gather pass + five-stage in-place pass; kiss_fft."""
    emu.emu_pair_fft_five.argtypes = [C.c_int, c64, c64, C.c_int]
    n = 1 << lgn
    x = oracle.synth_iq(500 + lgn + inv, 0, n)
    y = np.empty_like(x)
    assert emu.emu_pair_fft_five(lgn, x, y, inv) == 0
    assert np.array_equal(bits(y), bits(oracle.fft(x, inverse=bool(inv))))


def test_pair_fifth_stage_image_is_bank_conflict_free(emu):
    assert emu.emu_pair_x5_bank_conflicts() == 1


@pytest.mark.parametrize("nfft", [16384, 8192])
def test_four_wave_deal(emu, oracle, nfft):
    """fft16k_wave_kernel / fft8k_wave_kernel / ovsave8k_wave_kernel read their block in 512-byte runs and deal the samples to the four waves'
    sub-sequences x[4 n + q] through LDS (fft_big_core.h deal_write_cell / deal_read_cell): every register gets its own sample, no cell is
    written twice, and the writes (groups of 16 and of 32 lanes) and reads hit different bank pairs."""
    import ctypes as C
    x = oracle.synth_iq(0x16 + nfft, 0, nfft)
    emu.emu_four_wave_deal.restype = C.c_long
    assert emu.emu_four_wave_deal(C.c_int(nfft), x.ctypes.data_as(C.c_void_p)) == 0
    assert emu.emu_four_wave_deal_bank_conflicts() == 1


@pytest.mark.parametrize("lgn,inv", [(17, 0), (19, 0), (19, 1)])
def test_pair_g512_gather_pass(emu, oracle, lgn, inv):
    """G512 (fft_big_core.h; the device kernel is the next step): G128 on four wavefronts whose columns are N / 512 apart, then kissfft's
    radix-4 stage of sub-length 128 across them through the five-stage passes' workgroup image -- 2^19 points in TWO passes.  Every lane
    map, the extended twiddle copy and the store positions, one lane at a time: the oracle's kiss_fft, bit for bit."""
    n = 1 << lgn
    emu.emu_pair_g512_fft.argtypes = [C.c_int, c64, c64, C.c_int]
    x = oracle.synth_iq(0x512 + lgn + inv, 0, n)
    y = np.empty_like(x)
    assert emu.emu_pair_g512_fft(lgn, x, y, inv) > 0
    assert np.array_equal(bits(y), bits(oracle.fft(x, n, inverse=bool(inv))))


def test_channelizer_two_row_byte_loads_lane_map():
    """pfb_kernels.hip, IN_U8 == 2 (round 5): one wave instruction loads two 64-sample rows of u8 I/Q bytes, a dword per lane (lanes 0-31 the even
    row of the pair, 32-63 the odd one); lane l then takes ITS sample -- bytes (2l, 2l + 1) of the row -- from the dword of lane l / 2 (+ 32) with a
    ds_bpermute and a right shift by 16 (l & 1).  The map restated in numpy on random bytes, plus what the range-checked descriptor returns for a
    pair whose second row lies past the end of the stream (zeros: the lanes of that row only)."""
    rng = np.random.default_rng(12)
    rows = rng.integers(0, 256, (2, 128), dtype=np.uint8)                # two rows of 64 samples x (I, Q)
    dwords = rows.reshape(-1).view("<u4")                                # lane j holds bytes 4j .. 4j + 3 of the 256-byte pair
    assert dwords.shape == (64,)
    lane = np.arange(64)
    for ti in (0, 1):
        src_lane = (lane >> 1) + 32 * ti                                 # ds_bpermute address / 4
        w = dwords[src_lane] >> (16 * (lane & 1)).astype(np.uint32)
        i_byte, q_byte = w & 255, (w >> 8) & 255
        assert np.array_equal(i_byte, rows[ti, 0::2]) and np.array_equal(q_byte, rows[ti, 1::2])
    nbytes = 128                                                         # the stream ends after the first row of the pair
    off = 4 * lane
    got = np.where(off + 4 <= nbytes, dwords, 0)                         # raw buffer load: a lane out of range reads zero
    assert np.array_equal(got[:32], dwords[:32]) and not got[32:].any()
