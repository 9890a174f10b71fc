"""Parity of the HIP path (through the C ABI) against the oracle -- the parity tests proper.

Bar: bit-exact.  The FIR kernels accumulate in the reference's tap order (dsputils.rs:31); in
reference rounding (mul, add) they must equal the oracle's fold bit for bit, in fused rounding they
must equal the same fold written with fmaf.  The FFT kernels keep the published kissfft butterfly
order, so spectra must equal oracle_kiss.c bit for bit.  Tolerance vs float64 numpy is also checked
and written down where it is used.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


# ---------------------------------------------------------------- synthetic input
def test_synth_matches_host_hash(gpu, redio, oracle):
    for first, n in ((0, 1000), (12345, 4097), ((1 << 32) - 5, 64)):
        assert same_bits(redio.synth_iq(0x5EED0002, first, n).cpu().numpy(), oracle.synth_iq(0x5EED0002, first, n))
        assert same_bits(redio.synth_f32(7, first, n).cpu().numpy(), oracle.synth_f32(7, first, n))


# ---------------------------------------------------------------- A1 convolve (host drop-in)
@pytest.mark.parametrize("nu,nv", [(1, 1), (63, 63), (64, 63), (1024, 63), (1024, 127), (5000, 3), (4096, 200), (10, 11)])
def test_convolve_host_bit_exact(gpu, redio, oracle, nu, nv):
    u = oracle.synth_f32(1, 0, nu)
    v = oracle.synth_f32(2, 0, nv)
    got = redio.dsputils.convolve(u, v)
    want = oracle.convolve(u, v)
    assert same_bits(got, want)


def test_convolve_empty_taps_is_the_reference_panic(gpu, redio):
    with pytest.raises(redio.RedioError) as e:
        redio.dsputils.convolve([1.0, 2.0], [])
    assert e.value.code == -5  # REDIO_ERR_ASSERT: windows(0) panics in the reference


def test_convolve_with_quirk_lpf_taps_nan_position(gpu, redio, oracle):
    # lpf() as written has NaN at tap 1: every output that touches it is NaN
    taps = redio.dsputils.lpf(63, 0.1)
    assert np.isnan(taps[1]) and np.isnan(taps).sum() == 1
    u = oracle.synth_f32(3, 0, 300)
    got, want = redio.dsputils.convolve(u, taps), oracle.convolve(u, oracle.lpf(63, 0.1))
    assert np.isnan(got).all() and np.isnan(want).all() and len(got) == len(want)


# ---------------------------------------------------------------- device FIR
FIR_CASES = [(127, 5), (127, 1), (63, 1), (63, 5), (3, 1), (33, 2), (200, 7), (1, 1),
             # the chunked kernel: whole and partial 16-tap chunks for every compiled decimation, and its neighbours
             (64, 1), (16, 1), (15, 1), (17, 1), (100, 3), (48, 4), (31, 5), (129, 8), (255, 10), (2, 2), (40, 6), (1000, 1)]


@pytest.mark.parametrize("k,d", FIR_CASES)
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("cplx", [True, False])
def test_fir_bit_exact(gpu, redio, oracle, k, d, fused, cplx):
    taps = oracle.synth_f32(11, 0, k) if k not in (63, 127) else oracle.lpf_corrected(k, 0.08)
    for n in (k, k + 1, 4096 * d + k - 1, 20000, 33333):
        x = oracle.synth_iq(5, 0, n) if cplx else oracle.synth_f32(5, 0, n)
        plan = redio.Fir(taps, d, complex_input=cplx, fused=fused)
        got = plan(gpu.from_numpy(x).cuda()).cpu().numpy()
        want = oracle.fir(x, taps, d, fused)
        assert same_bits(got, want), (k, d, fused, cplx, n)


@pytest.mark.parametrize("d,chunk", [(3, 24), (5, 40), (10, 40), (4, 16), (8, 16), (2, 16)])
@pytest.mark.parametrize("fused", [False, True])
def test_fir_chunked_whole_chunk_seams(gpu, redio, oracle, d, chunk, fused):
    # the chunked kernel walks the taps in whole chunks of one lane stride (24 taps at / 3, 40 at / 5 and / 10) with immediate-offset window
    # reads, then the remainder's binary digits: tap counts either side of one, two and five whole chunks, a remainder with every digit set,
    # inputs that end inside a tile, exactly on one, and in the next; an unaligned view (scalar tile loads)
    for k in (chunk - 1, chunk, chunk + 1, 2 * chunk - 1, 2 * chunk, 2 * chunk + 1, 5 * chunk + chunk - 1):
        taps = oracle.synth_f32(100 + k, 0, k)
        tile_out = 256 * (2 if d >= 8 else 4)
        for n_out in (1, tile_out - 1, tile_out, tile_out + 1, 3 * tile_out + 17):
            n = (n_out - 1) * d + k
            x = oracle.synth_iq(7, 0, n + 1)
            dx = gpu.from_numpy(x).cuda()
            plan = redio.Fir(taps, d, complex_input=True, fused=fused)
            assert same_bits(plan(dx[:n]).cpu().numpy(), oracle.fir(x[:n], taps, d, fused)), (k, d, fused, n_out)
        assert same_bits(plan(dx[1:]).cpu().numpy(), oracle.fir(x[1:], taps, d, fused)), (k, d, fused, "view off the 16-byte grid")


def test_fir_short_input_is_empty(gpu, redio, oracle):
    plan = redio.Fir(oracle.lpf_corrected(63, 0.1), 1)
    assert plan.nout(62) == 0 and plan.nout(63) == 1
    assert plan(gpu.zeros(10, dtype=gpu.complex64, device="cuda")).numel() == 0


def test_fir_fused_vs_reference_rounding_tolerance(gpu, redio, oracle):
    # fused (fmaf) deviates from the reference fold by at most K * eps * sum|x*h| per output
    taps = oracle.lpf_corrected(127, 0.08)
    x = oracle.synth_iq(9, 0, 50000)
    a = redio.Fir(taps, 5, fused=True)(gpu.from_numpy(x).cuda()).cpu().numpy()
    b = oracle.fir(x, taps, 5, fused=False)
    bound = 127 * 2.0 ** -24 * np.abs(taps).sum()  # |x| < 1
    assert np.abs(a - b).max() <= bound
    ref = np.correlate(x.astype(np.complex128), taps.astype(np.float64), "valid")[::5]
    assert np.abs(a - ref).max() <= bound


def test_fir_decimate_is_stride_of_full(gpu, redio, oracle):
    taps = oracle.lpf_corrected(127, 0.08)
    x = gpu.from_numpy(oracle.synth_iq(4, 0, 30000)).cuda()
    full = redio.Fir(taps, 1)(x).cpu().numpy()
    dec = redio.Fir(taps, 5)(x).cpu().numpy()
    assert same_bits(dec, full[::5][: len(dec)])


# ---------------------------------------------------------------- FFT
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 15, 16, 30, 32, 64, 96, 100, 128, 160, 192, 243, 256, 320, 384, 512, 640, 768, 1000, 1024, 1280, 1536, 2048, 2560, 3072, 4096, 6144, 8192, 11, 221])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft_bit_exact(gpu, redio, oracle, n, inverse):
    nb = 5
    x = oracle.synth_iq(n, 0, n * nb)
    got = redio.Fft(n, inverse)(gpu.from_numpy(x).cuda()).cpu().numpy()
    want = oracle.fft(x, n, inverse)
    assert same_bits(got, want), (n, inverse)


CT_SIZES = [9, 10, 12, 15, 18, 20, 24, 25, 27, 30, 36, 40, 45, 48, 50, 54, 60, 72, 75, 80, 81, 90, 96, 100, 108, 120, 125, 135, 144, 150, 160, 162, 180, 192, 200, 216, 225, 240, 243, 250, 270, 288, 300, 320, 324, 360, 375, 384, 400, 405, 432, 450, 480, 486, 500, 540, 576, 600, 625, 640, 648, 675, 720, 729, 750, 768, 800, 810, 864, 900, 960, 972, 1000, 1080, 1125, 1152, 1200, 1215, 1250, 1280, 1296, 1350, 1440, 1458, 1500, 1536, 1600, 1620, 1728, 1800, 1875, 1920, 1944, 2000, 2025, 2160, 2187, 2250, 2304, 2400, 2430, 2500, 2560, 2592, 2700, 2880, 2916, 3000, 3072, 3125, 3200, 3240, 3375, 3456, 3600, 3645, 3750, 3840, 3888, 4000, 4050, 4320, 4374, 4500, 4608, 4800, 4860, 5000, 5120, 5184, 5400, 5625, 5760, 5832, 6000, 6075, 6144, 6250, 6400, 6480, 6561, 6750, 6912, 7200, 7290, 7500, 7680, 7776, 8000, 8100, 8640, 8748, 9000, 9216, 9375, 9600, 9720, 10000, 10125, 10240, 10368, 10800, 10935, 11250, 11520, 11664, 12000, 12150, 12288, 12500, 12800, 12960, 13122, 13500, 13824, 14400, 14580, 15000, 15360, 15552, 15625, 16000, 16200]  # every 5-smooth size up to 16384 with a compile-time kernel


@pytest.mark.parametrize("n", CT_SIZES)
def test_fft_compile_time_mixed_radix_sizes(gpu, redio, oracle, n):
    # every 2^a 3^b 5^c size that has its own compile-time kernel (fft_kernels.hip, REDIO_CT list), ragged batch counts
    nb = 7 if n < 2000 else 3
    x = oracle.synth_iq(n + 17, 0, n * nb)
    d = gpu.from_numpy(x).cuda()
    for inverse in (False, True):
        assert same_bits(redio.Fft(n, inverse)(d).cpu().numpy(), oracle.fft(x, n, inverse)), (n, inverse)
    want = oracle.fft(x, n, False)
    redio.Fft(n, False)(d, out=d)
    assert same_bits(d.cpu().numpy(), want)


@pytest.mark.parametrize("n", [16, 64, 100, 243, 256, 768, 1000, 1024, 1080, 1536, 2048, 4096, 6144, 8192, 8193, 16384, 65536])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft_strided_blocks(gpu, redio, oracle, n, inverse):
    # redio_fft_enqueue_strided through every kernel family: overlapping blocks (the overlap-save framing) and blocks with gaps
    nb = 6 if n < 8192 else 3
    for stride in (n - n // 3 if n > 3 else n, n + 5):
        x = oracle.synth_iq(n + stride, 0, (nb - 1) * stride + n)
        got = redio.Fft(n, inverse).strided(gpu.from_numpy(x).cuda(), nb, stride).cpu().numpy()
        want = np.concatenate([oracle.fft(x[b * stride: b * stride + n], n, inverse) for b in range(nb)])
        assert same_bits(got, want), (n, inverse, stride)


@pytest.mark.parametrize("n", [8193, 8209, 9 * 1031, 2 * 8191, 20011])
def test_fft_large_sizes_with_big_prime_factors(gpu, redio, oracle, n):
    # too large for the out-of-place generic butterfly in LDS: global-memory stages, the generic radix out of place
    x = oracle.synth_iq(n, 0, n * 2)
    d = gpu.from_numpy(x).cuda()
    want = oracle.fft(x, n, False)
    assert same_bits(redio.Fft(n, False)(d).cpu().numpy(), want)
    redio.Fft(n, False)(d, out=d)                      # in place
    assert same_bits(d.cpu().numpy(), want)
    xi = oracle.synth_iq(n + 1, 0, n)
    assert same_bits(redio.Fft(n, True)(gpu.from_numpy(xi).cuda()).cpu().numpy(), oracle.fft(xi, n, True))


@pytest.mark.parametrize("n", [16384, 32768, 65536, 131072, 262144, 524288, 1048576])
def test_fft_large_global_path(gpu, redio, oracle, n):
    x = oracle.synth_iq(n, 0, n * 2)
    d = gpu.from_numpy(x).cuda()
    got = redio.Fft(n, False)(d).cpu().numpy()
    assert same_bits(got, oracle.fft(x, n, False))
    # tolerance vs float64: relative L2 <= 2e-6
    ref = np.fft.fft(x.astype(np.complex128).reshape(2, n), axis=1).reshape(-1)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 2e-6
    # in place
    redio.Fft(n, False)(d, out=d)
    assert same_bits(d.cpu().numpy(), got)


@pytest.mark.parametrize("inverse", [False, True])
def test_fft_65536_large_batch(gpu, redio, oracle, inverse):
    """A batch of 65536-point transforms well past the Infinity Cache (401 transforms, 210 MB each way): the second pass walks the batch
    against the first (fft_kernels.hip launch_fftbig).  Transforms spread over the batch, the first and the last against the oracle; the
    whole batch against the same transforms computed in two calls; in place."""
    n, nb = 65536, 401
    x = redio.synth_iq(0x5EED0016, 0, n * nb)
    plan = redio.Fft(n, inverse)
    y = plan(x)
    for b in (0, 1, 127, 128, 129, 255, 256, 257, 383, 384, 385, nb - 1):
        xw = oracle.synth_iq(0x5EED0016, n * b, n)
        assert same_bits(y[n * b: n * (b + 1)].cpu().numpy(), oracle.fft(xw, n, inverse)), (inverse, b)
    half = 200 * n
    y2 = gpu.cat([plan(x[:half]), plan(x[half:])])
    assert gpu.equal(y, y2)
    z = x.clone()
    plan(z, out=z)  # in place
    assert gpu.equal(y, z)


@pytest.mark.parametrize("n", [16875, 17280, 18000, 20000, 30000, 48000, 50000, 65610, 78125, 100000, 196608, 250000, 1000000, 1594323])
def test_fft_large_mixed_radix_tile_passes(gpu, redio, oracle, n):
    # radix-2/3/4/5 sizes above 16384 that are not powers of two: one LDS tile pass per group of stages (fft_tile_pass_kernel)
    nb = 3 if n < 100000 else 1
    x = oracle.synth_iq(n & 0xFFFF, 0, n * nb)
    d = gpu.from_numpy(x).cuda()
    for inverse in (False, True):
        assert same_bits(redio.Fft(n, inverse)(d).cpu().numpy(), oracle.fft(x, n, inverse)), (n, inverse)
    want = oracle.fft(x, n, False)
    redio.Fft(n, False)(d, out=d)  # in place: staged by the C-ABI layer
    assert same_bits(d.cpu().numpy(), want)
    if n <= 50000:  # strided blocks
        xs = oracle.synth_iq(n + 1, 0, n + 2 * (n - 7))
        got = redio.Fft(n, False).strided(gpu.from_numpy(xs).cuda(), 3, n - 7).cpu().numpy()
        assert same_bits(got, np.concatenate([oracle.fft(xs[b * (n - 7): b * (n - 7) + n], n, False) for b in range(3)]))


@pytest.mark.parametrize("n", [1 << 15, 1 << 18, 1 << 19, 1 << 20, 1 << 22])
def test_fft_two_pass_five_stage_tiles(gpu, redio, oracle, n):
    # plan B of the multi-pass transforms (fft_kernels.hip, fftbig_plan_b): 2^15 = 3 + 5 stages, 4^9 = 4 + 5, 2^19 = 3 + 5 + 2,
    # 4^10 = 5 + 5, 4^11 = 5 + 5 + 1 (fftbig_first5_kernel / fftbig_mid5_kernel; 2^23 and 2^24 are in the test below):
    # both directions, several transforms per call, strided (overlapping) blocks, and the same plan size through the
    # overlap-save path, which keeps the four-stage passes and their tables
    nb = 3 if n <= (1 << 20) else 1
    x = oracle.synth_iq((n & 0xFFFF) + 5, 0, n * nb)
    d = gpu.from_numpy(x).cuda()
    for inverse in (False, True):
        assert same_bits(redio.Fft(n, inverse)(d).cpu().numpy(), oracle.fft(x, n, inverse)), (n, inverse)
    xs = oracle.synth_iq(n + 1, 0, n + 2 * (n - 5))
    got = redio.Fft(n, False).strided(gpu.from_numpy(xs).cuda(), 3, n - 5).cpu().numpy()
    assert same_bits(got, np.concatenate([oracle.fft(xs[b * (n - 5): b * (n - 5) + n], n, False) for b in range(3)]))
    taps = oracle.lpf_corrected(1001, 0.02)
    xo = oracle.synth_iq(9, 0, n + (n - 1000) + 33)
    assert same_bits(redio.OverlapSave(taps, n)(gpu.from_numpy(xo).cuda()).cpu().numpy(), oracle.overlap_save(xo, taps, n))


@pytest.mark.parametrize("n", [1 << 21, 1 << 22, 1 << 23, 1 << 24])
def test_fft_multi_pass_powers_of_four(gpu, redio, oracle, n):
    # 2 * 4^10, 4^11, 2 * 4^11, 4^12: gather pass, in-place four-stage passes, register-only last stages (fft_kernels.hip, fftbig_*)
    x = oracle.synth_iq(n & 0xFFFF, 0, n)
    d = gpu.from_numpy(x).cuda()
    for inverse in (False, True):
        assert same_bits(redio.Fft(n, inverse)(d).cpu().numpy(), oracle.fft(x, n, inverse)), (n, inverse)


def test_fft1024_many_blocks_and_inplace(gpu, redio, oracle):
    x = oracle.synth_iq(77, 0, 1024 * 37)
    d = gpu.from_numpy(x).cuda()
    plan = redio.Fft(1024)
    want = oracle.fft(x, 1024)
    assert same_bits(plan(d).cpu().numpy(), want)
    plan(d, out=d)
    assert same_bits(d.cpu().numpy(), want)


def test_fft_roundtrip_is_n_times_identity(gpu, redio, oracle):
    # kissfft is unnormalised in both directions: ifft(fft(x)) = N x
    x = oracle.synth_iq(8, 0, 1024 * 4)
    d = gpu.from_numpy(x).cuda()
    y = redio.Fft(1024, True)(redio.Fft(1024, False)(d)).cpu().numpy()
    assert np.abs(y / 1024 - x).max() <= 1e-5


def test_kiss_fft_dropin_symbols(gpu, redio, oracle):
    # the three C symbols the reference's Rust binds (kissfft.rs:13-15), host buffers, synchronous
    from libredio_amd import kissfft
    for n, inv in ((1024, 0), (64, 1), (30, 0)):
        cfg = kissfft.Cfg(n, inv)
        x = oracle.synth_iq(n + inv, 0, n)
        assert same_bits(cfg(x), oracle.fft(x, n, bool(inv)))
        # in place (fin == fout)
        buf = x.copy()
        redio.kisslib().kiss_fft(cfg._cfg, buf.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p))
        assert same_bits(buf, oracle.fft(x, n, bool(inv)))
        cfg.close()
    redio.kisslib().kiss_fft_cleanup()


def test_kiss_fft_placement_stride_and_helpers(gpu, redio, oracle):
    """kiss_fft_alloc's mem / lenmem protocol (the reference passes NULL, NULL -- kissfft.rs:19 -- but the C header promises both forms):
    a size query, a cfg placed in the caller's memory, too small a buffer; kiss_fft_stride; kiss_fft_next_fast_size; kiss_fft_free"""
    L = redio.kisslib()
    L.kiss_fft_alloc.restype = C.c_void_p
    L.kiss_fft_alloc.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    L.kiss_fft_stride.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.kiss_fft.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.kiss_fft_free.argtypes = [C.c_void_p]
    L.kiss_fft_next_fast_size.restype = C.c_int
    need = C.c_size_t(0)
    assert L.kiss_fft_alloc(256, 0, None, C.byref(need)) is None and 0 < need.value < 4096    # size query
    small = C.create_string_buffer(8); got = C.c_size_t(8)
    assert L.kiss_fft_alloc(256, 0, small, C.byref(got)) is None and got.value == need.value   # too small: the size comes back
    mem = C.create_string_buffer(need.value); got = C.c_size_t(need.value)
    cfg = L.kiss_fft_alloc(256, 1, mem, C.byref(got))
    assert C.addressof(mem) <= cfg < C.addressof(mem) + 16 and cfg % 8 == 0                   # placed in the caller's memory (first aligned address)
    x = oracle.synth_iq(91, 0, 256); out = np.empty_like(x)
    L.kiss_fft(cfg, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert same_bits(out, oracle.fft(x, 256, True))
    wide = oracle.synth_iq(92, 0, 256 * 3)                                                    # every third sample
    L.kiss_fft_stride(cfg, wide.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 3)
    assert same_bits(out, oracle.fft(np.ascontiguousarray(wide[::3]), 256, True))
    L.kiss_fft_free(cfg)                                                                      # releases the device side, not the caller's memory
    mem[0:4] = b"\0\0\0\0"                                                                    # still ours to write
    for n, want in ((1, 1), (7, 8), (17, 18), (1000, 1000), (1025, 1080), (4097, 4320)):      # next size with factors 2, 3, 5 only
        assert L.kiss_fft_next_fast_size(n) == want, n


def test_kiss_fft_misaligned_mem_and_completion_fallback(gpu, redio, oracle):
    """kiss_fft_alloc takes a caller's buffer of ANY alignment (the published contract: a cfg carved out of a byte arena) and places the state
    at the first aligned address inside it; a call whose completion poll is cut to zero takes the ordinary stream wait and returns the same
    bits (kissfft_shim.cpp: the poll is bounded by wall time, redio_kiss_fft_set_spin_ns is the hook)."""
    L = redio.kisslib()
    need = C.c_size_t(0)
    L.kiss_fft_alloc(1024, 0, None, C.byref(need))
    mem = C.create_string_buffer(need.value + 16)
    base = C.addressof(mem)
    odd = base + 1 if base % 2 == 0 else base + 2                                    # 1 (mod 2): never aligned for a pointer
    got = C.c_size_t(need.value)
    L.kiss_fft_alloc.restype = C.c_void_p
    L.kiss_fft_alloc.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    L.kiss_fft.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.kiss_fft_free.argtypes = [C.c_void_p]
    pc = L.kiss_fft_alloc(1024, 0, C.c_void_p(odd), C.byref(got))
    assert pc is not None and odd <= pc < odd + 8 and pc % 8 == 0 and pc + (need.value - 7) <= odd + need.value and got.value == need.value
    xo = oracle.synth_iq(78, 0, 1024); yo = np.empty_like(xo)
    L.kiss_fft(pc, xo.ctypes.data_as(C.c_void_p), yo.ctypes.data_as(C.c_void_p))
    assert same_bits(yo, oracle.fft(xo, 1024, False))
    L.kiss_fft_free(pc)
    from libredio_amd import kissfft
    cfg = kissfft.Cfg(1024, 0)
    x = oracle.synth_iq(77, 0, 1024)
    want = oracle.fft(x, 1024, False)
    try:
        L.redio_kiss_fft_set_spin_ns(0)                                                       # every call: the fall-back branch
        for _ in range(3):
            assert same_bits(cfg(x), want)
    finally:
        L.redio_kiss_fft_set_spin_ns(100000)
    assert same_bits(cfg(x), want)
    cfg.close()


def test_big_fft_into_an_odd_sample_view(gpu, redio, oracle):
    """Stand-alone transforms of 32768 points and more work in place in the caller's `out` with 16-byte accesses; `out` is only promised
    8-byte (one cf32 sample) alignment: a view that starts on an odd sample must give the oracle's bits (the pair tile
    program's accesses to caller-owned memory are declared 8-byte aligned)."""
    import torch
    for n, nb in ((32768, 3), (65536, 2), (1 << 17, 1), (1 << 19, 1)):
        xh = oracle.synth_iq(0x0DD + n, 0, n * nb)
        x = gpu.from_numpy(np.concatenate([np.zeros(1, np.complex64), xh])).cuda()[1:]   # input on an odd sample too
        buf = torch.zeros(n * nb + 3, dtype=torch.complex64, device="cuda")
        out = buf[1:1 + n * nb]
        assert out.data_ptr() % 16 == 8
        redio.Fft(n)(x, out=out)
        got = out.cpu().numpy()
        for b in range(nb):
            assert same_bits(got[b * n:(b + 1) * n], oracle.fft(xh[b * n:(b + 1) * n], n, False)), (n, b)
        assert buf[0].item() == 0 and buf[-1].item() == 0 and buf[-2].item() == 0


def test_kissfft_block_function(gpu, redio, oracle):
    import queue
    import threading
    from libredio_amd import kissfft
    pin, cout = queue.Queue(), queue.Queue()
    t = threading.Thread(target=kissfft.fft, args=(pin, cout, 256, 0))
    t.start()
    msgs = [oracle.synth_iq(i, 0, 256) for i in range(4)]
    for m in msgs:
        pin.put(m)
    pin.put(None)
    t.join()
    for m in msgs:
        assert same_bits(cout.get(), oracle.fft(m, 256))


# ---------------------------------------------------------------- C2 chain
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("nblocks", [1, 3, 4, 5, 9, 2051])
def test_chain_bit_exact(gpu, redio, oracle, fused, nblocks):
    taps = oracle.lpf_corrected(127, 0.08)
    n = nblocks * 1024 * 5 + 126 + 3  # a few samples that do not complete a block are dropped
    x = oracle.synth_iq(0x5EED0002, 0, n)
    d = gpu.from_numpy(x).cuda()
    chain = redio.Chain(taps, 5, 1024, fused=fused)
    assert chain.is_fused and chain.nblocks(n) == nblocks
    want = oracle.chain_fir_fft(x, taps, 5, 1024, fused=fused)
    got = chain(d).cpu().numpy()
    assert same_bits(got, want)
    chain.set_unfused(True)  # FIR kernel + FFT kernel through an intermediate: same bits
    chain.reserve(n)         # sized up front: enqueue then never allocates (graph-capturable)
    assert same_bits(chain(d).cpu().numpy(), want)
    # the diagnostic stamps are per plan: a stamped plan does not leak writes into another plan's launches
    chain.set_unfused(False)
    other = redio.Chain(taps, 5, 1024, fused=fused)
    stamps = gpu.zeros(4 * 4096, dtype=gpu.int64, device="cuda")
    chain.set_debug_stamps(stamps)
    assert same_bits(chain(d).cpu().numpy(), want) and int((stamps != 0).sum()) > 0
    chain.set_debug_stamps(None)
    stamps.zero_()
    assert same_bits(other(d).cpu().numpy(), want) and same_bits(chain(d).cpu().numpy(), want)
    assert int((stamps != 0).sum()) == 0


def test_chain_stamp_buffer_capacity_is_honoured(gpu, redio, oracle):
    """redio_chain_set_debug_stamps(buf, capacity): a launch with more wavefronts than the buffer has records (the launcher gives a
    wavefront at most 4 blocks, so 2^28 samples are 13107 wavefronts) stamps only the first `capacity` and writes nothing beyond."""
    taps = oracle.lpf_corrected(127, 0.08)
    nblocks = 5000
    n = nblocks * 5120 + 126
    x = redio.synth_iq(0x5EED0002, 0, n)
    chain = redio.Chain(taps, 5, 1024, fused=True)
    waves, bpw = chain.launch_waves(nblocks), chain.blocks_per_wave(nblocks)
    assert waves == -(-nblocks // bpw) and waves > 1024
    cap = 1024
    guard = gpu.full((4 * cap + 4096,), -7, dtype=gpu.int64, device="cuda")
    chain.set_debug_stamps(guard[: 4 * cap])
    out = chain(x)
    chain.set_debug_stamps(None)
    g = guard.cpu().numpy()
    assert (g[4 * cap:] == -7).all(), "a wavefront beyond the buffer's capacity wrote a stamp"
    assert (g[: 4 * cap].reshape(cap, 4)[:, 1] > 0).all(), "every wavefront below the capacity leaves its record"
    assert same_bits(out.cpu().numpy()[-2:], oracle.chain_fir_fft(oracle.synth_iq(0x5EED0002, (nblocks - 2) * 5120, 2 * 5120 + 126), taps, 5, 1024, fused=True))
    assert chain.kernel_name == "chain_v4_kernel<127,5,true,2,8,false,true,false,false>"
    two = redio.Chain(taps, 2, 64, fused=False)
    assert two.kernel_name is None and two.launch_waves(100) == 0 and two.blocks_per_wave(100) == 0


def test_chain_fused_rounding_vs_reference_rounding_tolerance(gpu, redio, oracle):
    """The stated f32 tolerance of the fmaf build of the chain against the REFERENCE arithmetic (Rust never
    contracts: separately rounded multiply and add, dsputils.rs:31).  Per FIR output the two folds differ
    by at most K * 2^-24 * sum|x*h|; the unnormalised 1024-point transform can at worst add
    those deviations coherently, so per spectrum bin |d| <= 1024 * K * 2^-24 * max_i sum_j |x[i+j] h[j]|,
    and in practice the relative L2 distance is ~1e-7 (asserted <= 1e-6)."""
    taps = oracle.lpf_corrected(127, 0.08)
    n = 64 * 5120 + 126
    x = oracle.synth_iq(0x5EED0002, 0, n)
    want = oracle.chain_fir_fft(x, taps, 5, 1024, fused=False)          # reference rounding
    got = redio.Chain(taps, 5, 1024, fused=True)(gpu.from_numpy(x).cuda()).cpu().numpy()
    assert not same_bits(got, want)                                      # the modes do differ ...
    bound = 1024 * 127 * 2.0 ** -24 * (np.abs(x).max() * np.sqrt(2) * np.abs(taps).sum())
    assert np.abs(got - want).max() <= bound
    rel = np.linalg.norm((got - want).ravel()) / np.linalg.norm(want.ravel())
    assert rel <= 1e-6, rel
    # ... and the reference-rounding build is bit-identical to the reference arithmetic
    exact = redio.Chain(taps, 5, 1024, fused=False)(gpu.from_numpy(x).cuda()).cpu().numpy()
    assert same_bits(exact, want)


@pytest.mark.parametrize("k,d", [(63, 5), (127, 1), (63, 1), (127, 3)])
@pytest.mark.parametrize("fused", [True, False])
def test_chain_other_fused_shapes(gpu, redio, oracle, k, d, fused):
    # the single-kernel chain is built for these tap / decimation pairs too (1024-point blocks)
    taps = oracle.lpf_corrected(k, 0.4 / max(d, 2))
    for nblocks in (1, 5, 37):
        n = nblocks * 1024 * d + (k - d) + 2
        x = oracle.synth_iq(0x5EED0002, 0, n + 1)
        dx = gpu.from_numpy(x).cuda()
        chain = redio.Chain(taps, d, 1024, fused=fused)
        assert chain.is_fused and chain.nblocks(n) == nblocks
        want = oracle.chain_fir_fft(x[:n], taps, d, 1024, fused=fused)
        assert same_bits(chain(dx[:n]).cpu().numpy(), want), (k, d, nblocks)
        # a stream that starts on an odd sample (8-byte aligned only) takes the two-kernel path: same bits
        want1 = oracle.chain_fir_fft(x[1:n + 1], taps, d, 1024, fused=fused)
        assert same_bits(chain(dx[1:n + 1]).cpu().numpy(), want1), (k, d, nblocks, "unaligned")


def test_chain_other_shape_runs_unfused(gpu, redio, oracle):
    taps = oracle.lpf_corrected(63, 0.1)
    x = oracle.synth_iq(3, 0, 64 * 2 * 10 + 62)
    chain = redio.Chain(taps, 2, 64, fused=False)
    assert not chain.is_fused
    got = chain(gpu.from_numpy(x).cuda()).cpu().numpy()
    assert same_bits(got, oracle.chain_fir_fft(x, taps, 2, 64, fused=False))


def test_chain_full_size_properties(gpu, redio, oracle):
    """BASELINE.json configs[1] at full size (2^28 samples): size-independent properties.
    (a) the first and last blocks equal the oracle on just their own input window (tile independence);
    (b) linearity: chain(a*x) == a*chain(x) exactly for a power of two;
    (c) a checksum of checksums is reproducible run to run."""
    taps = oracle.lpf_corrected(127, 0.08)
    n = 1 << 28
    x = redio.synth_iq(0x5EED0002, 0, n)
    for fused in (False, True):     # reference rounding, then the fmaf build bench.py times by default
        chain = redio.Chain(taps, 5, 1024, fused=fused)
        nb = chain.nblocks(n)
        assert nb == ((n - 127) // 5 + 1) // 1024
        out = chain(x)
        # a wave owns a run of blocks_per_wave consecutive blocks and carries the FIR halo inside LDS from one sub-tile to the
        # next (chain_v4.hip).  The run length comes from the launcher itself (redio_chain_blocks_per_wave), and EVERY block of
        # the launch -- so the first and the last block of every run, whatever that length is -- is compared with the oracle,
        # which computes each slice of 4096 blocks from nothing but that slice's own input window (tile independence)
        bpw, waves = chain.blocks_per_wave(nb), chain.launch_waves(nb)
        assert 1 <= bpw and waves == -(-nb // bpw) and waves > 4096, (bpw, waves)
        outh = out.cpu().numpy()
        step = 4096
        assert step % bpw == 0 or bpw > step  # slices start on run boundaries: a run's first block is some slice's block too
        for b0 in range(0, nb, step):
            cnt = min(step, nb - b0)
            xw = oracle.synth_iq(0x5EED0002, b0 * 5120, cnt * 5120 + 126)
            want = oracle.chain_fir_fft(xw, taps, 5, 1024, fused=fused)
            assert want.shape == (cnt, 1024)
            if not same_bits(outh[b0:b0 + cnt], want):
                bad = [b0 + i for i in range(cnt) if not same_bits(outh[b0 + i], want[i])]
                raise AssertionError((fused, "blocks that differ", bad[:8], "position in their run", [b % bpw for b in bad[:8]]))
        del outh
    s1 = gpu.view_as_real(out).view(gpu.int32).sum(dtype=gpu.int64).item()
    out2 = chain(x * 4.0)
    assert gpu.equal(out2, out * 4.0)
    out3 = chain(x)
    s3 = gpu.view_as_real(out3).view(gpu.int32).sum(dtype=gpu.int64).item()
    assert s1 == s3


@pytest.mark.gpu
@pytest.mark.parametrize("k,d", [(64, 1), (31, 2), (101, 3), (100, 4), (255, 5), (77, 8), (255, 10), (500, 1), (1000, 2), (16, 1), (1, 1), (5, 3)])
@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("fused", [False, True])
def test_fir_streaming_kernel_long_inputs(gpu, redio, oracle, k, d, cplx, fused):
    """Long inputs (hundreds of tiles per decimation class, 16-byte loads and the transposed stores on the aligned view, the
    element-wise paths on the unaligned one): dsputils::convolve's fold order (dsputils.rs:30-32) per output, checked on
    windows spread over the stream, at both ends and around the last tile boundaries."""
    n = 600 * 2048 * d + 12345 + k
    x = (oracle.synth_iq if cplx else oracle.synth_f32)(4242 + k, 0, n)
    taps = oracle.synth_f32(99 + d, 0, k)
    plan = redio.Fir(taps, d, complex_input=cplx, fused=fused)
    dx = gpu.from_numpy(x).cuda()
    got = plan(dx).cpu().numpy()
    nout = plan.nout(n)
    assert got.shape[0] == nout == (n - k) // d + 1
    rng = np.random.default_rng(k * 31 + d)
    starts = [0, nout - 3000, max(0, nout - 20000)] + [int(v) for v in rng.integers(0, nout - 3000, 12)]
    # the last tile boundaries (tiles are 512 ... 2048 outputs)
    starts += [max(0, (nout // 2048 - j) * 2048 - 1500) for j in range(0, 4)]
    for o0 in starts:
        o1 = min(nout, o0 + 3000)
        want = oracle.fir(x[o0 * d: (o1 - 1) * d + k], taps, d, fused)
        assert np.array_equal(bits(got[o0:o1]), bits(want)), (k, d, cplx, fused, o0)
    # an unaligned view of the same stream: same outputs, shifted
    off = 1
    got2 = plan(dx[off:]).cpu().numpy()
    m = min(len(got2), 5000)
    want2 = oracle.fir(x[off: off + (m - 1) * d + k], taps, d, fused)
    assert np.array_equal(bits(got2[:m]), bits(want2))
