"""The bit-exact ingest / slicing path on the MI355X against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_data_to_samples_all_256_byte_values(gpu, redio, oracle):
    d = np.repeat(np.arange(256, dtype=np.uint8), 2)
    d = np.concatenate([d, np.arange(256, dtype=np.uint8)[::-1], np.arange(256, dtype=np.uint8)])  # mixed pairs too
    got = redio.bitfount.data_to_samples(gpu.from_numpy(d).cuda()).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.data_to_samples(d)))
    with pytest.raises(redio.RedioError) as e:
        redio.bitfount.data_to_samples(gpu.zeros(7, dtype=gpu.uint8, device="cuda"))   # i[1] index panic
    assert e.value.code == -5


def test_norm_is_hypotf_on_the_whole_u8_domain(gpu, redio, oracle):
    a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    d = np.stack([a.reshape(-1), b.reshape(-1)], axis=1).reshape(-1)                 # all 65536 IQ byte pairs
    x = oracle.data_to_samples(d)
    want = oracle.norm(x)
    assert np.array_equal(bits(redio.bitfount.norm(gpu.from_numpy(x).cuda()).cpu().numpy()), bits(want))
    assert np.array_equal(bits(redio.bitfount.ingest_mag(gpu.from_numpy(d).cuda()).cpu().numpy()), bits(want))
    # long streams switch to the LDS-table kernel: the whole domain again, 65 times over plus a ragged tail
    big = np.concatenate([np.tile(d, 65), d[: 2 * 4099]])
    got = redio.bitfount.ingest_mag(gpu.from_numpy(big).cuda()).cpu().numpy()
    assert np.array_equal(bits(got), bits(np.concatenate([np.tile(want, 65), want[:4099]])))
    y = oracle.synth_iq(5, 0, 100003) * np.float32(1000.0)                              # and on random data
    assert np.array_equal(bits(redio.bitfount.norm(gpu.from_numpy(y).cuda()).cpu().numpy()), bits(oracle.norm(y)))


@pytest.mark.parametrize("block", [512, 64, 100, 1])
def test_block_sums_are_sequential_f32(gpu, redio, oracle, block):
    nb = 333
    x = (oracle.synth_f32(8, 0, nb * block) * np.float32(100.0)).astype(np.float32)
    got = redio.bitfount.block_sums(gpu.from_numpy(x).cuda(), block).cpu().numpy()
    want = np.array([oracle.block_sum(x[b * block:(b + 1) * block]) for b in range(nb)], np.float32)
    assert np.array_equal(bits(got), bits(want))


def test_discretize_bit_exact(gpu, redio, oracle):
    for n in (1, 513, 100000):
        x = np.abs(oracle.synth_f32(n, 0, n)) ** 2
        x[n // 2] = 3.0
        got = redio.bitfount.discretize(gpu.from_numpy(x).cuda()).cpu().numpy()
        assert np.array_equal(got, oracle.discretize(x).astype(np.uint8))
    # views that start off a 16-byte boundary, lengths around the 4- and 16-sample vector steps, the max in the head / tail
    base = np.abs(oracle.synth_f32(9, 0, 5000)).astype(np.float32)
    d = gpu.from_numpy(base).cuda()
    for off in (0, 1, 2, 3):
        for n in (2, 3, 4, 5, 15, 16, 17, 31, 33, 4097):
            for peak in (0, n // 2, n - 1):
                x = base[off:off + n].copy()
                x[peak] = 7.0
                d[off + peak] = 7.0
                got = redio.bitfount.discretize(d[off:off + n]).cpu().numpy()
                d[off + peak] = float(base[off + peak])
                assert np.array_equal(got, oracle.discretize(x).astype(np.uint8)), (off, n, peak)
    # a long stream with a ragged end: many workgroups of the four-step slicer, the scalar tail, the peak in the last workgroup
    n = (1 << 22) + 4 * 1024 * 3 + 7
    xl = redio.synth_f32(77, 0, n).abs()
    xl[n - 5] = 9.0
    assert np.array_equal(redio.bitfount.discretize(xl).cpu().numpy(), oracle.discretize(xl.cpu().numpy()).astype(np.uint8))
    # NaN is ignored by f32::max; all-negative input keeps max = 0.0 (the fold's seed)
    x = np.array([np.nan, -1.0, 0.5, 2.0, np.nan, 1.1], np.float32)
    assert redio.bitfount.discretize(gpu.from_numpy(x).cuda()).cpu().numpy().tolist() == oracle.discretize(x).tolist() == [0, 0, 0, 1, 0, 1]
    x = np.array([-3.0, -1.0], np.float32)
    assert redio.bitfount.discretize(gpu.from_numpy(x).cuda()).cpu().numpy().tolist() == oracle.discretize(x).tolist()


def test_trigger_state_machine_matches_the_reference_walk(gpu, redio, oracle):
    # quiet noise with three bursts; blocks of 512 like rtl_source_cmplx (bitfount.rs:17)
    rng = np.random.default_rng(7)
    nb = 600
    blocks = (0.05 * rng.random((nb, 512))).astype(np.float32)
    for start, ln in ((100, 7), (250, 60), (480, 3)):
        blocks[start:start + ln] += 1.0
    dev, ref = redio.bitfount.Trigger(), oracle.Trigger()
    got, want = [], []
    for lo, hi in ((0, 130), (130, 131), (131, 600)):          # state persists across calls
        got += [g.cpu().numpy() for g in dev.feed(gpu.from_numpy(blocks[lo:hi]).cuda())]
        want += ref.feed(blocks[lo:hi])
    assert len(got) == len(want) >= 3
    for g, w in zip(got, want):
        assert len(g) == len(w) and np.array_equal(bits(g), bits(w))
    assert want[0][0] == 0.0 and len(want[0]) % 512 == 1       # the first buffer starts as vec!(0.0) (bitfount.rs:43)


def test_trigger_refuses_a_result_that_does_not_fit_without_consuming(gpu, redio, oracle):
    """A completed trigger buffer is never dropped: when the caller's buffers are too small the call consumes nothing,
    reports what it needs, and the retry with enough room gives exactly the reference walk."""
    import ctypes as C
    rng = np.random.default_rng(8)
    blocks = (0.05 * rng.random((300, 512))).astype(np.float32)
    blocks[100:107] += 1.0
    d = gpu.from_numpy(blocks).cuda()
    want = oracle.Trigger().feed(blocks)
    assert len(want) == 1
    L = redio.lib()
    h = C.c_void_p()
    assert L.redio_trigger_create(C.byref(h)) == 0
    out = gpu.empty(len(want[0]) + 8, dtype=gpu.float32, device="cuda")
    lens = (C.c_size_t * 4)()
    ne, tot = C.c_size_t(0), C.c_size_t(0)
    small = len(want[0]) - 1
    rc = L.redio_trigger_feed(h, C.c_void_p(d.data_ptr()), 300, 512, C.c_void_p(out.data_ptr()), small, lens, 4, C.byref(ne), C.byref(tot), None)
    assert rc == -1 and ne.value == 1 and tot.value == len(want[0])      # REDIO_ERR_ARG + the capacities a retry needs
    rc = L.redio_trigger_feed(h, C.c_void_p(d.data_ptr()), 300, 512, C.c_void_p(out.data_ptr()), out.numel(), None, 0, C.byref(ne), C.byref(tot), None)
    assert rc == -1 and ne.value == 1                                      # no lens array for a call that emits
    rc = L.redio_trigger_feed(h, C.c_void_p(d.data_ptr()), 300, 512, C.c_void_p(out.data_ptr()), out.numel(), lens, 4, C.byref(ne), C.byref(tot), None)
    assert rc == 0 and ne.value == 1 and lens[0] == len(want[0])
    assert np.array_equal(bits(out[:lens[0]].cpu().numpy()), bits(want[0]))
    L.redio_trigger_destroy(h)


def test_shipped_graph_front_end_end_to_end(gpu, redio, oracle):
    """rtl bytes -> data_to_samples -> |x| -> trigger -> discretize, as src/ratpak.rs:60-76 wires them."""
    rng = np.random.default_rng(11)
    nb = 400
    raw = rng.integers(120, 136, size=(nb, 1024), dtype=np.uint8)       # 512 IQ samples per block near mid-scale
    raw[150:190] = rng.integers(0, 256, size=(40, 1024), dtype=np.uint8)  # an OOK burst
    mag_ref = oracle.norm(oracle.data_to_samples(raw.reshape(-1))).reshape(nb, 512)
    bufs_ref = oracle.Trigger().feed(mag_ref)
    mag = redio.bitfount.ingest_mag(gpu.from_numpy(raw.reshape(-1)).cuda()).reshape(nb, 512)
    bufs = redio.bitfount.Trigger().feed(mag)
    assert len(bufs) == len(bufs_ref) >= 1
    for b, r in zip(bufs, bufs_ref):
        assert np.array_equal(bits(b.cpu().numpy()), bits(r))
        assert np.array_equal(redio.bitfount.discretize(b).cpu().numpy(), oracle.discretize(r).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("shape", [(127, 5, 1024), (63, 5, 1024), (127, 1, 1024), (63, 1, 1024), (127, 3, 1024), (127, 5, 256), (31, 2, 64)])
def test_chain_from_u8_bytes(gpu, redio, oracle, shape, fused):
    """redio_chain_enqueue_u8: rtlsdr::data_to_samples (rtlsdr.rs:159-162) -> FIR (dsputils.rs:30-32) -> kissfft (kissfft.rs:20-29) from the
    receiver's u8 I/Q bytes.  (127, 5, 1024) is one kernel (the conversion happens on the way into the LDS image); other shapes and
    unaligned bytes convert first.  Same spectra as the oracle's data_to_samples followed by its chain, bit for bit: every byte value,
    ragged lengths, one block, no block, misaligned bytes, wave-run boundaries."""
    k, d, nfft = shape
    taps = oracle.synth_f32(17, 0, k)
    plan = redio.Chain(taps, d, nfft, fused=fused)
    rng = np.random.default_rng(k + d + nfft)
    for nblk, extra in ((0, 3), (1, 0), (3, d * nfft // 3), (70, d * nfft - 2), (1100, 5)):
        nsamp = (nblk * nfft - 1) * d + k + extra if nblk else k - 1
        raw = rng.integers(0, 256, 2 * nsamp + 4, dtype=np.uint8)
        if len(raw) >= 256:
            raw[:256] = np.arange(256, dtype=np.uint8)   # every byte value at least once
        for off in (0, 2, 1):   # 4-byte aligned, sample-aligned only, odd address
            view = raw[off: off + 2 * nsamp]
            dv = gpu.from_numpy(raw).cuda()[off: off + 2 * nsamp]
            got = plan.from_bytes(dv).cpu().numpy()
            want = oracle.chain_fir_fft(oracle.data_to_samples(view), taps, d, nfft, fused=fused)
            assert got.shape == want.shape == (nblk, nfft)
            assert np.array_equal(bits(got), bits(want)), (shape, fused, nblk, extra, off)
    with pytest.raises(Exception):
        plan.from_bytes(gpu.zeros(7, dtype=gpu.uint8, device="cuda"))   # odd byte count: rtlsdr.rs:160 would index out of bounds


@pytest.mark.gpu
def test_chain_from_u8_bytes_full_size(gpu, redio):
    """BASELINE.json configs[1] at 2^28 samples, from bytes: the one-kernel form gives the bits of the conversion kernel followed by
    the cf32 chain (itself checked against the oracle above and in test_gpu_parity.py) on every block."""
    import libredio_amd.bitfount as B
    n = 1 << 28
    taps = redio.dsputils.lpf_corrected(127, 0.08)
    plan = redio.Chain(taps, 5, 1024, fused=True)
    g = gpu.Generator(device="cuda"); g.manual_seed(5)
    raw = gpu.randint(0, 256, (2 * n,), dtype=gpu.uint8, device="cuda", generator=g)
    got = plan.from_bytes(raw)
    x = B.data_to_samples(raw)
    want = plan(x)
    del x
    assert got.shape == want.shape == (plan.nblocks(n), 1024)
    assert gpu.equal(gpu.view_as_real(got).view(gpu.int32), gpu.view_as_real(want).view(gpu.int32))


def test_trigger_buffer_reset_past_25_6_million_samples(gpu, redio, oracle):
    """bitfount.rs:52-54: a buffer that has grown past 1000 * 50 * 512 samples is dropped and restarts as vec!(0.0).  One quiet block, then
    50 010 loud ones (every one re-arms the 50-block counter), then silence until the counter runs out: the emitted buffer is what was
    collected AFTER the reset -- same length, same bits as the reference walk; fed in three calls that cut inside the run."""
    nb, bl = 50010 + 70, 512
    blocks = np.full((nb, bl), 1e-4, np.float32)
    blocks[1:50011] = (0.5 + 0.25 * np.sin(np.arange(bl, dtype=np.float32)))[None, :]
    blocks[1:50011, 0] += (np.arange(50010) % 7).astype(np.float32) * 0.125      # the blocks differ: a misplaced one shows
    dev, ref = redio.bitfount.Trigger(), oracle.Trigger()
    got, want = [], []
    for lo, hi in ((0, 20000), (20000, 50005), (50005, nb)):
        got += [g.cpu().numpy() for g in dev.feed(gpu.from_numpy(blocks[lo:hi]).cuda())]
        want += ref.feed(blocks[lo:hi])
    assert len(want) == 1 and len(got) == 1
    assert 1 < len(want[0]) < 1000 * 50 * 512 and want[0][0] == 0.0              # restarted once, then collected to the end of the burst
    assert len(got[0]) == len(want[0]) and np.array_equal(bits(got[0]), bits(want[0]))
