"""Streams beyond 2^32 bytes, 2^31 and 2^32 elements (an MI355X holds 288 GB: one call may be handed that much).

Every kernel here was written with 64-bit unit indices and 32-bit offsets inside a unit; nothing but a run at these sizes shows that no
32-bit product crept in.  One cf32 stream of 2^32 + a ragged tail samples (34 GB) is generated on the device by the closed-form hash
 and pushed through each operator; the oracle recomputes, from nothing but its own input window, the outputs next to every
boundary that matters: the start, byte offset 2^32 (sample 2^29), element 2^31, element 2^32, the end.  Bit-exact, as everywhere.
Skipped when the device has less than 120 GB free."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x5EED0009
N = (1 << 32) + 13 * 5120 + 1000  # cf32 samples


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def marks(total, unit=1):
    """unit indices next to the boundaries, for a stream of `total` units of `unit` samples each"""
    m = {0, total - 1}
    for p in (1 << 29, 1 << 31, 1 << 32):
        for q in (p // unit - 1, p // unit, p // unit + 1):
            if 0 <= q < total:
                m.add(q)
    return sorted(m)


@pytest.fixture(scope="module")
def big(gpu, redio):
    free, _ = gpu.cuda.mem_get_info()
    if free < 120 << 30:
        pytest.skip(f"needs 120 GB of free device memory, {free >> 30} GB free")
    x = redio.synth_iq(SEED, 0, N)
    yield x
    del x
    gpu.cuda.empty_cache()


def test_synth_past_2_32(gpu, redio, oracle, big):
    for p in marks(N):
        lo = max(0, p - 100)
        hi = min(N, p + 100)
        assert same_bits(big[lo:hi].cpu().numpy(), oracle.synth_iq(SEED, lo, hi - lo)), p


@pytest.mark.parametrize("k,d,fused", [(127, 5, True), (127, 5, False), (63, 1, False), (100, 3, False), (7, 13, False)])
def test_fir_past_2_32(gpu, redio, oracle, big, k, d, fused):
    """dsputils::convolve (dsputils.rs:30-32) with decimation: the chain kernel's FIR-only form, the tiled, the chunked and the direct kernel"""
    taps = oracle.lpf_corrected(k, 0.08)
    fir = redio.Fir(taps, d, complex_input=True, fused=fused)
    nout = fir.nout(N)
    assert nout == (N - k) // d + 1
    out = fir(big)
    assert out.numel() == nout
    for p in marks(N):
        o0 = max(0, min(p // d - 700, nout - 1400))
        cnt = min(1400, nout - o0)
        xw = oracle.synth_iq(SEED, o0 * d, (cnt - 1) * d + k)
        want = oracle.fir(xw, taps, d, fused=fused)
        assert same_bits(out[o0:o0 + cnt].cpu().numpy(), want), (k, d, p)
    del out


@pytest.mark.parametrize("nfft", [1024, 64, 4096, 16384, 65536, 1000])
def test_fft_past_2_32(gpu, redio, oracle, big, nfft):
    """kissfft::fft (kissfft.rs:18-31) on consecutive blocks of the stream"""
    nb = N // nfft
    plan = redio.Fft(nfft)
    out = plan(big[: nb * nfft])
    for b in marks(nb, nfft):
        lo = max(0, min(b - 1, nb - 3))
        cnt = min(3, nb - lo)
        want = oracle.fft(oracle.synth_iq(SEED, lo * nfft, cnt * nfft), nfft)
        assert same_bits(out[lo * nfft:(lo + cnt) * nfft].cpu().numpy().reshape(cnt, nfft), want.reshape(cnt, nfft)), (nfft, b)
    del out


def test_chain_past_2_32(gpu, redio, oracle, big):
    """BASELINE.json configs[1] sixteen times over: 838 873 blocks in one launch"""
    taps = oracle.lpf_corrected(127, 0.08)
    for fused in (True, False):
        chain = redio.Chain(taps, 5, 1024, fused=fused)
        nb = chain.nblocks(N)
        assert nb == ((N - 127) // 5 + 1) // 1024
        out = chain(big)
        for b in marks(nb, 5120):
            lo = max(0, min(b - 4, nb - 9))
            cnt = min(9, nb - lo)
            want = oracle.chain_fir_fft(oracle.synth_iq(SEED, lo * 5120, cnt * 5120 + 126), taps, 5, 1024, fused=fused)
            assert same_bits(out[lo:lo + cnt].cpu().numpy(), want), (fused, b)
        del out


def test_channelizer_past_2_32(gpu, redio, oracle, big):
    """BASELINE.json configs[3]: 64 channels, 16 taps per branch, 2^26 rows"""
    M, P = 64, 16
    h = oracle.lpf_corrected(M * P, 0.45 / M)
    plan = redio.Channelizer(h, M, P)
    rows = plan.nrows(N)
    out = plan(big)
    assert tuple(out.shape) == (rows, M)
    for r in marks(rows, M):
        lo = max(0, min(r - 20, rows - 40))
        cnt = min(40, rows - lo)
        want = oracle.pfb_channelizer(oracle.synth_iq(SEED, lo * M, (cnt + P - 1) * M), h, M, P, fused=True)
        assert same_bits(out[lo:lo + cnt].cpu().numpy(), want[:cnt]), r
    del out
    # the per-destination layout of the exchange step: [group][row][M / groups]
    g = 8
    outg = plan(big, ngroups=g)
    for r in marks(rows, M):
        lo = max(0, min(r - 20, rows - 40))
        cnt = min(40, rows - lo)
        want = oracle.pfb_channelizer(oracle.synth_iq(SEED, lo * M, (cnt + P - 1) * M), h, M, P, fused=True)[:cnt]
        got = outg[:, lo:lo + cnt, :].cpu().numpy()
        assert same_bits(np.ascontiguousarray(got.transpose(1, 0, 2)).reshape(cnt, M), want), r
    del outg


@pytest.mark.parametrize("nfft,k", [(65536, 8193), (4096, 127), (32768, 127)])
def test_overlap_save_past_2_32(gpu, redio, oracle, big, nfft, k):
    """BASELINE.json configs[4]: blocks of nfft points, hop nfft - k + 1, through the chunk loop tens of thousands of times"""
    h = oracle.lpf_corrected(k, 0.08)
    plan = redio.OverlapSave(h, nfft)
    hop = nfft - k + 1
    nout = plan.nout(N)
    out = plan(big)
    assert out.numel() == nout and nout % hop == 0
    nblk = nout // hop
    for b in marks(nblk, hop):
        lo = max(0, min(b - 1, nblk - 2))
        cnt = min(2, nblk - lo)
        want = oracle.overlap_save(oracle.synth_iq(SEED, lo * hop, (cnt - 1) * hop + nfft), h, nfft)
        assert same_bits(out[lo * hop:(lo + cnt) * hop].cpu().numpy(), want[: cnt * hop]), (nfft, b)
    del out


def test_ingest_past_2_32(gpu, redio, oracle, big):
    """rtlsdr::data_to_samples (rtlsdr.rs:159-162), norm, the 512-sample block sums and discretize (bitfount.rs:36-96) on 2^33 bytes / 2^32 samples"""
    raw = gpu.view_as_real(big).view(gpu.int32).reshape(-1)  # two words per sample
    d = ((raw >> 9) & 255).to(gpu.uint8)                      # one byte per word: 2 N bytes = N I/Q pairs
    del raw

    def host_bytes(first, n):  # the same bytes from the hash, samples [first, first + n)
        return ((oracle.synth_iq(SEED, first, n).view(np.int32).reshape(-1) >> 9) & 255).astype(np.uint8)

    x = redio.bitfount.data_to_samples(d)
    assert x.numel() == N
    for p in marks(N):
        lo, hi = max(0, p - 300), min(N, p + 300)
        assert same_bits(x[lo:hi].cpu().numpy(), oracle.data_to_samples(host_bytes(lo, hi - lo))), p
    del x
    mag = redio.bitfount.ingest_mag(d)
    for p in marks(N):
        lo, hi = max(0, p - 300), min(N, p + 300)
        assert same_bits(mag[lo:hi].cpu().numpy(), oracle.norm(oracle.data_to_samples(host_bytes(lo, hi - lo)))), p
    del d
    sums = redio.bitfount.block_sums(mag, 512)
    nb = N // 512
    assert sums.numel() == nb
    for b in marks(nb, 512):
        m = oracle.norm(oracle.data_to_samples(host_bytes(b * 512, 512)))
        assert bits(sums[b:b + 1].cpu().numpy())[0] == bits(np.array([oracle.block_sum(m)], np.float32))[0], b
    del sums
    mag[N - 5] = 3.0  # the maximum sits behind element 2^32
    sl = redio.bitfount.discretize(mag)
    assert sl.numel() == N
    for p in marks(N):
        lo, hi = max(0, p - 300), min(N, p + 300)
        m = oracle.norm(oracle.data_to_samples(host_bytes(lo, hi - lo)))
        if lo <= N - 5 < hi:
            m[N - 5 - lo] = 3.0
        assert np.array_equal(sl[lo:hi].cpu().numpy(), (m > np.float32(1.5)).astype(np.uint8)), p
    del sl, mag


def test_resampler_past_2_32(gpu, redio, oracle, big):
    """BASELINE.json configs[2] with 2^32 frames in one message: 256 channels x 2^24 frames, ratio 0.02"""
    nch, n, ratio = 256, 1 << 24, 0.02
    d = gpu.view_as_real(big).reshape(-1)[: nch * n].reshape(nch, n)  # channel c = f32 words [c n, (c + 1) n) of the stream
    src = redio.Src(nch, 1)
    out, used = src.process(d, ratio)
    assert used == n
    for c in (0, 127, 128, 255):  # channel 128 starts at element 2^31
        w = oracle.synth_iq(SEED, c * n // 2, n // 2).view(np.float32).reshape(-1)
        err, want, wused = oracle.Resampler(1).process(w, ratio, int(ratio * n + 1.0))
        assert err == 0 and wused == n
        assert same_bits(out[c].cpu().numpy(), want), c
