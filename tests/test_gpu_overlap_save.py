"""Overlap-save FFT convolution (BASELINE.json configs[4]) on the MI355X: bit-exact against the
oracle composition (kissfft-order transforms), and within the stated tolerance of the direct
dsputils::convolve fold it replaces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("nfft,k", [(1024, 127), (1024, 1), (1024, 1024), (1024, 500), (2048, 127), (2048, 1), (2048, 2048), (2048, 700), (8192, 127), (8192, 1), (8192, 8192), (8192, 3000), (32768, 1000), (32768, 32768), (131072, 127), (262144, 40000), (524288, 1), (2097152, 127), (4096, 63), (4096, 1025), (4096, 4096), (4096, 1), (16384, 127), (16384, 8193), (16384, 16384), (65536, 8193), (65536, 127), (256, 256), (64, 1)])
def test_overlap_save_bit_exact_and_close_to_direct(gpu, redio, oracle, nfft, k):
    taps = oracle.lpf_corrected(k, 0.02) if k > 1 else np.array([0.75], np.float32)
    hop = nfft - k + 1
    for n in (nfft, nfft + hop - 1, nfft + 2 * hop + 17):
        x = oracle.synth_iq(0x5EED0005, 0, n)
        plan = redio.OverlapSave(taps, nfft)
        want = oracle.overlap_save(x, taps, nfft)
        got = plan(gpu.from_numpy(x).cuda()).cpu().numpy()
        assert len(got) == len(want) == ((n - nfft) // hop + 1) * hop
        assert np.array_equal(bits(got), bits(want)), (nfft, k, n)
        # vs the direct fold (A1 semantics): f32 FFT round trip error, stated tolerance 2e-6 * sum|h| * sqrt(log2 N)
        direct = oracle.fir(x, taps, 1, False)[: len(got)]
        assert np.abs(got - direct).max() <= 2e-6 * np.abs(taps).sum() * np.sqrt(np.log2(nfft)) + 1e-7


def test_overlap_save_short_input_and_bad_args(gpu, redio, oracle):
    plan = redio.OverlapSave(oracle.lpf_corrected(127, 0.1), 1024)
    assert plan.nout(1023) == 0 and plan.nout(1024) == 898
    assert plan(gpu.zeros(100, dtype=gpu.complex64, device="cuda")).numel() == 0
    with pytest.raises(redio.RedioError):
        redio.OverlapSave(oracle.lpf_corrected(127, 0.1), 64)   # more taps than the block


def test_overlap_save_many_blocks_chunking(gpu, redio, oracle):
    # more blocks than one work-buffer chunk (64 MiB / (1024*8 B) = 8192 blocks)
    taps = oracle.lpf_corrected(127, 0.08)
    n = 1024 + 898 * 9000
    x = redio.synth_iq(3, 0, n)
    y = redio.OverlapSave(taps, 1024)(x)
    for b in (0, 8191, 8192, 9000):
        xw = oracle.synth_iq(3, 898 * b, 1024)
        assert np.array_equal(bits(y[898 * b: 898 * (b + 1)].cpu().numpy()), bits(oracle.overlap_save(xw, taps, 1024)))


@pytest.mark.parametrize("k", [8193, 127])
def test_overlap_save_65536_across_the_chunk_loop(gpu, redio, oracle, k):
    """BASELINE.json configs[4] shape beyond one work-buffer chunk: 65536-point blocks, two full 128-block chunks
    and a ragged third (261 blocks).  Every block is independent given its own 65536-sample window
    (dsputils.rs:30-32 semantics per window), so the blocks either side of every chunk seam, the first and the
    last are compared bit for bit with the oracle run on just that window."""
    nfft = 65536
    taps = oracle.lpf_corrected(k, 0.02)
    hop = nfft - k + 1
    nblk = 261
    n = nfft + hop * (nblk - 1) + 1234          # ragged tail that fills no block
    x = redio.synth_iq(0x5EED0005, 0, n)
    plan = redio.OverlapSave(taps, nfft)
    assert plan.nout(n) == nblk * hop
    y = plan(x)
    assert y.numel() == nblk * hop
    for b in (0, 1, 126, 127, 128, 129, 254, 255, 256, 257, nblk - 1):
        xw = oracle.synth_iq(0x5EED0005, hop * b, nfft)
        want = oracle.overlap_save(xw, taps, nfft)
        assert np.array_equal(bits(y[hop * b: hop * (b + 1)].cpu().numpy()), bits(want)), (k, b)
    # idempotence of the plan (work buffers reused by the second call) and a checksum over all blocks
    s1 = gpu.view_as_real(y).view(gpu.int32).sum(dtype=gpu.int64).item()
    y2 = plan(x)
    assert gpu.equal(y, y2) and s1 == gpu.view_as_real(y2).view(gpu.int32).sum(dtype=gpu.int64).item()


@pytest.mark.parametrize("nfft,k,nblk", [(32768, 127, 256 + 5), (32768, 4097, 257), (131072, 127, 64 + 3), (131072, 16385, 65)])
def test_overlap_save_big_blocks_across_the_chunk_loop(gpu, redio, oracle, nfft, k, nblk):
    """32768-point blocks (round 4: three passes -- G128 forward; forward in-place pass x conj H x inverse G128 on one tile; inverse in-place
    pass) and 131072-point blocks (four passes) over more blocks than one 64 MiB chunk of the work buffers: first block, the blocks on both
    sides of the chunk seam and the last block against the oracle on their own windows; even and odd hops."""
    taps = oracle.lpf_corrected(k, 0.06)
    hop = nfft - k + 1
    n = nfft + (nblk - 1) * hop + 7
    x = redio.synth_iq(0x5EED0005, 0, n)
    plan = redio.OverlapSave(taps, nfft)
    assert plan.nout(n) == nblk * hop
    out = plan(x)
    chunk = (64 << 20) // (nfft * 8)
    for b in sorted(b for b in {0, 1, chunk - 1, chunk, chunk + 1, nblk - 1} if b < nblk):
        want = oracle.overlap_save(oracle.synth_iq(0x5EED0005, b * hop, nfft), taps, nfft)
        assert np.array_equal(bits(out[b * hop:(b + 1) * hop].cpu().numpy()), bits(want)), (nfft, k, b)


@pytest.mark.parametrize("nblk", [1, 2, 127, 128, 129, 255, 256])
def test_overlap_save_65536_step_launch_edges(gpu, redio, oracle, nblk):
    """The chunk-step launches of the 65536-point scheme (fft_kernels.hip launch_ovsave64k: middle pass of chunk k, last pass of
    chunk k - 1 and gather pass of chunk k + 1 in one launch, work buffers doubled) at the block counts where a program drops out of
    a step: one block, one full 128-block chunk, a one-block second chunk, two full chunks.  First, last and seam blocks against the
    oracle on their own windows; the rest through a second plan fed the same stream in two calls."""
    nfft, k = 65536, 8193
    taps = oracle.lpf_corrected(k, 0.03)
    hop = nfft - k + 1
    n = nfft + hop * (nblk - 1)
    x = redio.synth_iq(0x5EED0035, 0, n)
    plan = redio.OverlapSave(taps, nfft)
    y = plan(x)
    assert y.numel() == nblk * hop
    for b in sorted({0, nblk // 2, max(nblk - 2, 0), nblk - 1, min(127, nblk - 1), min(128, nblk - 1)}):
        want = oracle.overlap_save(oracle.synth_iq(0x5EED0035, hop * b, nfft), taps, nfft)
        assert np.array_equal(bits(y[hop * b: hop * (b + 1)].cpu().numpy()), bits(want)), (nblk, b)
    if nblk > 2:  # the same blocks as two shorter calls (other chunk boundaries) must give the same bits
        cut = nblk // 3 + 1
        y2 = gpu.cat([plan(x[: nfft + hop * (cut - 1)]), plan(x[hop * cut:])])
        assert gpu.equal(y, y2)


def test_overlap_save_2_19_point_blocks_mix_the_two_plans(gpu, redio, oracle):
    """2^19-point blocks: the forward transform takes the two-pass plan (G512 + five-stage pass), the inverse -- spectrum product on the way in,
    masked store on the way out -- the three-pass plan of the same size; both read their tables from one plan allocation."""
    nfft, k = 1 << 19, 127
    h = oracle.lpf_corrected(k, 0.08)
    hop = nfft - k + 1
    x = oracle.synth_iq(19, 0, nfft + 2 * hop + 5)
    got = redio.OverlapSave(h, nfft)(gpu.from_numpy(x).cuda()).cpu().numpy()
    want = oracle.overlap_save(x, h, nfft)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
