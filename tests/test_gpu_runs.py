"""Run-length / bit-field stage (kpn::rle, dle, rld, dld, binconv) on the MI355X against the numpy /
C restatements in oracle/: exact integer work."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rle_streaming_state_and_unflushed_last_run(gpu, redio, oracle):
    rng = np.random.default_rng(3)
    x = np.repeat(rng.integers(0, 2, 4000), rng.integers(1, 40, 4000)).astype(np.uint8)
    dev, ref = redio.kpn_dev.Rle(), oracle.Rle()
    cuts = [0, 1, 2, 5000, 5001, 20000, len(x)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        v, c = dev.feed(gpu.from_numpy(x[lo:hi]).cuda())
        want = ref.feed(x[lo:hi])
        assert list(zip(v.cpu().tolist(), c.cpu().tolist())) == want, (lo, hi)
    # a constant tail emits nothing: the last run is never flushed (kpn.rs:17-29)
    v, c = dev.feed(gpu.full((1000,), int(x[-1]), dtype=gpu.uint8, device="cuda"))
    assert v.numel() == 0 and ref.feed(np.full(1000, x[-1])) == []
    v, c = dev.feed(gpu.tensor([1 - int(x[-1])], dtype=gpu.uint8, device="cuda"))
    assert list(zip(v.cpu().tolist(), c.cpu().tolist())) == ref.feed([1 - int(x[-1])])


def test_rle_large_and_multivalue(gpu, redio, oracle):
    rng = np.random.default_rng(5)
    x = np.repeat(rng.integers(0, 256, 300000), rng.integers(1, 9, 300000)).astype(np.uint8)
    v, c = redio.kpn_dev.Rle().feed(gpu.from_numpy(x).cuda())
    want = oracle.Rle().feed(x)
    assert v.numel() == len(want)
    assert np.array_equal(v.cpu().numpy(), np.array([w[0] for w in want], np.uint8))
    assert np.array_equal(c.cpu().numpy(), np.array([w[1] for w in want], np.int64))


@pytest.mark.parametrize("n", [16383, 16384, 16385, 3 * 16384 + 5, 20 * 16384])
def test_rle_changes_at_round_and_tile_seams(gpu, redio, oracle, n):
    # a workgroup walks a 16384-element tile in eight rounds of 2048 (runs.hip): value changes exactly at, just before and just after
    # every round and tile seam, long constant stretches over whole rounds, and nothing else
    x = np.zeros(n, np.uint8)
    for seam in range(2048, n, 2048):
        for d in (-1, 0, 1):
            if 0 < seam + d < n and (seam // 2048 + d) % 3 != 0:
                x[seam + d:] ^= 1
    v, c = redio.kpn_dev.Rle().feed(gpu.from_numpy(x).cuda())
    want = oracle.Rle().feed(x)
    assert list(zip(v.cpu().tolist(), c.cpu().tolist())) == want
    # one run spanning everything: no change at all
    v, c = redio.kpn_dev.Rle().feed(gpu.zeros(n, dtype=gpu.uint8, device="cuda"))
    assert v.numel() == 0


@pytest.mark.parametrize("off", [1, 3, 7])
@pytest.mark.parametrize("n", [1, 7, 8, 9, 2047, 2048, 2049, 70001])
def test_rle_views_off_the_8_byte_grid(gpu, redio, oracle, off, n):
    # the change detection reads 8 elements per load when it can: views that start anywhere, ragged ends
    rng = np.random.default_rng(off * 100 + n)
    x = np.repeat(rng.integers(0, 3, n + 8), rng.integers(1, 5, n + 8)).astype(np.uint8)[: off + n]
    d = gpu.from_numpy(x).cuda()
    v, c = redio.kpn_dev.Rle().feed(d[off:])
    want = oracle.Rle().feed(x[off:])
    assert list(zip(v.cpu().tolist(), c.cpu().tolist())) == want


def test_dle_rld_dld_round_trip(gpu, redio, oracle):
    runs = [(1, 51), (0, 512), (1, 90), (0, 1), (1, 1000)]
    vals = gpu.tensor([r[0] for r in runs], dtype=gpu.uint8, device="cuda")
    cts = gpu.tensor([r[1] for r in runs], dtype=gpu.int64, device="cuda")
    sec = redio.kpn_dev.dle(cts, 256000)                      # ratpak.rs: dle(.., 256000)
    want = oracle.dle(runs, 256000)
    assert np.array_equal(sec.cpu().numpy().view(np.uint32), np.array([w[1] for w in want], np.float32).view(np.uint32))
    assert redio.kpn_dev.rld(vals, cts).cpu().tolist() == oracle.rld(runs)
    got = redio.kpn_dev.dld(vals, sec, 256000.0, 4096).cpu().tolist()
    assert got == oracle.dld([(v, s) for (v, _), (_, s) in zip(runs, want)], 256000.0)


def test_binconv_ratpak_width_lists(gpu, redio, oracle):
    rng = np.random.default_rng(9)
    bits = rng.integers(0, 2, (500, 36)).astype(np.uint8)
    for widths in ([4, 8, 4, 12, 8], [4, 8, 2, 10, 12]):          # src/ratpak.rs:115,119
        got = redio.kpn_dev.binconv(gpu.from_numpy(bits).cuda(), widths).cpu().numpy()
        want = np.array([oracle.eat(b, widths) for b in bits], np.int64)
        assert np.array_equal(got, want)
    with pytest.raises(redio.RedioError) as e:
        redio.kpn_dev.binconv(gpu.from_numpy(bits).cuda(), [30, 10])   # slice out of bounds in the reference
    assert e.value.code == -5
    assert redio.kpn_dev.binconv(gpu.tensor([[1, 0, 1]], dtype=gpu.uint8, device="cuda"), [3]).item() == oracle.b2d([1, 0, 1]) == 5


def test_shipped_graph_discretize_to_runs(gpu, redio, oracle):
    """discretize -> rle -> dle, as src/ratpak.rs:73-87 wires them, device-resident end to end."""
    rng = np.random.default_rng(21)
    env = np.repeat(rng.integers(0, 2, 400), rng.integers(30, 200, 400)).astype(np.float32)
    buf = (0.05 * rng.random(len(env)) + env).astype(np.float32)
    bits_dev = redio.bitfount.discretize(gpu.from_numpy(buf).cuda())
    bits_ref = oracle.discretize(buf).astype(np.uint8)
    assert np.array_equal(bits_dev.cpu().numpy(), bits_ref)
    v, c = redio.kpn_dev.Rle().feed(bits_dev)
    want = oracle.Rle().feed(bits_ref)
    assert list(zip(v.cpu().tolist(), c.cpu().tolist())) == want
    sec = redio.kpn_dev.dle(c, 256000).cpu().numpy()
    assert np.array_equal(sec.view(np.uint32), np.array([w[1] for w in oracle.dle(want, 256000)], np.float32).view(np.uint32))


@pytest.mark.parametrize("n", [0, 1, 3, 4, 1023, 4096, 100003])
@pytest.mark.parametrize("cplx", [False, True])
def test_mul_vecs_sum_vecs_bit_exact(gpu, redio, oracle, n, cplx):
    # kpn::mul_vecs / sum_vecs (kpn.rs:198-203, 227-231): zip over the shorter length; aligned and unaligned views
    from libredio_amd import kpn_dev
    gen = oracle.synth_iq if cplx else oracle.synth_f32
    x, c = gen(31, 0, n + 5), gen(32, 0, n + 9)
    dx, dc = gpu.from_numpy(x).cuda(), gpu.from_numpy(c).cuda()
    for off in (0, 1):          # off = 1: 4- or 8-byte aligned only -> scalar path
        a, b = dx[off:off + n], dc[off:off + n + 2]
        got_m, got_s = kpn_dev.mul_vecs(a, b).cpu().numpy(), kpn_dev.sum_vecs(a, b).cpu().numpy()
        want_m = oracle.zip_vecs(x[off:off + n], c[off:off + n + 2], add=False)
        want_s = oracle.zip_vecs(x[off:off + n], c[off:off + n + 2], add=True)
        assert len(got_m) == n and len(got_s) == n
        assert np.array_equal(got_m.view(np.uint32), want_m.view(np.uint32))
        assert np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32))


@pytest.mark.parametrize("cplx", [False, True])
def test_mul_vecs_sum_vecs_operands_beyond_the_caches(gpu, redio, oracle, cplx):
    # 64 MB per operand and up takes the non-temporal instantiation, one 16-byte group per thread in dispatch order (elementwise.hip)
    from libredio_amd import kpn_dev
    n = (1 << 24) + 3 if not cplx else (1 << 23) + 1
    x = redio.synth_iq(41, 0, n) if cplx else redio.synth_f32(41, 0, n)
    c = redio.synth_iq(42, 0, n) if cplx else redio.synth_f32(42, 0, n)
    xn, cn = x.cpu().numpy(), c.cpu().numpy()
    for dev, add in ((kpn_dev.mul_vecs(x, c), False), (kpn_dev.sum_vecs(x, c), True)):
        want = oracle.zip_vecs(xn, cn, add=add)
        assert np.array_equal(dev.cpu().numpy().view(np.uint32), want.view(np.uint32)), (cplx, add)


@pytest.mark.gpu
@pytest.mark.parametrize("n,off", [(1, 0), (3, 1), (4, 0), (1023, 0), (1024, 3), (65537, 2), ((1 << 24) + 5, 1)])
def test_checksum_u32_is_the_exact_word_sum(gpu, redio, n, off):
    """redio_checksum_u32 (the checking sink of include/kpn_dev.hpp graphs): the order-free u64 sum of 32-bit words, accumulating, on views
    that start on and off the 16-byte grid."""
    import torch
    from libredio_amd import kpn_dev
    g = torch.Generator(device="cuda"); g.manual_seed(n)
    buf = torch.randint(-(1 << 31), 1 << 31, (n + off,), dtype=torch.int32, device="cuda", generator=g)
    x = buf[off:]
    want = int(x.cpu().numpy().view(np.uint32).astype(np.uint64).sum())
    acc = kpn_dev.checksum_u32(x)
    assert int(acc.cpu().numpy().view(np.uint64)[0]) == want
    kpn_dev.checksum_u32(x, acc)                               # accumulates
    assert int(acc.cpu().numpy().view(np.uint64)[0]) == (2 * want) % (1 << 64)
