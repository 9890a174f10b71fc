"""Carried history (redio_*_stream_*): a stream fed in ANY pieces gives exactly the bits of one stateless call on the
whole stream.  SURVEY.md 8d defines BASELINE.json configs[1] on a stream with the history carried, SURVEY.md 7.4.5 asks
for results independent of message tiling; the stateless plans keep dsputils::convolve's per-message semantics
(dsputils.rs:30-32), which drops ntaps-1 outputs at every seam."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def cuts(rng, n, style):
    if style == "one":
        return [n]
    if style == "tiny":       # many messages shorter than a window, some empty
        out, left = [], n
        while left:
            k = int(min(left, rng.integers(0, 40)))
            out.append(k); left -= k
        return out
    if style == "small":      # a few hundred samples per message: a window fills over several calls, the tail stays in place inside its staging
        out, left = [], n     # buffer (round 6) and is moved to the other buffer's front only when the next message would not fit behind it
        while left:
            k = int(min(left, rng.integers(1, 900)))
            out.append(k); left -= k
        return out
    if style == "odd":        # odd lengths: the body starts on an odd sample (8-byte aligned only)
        out, left = [], n
        while left:
            k = int(min(left, 2 * int(rng.integers(500, 40000)) + 1))
            out.append(k); left -= k
        return out
    out, left = [], n         # "mixed"
    while left:
        k = int(min(left, rng.choice([1, 7, 126, 127, 128, 1000, 5119, 5120, 5121, 5246, 20000, 65536, 100001])))
        out.append(k); left -= k
    return out


def feed(stream, x, pieces, gpu):
    outs, pos = [], 0
    for k in pieces:
        want = stream.nout(k)
        y = stream(x[pos:pos + k])
        assert y.numel() == want
        outs.append(y.clone())
        pos += k
    return gpu.cat(outs) if outs else None


@pytest.mark.parametrize("style", ["one", "tiny", "small", "odd", "mixed"])
@pytest.mark.parametrize("k,d,cplx,fused", [(127, 5, True, True), (127, 5, True, False), (63, 1, False, False), (100, 3, True, False), (1, 1, False, False), (17, 4, False, True), (5, 9, True, False)])
def test_fir_stream_any_segmentation(gpu, redio, oracle, k, d, cplx, fused, style):
    rng = np.random.default_rng(k * 131 + d)
    n = 30000 if style == "tiny" else (120000 if style == "small" else 300000)
    taps = oracle.synth_f32(5, 0, k)
    xh = (oracle.synth_iq if cplx else oracle.synth_f32)(77, 0, n)
    x = gpu.from_numpy(xh).cuda()
    plan = redio.Fir(taps, d, complex_input=cplx, fused=fused)
    want = oracle.fir(xh, taps, d, fused)
    st = redio.Stream(plan)
    got = feed(st, x, cuts(rng, n, style), gpu).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (k, d, style)
    assert st.pending == max(n - len(want) * d, 0) and st.pending < k
    # the handle is reusable after reset, and the stateless plan still gives the one-shot result
    st.reset()
    assert np.array_equal(bits(feed(st, x, [n], gpu).cpu().numpy()), bits(want))
    assert np.array_equal(bits(plan(x).cpu().numpy()), bits(want))


@pytest.mark.parametrize("style", ["one", "tiny", "small", "odd", "mixed"])
@pytest.mark.parametrize("k,d,nfft,fused", [(127, 5, 1024, True), (127, 5, 1024, False), (63, 1, 1024, True), (31, 4, 256, False)])
def test_chain_stream_any_segmentation(gpu, redio, oracle, k, d, nfft, fused, style):
    rng = np.random.default_rng(k + nfft)
    nb = 3 if style == "tiny" else 41
    n = nb * nfft * d + (k - d) + 777          # 777 trailing samples fill no block: they stay pending
    taps = oracle.lpf_corrected(k, 0.4 / max(d, 2))
    xh = oracle.synth_iq(0x5EED0002, 0, n)
    x = gpu.from_numpy(xh).cuda()
    want = oracle.chain_fir_fft(xh, taps, d, nfft, fused=fused)
    st = redio.Stream(redio.Chain(taps, d, nfft, fused=fused))
    got = feed(st, x, cuts(rng, n, style), gpu).cpu().numpy().reshape(-1, nfft)
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (k, d, nfft, style)
    assert st.pending == n - nb * nfft * d


@pytest.mark.parametrize("style", ["one", "small", "odd", "mixed"])
@pytest.mark.parametrize("nchan,p", [(64, 16), (64, 4), (32, 5)])
def test_channelizer_stream_any_segmentation(gpu, redio, oracle, nchan, p, style):
    rng = np.random.default_rng(nchan + p)
    n = nchan * 3000 + 13
    proto = oracle.synth_f32(9, 0, nchan * p)
    xh = oracle.synth_iq(0x5EED0004, 0, n)
    x = gpu.from_numpy(xh).cuda()
    want = oracle.pfb_channelizer(xh, proto, nchan, p, fused=False)
    st = redio.Stream(redio.Channelizer(proto, nchan, p, fused=False))
    got = feed(st, x, cuts(rng, n, style), gpu).cpu().numpy().reshape(-1, nchan)
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (nchan, p, style)


@pytest.mark.parametrize("style", ["one", "small", "odd", "mixed"])
@pytest.mark.parametrize("nfft,k", [(1024, 127), (4096, 1025), (65536, 8193), (1000, 100)])
def test_overlap_save_stream_any_segmentation(gpu, redio, oracle, nfft, k, style):
    rng = np.random.default_rng(nfft + k)
    hop = nfft - k + 1
    n = nfft + hop * (4 if nfft > 4096 else 37) + 55
    taps = oracle.lpf_corrected(k, 0.02)
    xh = oracle.synth_iq(0x5EED0005, 0, n)
    x = gpu.from_numpy(xh).cuda()
    want = oracle.overlap_save(xh, taps, nfft)
    st = redio.Stream(redio.OverlapSave(taps, nfft))
    got = feed(st, x, cuts(rng, n, style), gpu).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (nfft, k, style)


def test_chain_stream_large_pieces_at_full_rate_path(gpu, redio, oracle):
    """2^24-sample pieces of a 2^26-sample stream through the fused kernel: equal to the one-shot call, bit for bit."""
    taps = oracle.lpf_corrected(127, 0.08)
    n = 1 << 26
    x = redio.synth_iq(0x5EED0002, 0, n)
    chain = redio.Chain(taps, 5, 1024, fused=True)
    one = chain(x).reshape(-1)
    st = redio.Stream(chain)
    got = gpu.cat([st(x[i:i + (1 << 24)]).clone() for i in range(0, n, 1 << 24)])
    assert got.numel() == one.numel() and gpu.equal(got, one)


def test_stream_handle_edge_cases(gpu, redio, oracle):
    """Empty messages, reset in mid-stream, two streams on one plan, messages shorter than anything the kernels take."""
    import ctypes as C
    taps = oracle.lpf_corrected(127, 0.08)
    x = oracle.synth_iq(3, 0, 40000)
    d = gpu.from_numpy(x).cuda()
    plan = redio.Fir(taps, 5, complex_input=True, fused=True)
    a, b = redio.Stream(plan), redio.Stream(plan)          # independent histories on one plan
    assert a(d[:0]).numel() == 0 and a.pending == 0 and a.nout(0) == 0
    ya = gpu.cat([a(d[:100]), a(d[100:100]), a(d[100:20000]), a(d[20000:])])
    yb = gpu.cat([b(d[:33333]), b(d[33333:])])
    want = oracle.fir(x, taps, 5, True)
    assert np.array_equal(bits(ya.cpu().numpy()), bits(want)) and np.array_equal(bits(yb.cpu().numpy()), bits(want))
    # reset drops the carried tail: the next message starts a new stream (decimation phase 0 again)
    a.reset()
    assert a.pending == 0
    y2 = a(d[7:5007])
    assert np.array_equal(bits(y2.cpu().numpy()), bits(oracle.fir(x[7:5007], taps, 5, True)))
    # C ABI argument checks
    L = redio.lib()
    h = C.c_void_p()
    assert L.redio_fir_stream_create(C.byref(h), None) == -1
    assert L.redio_fir_stream_enqueue(None, None, 0, None, None, None) == -1
    assert L.redio_fir_stream_nout(None, 10) == 0 and L.redio_fir_stream_destroy(None) == 0
    got = C.c_size_t(7)
    assert L.redio_fir_stream_enqueue(a._h, None, 5, None, C.byref(got), None) == -1 and got.value == 0   # NULL data with n > 0


def test_exchange_argument_checks(gpu, redio):
    import ctypes as C
    L = redio.lib()
    assert L.redio_pfb_exchange(None, None, None, None, 8, None) == -1
    assert L.redio_comm_init_rank(C.byref(C.c_void_p()), 2, 5, C.create_string_buffer(128)) == -1        # rank outside the communicator
    assert L.redio_comm_rank(None) == -1 and L.redio_comm_size(None) == 0 and L.redio_comm_destroy(None) == 0
    comm = redio.Comm.single()
    rows = (C.c_size_t * 1)(4)
    assert L.redio_pfb_exchange(comm._h, None, None, rows, 8, None) == -1                                # rows announced, no buffers
    assert L.redio_pfb_exchange(comm._h, None, None, rows, 0, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["chain", "chain_small", "pfb", "pfb_generic"])
def test_u8_byte_streams(gpu, redio, oracle, kind):
    """redio_{chain,pfb}_stream_create_u8: the receiver's byte stream (rtlsdr.rs:127-162) cut into messages of awkward lengths, the
    history carried as bytes on the device.  The concatenated outputs are those of the oracle's data_to_samples + plan on the whole
    stream, whatever the cuts (odd sample counts put later windows on 2-byte boundaries: those take the converting path)."""
    rng = np.random.default_rng(len(kind))
    if kind.startswith("chain"):
        k, d, nfft = (127, 5, 1024) if kind == "chain" else (31, 2, 64)
        taps = oracle.synth_f32(3, 0, k)
        plan = redio.Chain(taps, d, nfft, fused=True)
        n = 9 * nfft * d + 777
        one = lambda v: oracle.chain_fir_fft(v, taps, d, nfft, fused=True).reshape(-1)
    else:
        M, P = (64, 16) if kind == "pfb" else (100, 5)
        h = oracle.synth_f32(4, 0, M * P)
        plan = redio.Channelizer(h, M, P, fused=True)
        n = M * 700 + 13
        one = lambda v: np.asarray(oracle.pfb_channelizer(v, h, M, P, True)).reshape(-1)
    raw = rng.integers(0, 256, 2 * n, dtype=np.uint8)
    want = one(oracle.data_to_samples(raw))
    draw = gpu.from_numpy(raw).cuda()
    for cuts in ([0, n], [0, 1, 2, 1001, 1002, 40000 % n, n], sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, 9)]))):
        cuts = sorted(set(cuts))
        st = redio.Stream(plan, u8=True)
        outs = [st(draw[2 * lo: 2 * hi]).clone() for lo, hi in zip(cuts[:-1], cuts[1:])]
        got = gpu.cat(outs).cpu().numpy()
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (kind, cuts)


def test_stream_handle_refuses_concurrent_entry_and_keeps_its_state_on_a_failed_call(gpu, redio, oracle):
    """One stream is fed from one thread (redio.h).  Two block threads that share a handle by mistake (the kpn twin spawns one
    thread per block, src/ratpak.rs:60-185) must get REDIO_ERR_ARG from the call that finds the other inside, never a corrupted
    count: every call that returned 0 consumed its samples exactly once, in some order.  And a call that fails before its
    kernels are launched (no output buffer although outputs are due) leaves the stream where it was."""
    import ctypes as C
    import threading
    import torch
    import libredio_amd as R
    L = R.lib()
    k = 33
    taps = oracle.synth_f32(5, 0, k)
    plan = redio.Fir(taps, 1, complex_input=False, fused=False)
    st = redio.Stream(plan)
    x = gpu.from_numpy(oracle.synth_f32(11, 0, 4096)).cuda()
    y = [torch.empty(4096, dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    stream = R.current_stream()
    ok_samples, ok_out, refused = [0, 0], [0, 0], [0, 0]
    go = threading.Barrier(2)

    def work(t):
        go.wait()
        for _ in range(3000):
            got = C.c_size_t(0)
            rc = L.redio_fir_stream_enqueue(st._h, C.c_void_p(x.data_ptr()), 64, C.c_void_p(y[t].data_ptr()), C.byref(got), stream)
            if rc == 0:
                ok_samples[t] += 64; ok_out[t] += got.value
            else:
                assert rc == -1, rc
                refused[t] += 1
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    total = sum(ok_samples)
    assert total > 0 and sum(ok_out) == total - (k - 1), (ok_samples, ok_out, refused)
    assert st.pending == k - 1
    # a failed call changes nothing: outputs are due, no output buffer
    before = st.pending
    got = C.c_size_t(0)
    assert L.redio_fir_stream_enqueue(st._h, C.c_void_p(x.data_ptr()), 100, None, C.byref(got), stream) == -1
    assert st.pending == before and st.nout(100) == 100
    out = st(x[:100])
    assert out.numel() == 100
