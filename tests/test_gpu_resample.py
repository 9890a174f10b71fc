"""samplerate::resample on the MI355X against the oracle (oracle/oracle_src.c): bit-exact, because
the kernel evaluates each output in the library's own accumulation order (double) and the host state
machine runs the same double recurrence.  "Parity unpinned" w.r.t. the real libsamplerate: its
coefficient tables cannot be reproduced here."""
import queue
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def tone(n, f=0.01):
    return np.sin(2 * np.pi * f * np.arange(n)).astype(np.float32)


def test_table_matches_oracle(gpu, redio, oracle):
    import ctypes as C
    for conv in (0, 1, 2):
        tab, half, inc = oracle.src_table(conv)
        out = np.empty(half + 2, np.float32); h = C.c_int(); i = C.c_int()
        assert redio.lib().redio_src_table(conv, out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(h), C.byref(i)) == 0
        assert (h.value, i.value) == (half, inc)
        assert np.array_equal(bits(out), bits(tab))


@pytest.mark.parametrize("ratio", [0.02, 0.5, 2.0, 48000 / 44100, 1.0, 1 / 256, 3.7])
@pytest.mark.parametrize("seg", [[6000], [1000, 2500, 1, 2499], [37] * 40])
def test_block_messages_bit_exact(gpu, redio, oracle, ratio, seg):
    from libredio_amd import samplerate
    n = sum(seg) if ratio >= 0.02 else 300000
    if ratio < 0.02:
        seg = [100000, 150000, 50000]
    x = oracle.synth_f32(11, 0, n)
    st, ref = samplerate.State(1, 1), oracle.Resampler(1)
    off = 0
    for m in seg:
        got = st.block(x[off:off + m], ratio)
        want = ref.block(x[off:off + m], ratio)
        assert len(got) == len(want)              # the count law per message
        assert np.array_equal(bits(got), bits(want)), (ratio, m)
        off += m
    st.close()


def test_commented_out_smoke_program(gpu, redio, oracle):
    # samplerate.rs:89-96: sin(x/1000), x in [0,1000), ratio 2.0; prints the output length
    from libredio_amd import samplerate
    v = np.sin(np.arange(1000, dtype=np.float32) / np.float32(1000.0)).astype(np.float32)
    got = samplerate.State().block(v, 2.0)
    want = oracle.Resampler().block(v, 2.0)
    assert len(got) == len(want) and np.array_equal(bits(got), bits(want))


def test_resample_block_function(gpu, redio, oracle):
    from libredio_amd import samplerate
    din, dout = queue.Queue(), queue.Queue()
    t = threading.Thread(target=samplerate.resample, args=(din, dout, 0.5))
    t.start()
    msgs = [oracle.synth_f32(i, 0, 3000) for i in range(3)]
    for m in msgs:
        din.put(m)
    din.put(None)
    t.join()
    ref = oracle.Resampler(1)
    for m in msgs:
        assert np.array_equal(bits(dout.get()), bits(ref.block(m, 0.5)))


def test_src_process_flags_and_counts(gpu, redio, oracle):
    from libredio_amd import samplerate
    x = tone(50000)
    st, ref = samplerate.State(1, 1), oracle.Resampler(1)
    # output capacity smaller than what the input could produce: input_frames_used < input_frames
    e1, o1, u1 = st.process(x, 0.5, 1000, 0)
    e2, o2, u2 = ref.process(x, 0.5, 1000, False)
    assert (e1, u1) == (e2, u2) and u1 < len(x) and np.array_equal(bits(o1), bits(o2))
    # end_of_input flushes the tail
    e1, o1, u1 = st.process(x[u1:u1 + 3000], 0.5, 5000, 1)
    e2, o2, u2 = ref.process(x[u2:u2 + 3000], 0.5, 5000, True)
    assert (e1, u1) == (e2, u2) and np.array_equal(bits(o1), bits(o2))
    st.close()


def test_varying_ratio_interpolates(gpu, redio, oracle):
    from libredio_amd import samplerate
    x = tone(20000)
    st, ref = samplerate.State(2, 1), oracle.Resampler(2)
    for r in (1.0, 1.5, 0.7):
        e1, o1, u1 = st.process(x, r, int(r * len(x)) + 10, 0)
        e2, o2, u2 = ref.process(x, r, int(r * len(x)) + 10, False)
        assert (e1, u1, len(o1)) == (e2, u2, len(o2)) and np.array_equal(bits(o1), bits(o2))


def test_errors_are_the_library_codes(gpu, redio):
    from libredio_amd import samplerate
    S = redio.samplerate_lib()
    with pytest.raises(samplerate.SrcError) as e:
        samplerate.State(5, 1)          # there are five converters (samplerate.rs:26-30)
    assert e.value.code == 10
    with pytest.raises(samplerate.SrcError) as e:
        samplerate.State(1, 0)          # channel count must be >= 1
    assert e.value.code == 11
    st = samplerate.State()
    err, _, _ = st.process(tone(100), 1000.0, 10)
    assert err == 6 and b"ratio" in S.src_strerror(err).lower()
    assert S.src_is_valid_ratio(0.02) == 1 and S.src_is_valid_ratio(300.0) == 0
    assert S.src_get_name(1) and S.src_get_description(1) and S.src_get_version()
    assert S.src_get_name(7) is None


def test_batched_channels_match_independent_states(gpu, redio, oracle):
    nch, n, ratio = 8, 60000, 0.02
    x = np.stack([oracle.synth_f32(100 + c, 0, n) for c in range(nch)])
    plan = redio.Src(nch, 1)
    d = gpu.from_numpy(x).cuda()
    got = []
    for lo, hi in ((0, 25000), (25000, 25001), (25001, 60000)):
        out, used = plan.process(d[:, lo:hi].contiguous(), ratio)
        assert used == hi - lo
        got.append(out.cpu().numpy())
    got = np.concatenate(got, axis=1)
    for c in range(nch):
        ref = oracle.Resampler(1)
        want = np.concatenate([ref.block(x[c, lo:hi], ratio) for lo, hi in ((0, 25000), (25000, 25001), (25001, 60000))])
        assert np.array_equal(bits(got[c]), bits(want)), c


def test_tone_in_tone_out_and_dc_gain(gpu, redio, oracle):
    from libredio_amd import samplerate
    fs_in, f = 2.4e6, 5e3
    x = np.sin(2 * np.pi * f / fs_in * np.arange(400000)).astype(np.float32)
    y = samplerate.State().block(x, 0.02)
    k = np.arange(len(y))
    ref = np.sin(2 * np.pi * f / fs_in * 50 * k)
    assert np.abs(y[200:] - ref[200:]).max() < 2e-5     # zero-phase, unity gain in the pass band
    dc = samplerate.State().block(np.ones(400000, np.float32), 0.02)
    assert np.abs(dc[200:] - 1.0).max() < 2e-5


SEGS = ((0, 25000), (25000, 25001), (25001, 60000), (60000, 60000), (60000, 200000))


@pytest.mark.parametrize("nch", [3, 70])      # 70: the lane-per-channel kernel (two channel groups, the second ragged)
@pytest.mark.parametrize("ratio,conv", [(0.02, 1), (0.5, 1), (1.0, 1), (1 / 256, 2), (0.1, 0), (0.25, 2)])
def test_single_launch_uniform_path_is_the_epoch_schedule(gpu, redio, oracle, ratio, conv, nch):
    # the one-launch form (default) against the literal one-launch-per-refill schedule and the oracle:
    # same counts, same bits, same carried state across messages of awkward lengths
    n = 200000 if nch == 3 else 60000
    x = np.stack([oracle.synth_f32(300 + c, 0, n) for c in range(nch)])
    d = gpu.from_numpy(x).cuda()
    one, lit = redio.Src(nch, conv), redio.Src(nch, conv, mode=redio.Src.EPOCHS)
    refs = {c: oracle.Resampler(conv) for c in range(nch)}
    for lo, hi in SEGS:
        lo, hi = min(lo, n), min(hi, n)
        a, ua = one.process(d[:, lo:hi].contiguous(), ratio)
        b, ub = lit.process(d[:, lo:hi].contiguous(), ratio)
        assert ua == ub and a.shape == b.shape
        a = a.cpu().numpy()
        assert np.array_equal(bits(a), bits(b.cpu().numpy())), (lo, hi)
        for c in (range(nch) if nch <= 8 else (0, 1, 31, 32, 63, 64, 69)):
            err, want, wused = refs[c].process(x[c, lo:hi], ratio, int(ratio * (hi - lo) + 1.0))
            assert err == 0 and wused == ua       # the library may leave input unread when the output side fills first
            assert np.array_equal(bits(a[c]), bits(want)), (c, lo, hi)


def test_single_launch_then_flush_and_ratio_change(gpu, redio, oracle):
    # a uniform message, then a varying-ratio one, then end_of_input: the image rebuilt by the
    # one-launch form must be exactly what the per-refill schedule would have left behind
    n = 90000
    x = oracle.synth_f32(77, 0, n)
    d = gpu.from_numpy(x[None, :]).cuda()
    plan, ref = redio.Src(1, 1), oracle.Resampler(1)
    steps = [(0, 40000, 0.05, 0), (40000, 70000, 0.07, 0), (70000, 90000, 0.07, 1)]
    for lo, hi, r, eoi in steps:
        cap = int(r * (hi - lo) + 1.0) + 400
        got, used = plan.process(d[:, lo:hi].contiguous(), r, output_frames=cap, end_of_input=bool(eoi))
        err, want, wused = ref.process(x[lo:hi], r, cap, bool(eoi))
        assert err == 0 and used == wused and got.shape[1] == len(want)
        assert np.array_equal(bits(got.cpu().numpy()[0]), bits(want)), (lo, hi, r)


def test_fast_mode_error_bound_and_state(gpu, redio, oracle):
    # FAST: f32 polyphase taps and accumulation for uniform-phase calls.  Tolerance: each of the
    # K = cl+cr+2 products carries one f32 tap rounding and the f32 sum at most K roundings:
    # |err| <= (K + 1) * 2^-24 * sum|h| * max|x| (loose first-order bound), checked against EXACT.
    nch, n, ratio = 4, 400000, 0.02
    x = np.stack([oracle.synth_f32(500 + c, 0, n) for c in range(nch)])
    d = gpu.from_numpy(x).cuda()
    exact, fast = redio.Src(nch, 1), redio.Src(nch, 1, mode=redio.Src.FAST)
    tab, half, inc = oracle.src_table(1)
    pos = np.arange(0.0, half, inc * ratio)                       # filter positions one wing visits
    K = 2 * len(pos)
    sum_h = 2 * ratio * np.abs(np.interp(pos, np.arange(half + 2), tab.astype(np.float64))).sum()
    bound = (K + 1) * 2.0 ** -24 * max(sum_h, 1.0) * np.abs(x).max()
    for lo, hi in ((0, 150000), (150000, 150003), (150003, 400000)):
        a, ua = exact.process(d[:, lo:hi].contiguous(), ratio)
        b, ub = fast.process(d[:, lo:hi].contiguous(), ratio)
        assert ua == ub and a.shape == b.shape
        if a.numel() == 0:
            continue
        err = (a - b).abs().max().item()
        assert err <= bound, (err, bound)
        assert err < 2e-5
    # the state FAST leaves behind is input samples only: switching back to EXACT continues bit-exactly
    fast.set_mode(redio.Src.EXACT)
    y = np.stack([oracle.synth_f32(900 + c, 0, 50000) for c in range(nch)])
    dy = gpu.from_numpy(y).cuda()
    a, _ = exact.process(dy, ratio)
    b, _ = fast.process(dy, ratio)
    assert np.array_equal(bits(a.cpu().numpy()), bits(b.cpu().numpy()))


@pytest.mark.parametrize("S", [2, 3, 4, 5, 7, 8, 10, 16, 25, 26, 31, 32, 33, 40, 41, 47, 48, 49, 50, 53, 64])
@pytest.mark.parametrize("conv,nch", [(1, 1), (1, 3), (2, 2)])
def test_fast_mode_phase_split_kernel_over_decimations(gpu, redio, oracle, S, conv, nch):
    """REDIO_SRC_FAST at ratio 1/S through the phase-split polyphase kernel (src_window_fastp_kernel: persistent tiles of 512
    outputs, eight phase groups, packed f32 multiply-adds) and, for the shapes it does not serve, the tap-range kernel:
    same frame counts as EXACT (samplerate.rs:59-87 drives both), values inside the f32 bound, across messages whose
    seams fall inside a tile, for one and several channels and two converters."""
    ratio = 1.0 / S
    n = 300 * S * 7 + 4001
    x = np.stack([oracle.synth_f32(700 + 13 * S + c, 0, n) for c in range(nch)])
    d = gpu.from_numpy(x).cuda()
    exact, fast = redio.Src(nch, conv), redio.Src(nch, conv, mode=redio.Src.FAST)
    tab, half, inc = oracle.src_table(conv)
    pos = np.arange(0.0, half, inc * ratio)
    K = 2 * len(pos)
    sum_h = 2 * ratio * np.abs(np.interp(pos, np.arange(half + 2), tab.astype(np.float64))).sum()
    bound = (K + 1) * 2.0 ** -24 * max(sum_h, 1.0) * np.abs(x).max()
    cuts = [0, 600 * S + 7, 600 * S + 8, n // 2 + 3, n]
    total = 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        a, ua = exact.process(d[:, lo:hi].contiguous(), ratio)
        b, ub = fast.process(d[:, lo:hi].contiguous(), ratio)
        assert ua == ub and a.shape == b.shape, (S, lo, hi)
        total += a.shape[1]
        if a.numel():
            assert (a - b).abs().max().item() <= bound, (S, lo, hi)
    assert total > 600


def _fast_bound(oracle, ratio, xmax, conv=1):
    tab, half, inc = oracle.src_table(conv)
    pos = np.arange(0.0, half, inc * ratio)
    K = 2 * len(pos)
    sum_h = 2 * ratio * np.abs(np.interp(pos, np.arange(half + 2), tab.astype(np.float64))).sum()
    return (K + 1) * 2.0 ** -24 * max(sum_h, 1.0) * xmax


def _oracle_channels(oracle, x, chans, ratio, cuts):
    """{channel: [output per message]} from one oracle converter state per channel fed the same messages; channels run on a thread pool
    (the C oracle releases the GIL)."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    def one(c):
        ref, outs = oracle.Resampler(1), []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            err, want, used = ref.process(x[c, lo:hi], ratio, int(ratio * (hi - lo) + 1.0))
            assert err == 0 and used == hi - lo
            outs.append(want)
        return c, outs

    with ThreadPoolExecutor(max_workers=min(32, len(os.sched_getaffinity(0)))) as ex:
        return dict(ex.map(one, chans))


def test_c3_all_256_channels_bit_exact(gpu, redio, oracle):
    """BASELINE.json configs[2] to the standard of test_chain_full_size_properties: ALL 256 channels (samplerate.rs:61: one mono state each)
    at 2^18 frames in two messages of unequal length, ratio 0.02 -- EXACT bit for bit against one oracle converter per channel, FAST
    inside its stated bound on every channel."""
    nch, n, ratio = 256, 1 << 18, 0.02
    x = np.stack([oracle.synth_f32(0x5EED0003 + c, 0, n) for c in range(nch)])
    d = gpu.from_numpy(x).cuda()
    exact, fast = redio.Src(nch, 1), redio.Src(nch, 1, mode=redio.Src.FAST)
    cuts = [0, 150001, n]
    want = _oracle_channels(oracle, x, range(nch), ratio, cuts)
    bound = _fast_bound(oracle, ratio, np.abs(x).max())
    for m, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
        a, ua = exact.process(d[:, lo:hi].contiguous(), ratio)
        b, ub = fast.process(d[:, lo:hi].contiguous(), ratio)
        assert ua == ub == hi - lo and a.shape == b.shape and a.shape[0] == nch
        an = a.cpu().numpy()
        bad = [c for c in range(nch) if an[c].shape != want[c][m].shape or not np.array_equal(bits(an[c]), bits(want[c][m]))]
        assert not bad, ("channels that differ from the oracle in message", m, bad[:16])
        assert (a - b).abs().max().item() <= bound


def test_c3_bench_size_channels_from_every_group(gpu, redio, oracle):
    """configs[2] at the size bench.py times (256 channels x 2^22 frames, one message): twelve channels spread over the four 64-channel
    groups (first / last of each, and some inside), EXACT bit for bit against the oracle, FAST inside its bound on ALL channels
    (against EXACT)."""
    nch, n, ratio = 256, 1 << 22, 0.02
    check = (0, 37, 63, 64, 101, 127, 128, 170, 191, 192, 230, 255)
    d = gpu.empty((nch, n), dtype=gpu.float32, device="cuda")
    for c in range(nch):
        d[c] = redio.synth_f32(100 + c, 0, n)
    x = {c: oracle.synth_f32(100 + c, 0, n) for c in check}
    xs = np.stack([x[c] for c in check])
    want = _oracle_channels(oracle, xs, range(len(check)), ratio, [0, n])
    exact, fast = redio.Src(nch, 1), redio.Src(nch, 1, mode=redio.Src.FAST)
    a, ua = exact.process(d, ratio)
    b, ub = fast.process(d, ratio)
    assert ua == ub == n and a.shape == b.shape
    for i, c in enumerate(check):
        assert np.array_equal(bits(a[c].cpu().numpy()), bits(want[i][0])), c
    assert (a - b).abs().max().item() <= _fast_bound(oracle, ratio, 1.0)


def test_c3_256_channels_two_messages(gpu, redio, oracle):
    """BASELINE.json configs[2] at its channel count: 256 independent mono states (samplerate.rs:61), 2.4 MS/s -> 48 kS/s
    (ratio 0.02), 2^18 frames per channel in two messages of unequal length.  EXACT is bit-identical to the oracle on
    channels spread over the whole batch (first / last of every 64-channel group and some inside); FAST stays inside its bound."""
    nch, n, ratio = 256, 1 << 18, 0.02
    x = np.stack([oracle.synth_f32(0x5EED0003 + c, 0, n) for c in range(nch)])
    d = gpu.from_numpy(x).cuda()
    exact, fast = redio.Src(nch, 1), redio.Src(nch, 1, mode=redio.Src.FAST)
    check = (0, 1, 31, 63, 64, 100, 127, 128, 191, 192, 200, 254, 255)
    refs = {c: oracle.Resampler(1) for c in check}
    tab, half, inc = oracle.src_table(1)
    pos = np.arange(0.0, half, inc * ratio)
    K = 2 * len(pos)
    sum_h = 2 * ratio * np.abs(np.interp(pos, np.arange(half + 2), tab.astype(np.float64))).sum()
    bound = (K + 1) * 2.0 ** -24 * max(sum_h, 1.0) * np.abs(x).max()
    total = wtotal = 0
    for lo, hi in ((0, 150001), (150001, n)):
        a, ua = exact.process(d[:, lo:hi].contiguous(), ratio)
        b, ub = fast.process(d[:, lo:hi].contiguous(), ratio)
        assert ua == ub == hi - lo and a.shape == b.shape and a.shape[0] == nch
        total += a.shape[1]
        an = a.cpu().numpy()
        for c in check:
            err, want, wused = refs[c].process(x[c, lo:hi], ratio, int(ratio * (hi - lo) + 1.0))
            assert err == 0 and wused == ua
            assert np.array_equal(bits(an[c]), bits(want)), (c, lo, hi)
        wtotal += len(want)
        assert (a - b).abs().max().item() <= bound
    # the output count over the whole stream: the oracle's, i.e. ratio * frames less the converter's start-up delay
    # (half the stretched filter: the first outputs wait for input that has not arrived)
    assert total == wtotal and 0 <= int(n * ratio) - total <= 64


@pytest.mark.parametrize("nch", [1, 5])
@pytest.mark.parametrize("ratio,conv,periodic", [(48000 / 44100, 1, True), (2.0, 1, True), (1.5, 1, True), (0.3, 1, True), (4 / 3, 2, True),
                                                 (44100 / 48000, 0, True), (0.75, 1, True), (0.0213, 1, False), (3.7, 1, None),
                                                 (2 ** 0.5, 1, False), (0.0917, 2, False), (3.14159265358979 / 3, 0, False)])
def test_rational_ratios_take_the_periodic_phase_kernel_bit_exactly(gpu, redio, oracle, ratio, conv, periodic, nch):
    """samplerate::resample takes any ratio: f64 (samplerate.rs:59).  A constant rational ratio makes the per-output
    (filter start index, position step) repeat every P outputs; those epochs run src_sinc_periodic_kernel (P sets of
    coefficients, LDS tiles) instead of the per-tap interpolating kernel.  Same bits as the literal one-launch-per-refill
    schedule (EPOCHS) and as the oracle, same counts, same carried state across messages of awkward lengths."""
    n = 130000
    x = np.stack([oracle.synth_f32(700 + c, 0, n) for c in range(nch)])
    d = gpu.from_numpy(x).cuda()
    new, lit = redio.Src(nch, conv), redio.Src(nch, conv, mode=redio.Src.EPOCHS)
    refs = [oracle.Resampler(conv) for _ in range(nch)]
    for lo, hi in ((0, 50000), (50000, 50001), (50001, 50120), (50120, n)):
        cap = int(ratio * (hi - lo) + 1.0)
        a, ua = new.process(d[:, lo:hi].contiguous(), ratio, output_frames=cap)
        b, ub = lit.process(d[:, lo:hi].contiguous(), ratio, output_frames=cap)
        assert ua == ub and a.shape == b.shape
        an = a.cpu().numpy()
        assert np.array_equal(bits(an), bits(b.cpu().numpy())), (ratio, lo, hi)
        for c in range(nch):
            err, want, wused = refs[c].process(x[c, lo:hi], ratio, cap)
            assert err == 0 and wused == ua and np.array_equal(bits(an[c]), bits(want)), (ratio, c, lo, hi)
    per, gen = new.path_counts()
    if periodic is True:
        assert per > 0 and per >= gen, (per, gen)     # the long epochs are periodic; only short leftovers may run per tap
    elif periodic is False:
        assert per == 0 and gen > 0
    assert lit.path_counts()[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("ratio,conv", [(48000 / 44100, 2), (1.5, 1), (2.0, 2), (0.3, 1), (4 / 3, 2)])
def test_periodic_ratios_long_messages(gpu, redio, oracle, ratio, conv):
    """Messages of several hundred thousand frames at periodic ratios -- dozens of buffer moves of the library's control flow inside one
    call, two channels, carried state: the oracle's bits and counts, and the literal per-refill schedule's.  (Round 5 ran these calls
    through a window form of the periodic-phase kernel, one launch per 65536 outputs with one set of tables per call: bit-identical and
    8 x fewer launches, but no faster -- such calls are bound by the host's per-output recurrence; profiles/r05_src_periodic_window_null.txt.)"""
    n = 420000
    x = np.stack([oracle.synth_f32(900 + c, 0, n) for c in range(2)])
    d = gpu.from_numpy(x).cuda()
    new, lit = redio.Src(2, conv), redio.Src(2, conv, mode=redio.Src.EPOCHS)
    refs = [oracle.Resampler(conv) for _ in range(2)]
    for lo, hi in ((0, 260000), (260000, 260300), (260300, n)):
        cap = int(ratio * (hi - lo) + 1.0)
        a, ua = new.process(d[:, lo:hi].contiguous(), ratio, output_frames=cap)
        b, ub = lit.process(d[:, lo:hi].contiguous(), ratio, output_frames=cap)
        assert ua == ub and a.shape == b.shape
        an = a.cpu().numpy()
        assert np.array_equal(bits(an), bits(b.cpu().numpy())), (ratio, lo, hi)
        for c in range(2):
            err, want, wused = refs[c].process(x[c, lo:hi], ratio, cap)
            assert err == 0 and wused == ua and np.array_equal(bits(an[c]), bits(want)), (ratio, c, lo, hi)
    assert new.path_counts()[0] > 0


def test_periodic_path_then_ratio_change_and_flush(gpu, redio, oracle):
    # a periodic message, then a varying-ratio one (general kernel), then end_of_input on a periodic ratio again
    n = 120000
    x = oracle.synth_f32(78, 0, n)
    d = gpu.from_numpy(x[None, :]).cuda()
    plan, ref = redio.Src(1, 1), oracle.Resampler(1)
    for lo, hi, r, eoi in ((0, 50000, 1.25, 0), (50000, 80000, 1.6, 0), (80000, n, 1.6, 1)):
        cap = int(r * (hi - lo) + 1.0) + 400
        got, used = plan.process(d[:, lo:hi].contiguous(), r, output_frames=cap, end_of_input=bool(eoi))
        err, want, wused = ref.process(x[lo:hi], r, cap, bool(eoi))
        assert err == 0 and used == wused and got.shape[1] == len(want)
        assert np.array_equal(bits(got.cpu().numpy()[0]), bits(want)), (lo, hi, r)
    assert plan.path_counts()[0] > 0


@pytest.mark.parametrize("conv", [3, 4])
@pytest.mark.parametrize("channels", [1, 2, 5])
@pytest.mark.parametrize("ratio", [2.0, 0.5, 48000 / 44100, 0.0213, 1.0, 3.7])
def test_zero_order_hold_and_linear_converters(gpu, redio, oracle, conv, channels, ratio):
    """SRC_ZERO_ORDER_HOLD / SRC_LINEAR (samplerate.rs:29-30; the reference only ever asks for converter 1) through the
    src_* drop-in, interleaved channels, state carried across messages, bit for bit against the oracle's restatement of
    the published src_zoh.c / src_linear.c."""
    from libredio_amd import samplerate
    n = 30000
    x = np.stack([oracle.synth_f32(40 + c, 0, n) for c in range(channels)], 1).reshape(-1)     # interleaved
    st, ref = samplerate.State(conv, channels), oracle.Resampler(conv, channels)
    pos = 0
    for frames in (7000, 1, 2, 9997, 13000):
        msg = x[pos * channels:(pos + frames) * channels]
        cap = int(ratio * frames + 1.0) + 3
        e1, o1, u1 = st.process(msg, ratio, cap, 0)
        e2, o2, u2 = ref.process(msg, ratio, cap, False)
        assert (e1, u1, len(o1)) == (e2, u2, len(o2)), (conv, channels, ratio, frames)
        assert np.array_equal(bits(o1), bits(o2)), (conv, channels, ratio, frames)
        pos += u1 if u1 else frames
    # a ratio change inside a call (linear interpolation of the ratio) and a reset
    msg = x[: 5000 * channels]
    for r in (ratio, min(ratio * 1.3, 200.0)):
        e1, o1, u1 = st.process(msg, r, int(r * 5000) + 10, 0)
        e2, o2, u2 = ref.process(msg, r, int(r * 5000) + 10, False)
        assert (e1, u1, len(o1)) == (e2, u2, len(o2)) and np.array_equal(bits(o1), bits(o2))
    st.reset(); ref = oracle.Resampler(conv, channels)
    e1, o1, u1 = st.process(msg, ratio, int(ratio * 5000) + 10, 0)
    e2, o2, u2 = ref.process(msg, ratio, int(ratio * 5000) + 10, False)
    assert (e1, u1, len(o1)) == (e2, u2, len(o2)) and np.array_equal(bits(o1), bits(o2))
    st.close()


@pytest.mark.parametrize("conv,channels,ratio", [(1, 2, 0.02), (1, 2, 48000 / 44100), (2, 3, 0.5), (0, 2, 2.0), (1, 4, 0.0213)])
def test_interleaved_channels_through_the_drop_in(gpu, redio, oracle, conv, channels, ratio):
    """src_new(converter, channels > 1): interleaved frames in and out; every channel is converted exactly as a mono
    stream (the reference itself always passes 1, samplerate.rs:61)."""
    from libredio_amd import samplerate
    n = 40000
    cols = [oracle.synth_f32(60 + c, 0, n) for c in range(channels)]
    x = np.stack(cols, 1).reshape(-1)
    st, ref = samplerate.State(conv, channels), oracle.Resampler(conv, channels)
    monos = [oracle.Resampler(conv, 1) for _ in range(channels)]
    pos = 0
    for frames in (15000, 3, 24997):
        msg = x[pos * channels:(pos + frames) * channels]
        cap = int(ratio * frames + 1.0)
        e1, o1, u1 = st.process(msg, ratio, cap, 0)
        e2, o2, u2 = ref.process(msg, ratio, cap, False)
        assert (e1, u1, len(o1)) == (e2, u2, len(o2))
        assert np.array_equal(bits(o1), bits(o2)), (conv, channels, ratio, frames)
        for c in range(channels):
            _, oc, _ = monos[c].process(cols[c][pos:pos + frames], ratio, cap, False)
            assert np.array_equal(bits(o1[c::channels]), bits(oc))
        pos += frames
    st.close()


@pytest.mark.gpu
def test_linear_converter_three_channels_boundary(gpu, redio, oracle):
    """Found by tests/fuzz_parity.py (seed 777001): converter 4 (linear), 3 interleaved channels, ratio 48000 / 44100, messages of
    6439 then 15317 frames -- the library compares in_used + channels * input_index with in_count in SAMPLE units (src_linear.c as
    published; samplerate.rs:26-30 declares the converter), and with a channel count that is not a power of two the frame form of
    the same inequality decided the last output of the second message differently (one frame more consumed and produced).
    Counts, outputs and carried state must be the oracle's for these and neighbouring lengths."""
    from libredio_amd import samplerate
    ratio = 48000 / 44100
    for conv in (4, 3):
        for first in (6439, 6438, 6440):
            st, ref = samplerate.State(conv, 3), oracle.Resampler(conv, 3)
            for m in (first, 15317, 1, 2, 4099):
                x = oracle.synth_f32(1000 + m, 0, m * 3)
                cap = int(ratio * m + 1.0)
                e1, a, u1 = st.process(x, ratio, cap, 0)
                e2, b, u2 = ref.process(x, ratio, cap, False)
                assert (e1, u1, len(a)) == (e2, u2, len(b)), (conv, first, m)
                assert np.array_equal(bits(a), bits(b)), (conv, first, m)
            st.close()


@pytest.mark.gpu
def test_batched_rows_zoh_linear_are_mono_streams(gpu, redio, oracle):
    """redio_src_process on [nchan][frames] rows: every row is its own mono stream (samplerate.rs:61 resamples one channel), so
    converters 3 / 4 must make the library's end-of-input decision with a channel count of ONE whatever nchan is -- the same
    message lengths as the interleaved boundary case above, 3 and 5 rows."""
    ratio = 48000 / 44100
    for conv in (4, 3):
        for nch in (3, 5):
            plan = redio.Src(nch, conv)
            refs = [oracle.Resampler(conv) for _ in range(nch)]
            for m in (6439, 15317, 1, 2, 4099):
                x = np.stack([oracle.synth_f32(77 + c + m, 0, m) for c in range(nch)])
                cap = int(ratio * m + 1.0)
                a, used = plan.process(gpu.from_numpy(x).cuda(), ratio, output_frames=cap)
                a = a.cpu().numpy()
                for c in range(nch):
                    err, want, wused = refs[c].process(x[c], ratio, cap)
                    assert err == 0 and wused == used and a.shape[1] == len(want), (conv, nch, m, c)
                    assert np.array_equal(bits(a[c]), bits(want)), (conv, nch, m, c)


@pytest.mark.gpu
def test_linear_converter_three_channels_ratio_glide(gpu, redio, oracle):
    """Found by tests/fuzz_parity.py (seed 20261003): when the ratio changes between messages the library glides to it inside the
    message, src_ratio = last_ratio + out_gen * (new - last) / out_count with out_gen / out_count counted in SAMPLES; with three
    interleaved channels the frame form of that quotient rounds differently.  The failing sequences of that run, converters 4 and 3."""
    from libredio_amd import samplerate
    cases = [([5675, 14488, 17186], [3.0, 3.0, 1.7]), ([15986, 2, 19949], [1.3503012770611056, 0.9, 1.3503012770611056]),
             ([17927, 11446, 6927, 16542], [0.6108250699566179, 0.4, 1.1, 0.6108250699566179]),
             ([8552, 17184, 17207, 7138], [1.4500575425380535, 2.2, 0.8, 1.4500575425380535]), ([10706, 3461, 1281, 11305], [3.0, 1.5, 3.0, 2.9])]
    for conv in (4, 3):
        for ch in (3, 5, 2):
            for sizes, ratios in cases:
                st, ref = samplerate.State(conv, ch), oracle.Resampler(conv, ch)
                for m, ratio in zip(sizes, ratios):
                    x = oracle.synth_f32(31 + m, 0, m * ch)
                    cap = int(ratio * m + 1.0)
                    e1, a, u1 = st.process(x, ratio, cap, 0)
                    e2, b, u2 = ref.process(x, ratio, cap, False)
                    assert (e1, u1, len(a)) == (e2, u2, len(b)), (conv, ch, sizes, m)
                    assert np.array_equal(bits(a), bits(b)), (conv, ch, sizes, m)
                st.close()


# the message sequence tests/fuzz_parity.py found in round 4 (seed 40426): the ratio falls between two calls, so the filter is wider than the
# history the library's buffer retains and the left wing reaches IN FRONT of the buffer (libsamplerate 0.1.8 reads the filter struct's own
# fields there).  Oracle and device define those samples as +0.0f (include/samplerate.h); the call runs the unclamped per-lane kernel.
RATIO_FALLS = [(1086, 708690820, '0x1.47ae147ae147bp-6', 22), (7842, 861348262, '0x1.e43d5e17e519ap-7', 116), (1103, 139141258, '0x1.e43d5e17e519ap-7', 17),
               (2719, 811523618, '0x1.ee08c42c7828dp-7', 41)]


# a second sequence (seed 60407, round 4): a long message at 1/50 runs the single-launch form, then the ratio falls to 1/100 over two short messages
# that produce nothing; the fourth message's left wing reads history the FIRST call left in the buffer, older than that call's own filter needed:
# the single-launch form must rebuild the image back to the widest filter's reach, not only to its own
RATIO_FALLS_2 = [(13762, 206420031, '0x1.47ae147ae147bp-6', 276), (327, 191814179, '0x1.484d44c389fe9p-7', 4), (1908, 356774819, '0x1.47ae147ae147bp-7', 20),
                 (2186, 528628988, '0x1.47ae147ae147bp-7', 22)]


@pytest.mark.gpu
@pytest.mark.parametrize("conv", [0, 1, 2])
@pytest.mark.parametrize("falls", [RATIO_FALLS, RATIO_FALLS_2])
def test_ratio_decrease_reaches_in_front_of_the_buffer(gpu, redio, oracle, conv, falls):
    from libredio_amd import samplerate
    for mode in ("dropin", redio.Src.EXACT, redio.Src.EPOCHS):
        st = samplerate.State(conv, 1) if mode == "dropin" else redio.Src(1, conv, mode=mode)
        ref = oracle.Resampler(conv, 1)
        total = 0
        for m, seed, rh, cap in falls:
            x = oracle.synth_f32(seed, 0, m); ratio = float.fromhex(rh)
            e2, want, u2 = ref.process(x, ratio, cap, False)
            if mode == "dropin":
                e1, got, u1 = st.process(x, ratio, cap, 0)
            else:
                a, u1 = st.process(gpu.from_numpy(x).cuda().view(1, -1), ratio, output_frames=cap, end_of_input=False)
                e1, got = 0, a.cpu().numpy()[0]
            assert (e1, u1, len(got)) == (e2, u2, len(want)), (conv, mode)
            assert np.array_equal(bits(got), bits(want)), (conv, mode, m)
            assert np.all(np.abs(want) < 4.0)          # samples are in [-1, 1): nothing from in front of the buffer leaks in
            total += len(want)
        if conv == 0 and falls is RATIO_FALLS:
            assert total > 40                           # the widened best-quality filter did produce outputs in the calls that reach back


# round 5: a call that moves nothing appends to the live image in place (src_host.hip, try_uniform_window / the tile form); a long run of small
# messages -- several buffer moves apart, the ratio falling and rising between them -- must read the same history the library's buffer holds
@pytest.mark.gpu
@pytest.mark.parametrize("conv,nch", [(0, 1), (1, 3), (2, 2)])
def test_many_small_messages_with_falling_and_rising_ratios(gpu, redio, oracle, conv, nch):
    rng = np.random.default_rng(500 + conv)
    for mode in (redio.Src.EXACT, redio.Src.EPOCHS):
        plan = redio.Src(nch, conv, mode=mode)
        refs = [oracle.Resampler(conv) for _ in range(nch)]
        ratio = 0.02
        rs = np.random.default_rng(77)
        for i in range(70):
            m = int(rs.integers(200, 5000)) if i % 9 else int(rs.integers(1, 40))
            if i and i % 7 == 0: ratio = float(rs.choice([0.02, 0.01, 0.005, 0.04, 1 / 64, 0.02, 0.0213]))
            x = np.stack([oracle.synth_f32(int(rng.integers(1, 1 << 30)), 0, m) for _ in range(nch)])
            cap = int(ratio * m + 1.0)
            got, used = plan.process(gpu.from_numpy(x).cuda(), ratio, output_frames=cap, end_of_input=False)
            got = got.cpu().numpy()
            for c in range(nch):
                e2, want, u2 = refs[c].process(x[c], ratio, cap, False)
                assert (0, used, got.shape[1]) == (e2, u2, len(want)), (conv, mode, i, m, ratio)
                assert np.array_equal(bits(got[c]), bits(want)), (conv, mode, i, m, ratio, c)


# found by the randomised run with ratios over the library's whole range (round 4): end_of_input at a ratio of 1 / 256 makes prepare_data's
# last move longer than the buffer (libsamplerate 0.1.8 overruns its allocation); defined as SRC_ERR_SINC_PREPARE_DATA_BAD_LEN (include/samplerate.h (iv))
EOI_SMALLEST = [(342, 474237719, '0x1.0000000000000p+0', 343, 0), (4, 835049986, '0x1.a074d40eaf3ffp-3', 1, 0),
                (5594, 866347677, '0x1.9905d40507092p-7', 70, 0), (14858, 933052245, '0x1.0000000000000p-8', 5323, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("ch", [1, 3])
def test_end_of_input_at_the_smallest_ratios(gpu, redio, oracle, ch):
    from libredio_amd import samplerate
    st, ref = samplerate.State(1, ch), oracle.Resampler(1, ch)
    errs = []
    for m, seed, rh, cap, eoi in EOI_SMALLEST:
        x = oracle.synth_f32(seed, 0, m * ch); ratio = float.fromhex(rh)
        e1, got, u1 = st.process(x, ratio, cap, eoi)
        e2, want, u2 = ref.process(x, ratio, cap, bool(eoi))
        assert (e1, u1, len(got)) == (e2, u2, len(want)), (ch, m)
        assert np.array_equal(bits(got), bits(want)), (ch, m)
        errs.append(e1)
    assert errs == [0, 0, 0, 21]                # SRC_ERR_SINC_PREPARE_DATA_BAD_LEN from the call the library cannot serve, nothing written
    if ch == 1:                                  # the batched interface reports the same code
        src, ref = redio.Src(1, 1), oracle.Resampler(1, 1)
        for m, seed, rh, cap, eoi in EOI_SMALLEST[:3]:
            x = oracle.synth_f32(seed, 0, m); ratio = float.fromhex(rh)
            a, u1 = src.process(gpu.from_numpy(x).cuda().view(1, -1), ratio, output_frames=cap, end_of_input=bool(eoi))
            e2, want, u2 = ref.process(x, ratio, cap, bool(eoi))
            assert e2 == 0 and u1 == u2 and np.array_equal(bits(a.cpu().numpy()[0]), bits(want))
        m, seed, rh, cap, eoi = EOI_SMALLEST[3]
        with pytest.raises(redio.RedioError) as e:
            src.process(gpu.from_numpy(oracle.synth_f32(seed, 0, m)).cuda().view(1, -1), float.fromhex(rh), output_frames=cap, end_of_input=True)
        assert e.value.code == 21
    # the smallest ratio from a fresh state: a whole stream with the flush at its end (whatever the library's flow decides, both sides agree)
    for n, cap_extra in ((40000, 0), (40000, 300), (70000, 500), (200, 50)):
        st, ref = samplerate.State(1, ch), oracle.Resampler(1, ch)
        x = oracle.synth_f32(77 + n, 0, n * ch)
        for lo, hi, eoi in ((0, n // 3, 0), (n // 3, n, 1)):
            cap = int((hi - lo) / 256 + 1.0) + (cap_extra if eoi else 0)
            e1, got, u1 = st.process(x[lo * ch:hi * ch], 1 / 256, cap, eoi)
            e2, want, u2 = ref.process(x[lo * ch:hi * ch], 1 / 256, cap, bool(eoi))
            assert (e1, u1, len(got)) == (e2, u2, len(want)), (ch, n, lo)
            assert np.array_equal(bits(got), bits(want)), (ch, n, lo)


@pytest.mark.gpu
@pytest.mark.parametrize("conv,ch", [(1, 1), (0, 1), (2, 2), (3, 1), (4, 2)])
def test_src_set_ratio_steps_instead_of_gliding(gpu, redio, oracle, conv, ch):
    """src_set_ratio (samplerate.rs:40) overwrites the ratio the next call starts from: a call at the new ratio then runs at it from its
    first output (no glide); a call at a third ratio glides from the SET one.  Bad arguments return the library's codes."""
    from libredio_amd import samplerate
    st, ref = samplerate.State(conv, ch), oracle.Resampler(conv, ch)
    assert st.set_ratio(300.0) == ref.set_ratio(300.0) == 6 and st.set_ratio(1e-3) == ref.set_ratio(1e-3) == 6   # SRC_ERR_BAD_SRC_RATIO
    seq = [(3000, 0.5, None), (3000, 0.25, 0.25), (2500, 0.4, 0.3), (1, 0.4, None), (4000, 1.5, 1.5), (3000, 0.02, 0.05), (6000, 0.02, None)]
    for i, (m, ratio, setr) in enumerate(seq):
        x = oracle.synth_f32(500 + i, 0, m * ch)
        if setr is not None:
            assert st.set_ratio(setr) == ref.set_ratio(setr) == 0
        cap = int(ratio * m + 1.0)
        e1, got, u1 = st.process(x, ratio, cap, 0)
        e2, want, u2 = ref.process(x, ratio, cap, False)
        assert (e1, u1, len(got)) == (e2, u2, len(want)) and e1 == 0, (conv, ch, i)
        assert np.array_equal(bits(got), bits(want)), (conv, ch, i)
    # the step is visible: the same two messages WITHOUT the set glide, and differ
    if conv < 3:
        a, b = oracle.Resampler(conv, ch), oracle.Resampler(conv, ch)
        x0, x1 = oracle.synth_f32(1, 0, 3000 * ch), oracle.synth_f32(2, 0, 3000 * ch)
        a.process(x0, 0.5, 1501, False); b.process(x0, 0.5, 1501, False)
        b.set_ratio(0.25)
        ya, yb = a.process(x1, 0.25, 751, False)[1], b.process(x1, 0.25, 751, False)[1]
        assert len(ya) != len(yb) or not np.array_equal(bits(ya), bits(yb))
