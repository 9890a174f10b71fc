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
        samplerate.State(3, 1)          # zero-order hold: not built
    assert e.value.code == 10
    with pytest.raises(samplerate.SrcError) as e:
        samplerate.State(1, 2)          # channels > 1: not built
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
