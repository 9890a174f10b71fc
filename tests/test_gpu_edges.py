"""Edge cases of the device paths: empty and ragged inputs, unaligned device pointers (scalar-load
variants), shapes without a specialised kernel, maximum tap counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_empty_inputs_everywhere(gpu, redio, oracle):
    z = gpu.zeros(0, dtype=gpu.complex64, device="cuda")
    taps = oracle.lpf_corrected(127, 0.08)
    assert redio.Fir(taps, 5)(z).numel() == 0
    assert redio.Fft(1024)(z).numel() == 0
    assert redio.Chain(taps, 5, 1024)(z).numel() == 0
    assert redio.OverlapSave(taps, 1024)(z).numel() == 0
    assert redio.Channelizer(oracle.lpf_corrected(1024, 0.007))(z).numel() == 0
    assert redio.bitfount.data_to_samples(gpu.zeros(0, dtype=gpu.uint8, device="cuda")).numel() == 0
    assert redio.bitfount.discretize(gpu.zeros(0, dtype=gpu.float32, device="cuda")).numel() == 0
    v, c = redio.kpn_dev.Rle().feed(gpu.zeros(0, dtype=gpu.uint8, device="cuda"))
    assert v.numel() == 0 and c.numel() == 0
    out, used = redio.Src(2).process(gpu.zeros((2, 0), dtype=gpu.float32, device="cuda"), 0.5)
    assert out.shape == (2, 0) and used == 0
    assert len(redio.dsputils.convolve(np.zeros(0, np.float32), np.ones(3, np.float32))) == 0


@pytest.mark.parametrize("k,d", [(127, 5), (63, 1), (33, 2)])
def test_fir_unaligned_device_pointers(gpu, redio, oracle, k, d):
    # a view that starts 8 bytes into an allocation: the 16-byte vector paths must not be taken
    taps = oracle.lpf_corrected(k, 0.08) if k != 33 else oracle.synth_f32(1, 0, 33)
    x = oracle.synth_iq(17, 0, 20001)
    dx = gpu.from_numpy(x).cuda()
    for fused in (False, True):
        got = redio.Fir(taps, d, fused=fused)(dx[1:]).cpu().numpy()
        assert np.array_equal(bits(got), bits(oracle.fir(x[1:], taps, d, fused)))
    # unaligned OUTPUT as well
    plan = redio.Fir(taps, d)
    buf = gpu.zeros(plan.nout(20000) + 1, dtype=gpu.complex64, device="cuda")
    got = plan(dx[1:], out=buf[1:]).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.fir(x[1:], taps, d, False)))


def test_chain_unaligned_input_takes_a_correct_path(gpu, redio, oracle):
    taps = oracle.lpf_corrected(127, 0.08)
    x = oracle.synth_iq(3, 0, 3 * 5120 + 126 + 1)
    dx = gpu.from_numpy(x).cuda()
    for fused in (True, False):
        got = redio.Chain(taps, 5, 1024, fused=fused)(dx[1:]).cpu().numpy()
        assert np.array_equal(bits(got), bits(oracle.chain_fir_fft(x[1:], taps, 5, 1024, fused)))


def test_fir_long_filter_direct_path(gpu, redio, oracle):
    # 8193 taps (the overlap-save size) through the direct kernel; decimation larger than the filter
    taps = oracle.lpf_corrected(8193, 0.01)
    x = oracle.synth_iq(5, 0, 8193 + 700)
    assert np.array_equal(bits(redio.Fir(taps, 1)(gpu.from_numpy(x).cuda()).cpu().numpy()), bits(oracle.fir(x, taps, 1, False)))
    t3 = np.array([0.25, 0.5, 0.25], np.float32)
    y = oracle.synth_f32(6, 0, 1000)
    assert np.array_equal(bits(redio.Fir(t3, 7, complex_input=False)(gpu.from_numpy(y).cuda()).cpu().numpy()), bits(oracle.fir(y, t3, 7, False)))


def test_fir_real_and_complex_full_tiles_and_ragged_tails(gpu, redio, oracle):
    taps = oracle.lpf_corrected(127, 0.08)
    for n in (127 + 5 * 1023, 127 + 5 * 1024, 127 + 5 * 1025, 127 + 5 * 4095 + 4):   # around one and four v4 blocks
        x = oracle.synth_iq(n, 0, n)
        for fused in (True, False):
            got = redio.Fir(taps, 5, fused=fused)(gpu.from_numpy(x).cuda()).cpu().numpy()
            assert np.array_equal(bits(got), bits(oracle.fir(x, taps, 5, fused))), (n, fused)


def test_fft_in_place_all_paths(gpu, redio, oracle):
    for n in (64, 1024, 4096, 65536):
        x = oracle.synth_iq(n, 0, n * 3)
        d = gpu.from_numpy(x).cuda()
        redio.Fft(n)(d, out=d)
        assert np.array_equal(bits(d.cpu().numpy()), bits(oracle.fft(x, n)))


def test_resampler_extreme_ratios_and_tiny_messages(gpu, redio, oracle):
    from libredio_amd import samplerate
    x = oracle.synth_f32(4, 0, 70000)
    for ratio in (1 / 256, 256.0):
        st, ref = samplerate.State(2, 1), oracle.Resampler(2)
        n = 70000 if ratio < 1 else 300
        a, b = st.block(x[:n], ratio), ref.block(x[:n], ratio)
        assert len(a) == len(b) and np.array_equal(bits(a), bits(b))
    st, ref = samplerate.State(1, 1), oracle.Resampler(1)
    for m in (1, 1, 2, 3, 5000, 1):                         # messages shorter than anything useful
        a, b = st.block(x[:m], 0.5), ref.block(x[:m], 0.5)
        assert len(a) == len(b) and np.array_equal(bits(a), bits(b))
