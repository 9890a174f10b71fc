"""Known answers that pin the bit-exact oracle paths (oracle/oracle_bits.c) and the SRC oracle's
contract behaviour -- CPU only."""
import numpy as np
import pytest


def test_b2d_eat_known_answers(oracle):
    assert oracle.b2d([1, 0, 1]) == 5
    assert oracle.b2d([]) == 0 and oracle.b2d([1] * 12) == 4095
    bits = [0, 0, 0, 1] + [0] * 7 + [1] + [0, 0, 1, 0] + [0] * 12 + [1, 1, 1, 1, 1, 1, 1, 1]
    assert oracle.eat(bits, [4, 8, 4, 12, 8]) == [1, 1, 2, 0, 255]      # src/ratpak.rs:115
    assert oracle.eat(bits, [4, 8, 2, 10, 12]) == [1, 1, 0, 512, 255]   # src/ratpak.rs:119
    with pytest.raises(IndexError):
        oracle.eat(bits, [30, 10])


def test_data_to_samples_is_i_over_127_minus_1(oracle):
    d = np.arange(256, dtype=np.uint8).repeat(2)
    s = oracle.data_to_samples(d)
    want = (np.arange(256, dtype=np.float32) / np.float32(127.0) - np.float32(1.0)).astype(np.float32)
    assert np.array_equal(s.real, want) and np.array_equal(s.imag, want)
    assert s[0] == -1 - 1j and s[127] == 0 and s[254] == 1 + 1j and s[255].real > 1   # 255/127 - 1 overshoots
    with pytest.raises(IndexError):
        oracle.data_to_samples(np.zeros(3, np.uint8))


def test_discretize_and_block_sum(oracle):
    assert oracle.discretize([0.0, 1.0, 0.4, 0.6, 2.0]).tolist() == [0, 0, 0, 0, 0][:0] + [0, 0, 0, 0, 1]
    assert oracle.discretize([1.0, 0.6, 0.5, 0.4]).tolist() == [1, 1, 0, 0]
    assert oracle.discretize([-1.0, -2.0]).tolist() == [0, 0]            # max stays at the seed 0.0: x > 0 is false
    x = np.array([1e8, 1.0, -1e8], np.float32)
    assert oracle.block_sum(x) == 0.0                                     # sequential: (1e8 + 1) - 1e8 == 0 in f32
    assert oracle.norm(np.array([3 + 4j], np.complex64))[0] == 5.0


def test_trigger_walk(oracle):
    t = oracle.Trigger()
    quiet = np.full((200, 512), 0.01, np.float32)
    assert t.feed(quiet) == []                                            # never triggers on a flat floor
    loud = quiet.copy(); loud[10:15] = 1.0
    out = t.feed(loud)
    assert len(out) == 1
    # 0.0 seed + blocks from the first loud block until the counter runs from 50 down to 2 after the last one
    assert out[0][0] == 0.0 and (len(out[0]) - 1) % 512 == 0
    assert (len(out[0]) - 1) // 512 == 5 + 48


def test_resampler_contract(oracle):
    # samplerate.rs:64: capacity = (ratio*len + 1) as usize; steady-state count law floor..floor+1
    r = oracle.Resampler(1)
    outs = [len(r.block(np.zeros(20000, np.float32), 0.02)) for _ in range(6)]
    assert outs[0] < 400 and all(o in (400, 401) for o in outs[1:])       # start-up latency, then ratio*len
    err, y, used = oracle.Resampler(1).process(np.zeros(100, np.float32), 1000.0, 10)
    assert err == 6                                                       # SRC_ERR_BAD_SRC_RATIO
    with pytest.raises(ValueError):
        oracle.Resampler(5)                                               # five converters (samplerate.rs:26-30)
    # converters 3 / 4 against closed forms: hold repeats the sample before the output instant, linear interpolates
    ramp = np.arange(1000, dtype=np.float32)
    e, z, u = oracle.Resampler(3).process(ramp, 0.25, 300)
    assert e == 0 and u == 1000 and np.array_equal(z[1:5], np.float32([3, 7, 11, 15]))
    e, l, u = oracle.Resampler(4).process(ramp, 4.0, 4100)
    assert e == 0 and np.allclose(l[4:40], np.arange(36) * 0.25, atol=0)          # exact: dyadic steps on integers
    # interleaved channels are independent mono streams
    a2 = oracle.synth_f32(5, 0, 4000); b2 = oracle.synth_f32(6, 0, 4000)
    e, y2, u2 = oracle.Resampler(1, 2).process(np.stack([a2, b2], 1).reshape(-1), 0.5, 2100)
    _, ya, _ = oracle.Resampler(1).process(a2, 0.5, 2100)
    _, yb, _ = oracle.Resampler(1).process(b2, 0.5, 2100)
    assert e == 0 and np.array_equal(y2[0::2], ya) and np.array_equal(y2[1::2], yb)
    # streaming invariance to message segmentation
    x = oracle.synth_f32(3, 0, 9000)
    a = oracle.Resampler(1).block(x, 0.5)
    rb = oracle.Resampler(1)
    b = np.concatenate([rb.block(x[:1234], 0.5), rb.block(x[1234:1235], 0.5), rb.block(x[1235:], 0.5)])
    n = min(len(a), len(b))
    assert n > 4000 and np.array_equal(a[:n], b[:n])


def test_src_table_shape(oracle):
    tab, half, inc = oracle.src_table(1)
    assert (half, inc, len(tab)) == (22436, 491, 22438)                   # sizes of the library's medium table
    assert abs(tab[0] - 0.9425) < 1e-6 and tab[-1] == 0 and np.abs(tab[2000:]).max() < 0.2
    # documented quality class: >= 120 dB stop band, DC gain 1 at ratio 1 (table sampled every `inc`)
    h = np.concatenate([tab[inc:half:inc][::-1], tab[0:half:inc]])
    assert abs(h.sum() - 1.0) < 1e-3
