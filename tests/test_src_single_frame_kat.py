"""Known answers for single-frame messages through the zero-order-hold and linear converters (samplerate.rs:26-30 declares them).

oracle/oracle_src.c and the device path define one case by agreement rather than by the published text: a message of ONE frame
reaches the main loop of src_zoh.c / src_linear.c (libsamplerate 0.1.8) with in_used == 0, where the published code reads
data_in[-channels], before the array; both sides use the value carried from the previous message instead.  Every other parity
test compares those two sides with each other, so this file pins the case on its own: the expected outputs below were derived
BY HAND from the converters' recurrence (first loop: outputs before the first sample of the message; main loop: a + idx * (b - a)
between the previous and the current frame; position advanced by 1 / ratio and reduced modulo 1) and are written out as explicit
index lists and float32 expressions -- none of them is produced by oracle/ or by the library.  A stream fed one frame at a
time must also give the outputs of the same stream fed in one message (the converters are streaming), which the table shows too.
"""
import numpy as np
import pytest

X = np.array([0.625, -1.5, 2.25, 0.0078125, -3.0, 1.75, 0.5], np.float32)   # exact in float32; differences are exact too


def lin(a, b, frac):
    """(float)(a + frac * (b - a)) with the difference formed in float, as the C expression of src_linear.c does"""
    d = np.float32(np.float32(b) - np.float32(a))
    return np.float32(np.float64(np.float32(a)) + np.float64(frac) * np.float64(d))


# converter, ratio -> what message k (the single frame X[k]) must emit
KAT = {
    # ratio 1: both converters run one input frame behind; ZOH emits the first frame twice in the first message
    (3, 1.0): [[X[0], X[0]], [X[1]], [X[2]], [X[3]], [X[4]], [X[5]], [X[6]]],
    (4, 1.0): [[X[0]], [X[0]], [X[1]], [X[2]], [X[3]], [X[4]], [X[5]]],
    # ratio 1/2 (one output per two frames): every other message is silent
    (3, 0.5): [[X[0]], [X[1]], [], [X[3]], [], [X[5]], []],
    (4, 0.5): [[X[0]], [], [X[1]], [], [X[3]], [], [X[5]]],
    # ratio 2/3 (step 1.5): the linear converter meets fractional positions 0.5 inside single-frame messages:
    # the interpolation runs between the frame CARRIED from the previous message and the message's only frame
    (4, 2.0 / 3.0): [[X[0]], [lin(X[0], X[1], 0.5)], [], [X[2]], [lin(X[3], X[4], 0.5)], [], [X[5]]],
}


def run_single_frames(make_state, conv, ratio):
    st = make_state(conv)
    got = []
    for k in range(len(X)):
        err, out, used = st.process(X[k:k + 1], ratio, 4, False)
        assert err == 0 and used == 1, (conv, ratio, k, err, used)
        got.append(list(out))
    return got


def check(got, conv, ratio):
    want = KAT[(conv, ratio)]
    for k, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w), (conv, ratio, k, g, w)
        assert np.array_equal(np.array(g, np.float32).view(np.uint32), np.array(w, np.float32).view(np.uint32)), (conv, ratio, k, g, w)


@pytest.mark.parametrize("conv,ratio", sorted(KAT))
def test_oracle_single_frame_messages_match_the_hand_derived_answers(oracle, conv, ratio):
    check(run_single_frames(lambda c: oracle.Resampler(c), conv, ratio), conv, ratio)
    # and the same stream in ONE message gives the concatenation (streaming invariance; capacity is ample)
    err, out, used = oracle.Resampler(conv).process(X, ratio, 16, False)
    flat = [v for m in KAT[(conv, ratio)] for v in m]
    assert err == 0 and used == len(X)
    assert np.array_equal(out[: len(flat)].view(np.uint32), np.array(flat, np.float32).view(np.uint32)), (conv, ratio, out, flat)


@pytest.mark.gpu
@pytest.mark.parametrize("conv,ratio", sorted(KAT))
def test_device_single_frame_messages_match_the_hand_derived_answers(gpu, redio, conv, ratio):
    """the libsamplerate.so drop-in (src_new / src_process, samplerate.rs:32-42) on the same messages"""
    import libredio_amd.samplerate as S
    got = run_single_frames(lambda c: S.State(c, 1), conv, ratio)
    check(got, conv, ratio)
