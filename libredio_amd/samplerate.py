"""Mirror of the reference's samplerate crate (src/samplerate/src/samplerate.rs) over libsamplerate.so.

`resample(din, dout, ratio)` keeps the reference block's signature (samplerate.rs:59): one
src_new(SRC_SINC_MEDIUM_QUALITY, 1) state for the life of the block (:61); per message the output
capacity is `(ratio*len + 1) as usize` (:64), end_of_input = 0 (:73), the block sends
output_frames_gen samples (:84) and never looks at input_frames_used (:71); a non-zero return is the
reference's panic with the src_strerror text (:77-83).  Channels are anything with get()/put();
`None` ends the block.
"""
import ctypes as C

import numpy as np

from . import samplerate_lib

SRC_SINC_BEST_QUALITY, SRC_SINC_MEDIUM_QUALITY, SRC_SINC_FASTEST, SRC_ZERO_ORDER_HOLD, SRC_LINEAR = range(5)


class SRC_DATA(C.Structure):  # samplerate.rs:15-24, C layout
    _fields_ = [("data_in", C.c_void_p), ("data_out", C.c_void_p),
                ("input_frames", C.c_long), ("output_frames", C.c_long),
                ("input_frames_used", C.c_long), ("output_frames_gen", C.c_long),
                ("end_of_input", C.c_int), ("src_ratio", C.c_double)]


class SrcError(RuntimeError):
    def __init__(self, code):
        self.code = code
        msg = samplerate_lib().src_strerror(code)
        super().__init__(msg.decode() if msg else f"src error {code}")


class State:
    """src_new(converter, channels, &error) (samplerate.rs:61)."""

    def __init__(self, converter=SRC_SINC_MEDIUM_QUALITY, channels=1):
        err = C.c_int(0)
        self.channels = int(channels)
        self._s = samplerate_lib().src_new(converter, channels, C.byref(err))
        if not self._s:
            raise SrcError(err.value)

    def process(self, vin, ratio, output_frames, end_of_input=0):
        """src_process(state, &SRC_DATA): vin holds interleaved frames (len = frames * channels);
        returns (error, interleaved output[:gen * channels], input_frames_used)."""
        ch = self.channels
        vin = np.ascontiguousarray(vin, dtype=np.float32)
        vout = np.empty(max(int(output_frames), 1) * ch, np.float32)
        d = SRC_DATA(vin.ctypes.data, vout.ctypes.data, len(vin) // ch, int(output_frames), 0, 0, int(end_of_input), float(ratio))
        err = samplerate_lib().src_process(self._s, C.byref(d))
        return err, vout[: d.output_frames_gen * ch].copy(), d.input_frames_used

    def block(self, vin, ratio):
        """One message of the resample block (samplerate.rs:63-85)."""
        lout = int(ratio * len(vin) + 1.0)
        err, out, _ = self.process(vin, ratio, lout, 0)
        if err != 0:
            raise SrcError(err)  # panic!(src_strerror(error))
        return out

    def reset(self):
        return samplerate_lib().src_reset(self._s)

    def set_ratio(self, ratio):
        return samplerate_lib().src_set_ratio(self._s, float(ratio))

    def close(self):
        if getattr(self, "_s", None):
            samplerate_lib().src_delete(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resample(din, dout, ratio):
    """samplerate::resample(din, dout, ratio) (samplerate.rs:59-87)."""
    ctx = State(SRC_SINC_MEDIUM_QUALITY, 1)
    try:
        while True:
            vin = din.get()
            if vin is None:
                break
            dout.put(ctx.block(vin, ratio))
    finally:
        ctx.close()
