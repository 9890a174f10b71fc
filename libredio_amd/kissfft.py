"""Mirror of the reference's kissfft crate (src/kissfft/src/kissfft.rs) over libkissfft.so.

`fft(pin, cout, block_size, inv)` keeps the reference block's signature (kissfft.rs:18): it owns one
cfg for its life (:19), asserts every message is exactly block_size long (:24), and sends a freshly
allocated output per message (:21-27).  Channels are anything with get()/put() (queue.Queue); a
`None` message ends the block (the reference loops until its channel hangs up and panics).
"""
import ctypes as C

import numpy as np

from . import kisslib


class Cfg:
    """kiss_fft_alloc(nfft, inverse, NULL, NULL) (kissfft.rs:19)."""

    def __init__(self, nfft, inverse=0):
        self.nfft = int(nfft)
        self._cfg = kisslib().kiss_fft_alloc(self.nfft, int(inverse), None, None)
        if not self._cfg:
            raise RuntimeError(f"kiss_fft_alloc({nfft}) failed (no HIP device or bad size)")

    def __call__(self, din):
        """kiss_fft(cfg, fin, fout) (kissfft.rs:26): host complex64 in, new host complex64 out."""
        din = np.ascontiguousarray(din, dtype=np.complex64)
        assert len(din) == self.nfft, "din.len() == block_size (kissfft.rs:24)"
        fout = np.empty(self.nfft, np.complex64)
        kisslib().kiss_fft(self._cfg, din.ctypes.data_as(C.c_void_p), fout.ctypes.data_as(C.c_void_p))
        return fout

    def close(self):
        if self._cfg:
            kisslib().kiss_fft_free(self._cfg)
            self._cfg = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fft(pin, cout, block_size, inv):
    """kissfft::fft(pin, cout, block_size, inv) (kissfft.rs:18-31)."""
    cfg = Cfg(block_size, inv)
    try:
        while True:
            din = pin.get()
            if din is None:
                break
            cout.put(cfg(din))
    finally:
        cfg.close()
        kisslib().kiss_fft_cleanup()
