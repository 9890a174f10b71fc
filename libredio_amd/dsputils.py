"""Mirror of the reference's dsputils crate (src/dsputils/src/dsputils.rs) over libredio.so.

convolve() runs on the MI355X (no CPU path); the tap generators are host-side one-offs, exactly as in
the reference, and reproduce it as written (lpf()[1] is NaN -- SURVEY.md 0.6); lpf_corrected() is the
documented-deviation designer used for benchmark taps.
"""
import ctypes as C

import numpy as np

from . import RedioError, check, lib

_pf = C.POINTER(C.c_float)


def _ptr(a):
    return a.ctypes.data_as(_pf)


def convolve(u, v):
    """dsputils::convolve(u, v) -> Vec (dsputils.rs:30-32): valid-mode correlation, taps not reversed.

    Host arrays in, host array out, synchronous.  len(v) == 0 raises (windows(0) panics in the
    reference); len(u) < len(v) returns an empty array.
    """
    u = np.ascontiguousarray(u, dtype=np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.empty(max(len(u) - len(v) + 1, 1), np.float32)
    n = C.c_size_t(0)
    check(lib().redio_convolve_f32(_ptr(u), len(u), _ptr(v), len(v), _ptr(out), C.byref(n)), "convolve")
    return out[: n.value]


def _gen(name, m, *fcs, extra=0):
    out = np.empty(max(m + extra, 1), np.float32)
    check(getattr(lib(), name)(m, *[float(f) for f in fcs], _ptr(out)), name)
    return out[: m + extra]


def window(m):
    """dsputils::window(m) (dsputils.rs:38-51): m+1 values, [1] is NaN."""
    return _gen("redio_window", m, extra=1)


def sinc(m, fc): return _gen("redio_sinc", m, fc)            # dsputils.rs:53-63
def lpf(m, fc): return _gen("redio_lpf", m, fc)              # dsputils.rs:66-71
def hpf(m, fc): return _gen("redio_hpf", m, fc)              # dsputils.rs:74-79
def bsf(m, fc1, fc2): return _gen("redio_bsf", m, fc1, fc2)  # dsputils.rs:82-88
def bpf(m, fc1, fc2): return _gen("redio_bpf", m, fc1, fc2)  # dsputils.rs:91-94
def lpf_corrected(m, fc): return _gen("redio_lpf_corrected", m, fc)


__all__ = ["convolve", "window", "sinc", "lpf", "hpf", "bsf", "bpf", "lpf_corrected", "RedioError"]
