"""Device-resident mirrors of the run-length / bit-field blocks of src/kpn/src/kpn.rs that sit
downstream of discretize in the shipped graph (src/ratpak.rs:77-119): rle, dle, rld, dld, binconv.
torch CUDA tensors in and out; values are one byte (discretize emits 0/1), run lengths uint64 (held in
int64 tensors), durations float32.  Bit-exact integer work; no CPU path."""
import ctypes as C

from . import check, lib
from .plans import _dev_ptr, _safe_destroy, current_stream


class Rle:
    """kpn::rle (kpn.rs:17-29): stateful; the open run is carried across calls and never flushed."""

    def __init__(self):
        self._h = C.c_void_p()
        check(lib().redio_rle_create(C.byref(self._h)), "rle_create")

    def feed(self, x):
        import torch
        assert x.dtype == torch.uint8 and x.is_contiguous()
        n = x.numel()
        vals = torch.empty(max(n, 1), dtype=torch.uint8, device=x.device)
        counts = torch.empty(max(n, 1), dtype=torch.int64, device=x.device)
        nr = C.c_size_t(0)
        check(lib().redio_rle_feed(self._h, _dev_ptr(x), n, _dev_ptr(vals), _dev_ptr(counts), max(n, 1), C.byref(nr), current_stream()), "rle_feed")
        return vals[: nr.value], counts[: nr.value]

    def __del__(self, _safe_destroy=_safe_destroy):  # bound at definition: module globals may be gone at shutdown
        if getattr(self, "_h", None):
            _safe_destroy("redio_rle_destroy", self._h)
            self._h = None


def dle(counts, s_rate):
    """kpn::dle (kpn.rs:32-38)."""
    import torch
    out = torch.empty(counts.numel(), dtype=torch.float32, device=counts.device)
    check(lib().redio_dle(_dev_ptr(counts), counts.numel(), int(s_rate), _dev_ptr(out), current_stream()), "dle")
    return out


def rld(vals, counts):
    """kpn::rld (kpn.rs:50-56)."""
    import torch
    n = vals.numel()
    total = int(counts.sum().item()) if n else 0
    out = torch.empty(max(total, 1), dtype=torch.uint8, device=vals.device)
    scratch = torch.empty(n + 1, dtype=torch.int64, device=vals.device)
    no = C.c_size_t(0)
    check(lib().redio_rld(_dev_ptr(vals), _dev_ptr(counts), n, _dev_ptr(out), max(total, 1), _dev_ptr(scratch), C.byref(no), current_stream()), "rld")
    return out[: no.value]


def dld(vals, seconds, s_rate, cap):
    """kpn::dld (kpn.rs:41-47): n = (dur*s_rate) as usize per run; cap bounds the output length."""
    import torch
    n = vals.numel()
    out = torch.empty(max(cap, 1), dtype=torch.uint8, device=vals.device)
    scratch = torch.empty(2 * n + 1, dtype=torch.int64, device=vals.device)
    no = C.c_size_t(0)
    check(lib().redio_dld(_dev_ptr(vals), _dev_ptr(seconds), n, float(s_rate), _dev_ptr(out), max(cap, 1), _dev_ptr(scratch), C.byref(no),
                          current_stream()), "dld")
    return out[: no.value]


def binconv(bits, widths):
    """kpn::binconv (kpn.rs:295-299): bits is uint8 [nmsg, nbits] of binary digits; returns int64 [nmsg, len(widths)]."""
    import torch
    assert bits.dtype == torch.uint8 and bits.dim() == 2 and bits.is_contiguous()
    nmsg, nbits = bits.shape
    w = (C.c_size_t * len(widths))(*widths)
    out = torch.empty((nmsg, len(widths)), dtype=torch.int64, device=bits.device)
    check(lib().redio_binconv(_dev_ptr(bits), nmsg, nbits, w, len(widths), _dev_ptr(out), current_stream()), "binconv")
    return out


def _zip(name_f32, name_c32, x, c):
    import torch
    assert x.dtype == c.dtype and x.dtype in (torch.float32, torch.complex64) and x.dim() == 1 and c.dim() == 1
    assert x.is_contiguous() and c.is_contiguous()
    n = min(x.numel(), c.numel())
    out = torch.empty(n, dtype=x.dtype, device=x.device)
    f = getattr(lib(), name_f32 if x.dtype == torch.float32 else name_c32)
    check(f(_dev_ptr(x), _dev_ptr(c), _dev_ptr(out), n, current_stream()), name_f32)
    return out


def mul_vecs(x, c):
    """kpn::mul_vecs (kpn.rs:198-203) on device tensors: x[i] * c[i] over the shorter length."""
    return _zip("redio_mul_f32", "redio_mul_c32", x, c)


def sum_vecs(x, c):
    """kpn::sum_vecs (kpn.rs:227-231) on device tensors: x[i] + c[i] over the shorter length."""
    return _zip("redio_add_f32", "redio_add_c32", x, c)


def checksum_u32(x, acc=None):
    """The checking sink of a device-resident graph (redio_checksum_u32): the order-free 64-bit sum of every 32-bit word of x, ADDED to
    acc (an int64 device tensor of one element; created zeroed when None).  Returns acc."""
    import torch
    assert x.is_contiguous() and (x.numel() * x.element_size()) % 4 == 0
    if acc is None:
        acc = torch.zeros(1, dtype=torch.int64, device=x.device)
    check(lib().redio_checksum_u32(_dev_ptr(x), x.numel() * x.element_size() // 4, _dev_ptr(acc), current_stream()), "checksum_u32")
    return acc
