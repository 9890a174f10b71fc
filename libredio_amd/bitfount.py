"""Mirror of the ingest / slicing blocks of the one graph the reference ships (src/ratpak.rs:60-76):
rtlsdr::data_to_samples (src/rtlsdr/src/rtlsdr.rs:159-162), the |x| map (ratpak.rs:64-68),
bitfount::trigger (src/bitfount/src/bitfount.rs:36-85) and bitfount::discretize (:87-96), on torch CUDA
tensors.  Bit-exact with the reference arithmetic; no CPU path."""
import ctypes as C

from . import check, lib
from .plans import _dev_ptr, current_stream


def data_to_samples(data, out=None):
    """uint8 CUDA tensor of IQ byte pairs -> complex64 (i as f32/127.0 - 1.0). Odd length raises."""
    import torch
    assert data.dtype == torch.uint8
    if out is None:
        out = torch.empty(data.numel() // 2, dtype=torch.complex64, device=data.device)
    assert out.dtype == torch.complex64 and out.numel() >= data.numel() // 2
    check(lib().redio_data_to_samples(_dev_ptr(data), data.numel(), _dev_ptr(out), current_stream()), "data_to_samples")
    return out


def norm(x):
    """|x| = Complex::norm() = hypotf(re, im)."""
    import torch
    assert x.dtype == torch.complex64
    out = torch.empty(x.numel(), dtype=torch.float32, device=x.device)
    check(lib().redio_norm_c32(_dev_ptr(x), x.numel(), _dev_ptr(out), current_stream()), "norm")
    return out


def ingest_mag(data):
    """data_to_samples + norm fused: uint8 IQ -> float32 magnitude."""
    import torch
    assert data.dtype == torch.uint8
    out = torch.empty(data.numel() // 2, dtype=torch.float32, device=data.device)
    check(lib().redio_ingest_u8_mag(_dev_ptr(data), data.numel(), _dev_ptr(out), current_stream()), "ingest_u8_mag")
    return out


def block_sums(x, block=512):
    import torch
    nb = x.numel() // block
    out = torch.empty(nb, dtype=torch.float32, device=x.device)
    check(lib().redio_block_sums(_dev_ptr(x), nb, block, _dev_ptr(out), current_stream()), "block_sums")
    return out


def discretize(sample_buffer):
    """bitfount::discretize on one buffer: uint8 tensor of 0/1 (the reference sends usize per sample)."""
    import torch
    assert sample_buffer.dtype == torch.float32
    out = torch.empty(sample_buffer.numel(), dtype=torch.uint8, device=sample_buffer.device)
    scratch = torch.zeros(1, dtype=torch.int32, device=sample_buffer.device)
    check(lib().redio_discretize(_dev_ptr(sample_buffer), sample_buffer.numel(), _dev_ptr(out), _dev_ptr(scratch), current_stream()), "discretize")
    return out


class Trigger:
    """bitfount::trigger: energy-gated block collector with persistent state."""

    def __init__(self):
        self._h = C.c_void_p()
        check(lib().redio_trigger_create(C.byref(self._h)), "trigger_create")

    def feed(self, blocks):
        """blocks: float32 CUDA tensor [nblocks, block]. Returns the list of emitted buffers."""
        import torch
        assert blocks.dtype == torch.float32 and blocks.dim() == 2 and blocks.is_contiguous()
        nb, bl = blocks.shape
        cap = 1000 * 50 * 512 + nb * bl + 1024  # a buffer can carry blocks collected by earlier calls
        out = torch.empty(cap, dtype=torch.float32, device=blocks.device)
        lens = (C.c_size_t * (nb + 1))()
        ne, tot = C.c_size_t(0), C.c_size_t(0)
        check(lib().redio_trigger_feed(self._h, _dev_ptr(blocks), nb, bl, _dev_ptr(out), cap, lens, nb + 1,
                                       C.byref(ne), C.byref(tot), current_stream()), "trigger_feed")
        res, off = [], 0
        for i in range(ne.value):
            res.append(out[off:off + lens[i]].clone())
            off += lens[i]
        return res

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().redio_trigger_destroy(self._h)
            except Exception:
                pass
            self._h = None
