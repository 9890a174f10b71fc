"""Device-resident plans of include/redio.h driven from torch tensors (torch = device memory and
streams only).  Inputs/outputs are torch CUDA tensors; kernels are enqueued on torch's current
stream, nothing synchronises."""
import ctypes as C

import numpy as np

from . import REDIO_FIR_COMPLEX, REDIO_FIR_FUSED, check, lib

_pf = C.POINTER(C.c_float)


def _safe_destroy(fn, h):
    """Plan destructors also run at interpreter shutdown, when module globals may already be gone."""
    try:
        getattr(lib(), fn)(h)
    except Exception:
        pass


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_ptr(t):
    assert t.is_cuda and t.is_contiguous(), "device-resident, contiguous tensors only"
    return C.c_void_p(t.data_ptr())


def _taps(taps):
    t = np.ascontiguousarray(taps, dtype=np.float32)
    return t, t.ctypes.data_as(_pf)


class Fir:
    """redio_fir_*: valid-mode FIR with the fold order of dsputils::convolve (dsputils.rs:30-32),
    optional decimation, real (float32) or interleaved complex (complex64) streams."""

    def __init__(self, taps, decim=1, complex_input=True, fused=False):
        t, p = _taps(taps)
        self.ntaps, self.decim, self.complex_input = len(t), int(decim), complex_input
        flags = (REDIO_FIR_COMPLEX if complex_input else 0) | (REDIO_FIR_FUSED if fused else 0)
        self._h = C.c_void_p()
        check(lib().redio_fir_create(C.byref(self._h), p, len(t), self.decim, flags), "fir_create")

    def nout(self, n_in):
        return lib().redio_fir_nout(self._h, n_in)

    def __call__(self, x, out=None):
        import torch
        want = torch.complex64 if self.complex_input else torch.float32
        assert x.dtype == want, f"expected {want}"
        n = self.nout(x.numel())
        if out is None:
            out = torch.empty(n, dtype=want, device=x.device)
        assert out.numel() >= n
        check(lib().redio_fir_enqueue(self._h, _dev_ptr(x), x.numel(), _dev_ptr(out), current_stream()), "fir_enqueue")
        return out[:n]

    def __del__(self, _safe_destroy=_safe_destroy):  # bound at definition: module globals may be gone at shutdown
        if getattr(self, "_h", None):
            _safe_destroy("redio_fir_destroy", self._h)
            self._h = None


class Fft:
    """redio_fft_*: batched kissfft::fft blocks (kissfft.rs:18-31), unnormalised."""

    def __init__(self, nfft, inverse=False):
        self.nfft = int(nfft)
        self._h = C.c_void_p()
        check(lib().redio_fft_create(C.byref(self._h), self.nfft, int(bool(inverse))), "fft_create")

    def __call__(self, x, out=None):
        import torch
        assert x.dtype == torch.complex64 and x.numel() % self.nfft == 0, "messages of exactly nfft samples"
        if out is None:
            out = torch.empty_like(x)
        check(lib().redio_fft_enqueue(self._h, _dev_ptr(x), _dev_ptr(out), x.numel() // self.nfft, current_stream()), "fft_enqueue")
        return out

    def strided(self, x, nbatch, in_stride, out=None):
        """redio_fft_enqueue_strided: block b is x[b*in_stride : b*in_stride + nfft] (overlapping when in_stride < nfft,
        the overlap-save framing); the outputs are packed."""
        import torch
        assert x.dtype == torch.complex64 and in_stride > 0 and (nbatch == 0 or (nbatch - 1) * in_stride + self.nfft <= x.numel())
        if out is None:
            out = torch.empty(nbatch * self.nfft, dtype=torch.complex64, device=x.device)
        check(lib().redio_fft_enqueue_strided(self._h, _dev_ptr(x), _dev_ptr(out), nbatch, in_stride, current_stream()), "fft_enqueue_strided")
        return out

    def __del__(self, _safe_destroy=_safe_destroy):  # bound at definition: module globals may be gone at shutdown
        if getattr(self, "_h", None):
            _safe_destroy("redio_fft_destroy", self._h)
            self._h = None


class Chain:
    """redio_chain_*: FIR(ntaps, decimate) -> nfft-point forward FFT over consecutive blocks."""

    def __init__(self, taps, decim, nfft, fused=True):
        t, p = _taps(taps)
        self.ntaps, self.decim, self.nfft = len(t), int(decim), int(nfft)
        self._h = C.c_void_p()
        check(lib().redio_chain_create(C.byref(self._h), p, len(t), self.decim, self.nfft,
                                       REDIO_FIR_FUSED if fused else 0), "chain_create")

    def nblocks(self, n_in):
        return lib().redio_chain_nblocks(self._h, n_in)

    @property
    def is_fused(self):
        return bool(lib().redio_chain_is_fused(self._h))

    def set_unfused(self, unfused):
        check(lib().redio_chain_set_unfused(self._h, int(bool(unfused))), "chain_set_unfused")

    def reserve(self, n_in):
        """Size the two-kernel path's intermediate up front (enqueue then never allocates)."""
        check(lib().redio_chain_reserve(self._h, int(n_in)), "chain_reserve")

    def blocks_per_wave(self, nblocks):
        """redio_chain_blocks_per_wave: consecutive blocks one wavefront of the fused launch over `nblocks` blocks owns (0: two kernels)."""
        return lib().redio_chain_blocks_per_wave(self._h, int(nblocks))

    def launch_waves(self, nblocks):
        """redio_chain_launch_waves: wavefronts of the fused launch over `nblocks` blocks (0: two kernels)."""
        return lib().redio_chain_launch_waves(self._h, int(nblocks))

    @property
    def kernel_name(self):
        """redio_chain_kernel_name: the fused kernel as rocprofv3 names it, spaces removed (None for a two-kernel plan)."""
        n = lib().redio_chain_kernel_name(self._h)
        return n.decode() if n else None

    def set_debug_stamps(self, buf):
        """Diagnostic per-wave stamps of this plan's fused launches into `buf` (int64 CUDA tensor, 4 per wave: size it with
        4 * launch_waves(nblocks); wavefronts beyond its capacity leave no stamp); None = off."""
        if buf is None:
            check(lib().redio_chain_set_debug_stamps(self._h, None, 0), "chain_set_debug_stamps")
        else:
            check(lib().redio_chain_set_debug_stamps(self._h, C.c_void_p(buf.data_ptr()), buf.numel() // 4), "chain_set_debug_stamps")

    def __call__(self, x, out=None):
        import torch
        assert x.dtype == torch.complex64
        nb = self.nblocks(x.numel())
        if out is None:
            out = torch.empty((nb, self.nfft), dtype=torch.complex64, device=x.device)
        assert out.numel() >= nb * self.nfft
        check(lib().redio_chain_enqueue(self._h, _dev_ptr(x), x.numel(), _dev_ptr(out), current_stream()), "chain_enqueue")
        return out

    def reserve_u8(self, nbytes):
        """Size what from_bytes needs beyond the one-kernel form (other shapes, unaligned bytes) for messages of up to nbytes bytes."""
        check(lib().redio_chain_reserve_u8(self._h, int(nbytes)), "chain_reserve_u8")

    def from_bytes(self, raw, out=None):
        """redio_chain_enqueue_u8: the receiver's interleaved u8 I/Q bytes (rtlsdr::data_to_samples, rtlsdr.rs:159-162)
        straight into the chain; the spectra of bitfount.data_to_samples(raw) followed by this plan, bit for bit."""
        import torch
        assert raw.dtype == torch.uint8 and raw.numel() % 2 == 0
        nb = self.nblocks(raw.numel() // 2)
        if out is None:
            out = torch.empty((nb, self.nfft), dtype=torch.complex64, device=raw.device)
        assert out.numel() >= nb * self.nfft
        check(lib().redio_chain_enqueue_u8(self._h, _dev_ptr(raw), raw.numel(), _dev_ptr(out), current_stream()), "chain_enqueue_u8")
        return out

    def __del__(self, _safe_destroy=_safe_destroy):  # bound at definition: module globals may be gone at shutdown
        if getattr(self, "_h", None):
            _safe_destroy("redio_chain_destroy", self._h)
            self._h = None


class Stream:
    """redio_{fir,chain,pfb,ovsave}_stream_*: a plan fed as a STREAM with the history carried on the device, so that any
    segmentation of the input gives the bits of one stateless call on the whole stream (the stateless plans keep the
    reference's per-message semantics, dsputils.rs:30-32).  `plan` is a Fir, Chain, Channelizer or OverlapSave."""

    def __init__(self, plan, u8=False):
        """u8=True (Chain and Channelizer): the stream arrives as the receiver's interleaved u8 I/Q bytes (uint8 tensors, two
        bytes per sample; redio_{chain,pfb}_stream_create_u8)."""
        kind = {Fir: "fir", Chain: "chain"}.get(type(plan)) or {"Channelizer": "pfb", "OverlapSave": "ovsave"}[type(plan).__name__]
        assert not u8 or kind in ("chain", "pfb")
        self._kind, self._plan, self._u8 = kind, plan, bool(u8)          # the plan must outlive the stream handle
        self._h = C.c_void_p()
        check(getattr(lib(), f"redio_{kind}_stream_create" + ("_u8" if u8 else ""))(C.byref(self._h), plan._h), f"{kind}_stream_create")

    def _f(self, name):
        return getattr(lib(), f"redio_{self._kind}_stream_{name}")

    def nout(self, n_new):
        return self._f("nout")(self._h, int(n_new))

    @property
    def pending(self):
        return self._f("pending")(self._h)

    def reset(self):
        check(self._f("reset")(self._h), "stream_reset")

    def __call__(self, x, out=None):
        """Feed the next piece of the stream; returns the output samples that became computable (flat tensor; the
        chain's are whole spectra of nfft samples, the channelizer's whole rows of nchan samples)."""
        import torch
        real = self._kind == "fir" and not self._plan.complex_input
        want = torch.float32 if real else torch.complex64
        if self._u8:
            assert x.dtype == torch.uint8 and x.numel() % 2 == 0, "expected an even number of uint8 bytes"
            nsamp = x.numel() // 2
        else:
            assert x.dtype == want, f"expected {want}"
            nsamp = x.numel()
        n = self.nout(nsamp)
        if out is None:
            out = torch.empty(max(n, 1), dtype=want, device=x.device)
        assert out.numel() >= n
        got = C.c_size_t(0)
        check(self._f("enqueue")(self._h, _dev_ptr(x) if x.numel() else None, nsamp, _dev_ptr(out), C.byref(got), current_stream()),
              f"{self._kind}_stream_enqueue")
        assert got.value == n
        return out[:n]

    def __del__(self, _safe_destroy=_safe_destroy):
        if getattr(self, "_h", None):
            _safe_destroy(f"redio_{self._kind}_stream_destroy", self._h)
            self._h = None


class Src:
    """redio_src_*: nchan independent mono streams through the libsamplerate-style sinc converter
    (samplerate.rs:59-87 semantics per stream), device-resident rows [nchan][frames]."""

    EXACT, FAST, EPOCHS = 0, 1, 2

    def __init__(self, nchan, converter=1, mode=0):
        self.nchan = int(nchan)
        self._h = C.c_void_p()
        check(lib().redio_src_create(C.byref(self._h), int(converter), self.nchan), "src_create")
        if mode:
            self.set_mode(mode)

    def set_mode(self, mode):
        """EXACT (bit-identical, default) / FAST (f32 polyphase for uniform-phase calls) / EPOCHS
        (EXACT, one launch per buffer refill)."""
        check(lib().redio_src_set_mode(self._h, int(mode)), "src_set_mode")

    def process(self, x, ratio, output_frames=None, end_of_input=False):
        """x: float32 CUDA tensor [nchan, frames]. Returns (out[nchan, gen], input_frames_used)."""
        import torch
        assert x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] == self.nchan and x.is_contiguous()
        frames = x.shape[1]
        cap = int(ratio * frames + 1.0) if output_frames is None else int(output_frames)
        out = torch.empty((self.nchan, max(cap, 1)), dtype=torch.float32, device=x.device)
        used, gen = C.c_long(0), C.c_long(0)
        rc = lib().redio_src_process(self._h, _dev_ptr(x), frames, x.stride(0), _dev_ptr(out), cap, out.stride(0),
                                     float(ratio), int(bool(end_of_input)), C.byref(used), C.byref(gen), current_stream())
        check(rc, "src_process")
        return out[:, : gen.value], used.value

    def reset(self):
        check(lib().redio_src_reset(self._h), "src_reset")

    def path_counts(self):
        """(epochs through the periodic-phase kernel, epochs through the general per-tap kernel) so far."""
        a, b = C.c_long(0), C.c_long(0)
        check(lib().redio_src_path_counts(self._h, C.byref(a), C.byref(b)), "src_path_counts")
        return a.value, b.value

    def __del__(self, _safe_destroy=_safe_destroy):  # bound at definition: module globals may be gone at shutdown
        if getattr(self, "_h", None):
            _safe_destroy("redio_src_destroy", self._h)
            self._h = None


class Graph:
    """redio_graph_*: record the enqueue calls issued inside the `with` block on a side stream and replay
    them with one submission (launch-bound pipelines of many small messages).  Allocate every output
    before entering and run the sequence once un-captured first.

        g = Graph()
        with g:                      # torch's current stream is the capture stream inside the block
            for i in range(64):
                fir(x[i], out=y[i]); fft(y[i], out=z[i])
        g.launch(); g.launch()
    """

    def __init__(self):
        import torch
        self.stream = torch.cuda.Stream()
        self._g = C.c_void_p()
        self._ctx = None

    def __enter__(self):
        import torch
        self.stream.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        check(lib().redio_graph_begin(C.c_void_p(self.stream.cuda_stream)), "graph_begin")
        return self

    def __exit__(self, et, ev, tb):
        rc = lib().redio_graph_end(C.c_void_p(self.stream.cuda_stream), C.byref(self._g))
        self._ctx.__exit__(et, ev, tb)
        self._ctx = None
        if et is None:
            check(rc, "graph_end")
        return False

    def launch(self, stream=None):
        """Replay on `stream` (default: torch's current stream)."""
        check(lib().redio_graph_launch(self._g, current_stream() if stream is None else C.c_void_p(stream.cuda_stream)), "graph_launch")

    def __del__(self, _safe_destroy=_safe_destroy):
        if getattr(self, "_g", None):
            _safe_destroy("redio_graph_destroy", self._g)
            self._g = None


def synth_iq(seed, first, n, device="cuda"):
    """Hash-generated cf32 IQ in [-1,1), generated on the device."""
    import torch
    out = torch.empty(n, dtype=torch.complex64, device=device)
    check(lib().redio_synth_iq(_dev_ptr(out), seed, first, n, current_stream()), "synth_iq")
    return out


def synth_f32(seed, first, n, device="cuda"):
    import torch
    out = torch.empty(n, dtype=torch.float32, device=device)
    check(lib().redio_synth_f32(_dev_ptr(out), seed, first, n, current_stream()), "synth_f32")
    return out


class Channelizer:
    """redio_pfb_*: polyphase channelizer (BASELINE.json configs[3]): branch FIRs with the dsputils fold + kissfft
    across branches.  64 channels with 4 / 8 / 16 taps per branch run one fused kernel; any other shape runs the
    branch filters and the plan's transform as two passes.  out[row][channel], or [group][row][nchan/ngroups] for
    the multi-GPU regrouping (see channelizer_all_to_all)."""

    def __init__(self, proto, nchan=64, taps_per_branch=16, fused=True):
        t, p = _taps(proto)
        assert len(t) == nchan * taps_per_branch
        self.nchan, self.taps_per_branch = int(nchan), int(taps_per_branch)
        self._h = C.c_void_p()
        check(lib().redio_pfb_create(C.byref(self._h), p, self.nchan, self.taps_per_branch,
                                     REDIO_FIR_FUSED if fused else 0), "pfb_create")

    def nrows(self, n_in):
        return lib().redio_pfb_nrows(self._h, n_in)

    def __call__(self, x, ngroups=1, out=None):
        import torch
        assert x.dtype == torch.complex64
        rows = self.nrows(x.numel())
        if out is None:
            shape = (rows, self.nchan) if ngroups == 1 else (ngroups, rows, self.nchan // ngroups)
            out = torch.empty(shape, dtype=torch.complex64, device=x.device)
        check(lib().redio_pfb_enqueue(self._h, _dev_ptr(x), x.numel(), _dev_ptr(out), int(ngroups), current_stream()), "pfb_enqueue")
        return out

    def reserve(self, n_in, ngroups=1, two_pass=False):
        """redio_pfb_reserve / redio_pfb_reserve_two_pass: plan scratch for inputs of up to n_in samples (so that a call never allocates,
        e.g. inside a Graph); two_pass: also for the one-kernel shapes, whose fall-back for an output that is not 16-byte aligned needs it."""
        f = lib().redio_pfb_reserve_two_pass if two_pass else lib().redio_pfb_reserve
        check(f(self._h, int(n_in), int(ngroups)), "pfb_reserve")

    def from_bytes(self, raw, ngroups=1, out=None):
        """redio_pfb_enqueue_u8: the receiver's interleaved u8 I/Q bytes (rtlsdr::data_to_samples, rtlsdr.rs:159-162) straight
        into the channelizer; the rows of bitfount.data_to_samples(raw) followed by this plan, bit for bit."""
        import torch
        assert raw.dtype == torch.uint8 and raw.numel() % 2 == 0
        rows = self.nrows(raw.numel() // 2)
        if out is None:
            shape = (rows, self.nchan) if ngroups == 1 else (ngroups, rows, self.nchan // ngroups)
            out = torch.empty(shape, dtype=torch.complex64, device=raw.device)
        check(lib().redio_pfb_enqueue_u8(self._h, _dev_ptr(raw), raw.numel(), _dev_ptr(out), int(ngroups), current_stream()), "pfb_enqueue_u8")
        return out

    def __del__(self, _safe_destroy=_safe_destroy):  # bound at definition: module globals may be gone at shutdown
        if getattr(self, "_h", None):
            _safe_destroy("redio_pfb_destroy", self._h)
            self._h = None


def channelizer_all_to_all(grouped, group=None):
    """The one exchange step of the time-sharded channelizer: every rank holds
    grouped[g] = [its rows][channels of rank g]; after the all-to-all rank g holds
    [all rows, in rank (= time) order][its channels].  Rows per rank may differ (ragged split sizes).
    Works with any torch.distributed backend (nccl = RCCL over xGMI on the GPU box, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    assert grouped.shape[0] == world
    my_rows, cpg = grouped.shape[1], grouped.shape[2]
    rows = [torch.zeros(1, dtype=torch.int64, device=grouped.device) for _ in range(world)]
    dist.all_gather(rows, torch.tensor([my_rows], dtype=torch.int64, device=grouped.device), group=group)
    rows = [int(r) for r in rows]
    send = torch.view_as_real(grouped.contiguous()).reshape(world * my_rows, cpg * 2)
    recv = torch.empty((sum(rows), cpg * 2), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(recv, send, output_split_sizes=rows, input_split_sizes=[my_rows] * world, group=group)
    return torch.view_as_complex(recv.reshape(sum(rows), cpg, 2))


class Comm:
    """redio_comm_* / redio_pfb_exchange: the channelizer's regrouping step through the C ABI (RCCL ncclSend/ncclRecv
    inside one group) -- what a Rust or C++ kpn host calls.  channelizer_all_to_all above is the torch.distributed
    twin (and the only form the gloo CPU tests can run)."""

    def __init__(self, handle, rank, size):
        self._h, self.rank, self.size = handle, rank, size

    @classmethod
    def from_torch_distributed(cls, group=None):
        """One rank per process under torch.distributed.run: rank 0 makes the RCCL id, the process group carries the
        128 bytes to the others (any backend), every rank joins on its current device."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = C.create_string_buffer(128)
        if rank == 0:
            check(lib().redio_comm_unique_id(buf), "comm_unique_id")
        box = [buf.raw]
        dist.broadcast_object_list(box, src=0, group=group)
        h = C.c_void_p()
        check(lib().redio_comm_init_rank(C.byref(h), world, rank, C.create_string_buffer(box[0], 128)), "comm_init_rank")
        return cls(h, rank, world)

    @classmethod
    def single(cls):
        """A communicator of one rank on the current device (1-GPU boxes; the exchange degenerates to a copy to self)."""
        buf = C.create_string_buffer(128)
        check(lib().redio_comm_unique_id(buf), "comm_unique_id")
        h = C.c_void_p()
        check(lib().redio_comm_init_rank(C.byref(h), 1, 0, buf), "comm_init_rank")
        return cls(h, 0, 1)

    @classmethod
    def init_all(cls, devices=None):
        """Every visible device (or `devices`) as ranks of ONE process: returns the list of communicators."""
        import torch
        devs = list(range(torch.cuda.device_count())) if devices is None else list(devices)
        hs = (C.c_void_p * len(devs))()
        check(lib().redio_comm_init_all(hs, len(devs), (C.c_int * len(devs))(*devs)), "comm_init_all")
        return [cls(C.c_void_p(hs[i]), i, len(devs)) for i in range(len(devs))]

    def exchange(self, grouped, rows_per_rank, out=None):
        """grouped: [size][my rows][channels per rank] complex64 on this rank's device -> [sum(rows)][channels per rank]."""
        import torch
        assert grouped.dtype == torch.complex64 and grouped.dim() == 3 and grouped.shape[0] == self.size
        rows = [int(r) for r in rows_per_rank]
        assert len(rows) == self.size and rows[self.rank] == grouped.shape[1]
        cpg = grouped.shape[2]
        if out is None:
            out = torch.empty((sum(rows), cpg), dtype=torch.complex64, device=grouped.device)
        check(lib().redio_pfb_exchange(self._h, _dev_ptr(grouped), _dev_ptr(out), (C.c_size_t * self.size)(*rows), cpg, current_stream()),
              "pfb_exchange")
        return out

    def exchange_at(self, grouped, rows_per_rank, out, out_row_offset, stream=None):
        """redio_pfb_exchange_at: rank q's rows land at row out_row_offset[q] of `out`; enqueued on `stream` (a torch stream;
        default: the current one) -- the piece-wise form that overlaps the exchange with the next piece's analysis."""
        rows = [int(r) for r in rows_per_rank]
        offs = [int(o) for o in out_row_offset]
        st = current_stream() if stream is None else C.c_void_p(stream.cuda_stream)
        check(lib().redio_pfb_exchange_at(self._h, _dev_ptr(grouped), _dev_ptr(out), (C.c_size_t * self.size)(*rows),
                                          (C.c_size_t * self.size)(*offs), grouped.shape[2], st), "pfb_exchange_at")
        return out

    def __del__(self, _safe_destroy=_safe_destroy):
        if getattr(self, "_h", None):
            _safe_destroy("redio_comm_destroy", self._h)
            self._h = None


def exchange_all(comms, grouped, rows_per_rank, outs=None):
    """redio_pfb_exchange_all: all ranks of one process (Comm.init_all) in one RCCL group.  grouped[g] lives on device g."""
    import torch
    n = len(comms)
    rows = [int(r) for r in rows_per_rank]
    cpg = grouped[0].shape[2]
    if outs is None:
        outs = [torch.empty((sum(rows), cpg), dtype=torch.complex64, device=grouped[g].device) for g in range(n)]
    streams = [C.c_void_p(torch.cuda.current_stream(grouped[g].device).cuda_stream) for g in range(n)]
    check(lib().redio_pfb_exchange_all((C.c_void_p * n)(*[c._h for c in comms]), n, (C.c_void_p * n)(*[_dev_ptr(t) for t in grouped]),
                                       (C.c_void_p * n)(*[_dev_ptr(t) for t in outs]), (C.c_size_t * n)(*rows), cpg,
                                       (C.c_void_p * n)(*streams)), "pfb_exchange_all")
    return outs


class OverlapSave:
    """redio_ovsave_*: overlap-save FFT convolution (BASELINE.json configs[4]) with the semantics of
    dsputils::convolve (valid-mode correlation, dsputils.rs:30-32) on complex64 streams."""

    def __init__(self, taps, nfft=65536):
        t, p = _taps(taps)
        self.ntaps, self.nfft = len(t), int(nfft)
        self._h = C.c_void_p()
        check(lib().redio_ovsave_create(C.byref(self._h), p, len(t), self.nfft), "ovsave_create")

    def nout(self, n_in):
        return lib().redio_ovsave_nout(self._h, n_in)

    def __call__(self, x, out=None):
        import torch
        assert x.dtype == torch.complex64
        n = self.nout(x.numel())
        if out is None:
            out = torch.empty(n, dtype=torch.complex64, device=x.device)
        check(lib().redio_ovsave_enqueue(self._h, _dev_ptr(x), x.numel(), _dev_ptr(out), current_stream()), "ovsave_enqueue")
        return out[:n]

    def __del__(self, _safe_destroy=_safe_destroy):  # bound at definition: module globals may be gone at shutdown
        if getattr(self, "_h", None):
            _safe_destroy("redio_ovsave_destroy", self._h)
            self._h = None
