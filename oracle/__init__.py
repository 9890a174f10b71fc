"""ctypes front-end of the CPU oracle (oracle/_build/libredio_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker -- never by libredio_amd.  "Parity unpinned": see oracle/redio_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# REDIO_ORACLE_SAN=1: the AddressSanitizer + UBSan build (make -C oracle SAN=1; the process needs libasan preloaded: tests/san_check.sh)
_SAN = os.environ.get("REDIO_ORACLE_SAN") == "1"
_BUILD = os.path.join(_HERE, "_build_san" if _SAN else "_build")
_SO = os.path.join(_BUILD, "libredio_oracle.so")
_KPN_SO = os.path.join(_BUILD, "libredio_kpn_baseline.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".cpp"))]
    if (not force and os.path.exists(_SO) and os.path.exists(_KPN_SO) and os.path.getmtime(_KPN_SO) >= os.path.getmtime(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["SAN=1"] if _SAN else []))
    return _SO


_lib = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_c64p = np.ctypeslib.ndpointer(np.complex64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_szp = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


class SrcData(C.Structure):
    _fields_ = [("data_in", C.c_void_p), ("data_out", C.c_void_p),
                ("input_frames", C.c_long), ("output_frames", C.c_long),
                ("input_frames_used", C.c_long), ("output_frames_gen", C.c_long),
                ("end_of_input", C.c_int), ("src_ratio", C.c_double)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    sz = C.c_size_t
    L.orc_convolve_f32.argtypes = [_f32p, sz, _f32p, sz, _f32p]; L.orc_convolve_f32.restype = sz
    L.orc_convolve_f64.argtypes = [_f64p, sz, _f64p, sz, _f64p]; L.orc_convolve_f64.restype = sz
    L.orc_fir_c32.argtypes = [_c64p, sz, _f32p, sz, sz, C.c_int, _c64p]; L.orc_fir_c32.restype = sz
    L.orc_fir_f32.argtypes = [_f32p, sz, _f32p, sz, sz, C.c_int, _f32p]; L.orc_fir_f32.restype = sz
    L.orc_window.argtypes = [sz, _f32p]; L.orc_window.restype = None
    for n in ("orc_sinc", "orc_lpf", "orc_hpf", "orc_lpf_corrected"):
        getattr(L, n).argtypes = [sz, C.c_float, _f32p]; getattr(L, n).restype = C.c_int
    for n in ("orc_bsf", "orc_bpf"):
        getattr(L, n).argtypes = [sz, C.c_float, C.c_float, _f32p]; getattr(L, n).restype = C.c_int
    L.orc_kiss_fft_alloc.argtypes = [C.c_int, C.c_int]; L.orc_kiss_fft_alloc.restype = C.c_void_p
    L.orc_kiss_fft.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; L.orc_kiss_fft.restype = None
    L.orc_kiss_fft_free.argtypes = [C.c_void_p]; L.orc_kiss_fft_free.restype = None
    L.orc_kiss_factors.argtypes = [C.c_int, C.POINTER(C.c_int)]; L.orc_kiss_factors.restype = C.c_int
    L.orc_fft_blocks.argtypes = [C.c_int, C.c_int, _c64p, _c64p, sz]; L.orc_fft_blocks.restype = None
    L.orc_hash32.argtypes = [C.c_uint32, C.c_uint64]; L.orc_hash32.restype = C.c_uint32
    L.orc_synth_iq.argtypes = [C.c_uint32, C.c_uint64, sz, _c64p]; L.orc_synth_iq.restype = None
    L.orc_synth_f32.argtypes = [C.c_uint32, C.c_uint64, sz, _f32p]; L.orc_synth_f32.restype = None
    L.orc_chain_fir_fft.argtypes = [_c64p, sz, _f32p, sz, sz, C.c_int, C.c_int, _c64p]
    L.orc_chain_fir_fft.restype = sz
    L.orc_pfb_channelizer.argtypes = [_c64p, sz, _f32p, C.c_int, C.c_int, C.c_int, _c64p]
    L.orc_pfb_channelizer.restype = sz
    L.orc_overlap_save.argtypes = [_c64p, sz, _f32p, sz, C.c_int, _c64p]; L.orc_overlap_save.restype = sz
    L.orc_src_new.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]; L.orc_src_new.restype = C.c_void_p
    L.orc_src_delete.argtypes = [C.c_void_p]; L.orc_src_delete.restype = None
    L.orc_src_process.argtypes = [C.c_void_p, C.POINTER(SrcData)]; L.orc_src_process.restype = C.c_int
    L.orc_src_reset.argtypes = [C.c_void_p]; L.orc_src_reset.restype = C.c_int
    L.orc_src_set_ratio.argtypes = [C.c_void_p, C.c_double]; L.orc_src_set_ratio.restype = C.c_int
    L.orc_src_table.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_src_table.restype = C.c_int
    L.orc_resample_block.argtypes = [C.c_void_p, _f32p, C.c_long, C.c_double, _f32p, C.c_long]
    L.orc_resample_block.restype = C.c_long
    L.orc_b2d.argtypes = [_szp, sz]; L.orc_b2d.restype = sz
    L.orc_eat.argtypes = [_szp, sz, _szp, sz, _szp]; L.orc_eat.restype = C.c_int
    L.orc_discretize.argtypes = [_f32p, sz, _szp]; L.orc_discretize.restype = None
    L.orc_data_to_samples.argtypes = [_u8p, sz, _c64p]; L.orc_data_to_samples.restype = C.c_int
    L.orc_trigger_new.argtypes = []; L.orc_trigger_new.restype = C.c_void_p
    L.orc_trigger_free.argtypes = [C.c_void_p]; L.orc_trigger_free.restype = None
    L.orc_trigger_feed.argtypes = [C.c_void_p, _f32p, sz, sz, _f32p, sz, _szp, sz, C.POINTER(sz)]
    L.orc_trigger_feed.restype = sz
    L.orc_block_sum.argtypes = [_f32p, sz]; L.orc_block_sum.restype = C.c_float
    L.orc_norm_c32.argtypes = [_c64p, sz, _f32p]; L.orc_norm_c32.restype = None
    L.orc_zip_f32.argtypes = [_f32p, _f32p, sz, C.c_int, _f32p]; L.orc_zip_f32.restype = None
    L.orc_zip_c32.argtypes = [_c64p, _c64p, sz, C.c_int, _c64p]; L.orc_zip_c32.restype = None
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


# ---- A1 -------------------------------------------------------------------------------------
def convolve(u, v):
    """dsputils::convolve (src/dsputils/src/dsputils.rs:30-32). Raises on empty taps (Rust panics)."""
    u = np.asarray(u)
    if u.dtype == np.float64:
        u = np.ascontiguousarray(u); v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.empty(max(len(u) - len(v) + 1, 0) if len(v) else 0, np.float64)
        n = lib().orc_convolve_f64(u, len(u), v, len(v), out)
    else:
        u = _f32(u); v = _f32(v)
        out = np.empty(max(len(u) - len(v) + 1, 0) if len(v) else 0, np.float32)
        n = lib().orc_convolve_f32(u, len(u), v, len(v), out)
    if n == C.c_size_t(-1).value:
        raise ValueError("convolve: empty taps (windows(0) panics in the reference)")
    return out[:n]


def fir(x, taps, decim=1, fused=False):
    """FIR with the fold of dsputils.rs:31, real or interleaved-complex input, keep out[decim*i]."""
    taps = _f32(taps)
    x = np.asarray(x)
    k = len(taps)
    nout = 0 if len(x) < k or k == 0 else (len(x) - k) // decim + 1
    if np.iscomplexobj(x):
        x = _c64(x); out = np.empty(nout, np.complex64)
        n = lib().orc_fir_c32(x, len(x), taps, k, decim, int(fused), out)
    else:
        x = _f32(x); out = np.empty(nout, np.float32)
        n = lib().orc_fir_f32(x, len(x), taps, k, decim, int(fused), out)
    if n == C.c_size_t(-1).value:
        raise ValueError("fir: empty taps or zero decimation")
    return out[:n]


# ---- A2-A4 ----------------------------------------------------------------------------------
def window(m):
    out = np.empty(m + 1, np.float32); lib().orc_window(m, out); return out


def _gen(name, m, *fcs):
    out = np.empty(max(m, 1), np.float32)
    rc = getattr(lib(), name)(m, *[float(f) for f in fcs], out)
    if rc:
        raise ValueError(f"{name}({m}, {fcs}) panics in the reference (fc>=0.5 or m<2)")
    return out[:m]


def sinc(m, fc): return _gen("orc_sinc", m, fc)
def lpf(m, fc): return _gen("orc_lpf", m, fc)
def hpf(m, fc): return _gen("orc_hpf", m, fc)
def bsf(m, fc1, fc2): return _gen("orc_bsf", m, fc1, fc2)
def bpf(m, fc1, fc2): return _gen("orc_bpf", m, fc1, fc2)
def lpf_corrected(m, fc): return _gen("orc_lpf_corrected", m, fc)


# ---- A5 -------------------------------------------------------------------------------------
def kiss_factors(n):
    buf = (C.c_int * 64)()
    ns = lib().orc_kiss_factors(n, buf)
    return [(buf[2 * i], buf[2 * i + 1]) for i in range(ns)]


def fft(x, nfft=None, inverse=False):
    """kissfft::fft block (src/kissfft/src/kissfft.rs:18-31) over consecutive nfft-sized messages."""
    x = _c64(x)
    nfft = nfft or len(x)
    assert len(x) % nfft == 0, "every message must be exactly block_size long (kissfft.rs:24)"
    out = np.empty_like(x)
    lib().orc_fft_blocks(nfft, int(bool(inverse)), x, out, len(x) // nfft)
    return out


# ---- synthetic input --------------------------------------------------------------------------
def synth_iq(seed, first, n):
    out = np.empty(n, np.complex64); lib().orc_synth_iq(seed, first, n, out); return out


def synth_f32(seed, first, n):
    out = np.empty(n, np.float32); lib().orc_synth_f32(seed, first, n, out); return out


def chain_fir_fft(x, taps, decim, nfft, fused=False):
    x = _c64(x); taps = _f32(taps)
    k = len(taps)
    ny = 0 if len(x) < k else (len(x) - k) // decim + 1
    out = np.empty((ny // nfft) * nfft, np.complex64)
    nb = lib().orc_chain_fir_fft(x, len(x), taps, k, decim, nfft, int(fused), out)
    return out[: nb * nfft].reshape(nb, nfft)


def overlap_save(x, h, nfft):
    """convolve semantics (valid-mode correlation) through kissfft blocks of nfft samples."""
    x = _c64(x); h = _f32(h)
    k = len(h)
    hop = nfft - k + 1
    nblk = 0 if len(x) < nfft else (len(x) - nfft) // hop + 1
    out = np.empty(nblk * hop, np.complex64)
    if nblk:
        n = lib().orc_overlap_save(x, len(x), h, k, nfft, out)
        assert n == nblk * hop
    return out


def pfb_channelizer(x, h, M, P, fused=False):
    """M-channel polyphase channelizer: branch FIRs (dsputils fold) + kissfft across branches."""
    x = _c64(x); h = _f32(h)
    assert len(h) == M * P
    T = len(x) // M
    rows = max(T - P + 1, 0)
    out = np.empty((rows, M), np.complex64)
    if rows:
        n = lib().orc_pfb_channelizer(x, len(x), h, M, P, int(fused), out)
        assert n == rows
    return out


# ---- A6 -------------------------------------------------------------------------------------
class Resampler:
    """samplerate::resample state (src/samplerate/src/samplerate.rs:59-87): src_new(1, 1)."""

    def __init__(self, converter=1, channels=1):
        err = C.c_int(0)
        self.channels = int(channels)
        self._s = lib().orc_src_new(converter, channels, C.byref(err))
        if not self._s:
            raise ValueError(f"src_new failed with error {err.value}")

    def __del__(self):
        try:   # module globals may already be gone at interpreter shutdown
            if getattr(self, "_s", None):
                lib().orc_src_delete(self._s); self._s = None
        except Exception:
            pass

    def block(self, vin, ratio):
        vin = _f32(vin)
        cap = int(ratio * len(vin) + 1.0)
        out = np.empty(max(cap, 1), np.float32)
        n = lib().orc_resample_block(self._s, vin, len(vin), float(ratio), out, cap)
        if n < 0:
            raise RuntimeError(f"src_process error {-(n + 1000)}")
        return out[:n].copy()

    def set_ratio(self, ratio):
        """src_set_ratio (samplerate.rs:40): the next call starts at this ratio instead of gliding to it."""
        return lib().orc_src_set_ratio(self._s, float(ratio))

    def process(self, vin, ratio, out_frames, end_of_input=False):
        """vin: interleaved frames (len = frames * channels); returns (error, interleaved output, input FRAMES used)."""
        ch = self.channels
        vin = _f32(vin); out = np.empty(max(out_frames, 1) * ch, np.float32)
        d = SrcData(vin.ctypes.data, out.ctypes.data, len(vin) // ch, out_frames, 0, 0, int(end_of_input), ratio)
        err = lib().orc_src_process(self._s, C.byref(d))
        return err, out[: d.output_frames_gen * ch].copy(), d.input_frames_used


def src_table(converter):
    p = C.c_void_p(); h = C.c_int(); inc = C.c_int()
    rc = lib().orc_src_table(converter, C.byref(p), C.byref(h), C.byref(inc))
    if rc:
        raise ValueError("bad converter")
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(h.value + 2,))
    return arr.copy(), h.value, inc.value


# ---- A9 -------------------------------------------------------------------------------------
def b2d(bits):
    b = np.ascontiguousarray(bits, dtype=np.uint64); return int(lib().orc_b2d(b, len(b)))


def eat(bits, widths):
    b = np.ascontiguousarray(bits, dtype=np.uint64); w = np.ascontiguousarray(widths, dtype=np.uint64)
    out = np.empty(len(w), np.uint64)
    if lib().orc_eat(b, len(b), w, len(w), out):
        raise IndexError("eat: widths overrun the input (slice panic in the reference)")
    return [int(v) for v in out]


def discretize(x):
    x = _f32(x); out = np.empty(len(x), np.uint64); lib().orc_discretize(x, len(x), out); return out


def data_to_samples(d):
    d = np.ascontiguousarray(d, dtype=np.uint8)
    out = np.empty(len(d) // 2, np.complex64)
    if lib().orc_data_to_samples(d, len(d), out):
        raise IndexError("data_to_samples: odd byte count (index panic in the reference)")
    return out


def norm(x):
    """|x| = Complex::norm() = hypotf(re, im) (the map of src/ratpak.rs:64-68)."""
    x = _c64(x); out = np.empty(len(x), np.float32); lib().orc_norm_c32(x, len(x), out); return out


def block_sum(x):
    x = _f32(x); return np.float32(lib().orc_block_sum(x, len(x)))


class Trigger:
    """bitfount::trigger (src/bitfount/src/bitfount.rs:36-85)."""

    def __init__(self):
        self._t = lib().orc_trigger_new()
        self._fed = 0

    def __del__(self, _lib=lib):  # bound at definition: module globals may be gone at interpreter shutdown
        if getattr(self, "_t", None):
            try:
                _lib().orc_trigger_free(self._t)
            except Exception:
                pass
            self._t = None

    def feed(self, blocks):
        blocks = _f32(blocks)
        nb, bl = blocks.shape
        self._fed += nb * bl
        cap = self._fed + nb + 64   # a buffer can carry blocks collected by earlier calls
        out = np.empty(cap, np.float32); lens = np.zeros(nb + 1, np.uint64); tot = C.c_size_t(0)
        ne = lib().orc_trigger_feed(self._t, blocks, nb, bl, out, cap, lens, len(lens), C.byref(tot))
        res, off = [], 0
        for i in range(ne):
            n = int(lens[i]); res.append(out[off:off + n].copy()); off += n
        return res


# ---- run-length blocks (numpy restatements; integer arithmetic with a single possible result) --------
class Rle:
    """kpn::rle, src/kpn/src/kpn.rs:17-29: x = first value, i = 1; for every later y: if y != x emit
    (x, i) and i = 1 else i += 1; x = y.  The open run is never flushed."""

    def __init__(self):
        self.x = None
        self.i = 0

    def feed(self, vals):
        out = []
        for y in np.asarray(vals).tolist():
            if self.x is None:
                self.x, self.i = y, 1
                continue
            if y != self.x:
                out.append((self.x, self.i))
                self.i = 1
            else:
                self.i += 1
            self.x = y
        return out


def dle(runs, s_rate):
    """kpn::dle, kpn.rs:32-38: (x, ct) -> (x, ct as f32 / s_rate as f32)."""
    return [(x, np.float32(np.float32(ct) / np.float32(s_rate))) for x, ct in runs]


def rld(runs):
    """kpn::rld, kpn.rs:50-56."""
    return [x for x, ct in runs for _ in range(int(ct))]


def dld(runs, s_rate):
    """kpn::dld, kpn.rs:41-47: repeat x (dur*s_rate) as usize times (f32 product, truncating cast)."""
    out = []
    for x, dur in runs:
        out += [x] * int(np.float32(np.float32(dur) * np.float32(s_rate)))
    return out


def zip_vecs(a, b, add=False):
    """kpn::mul_vecs / sum_vecs (kpn.rs:198-203, 227-231): elementwise over the shorter length."""
    n = min(len(a), len(b))
    if np.iscomplexobj(a) or np.iscomplexobj(b):
        a, b = _c64(a[:n]), _c64(b[:n]); out = np.empty(n, np.complex64)
        lib().orc_zip_c32(a, b, n, int(add), out)
    else:
        a, b = _f32(a[:n]), _f32(b[:n]); out = np.empty(n, np.float32)
        lib().orc_zip_f32(a, b, n, int(add), out)
    return out


def kpn_chain_baseline(seconds, log2n, seed, taps, decim, nfft):
    """bench.py's cpu_baseline leg in the reference's structure (oracle/kpn_baseline.cpp): one thread per block, one heap
    Vec per message, queue hand-off.  Returns (input samples whose spectra reached the sink, messages, wall seconds)."""
    lib()
    K = C.CDLL(_KPN_SO)
    K.orc_kpn_chain_baseline.restype = C.c_double
    K.orc_kpn_chain_baseline.argtypes = [C.c_double, C.c_int, C.c_uint32, _f32p, C.c_size_t, C.c_size_t, C.c_int,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    t = np.ascontiguousarray(taps, dtype=np.float32)
    done, msgs = C.c_uint64(0), C.c_uint64(0)
    wall = K.orc_kpn_chain_baseline(float(seconds), int(log2n), int(seed), t, len(t), int(decim), int(nfft), C.byref(done), C.byref(msgs))
    return done.value, msgs.value, wall
