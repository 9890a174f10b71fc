"""libredio_amd -- host-side mirror of LibRedio's DSP-block interface over libredio.so (gfx950).

The product is the C-ABI library built from libredio_amd/csrc (include/redio.h, include/kiss_fft.h).
This package is the thin Python host layer used by tests and bench.py: it loads the library with
ctypes and mirrors the reference's names -- dsputils.convolve / window / sinc / lpf / hpf / bsf / bpf
(src/dsputils/src/dsputils.rs), kissfft.fft (src/kissfft/src/kissfft.rs:18-31) -- plus the
device-resident plans.  There is NO CPU fallback: if the library is missing the import of `lib()`
raises, and every compute entry point needs a HIP device.  Nothing here imports oracle/.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# REDIO_BUILD_DIR: measurement tools only (tools/ablate.sh loads timing-only experimental builds from another directory)
_BUILD = os.environ.get("REDIO_BUILD_DIR") or os.path.join(_HERE, "_build")
LIBREDIO = os.path.join(_BUILD, "libredio.so")
LIBKISSFFT = os.path.join(_BUILD, "libkissfft.so")
LIBSAMPLERATE = os.path.join(_BUILD, "libsamplerate.so")

REDIO_FIR_COMPLEX = 1
REDIO_FIR_FUSED = 2


class RedioError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib().redio_strerror(code).decode() if _lib is not None else str(code)
        super().__init__(f"{what}: redio error {code}: {msg}")


def build(verbose=False):
    """Compile every HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return LIBREDIO


_lib = None
_kiss = None
_src = None


def _sig(f, res, *args):
    f.restype = res
    f.argtypes = list(args)


def lib():
    """The loaded libredio.so (raises if it was not built: no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBREDIO):
        raise ImportError(f"{LIBREDIO} is missing: run libredio_amd.build() / make -C libredio_amd/csrc")
    L = C.CDLL(LIBREDIO, mode=C.RTLD_GLOBAL)
    vp, sz, i, u, f = C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_float
    pf = C.POINTER(C.c_float)
    _sig(L.redio_strerror, C.c_char_p, i)
    _sig(L.redio_version, C.c_char_p)
    _sig(L.redio_device_count, i, C.POINTER(i))
    _sig(L.redio_set_device, i, i)
    _sig(L.redio_get_device, i, C.POINTER(i))
    _sig(L.redio_malloc, i, C.POINTER(vp), sz)
    _sig(L.redio_free, i, vp)
    _sig(L.redio_upload, i, vp, vp, sz, vp)
    _sig(L.redio_download, i, vp, vp, sz, vp)
    _sig(L.redio_copy, i, vp, vp, sz, vp)
    _sig(L.redio_stream_create, i, C.POINTER(vp))
    _sig(L.redio_stream_destroy, i, vp)
    _sig(L.redio_stream_sync, i, vp)
    _sig(L.redio_stream_signal, i, vp, vp, C.c_uint32)
    _sig(L.redio_event_create, i, C.POINTER(vp))
    _sig(L.redio_event_destroy, i, vp)
    _sig(L.redio_event_record, i, vp, vp)
    _sig(L.redio_event_elapsed_ms, i, vp, vp, C.POINTER(f))
    _sig(L.redio_event_create_sync, i, C.POINTER(vp))
    _sig(L.redio_stream_wait_event, i, vp, vp)
    _sig(L.redio_event_sync, i, vp)
    _sig(L.redio_malloc_count, C.c_ulonglong)
    _sig(L.redio_checksum_u32, i, vp, sz, vp, vp)
    _sig(L.redio_window, i, sz, pf)
    for n in ("redio_sinc", "redio_lpf", "redio_hpf", "redio_lpf_corrected"):
        _sig(getattr(L, n), i, sz, f, pf)
    for n in ("redio_bsf", "redio_bpf"):
        _sig(getattr(L, n), i, sz, f, f, pf)
    _sig(L.redio_convolve_f32, i, pf, sz, pf, sz, pf, C.POINTER(sz))
    _sig(L.redio_fir_create, i, C.POINTER(vp), pf, sz, sz, u)
    _sig(L.redio_fir_destroy, i, vp)
    _sig(L.redio_fir_nout, sz, vp, sz)
    _sig(L.redio_fir_enqueue, i, vp, vp, sz, vp, vp)
    _sig(L.redio_fft_create, i, C.POINTER(vp), i, i)
    _sig(L.redio_fft_destroy, i, vp)
    _sig(L.redio_fft_enqueue, i, vp, vp, vp, sz, vp)
    _sig(L.redio_chain_create, i, C.POINTER(vp), pf, sz, sz, i, u)
    _sig(L.redio_chain_destroy, i, vp)
    _sig(L.redio_chain_nblocks, sz, vp, sz)
    _sig(L.redio_chain_is_fused, i, vp)
    _sig(L.redio_chain_set_unfused, i, vp, i)
    _sig(L.redio_chain_reserve, i, vp, sz)
    _sig(L.redio_chain_set_debug_stamps, i, vp, vp, sz)
    _sig(L.redio_chain_blocks_per_wave, sz, vp, sz)
    _sig(L.redio_chain_launch_waves, sz, vp, sz)
    _sig(L.redio_chain_kernel_name, C.c_char_p, vp)
    _sig(L.redio_fft_reserve, i, vp, sz)
    _sig(L.redio_pfb_reserve, i, vp, sz, i)
    _sig(L.redio_pfb_reserve_two_pass, i, vp, sz, i)
    for n in ("fir", "chain", "pfb", "ovsave"):
        _sig(getattr(L, f"redio_{n}_stream_create"), i, C.POINTER(vp), vp)
        _sig(getattr(L, f"redio_{n}_stream_destroy"), i, vp)
        _sig(getattr(L, f"redio_{n}_stream_reset"), i, vp)
        _sig(getattr(L, f"redio_{n}_stream_nout"), sz, vp, sz)
        _sig(getattr(L, f"redio_{n}_stream_pending"), sz, vp)
        _sig(getattr(L, f"redio_{n}_stream_enqueue"), i, vp, vp, sz, vp, C.POINTER(sz), vp)
    for n in ("chain", "pfb"):
        _sig(getattr(L, f"redio_{n}_stream_create_u8"), i, C.POINTER(vp), vp)
    _sig(L.redio_chain_enqueue, i, vp, vp, sz, vp, vp)
    _sig(L.redio_chain_enqueue_u8, i, vp, vp, sz, vp, vp)
    _sig(L.redio_chain_reserve_u8, i, vp, sz)
    _sig(L.redio_pfb_reserve_u8, i, vp, sz, i)
    pl = C.POINTER(C.c_long)
    _sig(L.redio_graph_begin, i, vp)
    _sig(L.redio_graph_end, i, vp, C.POINTER(vp))
    _sig(L.redio_graph_launch, i, vp, vp)
    _sig(L.redio_graph_destroy, i, vp)
    _sig(L.redio_src_create, i, C.POINTER(vp), i, i)
    _sig(L.redio_src_destroy, i, vp)
    _sig(L.redio_src_reset, i, vp)
    _sig(L.redio_src_set_ratio, i, vp, C.c_double)
    _sig(L.redio_src_set_mode, i, vp, i)
    _sig(L.redio_src_process, i, vp, vp, C.c_long, C.c_long, vp, C.c_long, C.c_long, C.c_double, i, pl, pl, vp)
    _sig(L.redio_src_process_host, i, vp, pf, C.c_long, pf, C.c_long, C.c_double, i, pl, pl)
    _sig(L.redio_src_table, i, i, pf, C.POINTER(i), C.POINTER(i))
    _sig(L.redio_src_path_counts, i, vp, pl, pl)
    _sig(L.redio_fft_enqueue_strided, i, vp, vp, vp, sz, C.c_long, vp)
    _sig(L.redio_ovsave_create, i, C.POINTER(vp), pf, sz, i)
    _sig(L.redio_ovsave_destroy, i, vp)
    _sig(L.redio_ovsave_nout, sz, vp, sz)
    _sig(L.redio_ovsave_enqueue, i, vp, vp, sz, vp, vp)
    psz = C.POINTER(sz)
    _sig(L.redio_data_to_samples, i, vp, sz, vp, vp)
    _sig(L.redio_norm_c32, i, vp, sz, vp, vp)
    _sig(L.redio_ingest_u8_mag, i, vp, sz, vp, vp)
    _sig(L.redio_block_sums, i, vp, sz, sz, vp, vp)
    _sig(L.redio_discretize, i, vp, sz, vp, vp, vp)
    _sig(L.redio_trigger_create, i, C.POINTER(vp))
    _sig(L.redio_trigger_destroy, i, vp)
    _sig(L.redio_trigger_feed, i, vp, vp, sz, sz, vp, sz, psz, sz, psz, psz, vp)
    _sig(L.redio_rle_create, i, C.POINTER(vp))
    _sig(L.redio_rle_destroy, i, vp)
    _sig(L.redio_rle_feed, i, vp, vp, sz, vp, vp, sz, psz, vp)
    _sig(L.redio_dle, i, vp, sz, sz, vp, vp)
    _sig(L.redio_rld, i, vp, vp, sz, vp, sz, vp, psz, vp)
    _sig(L.redio_dld, i, vp, vp, sz, f, vp, sz, vp, psz, vp)
    _sig(L.redio_binconv, i, vp, sz, sz, psz, sz, vp, vp)
    for f in (L.redio_mul_f32, L.redio_add_f32, L.redio_mul_c32, L.redio_add_c32):
        _sig(f, i, vp, vp, vp, sz, vp)
    _sig(L.redio_pfb_create, i, C.POINTER(vp), pf, i, i, u)
    _sig(L.redio_pfb_destroy, i, vp)
    _sig(L.redio_pfb_nrows, sz, vp, sz)
    _sig(L.redio_pfb_enqueue, i, vp, vp, sz, vp, i, vp)
    _sig(L.redio_pfb_enqueue_u8, i, vp, vp, sz, vp, i, vp)
    _sig(L.redio_comm_unique_id, i, vp)
    _sig(L.redio_comm_init_rank, i, C.POINTER(vp), i, i, vp)
    _sig(L.redio_comm_init_all, i, C.POINTER(vp), i, C.POINTER(i))
    _sig(L.redio_comm_destroy, i, vp)
    _sig(L.redio_comm_rank, i, vp)
    _sig(L.redio_comm_size, i, vp)
    _sig(L.redio_comm_last_error, C.c_char_p)
    _sig(L.redio_pfb_exchange, i, vp, vp, vp, psz, sz, vp)
    _sig(L.redio_pfb_exchange_at, i, vp, vp, vp, psz, psz, sz, vp)
    _sig(L.redio_pfb_exchange_all, i, C.POINTER(vp), i, C.POINTER(vp), C.POINTER(vp), psz, sz, C.POINTER(vp))
    _sig(L.redio_synth_iq, i, vp, C.c_uint32, C.c_uint64, sz, vp)
    _sig(L.redio_synth_f32, i, vp, C.c_uint32, C.c_uint64, sz, vp)
    _lib = L
    return L


def kisslib():
    """The loaded libkissfft.so drop-in (kiss_fft_alloc / kiss_fft / kiss_fft_cleanup)."""
    global _kiss
    if _kiss is not None:
        return _kiss
    lib()
    if not os.path.exists(LIBKISSFFT):
        raise ImportError(f"{LIBKISSFFT} is missing: run libredio_amd.build()")
    K = C.CDLL(LIBKISSFFT)
    _sig(K.kiss_fft_alloc, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_size_t))
    _sig(K.kiss_fft, None, C.c_void_p, C.c_void_p, C.c_void_p)
    _sig(K.kiss_fft_stride, None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)
    _sig(K.kiss_fft_cleanup, None)
    _sig(K.kiss_fft_next_fast_size, C.c_int, C.c_int)
    _sig(K.kiss_fft_free, None, C.c_void_p)
    _sig(K.redio_kiss_fft_set_spin_ns, None, C.c_long)
    _kiss = K
    return K


def samplerate_lib():
    """The loaded libsamplerate.so drop-in (the nine src_* symbols of samplerate.rs:32-42)."""
    global _src
    if _src is not None:
        return _src
    lib()
    if not os.path.exists(LIBSAMPLERATE):
        raise ImportError(f"{LIBSAMPLERATE} is missing: run libredio_amd.build()")
    S = C.CDLL(LIBSAMPLERATE)
    _sig(S.src_new, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int))
    _sig(S.src_delete, C.c_void_p, C.c_void_p)
    _sig(S.src_process, C.c_int, C.c_void_p, C.c_void_p)
    _sig(S.src_get_name, C.c_char_p, C.c_int)
    _sig(S.src_get_description, C.c_char_p, C.c_int)
    _sig(S.src_get_version, C.c_char_p)
    _sig(S.src_set_ratio, C.c_int, C.c_void_p, C.c_double)
    _sig(S.src_is_valid_ratio, C.c_int, C.c_double)
    _sig(S.src_strerror, C.c_char_p, C.c_int)
    _sig(S.src_reset, C.c_int, C.c_void_p)
    _sig(S.src_error, C.c_int, C.c_void_p)
    _sig(S.src_simple, C.c_int, C.c_void_p, C.c_int, C.c_int)
    _src = S
    return S


def check(code, what="redio"):
    if code != 0:
        raise RedioError(code, what)


from . import bitfount, dsputils, kissfft, kpn_dev, plans, samplerate  # noqa: E402,F401
from .plans import Chain, Channelizer, Comm, Fft, Fir, Graph, OverlapSave, Src, Stream, channelizer_all_to_all, current_stream, synth_f32, synth_iq  # noqa: E402,F401
