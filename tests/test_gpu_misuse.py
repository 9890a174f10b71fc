"""The C ABI never aborts ("C ABI never aborts; returns codes"): NULL handles, NULL buffers with a non-zero count, shapes no
plan can have.  Every call below returns a negative REDIO_ERR_* (or 0 from a size query on a NULL handle) and the library keeps working."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_null_handles_and_null_buffers_return_codes(gpu, redio, oracle):
    L = redio.lib()
    N = None
    one = gpu.zeros(4096, dtype=gpu.float32, device="cuda")
    p = C.c_void_p(one.data_ptr())
    h = C.c_void_p()
    taps = (C.c_float * 4)(0.25, 0.25, 0.25, 0.25)
    sz, ln = C.c_size_t(0), C.c_long(0)
    neg = [
        # creation: NULL out-pointer, no taps, zero decimation, impossible sizes
        ("redio_fir_create", (N, taps, 4, 1, 0)), ("redio_fir_create", (C.byref(h), N, 4, 1, 0)), ("redio_fir_create", (C.byref(h), taps, 0, 1, 0)),
        ("redio_fir_create", (C.byref(h), taps, 4, 0, 0)),
        ("redio_fft_create", (N, 64, 0)), ("redio_fft_create", (C.byref(h), 0, 0)), ("redio_fft_create", (C.byref(h), -5, 0)),
        ("redio_chain_create", (N, taps, 4, 1, 64, 0)), ("redio_chain_create", (C.byref(h), N, 4, 1, 64, 0)), ("redio_chain_create", (C.byref(h), taps, 0, 1, 64, 0)),
        ("redio_chain_create", (C.byref(h), taps, 4, 0, 64, 0)), ("redio_chain_create", (C.byref(h), taps, 4, 1, 0, 0)),
        ("redio_ovsave_create", (N, taps, 4, 64)), ("redio_ovsave_create", (C.byref(h), N, 4, 64)), ("redio_ovsave_create", (C.byref(h), taps, 0, 64)),
        ("redio_ovsave_create", (C.byref(h), taps, 4, 2)), ("redio_ovsave_create", (C.byref(h), taps, 4, 0)),
        ("redio_pfb_create", (N, taps, 2, 2, 0)), ("redio_pfb_create", (C.byref(h), N, 2, 2, 0)), ("redio_pfb_create", (C.byref(h), taps, 0, 2, 0)),
        ("redio_pfb_create", (C.byref(h), taps, 2, 0, 0)),
        ("redio_src_create", (N, 1, 1)), ("redio_src_create", (C.byref(h), 9, 1)), ("redio_src_create", (C.byref(h), 1, 0)),
        ("redio_trigger_create", (N,)), ("redio_rle_create", (N,)),
        ("redio_fir_stream_create", (C.byref(h), N)), ("redio_chain_stream_create", (C.byref(h), N)), ("redio_pfb_stream_create", (C.byref(h), N)),
        ("redio_ovsave_stream_create", (C.byref(h), N)),
        # NULL handles
        ("redio_fir_enqueue", (N, p, 16, p, N)), ("redio_fft_enqueue", (N, p, p, 1, N)), ("redio_fft_enqueue_strided", (N, p, p, 1, 64, N)),
        ("redio_fft_reserve", (N, 1)), ("redio_chain_enqueue", (N, p, 16, p, N)), ("redio_chain_enqueue_u8", (N, p, 16, p, N)),
        ("redio_chain_reserve", (N, 16)), ("redio_chain_set_unfused", (N, 1)), ("redio_chain_set_debug_stamps", (N, p, 4)),
        ("redio_ovsave_enqueue", (N, p, 16, p, N)), ("redio_pfb_enqueue", (N, p, 16, p, 1, N)), ("redio_pfb_enqueue_u8", (N, p, 16, p, 1, N)),
        ("redio_pfb_reserve", (N, 16, 1)), ("redio_src_reset", (N,)), ("redio_src_set_ratio", (N, 1.0)), ("redio_src_set_mode", (N, 0)),
        ("redio_src_process", (N, p, 16, 16, p, 16, 16, 1.0, 0, C.byref(ln), C.byref(ln), N)),
        ("redio_fir_stream_enqueue", (N, p, 16, p, C.byref(sz), N)), ("redio_chain_stream_enqueue", (N, p, 16, p, C.byref(sz), N)),
        ("redio_pfb_stream_enqueue", (N, p, 16, p, C.byref(sz), N)), ("redio_ovsave_stream_enqueue", (N, p, 16, p, C.byref(sz), N)),
        ("redio_fir_stream_reset", (N,)), ("redio_graph_launch", (N, N)), ("redio_graph_end", (N, N)),
        # NULL buffers with a non-zero count on the plan-less kernels
        ("redio_data_to_samples", (N, 16, p, N)), ("redio_data_to_samples", (p, 16, N, N)), ("redio_data_to_samples", (p, 15, p, N)),
        ("redio_norm_c32", (N, 8, p, N)), ("redio_norm_c32", (p, 8, N, N)), ("redio_ingest_u8_mag", (N, 16, p, N)), ("redio_ingest_u8_mag", (p, 16, N, N)),
        ("redio_block_sums", (N, 2, 8, p, N)), ("redio_block_sums", (p, 2, 8, N, N)), ("redio_block_sums", (p, 2, 0, p, N)),
        ("redio_discretize", (N, 8, p, p, N)), ("redio_discretize", (p, 8, N, p, N)), ("redio_discretize", (p, 8, p, N, N)),
        ("redio_mul_f32", (N, p, p, 8, N)), ("redio_add_c32", (p, p, N, 8, N)), ("redio_synth_iq", (N, 1, 0, 8, N)), ("redio_synth_f32", (N, 1, 0, 8, N)),
        ("redio_dle", (N, 4, 100, p, N)),  # (a zero rate is not an error: kpn.rs:36 divides f32s, the result is inf)
        ("redio_upload", (N, p, 16, N)), ("redio_download", (N, p, 16, N)), ("redio_copy", (N, p, 16, N)), ("redio_malloc", (N, 16)),
        ("redio_stream_create", (N,)), ("redio_event_create", (N,)), ("redio_event_elapsed_ms", (N, N, N)),
        ("redio_convolve_f32", (N, 8, taps, 4, taps, C.byref(sz))), ("redio_window", (4, N)), ("redio_lpf", (4, 0.1, N)),
        ("redio_src_table", (9, N, N, N)),
    ]
    for name, args in neg:
        rc = getattr(L, name)(*args)
        if name.startswith("redio_src_"):   # the resampler answers with libsamplerate's own (positive) codes
            assert rc != 0, (name, args, rc)
            continue
        assert rc < 0, (name, args, rc)
        assert redio.lib().redio_strerror(rc)
    # size queries on a NULL handle answer 0
    for name, args in (("redio_fir_nout", (N, 100)), ("redio_chain_nblocks", (N, 100)), ("redio_ovsave_nout", (N, 100)), ("redio_pfb_nrows", (N, 100)),
                       ("redio_fir_stream_nout", (N, 100)), ("redio_fir_stream_pending", (N,)), ("redio_chain_blocks_per_wave", (N, 100)),
                       ("redio_chain_launch_waves", (N, 100))):
        assert getattr(L, name)(*args) == 0, name
    # destroying nothing is not an error worth a crash
    for name in ("redio_fir_destroy", "redio_fft_destroy", "redio_chain_destroy", "redio_ovsave_destroy", "redio_pfb_destroy", "redio_src_destroy",
                 "redio_trigger_destroy", "redio_rle_destroy", "redio_fir_stream_destroy", "redio_graph_destroy", "redio_free", "redio_host_free"):
        getattr(L, name)(N)
    # live plans: NULL buffers with a non-zero count
    fir = redio.Fir(np.float32([0.5, 0.5]), 1)
    assert L.redio_fir_enqueue(fir._h, N, 16, p, N) < 0 and L.redio_fir_enqueue(fir._h, p, 16, N, N) < 0
    fft = redio.Fft(64)
    assert L.redio_fft_enqueue(fft._h, N, p, 1, N) < 0 and L.redio_fft_enqueue(fft._h, p, N, 1, N) < 0
    ch = redio.Chain(oracle.lpf_corrected(127, 0.08), 5, 1024)
    assert L.redio_chain_enqueue(ch._h, N, 1 << 16, p, N) < 0 and L.redio_chain_enqueue(ch._h, p, 1 << 16, N, N) < 0
    ov = redio.OverlapSave(oracle.lpf_corrected(31, 0.1), 256)
    assert L.redio_ovsave_enqueue(ov._h, N, 4096, p, N) < 0 and L.redio_ovsave_enqueue(ov._h, p, 4096, N, N) < 0
    pf = redio.Channelizer(oracle.lpf_corrected(64 * 4, 0.45 / 64), 64, 4)
    assert L.redio_pfb_enqueue(pf._h, N, 4096, p, 1, N) < 0 and L.redio_pfb_enqueue(pf._h, p, 4096, N, 1, N) < 0
    assert L.redio_pfb_enqueue(pf._h, p, 4096, p, 0, N) < 0 and L.redio_pfb_enqueue(pf._h, p, 4096, p, 7, N) < 0   # groups must divide the channels
    src = redio.Src(1, 1)
    assert L.redio_src_process(src._h, N, 16, 16, p, 16, 16, 1.0, 0, C.byref(ln), C.byref(ln), N) != 0
    assert L.redio_src_process(src._h, p, 16, 16, N, 16, 16, 1.0, 0, C.byref(ln), C.byref(ln), N) != 0
    assert L.redio_src_process(src._h, p, 16, 16, p, 16, 16, 1e9, 0, C.byref(ln), C.byref(ln), N) == 6   # SRC_ERR_BAD_SRC_RATIO
    # and the library still computes
    x = oracle.synth_iq(3, 0, 64)
    assert np.array_equal(fft(gpu.from_numpy(x).cuda()).cpu().numpy().view(np.uint32), oracle.fft(x, 64).view(np.uint32))
