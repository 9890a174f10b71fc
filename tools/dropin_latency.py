"""Per-call cost of the host-buffer drop-ins (kiss_fft, src_process, redio_convolve_f32): PCIe + launch bound."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import libredio_amd as R
K = R.kisslib()
K.kiss_fft_alloc.restype = C.c_void_p
K.kiss_fft_alloc.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
K.kiss_fft.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
for n in (64, 1024, 4096, 65536):
    cfg = K.kiss_fft_alloc(n, 0, None, None)
    x = (np.random.rand(n) + 1j * np.random.rand(n)).astype(np.complex64); y = np.empty_like(x)
    for _ in range(20): K.kiss_fft(cfg, x.ctypes.data, y.ctypes.data)
    t0 = time.perf_counter()
    reps = 500
    for _ in range(reps): K.kiss_fft(cfg, x.ctypes.data, y.ctypes.data)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(50): np.fft.fft(x)
    cpu = (time.perf_counter() - t0) / 50
    print(f"kiss_fft drop-in N={n}: {dt*1e6:.1f} us per call ({n/dt/1e6:.1f} MS/s); numpy.fft on this host: {cpu*1e6:.1f} us")
from libredio_amd import dsputils
u = np.random.rand(1 << 20).astype(np.float32); v = dsputils.lpf_corrected(63, 0.1)
dsputils.convolve(u, v)
t0 = time.perf_counter()
for _ in range(20): dsputils.convolve(u, v)
dt = (time.perf_counter() - t0) / 20
print(f"redio_convolve_f32 2^20 x 63 taps (host buffers): {dt*1e3:.2f} ms per call ({len(u)/dt/1e6:.0f} MS/s)")
for m in (1024, 16384):
    um = u[:m].copy()
    dsputils.convolve(um, v)
    t0 = time.perf_counter()
    for _ in range(300): dsputils.convolve(um, v)
    dt = (time.perf_counter() - t0) / 300
    t0 = time.perf_counter()
    for _ in range(50): np.correlate(um, v, "valid")
    cpu = (time.perf_counter() - t0) / 50
    print(f"redio_convolve_f32 {m} x 63 taps per message: {dt*1e6:.1f} us per call; numpy.correlate on this host: {cpu*1e6:.1f} us")
from libredio_amd import samplerate
for ratio, frames in ((0.02, 4096), (0.02, 65536), (0.5, 4096), (2.0, 4096), (48000 / 44100, 4096)):
    st = samplerate.State(1, 1)
    x = np.random.rand(frames).astype(np.float32)
    for _ in range(5): st.block(x, ratio)
    t0 = time.perf_counter()
    reps = 100
    for _ in range(reps): st.block(x, ratio)
    dt = (time.perf_counter() - t0) / reps
    print(f"src_process drop-in ratio {ratio:.4f}, {frames} frames per message: {dt*1e6:.1f} us per call ({frames/dt/1e6:.1f} MS/s in)")
    st.close()
