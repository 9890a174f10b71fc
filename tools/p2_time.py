"""Kernels on the FftP2 workgroup image (fft_p2_kernel for 128 / 512 points, pfb_p2_kernel for every one-kernel channelizer shape) on 2^28 samples, best of three
bursts: run once per build directory by tools/ab_old_build.sh (round 6: one pad element per 16 instead of per 8 for 128 points and more)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, libredio_amd as R
n = 1 << 28
x = R.synth_iq(0x5EED0004, 0, n)
def timed(f, reps=20):
    for _ in range(40): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for M, P in ((256, 16), (128, 16), (256, 8), (128, 8), (512, 8), (1024, 4), (32, 16), (128, 4), (256, 4)):
    plan = R.Channelizer(R.dsputils.lpf_corrected(M * P, 0.45 / M), M, P)
    out = torch.empty((plan.nrows(n), M), dtype=torch.complex64, device="cuda")
    t = min(timed(lambda: plan(x, out=out)) for _ in range(3))
    print(f"channelizer M={M} P={P}: {t:.4f} ms ({16.0*n/t/1e6/8000:.1%} of 8 TB/s)", flush=True)
    del out, plan
out = torch.empty_like(x)
for nfft in (128, 512, 32):
    plan = R.Fft(nfft)
    t = min(timed(lambda: plan(x, out=out)) for _ in range(3))
    print(f"FFT {nfft}: {t:.4f} ms ({16.0*n/t/1e6/8000:.1%} of 8 TB/s)", flush=True)
