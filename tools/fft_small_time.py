"""The batched one-wave transforms (1024 / 256 / 64 points; kissfft::fft's sizes at BASELINE configs[1] and configs[3]) on 2^28 points, out of place
and in place, best of three bursts: run once per build directory by tools/ab_old_build.sh."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, libredio_amd as R
n = 1 << 28
x = R.synth_iq(2, 0, n)
out = torch.empty_like(x)
def timed(f, reps=20):
    for _ in range(30): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for nfft in (1024, 256, 64, 4096, 2048):
    plan = R.Fft(nfft)
    t = min(timed(lambda: plan(x, out=out)) for _ in range(3))
    print(f"FFT {nfft}, out of place: {t:.4f} ms ({16.0*n/t/1e6/8000:.1%} of 8 TB/s)", flush=True)
plan = R.Fft(1024)
m = 52428 * 1024
t = min(timed(lambda: plan(x[:m], out=out[:m])) for _ in range(3))
print(f"FFT 1024, the chain's 52428 decimated blocks: {t:.4f} ms ({16.0*m/t/1e6/8000:.1%} of 8 TB/s)", flush=True)
