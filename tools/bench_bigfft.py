import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libredio_amd as R
def timeit(f, n=5, warm=2):
    for _ in range(warm): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
n=1<<26
for nfft in (9000, 10000, 12288, 15000, 16200, 20000, 50000, 100000, 1000000, 32768, 131072, 262144, 524288, 1048576, 2097152, 4194304, 8388608, 16777216):
    x = R.synth_iq(2, 0, n)[: n // nfft * nfft]
    plan = R.Fft(nfft); out = torch.empty_like(x)
    ms = timeit(lambda: plan(x, out=out))
    print(f"FFT {nfft}: {ms:.3f} ms {x.numel()/ms/1e6:.1f} GS/s ({16*x.numel()/ms/1e6/8000:.1%})")
