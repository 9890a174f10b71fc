import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, libredio_amd as R
torch.zeros(1, device="cuda")
for n in (1024, 4096, 65536, 1 << 20, 1 << 22, 1 << 24, 10000, 16200):
    t0 = time.perf_counter(); p = R.Fft(n); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"plan {n}: {(t1 - t0) * 1e3:.1f} ms")
    del p
