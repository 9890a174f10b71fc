"""Overlap-save with blocks that fit one CU (1024 ... 16384 points), 2^28 samples: python tools/ovsave_small_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, libredio_amd as R
n = 1 << 28
x = R.synth_iq(0x5EED0005, 0, n)
def timed(f, reps=20):
    for _ in range(40): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for nfft, k in ((1024, 127), (1024, 63), (2048, 127), (4096, 127), (8192, 127), (16384, 127)):
    plan = R.OverlapSave(R.dsputils.lpf_corrected(k, 0.08), nfft)
    out = torch.empty(plan.nout(n), dtype=torch.complex64, device="cuda")
    ms = min(timed(lambda: plan(x, out=out)) for _ in range(3))
    b = 8 * nfft / (nfft - k + 1) + 8
    print(f"overlap-save N={nfft} K={k}: {ms:.4f} ms  {out.numel()/ms/1e6:.1f} GS/s out  ({b*out.numel()/ms/1e6/8000:.1%} of 8 TB/s)", flush=True)
