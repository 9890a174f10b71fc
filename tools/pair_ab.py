"""The multi-pass transforms and the 65536-point overlap-save in ONE build (REDIO_BUILD_DIR selects it: the product build has the
two-column "pair" tile program, make OUT=../_build_onecol EXTRA=-DREDIO_TILE_PAIR=0 the one-column program it replaced).  Run the two
builds alternately (A B A B) and compare.  usage: python tools/pair_ab.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, libredio_amd as R

tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("REDIO_BUILD_DIR", "product"))


def timeit(f, n=30, warm=30):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


n = 1 << 28
x = R.synth_iq(0x5EED0005, 0, n)
for nfft, k in ((65536, 8193), (65536, 127), (65536, 8192), (32768, 127), (131072, 127)):
    taps = R.dsputils.lpf_corrected(k, 0.08)
    plan = R.OverlapSave(taps, nfft)
    out = torch.empty(plan.nout(n), dtype=torch.complex64, device="cuda")
    ms = timeit(lambda: plan(x, out=out), n=20, warm=20)
    b = 8 * nfft / (nfft - k + 1) + 8
    print(f"[{tag}] overlap-save N={nfft} K={k}: {ms:.3f} ms  {out.numel()/ms/1e6:.1f} GS/s out  ({b*out.numel()/ms/1e6/8000:.1%} of 8 TB/s)", flush=True)
    del plan, out
for lg, nn in ((16, 28), (16, 26), (18, 26), (20, 26), (22, 26), (24, 26), (15, 26), (17, 26), (19, 26), (21, 26), (23, 26)):
    nfft = 1 << lg
    xx = x[: 1 << nn]
    plan = R.Fft(nfft)
    out = torch.empty_like(xx)
    ms = timeit(lambda: plan(xx, out=out), n=20, warm=20)
    print(f"[{tag}] FFT 2^{lg} over 2^{nn} samples: {ms:.3f} ms  {xx.numel()/ms/1e6:.1f} GS/s  ({16*xx.numel()/ms/1e6/8000:.1%} of 8 TB/s)", flush=True)
    del plan, out
