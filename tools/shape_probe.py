#!/usr/bin/env python3
"""Run ONE plan shape a few times (for rocprofv3 passes on a secondary kernel):
    python3 tools/shape_probe.py fir|firr|firx|firrx K D [log2n] [launches]
    python3 tools/shape_probe.py pfb|pfbu8 M P [log2n] [launches]      (pfbu8: the 64-channel channelizer from u8 I/Q bytes)
    python3 tools/shape_probe.py src|srcfast CHANNELS LOG2FRAMES
    python3 tools/shape_probe.py fft N 0 [log2n] [launches]
Prints the HIP-event mean per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libredio_amd as R

kind, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = 1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 26)
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
x = R.synth_iq(1, 0, n) if kind not in ("src", "srcfast", "u8chain", "pfbu8") else None
if kind in ("fir", "firr", "firx", "firrx"):   # r: real samples; x: reference rounding (multiply and add rounded separately)
    cplx = kind in ("fir", "firx")
    if not cplx:
        x = R.synth_f32(1, 0, n)
    plan = R.Fir(R.dsputils.lpf_corrected(a, 0.4 / b if b > 1 else 0.2), b, complex_input=cplx, fused=kind in ("fir", "firr"))
    out = torch.empty(plan.nout(n), dtype=x.dtype, device="cuda")
    run = lambda: plan(x, out=out)
elif kind == "u8chain":   # the north-star chain from u8 I/Q bytes (a, b ignored)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    raw = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
    plan = R.Chain(R.dsputils.lpf_corrected(127, 0.08), 5, 1024, fused=True)
    out = torch.empty((plan.nblocks(n), 1024), dtype=torch.complex64, device="cuda")
    run = lambda: plan.from_bytes(raw, out=out)
elif kind in ("src", "srcfast"):   # a = channels, b = log2 frames per channel; ratio 1/50 (BASELINE.json configs[2]); srcfast: REDIO_SRC_FAST
    xr = torch.stack([R.synth_f32(100 + c, 0, 1 << b) for c in range(a)])
    plan = R.Src(a, 1, mode=R.Src.FAST if kind == "srcfast" else R.Src.EXACT)
    n = a << b
    def run():
        plan.reset(); plan.process(xr, 0.02)
elif kind == "pfbu8":
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    raw = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
    plan = R.Channelizer(R.dsputils.lpf_corrected(a * b, 0.45 / a), a, b)
    out = torch.empty((plan.nrows(n), a), dtype=torch.complex64, device="cuda")
    run = lambda: plan.from_bytes(raw, out=out)
elif kind == "pfb":
    plan = R.Channelizer(R.dsputils.lpf_corrected(a * b, 0.45 / a), a, b)
    out = torch.empty((plan.nrows(n), a), dtype=torch.complex64, device="cuda")
    run = lambda: plan(x, out=out)
else:
    x = x[: n // a * a]
    plan = R.Fft(a)
    out = torch.empty_like(x)
    run = lambda: plan(x, out=out)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{kind} {a} {b}: {ms:.3f} ms per launch, {n / ms / 1e6:.1f} GS/s")
