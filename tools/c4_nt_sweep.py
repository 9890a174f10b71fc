"""The 64-channel channelizer kernel (pfb64_kernel, 16 taps per branch) under each cache policy of its two streams (REDIO_PFB_NT: bit 0
non-temporal row loads, bit 1 non-temporal row stores; measurement build), every input format and output layout, interleaved in ONE
process on 2^28 samples: what the per-variant choice in launch_pfb_t rests on.  usage: python3 tools/c4_nt_sweep.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("REDIO_BUILD_DIR", os.path.join(ROOT, "libredio_amd", "_build_measure"))
import torch, libredio_amd as R
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 1 << 28
x = R.synth_iq(0x5EED0004, 0, n)
g = torch.Generator(device="cuda"); g.manual_seed(4)
raw = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
plan = R.Channelizer(R.dsputils.lpf_corrected(1024, 0.45 / 64))
rows = plan.nrows(n)
o = torch.empty((rows, 64), dtype=torch.complex64, device="cuda")
og = torch.empty((8, rows, 8), dtype=torch.complex64, device="cuda")
def timed(f, reps=30):
    for _ in range(40): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cases = (("cf32 natural", lambda: plan(x, out=o), 16.0, o), ("cf32 grouped x8", lambda: plan(x, ngroups=8, out=og), 16.0, og),
         ("u8 natural", lambda: plan.from_bytes(raw, out=o), 10.0, o), ("u8 grouped x8", lambda: plan.from_bytes(raw, ngroups=8, out=og), 10.0, og))
ref = {}
for r in range(rounds):
    for name, f, b, buf in cases:
        for nt in ("product", "0", "1", "2", "3"):
            if nt == "product": os.environ.pop("REDIO_PFB_NT", None)
            else: os.environ["REDIO_PFB_NT"] = nt
            t = min(timed(f) for _ in range(2))
            if name not in ref: ref[name] = buf.clone()
            same = torch.equal(buf.view(torch.int32), ref[name].view(torch.int32))
            print(f"round {r}: C4 64 ch x 16 taps {name:16s} policy {nt:7s}: {t:.4f} ms ({b*n/t/1e6/8000:.1%} of 8 TB/s)  same bits: {same}", flush=True)
            assert same
