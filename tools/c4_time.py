import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, libredio_amd as R
n = 1 << 28
x = R.synth_iq(0x5EED0004, 0, n)
g = torch.Generator(device="cuda"); g.manual_seed(4)
raw = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
plan = R.Channelizer(R.dsputils.lpf_corrected(1024, 0.45 / 64))
rows = plan.nrows(n)
def timed(f, reps=30):
    for _ in range(60): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
o = torch.empty((rows, 64), dtype=torch.complex64, device="cuda")
og = torch.empty((8, rows, 8), dtype=torch.complex64, device="cuda")
for name, f, b in (("cf32 natural", lambda: plan(x, out=o), 16.0), ("cf32 grouped x8", lambda: plan(x, ngroups=8, out=og), 16.0), ("u8 natural", lambda: plan.from_bytes(raw, out=o), 10.0)):
    t = min(timed(f) for _ in range(3))
    print(f"C4 64 ch x 16 taps {name}: {t:.4f} ms ({b*n/t/1e6/8000:.1%} of 8 TB/s)", flush=True)
for P in (8, 4):
    pl = R.Channelizer(R.dsputils.lpf_corrected(64 * P, 0.45 / 64), 64, P)
    t = min(timed(lambda: pl(x, out=o)) for _ in range(3))
    print(f"C4 64 ch x {P} taps cf32 natural: {t:.4f} ms ({16.0*n/t/1e6/8000:.1%})", flush=True)
