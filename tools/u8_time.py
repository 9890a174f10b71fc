"""The byte-input forms (chain and 64-channel channelizer from u8 I/Q bytes) on 2^28 samples, best of three bursts: run once per build directory by
tools/ab_old_build.sh (round 6: the byte -> f32 conversion i2f in two operations instead of three)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, libredio_amd as R
from libredio_amd import bitfount as B
n = 1 << 28
g = torch.Generator(device="cuda"); g.manual_seed(4)
raw = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
def timed(f, reps=30):
    for _ in range(60): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
chain = R.Chain(R.dsputils.lpf_corrected(127, 0.08), 5, 1024, fused=True)
oc = torch.empty((chain.nblocks(n), 1024), dtype=torch.complex64, device="cuda")
plan = R.Channelizer(R.dsputils.lpf_corrected(1024, 0.45 / 64))
o = torch.empty((plan.nrows(n), 64), dtype=torch.complex64, device="cuda")
og = torch.empty((8, plan.nrows(n), 8), dtype=torch.complex64, device="cuda")
conv = torch.empty(n, dtype=torch.complex64, device="cuda")
mag = torch.empty(n, dtype=torch.float32, device="cuda")
for name, f in (("u8 -> chain (fmaf), one kernel", lambda: chain.from_bytes(raw, oc)), ("u8 -> C4 natural", lambda: plan.from_bytes(raw, out=o)),
                ("u8 -> C4 grouped x8", lambda: plan.from_bytes(raw, ngroups=8, out=og)), ("data_to_samples", lambda: B.data_to_samples(raw, out=conv)),
                ("ingest u8 -> |x|", lambda: B.ingest_mag(raw))):
    t = min(timed(f) for _ in range(3))
    print(f"{name}: {t:.4f} ms  {n / t / 1e6:.1f} GS/s", flush=True)
