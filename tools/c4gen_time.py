"""The one-kernel channelizer shapes (32 ... 1024 channels) on 2^28 samples, best of three timed bursts: run once per build directory
(REDIO_BUILD_DIR) by tools/ab_old_build.sh to A/B a kernel change between alternating processes on one box.  The u8 and 64-channel lines
ride along because round 6 moved the u8 row offsets into the vector offset (pfb_kernels.hip)."""
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
for M, P in ((32, 16), (128, 16), (256, 16), (128, 8), (256, 8), (32, 4), (512, 8), (1024, 4)):
    plan = R.Channelizer(R.dsputils.lpf_corrected(M * P, 0.45 / M), M, P)
    out = torch.empty((plan.nrows(n), M), dtype=torch.complex64, device="cuda")
    t = min(timed(lambda: plan(x, out=out)) for _ in range(3))
    print(f"channelizer M={M} P={P}: {t:.4f} ms ({16.0*n/t/1e6/8000:.1%} of 8 TB/s)", flush=True)
    del out, plan
g = torch.Generator(device="cuda"); g.manual_seed(4)
raw = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
plan = R.Channelizer(R.dsputils.lpf_corrected(1024, 0.45 / 64))
o = torch.empty((plan.nrows(n), 64), dtype=torch.complex64, device="cuda")
og = torch.empty((8, plan.nrows(n), 8), dtype=torch.complex64, device="cuda")
for name, f, b in (("cf32 natural", lambda: plan(x, out=o), 16.0), ("u8 natural", lambda: plan.from_bytes(raw, out=o), 10.0), ("u8 grouped x8", lambda: plan.from_bytes(raw, ngroups=8, out=og), 10.0)):
    t = min(timed(f, 30) for _ in range(3))
    print(f"C4 64 ch x 16 taps {name}: {t:.4f} ms ({b*n/t/1e6/8000:.1%} of 8 TB/s)", flush=True)
