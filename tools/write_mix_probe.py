"""What does a write-dominated stream sustain on this box?  Yardsticks for the u8 -> channelizer path (2 B read + 8 B written per sample):
the repository's own u8 -> cf32 conversion kernel (the same byte mix with next to no arithmetic), a plain write (fill), a plain copy,
then the two channelizer entries -- one process, one box."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, libredio_amd as R
from libredio_amd import bitfount
n = 1 << 28
g = torch.Generator(device="cuda"); g.manual_seed(4)
raw = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
x = R.synth_iq(0x5EED0004, 0, n)
def timed(f, reps=30):
    for _ in range(60): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
y = torch.empty(n, dtype=torch.complex64, device="cuda")
plan = R.Channelizer(R.dsputils.lpf_corrected(1024, 0.45 / 64))
o = torch.empty((plan.nrows(n), 64), dtype=torch.complex64, device="cuda")
yr = torch.view_as_real(y)
for name, f, b in (("u8 -> cf32 conversion (redio_data_to_samples)", lambda: bitfount.data_to_samples(raw, out=y), 10.0),
                   ("plain write (torch fill_)", lambda: yr.fill_(1.0), 8.0),
                   ("plain copy cf32 (torch copy_)", lambda: y.copy_(x), 16.0),
                   ("C4 64 ch x 16 taps from u8", lambda: plan.from_bytes(raw, out=o), 10.0),
                   ("C4 64 ch x 16 taps from cf32", lambda: plan(x, out=o), 16.0)):
    t = min(timed(f) for _ in range(3))
    print(f"{name}: {t:.4f} ms  {b * n / t / 1e6:.0f} GB/s ({b * n / t / 1e6 / 8000:.1%} of 8 TB/s; written {8.0 * n / t / 1e6:.0f} GB/s)", flush=True)
