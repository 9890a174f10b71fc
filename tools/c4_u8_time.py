"""The channelizer's u8 entry (64 channels x 16 taps, 2^28 samples), row-major and grouped x 8 layouts; REDIO_BUILD_DIR selects another build of the
library (tools/ab_old_build.sh alternates two builds on one box).  profiles/r05_channelizer_u8_two_row_loads.txt, r05_channelizer_grouped_shift_null.txt."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, libredio_amd as R
n = 1 << 28
g = torch.Generator(device="cuda"); g.manual_seed(4)
raw = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
plan = R.Channelizer(R.dsputils.lpf_corrected(1024, 0.45 / 64))
rows = plan.nrows(n)
o = torch.empty((rows, 64), dtype=torch.complex64, device="cuda")
og = torch.empty((8, rows, 8), dtype=torch.complex64, device="cuda")
def timed(f, reps=30):
    for _ in range(60): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t = min(timed(lambda: plan.from_bytes(raw, out=o)) for _ in range(3))
tg = min(timed(lambda: plan.from_bytes(raw, ngroups=8, out=og)) for _ in range(3))
print(f"{os.environ.get('REDIO_BUILD_DIR', 'product').split('/')[-1]}: u8 natural {t:.4f} ms ({10.0*n/t/1e6/8000:.1%})  u8 grouped x8 {tg:.4f} ms", flush=True)
