import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, libredio_amd as R
n = 1 << 28
taps = R.dsputils.lpf_corrected(8193, 0.08)
x = R.synth_iq(0x5EED0005, 0, n)
plan = R.OverlapSave(taps, 65536)
out = torch.empty(plan.nout(n), dtype=torch.complex64, device="cuda")
for _ in range(6): plan(x, out=out)
torch.cuda.synchronize()
