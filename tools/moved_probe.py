"""Exactly NCALLS calls of one multi-pass config, nothing else on the device: the workload of tools/pmc_moved.sh (rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE passes), whose counter totals divided by NCALLS are the HBM bytes one call moves (bench.py's other_configs sizes).
usage: python3 tools/moved_probe.py c5|fft65536 NCALLS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, libredio_amd as R

which, ncalls = sys.argv[1], int(sys.argv[2])
n = 1 << 28
x = R.synth_iq(0x5EED0002, 0, n)
if which == "c5":
    plan = R.OverlapSave(R.dsputils.lpf_corrected(8193, 0.08), 65536)
    out = torch.empty(plan.nout(n), dtype=torch.complex64, device="cuda")
    f = lambda: plan(x, out=out)
else:
    m = 1 << 26
    plan = R.Fft(65536)
    xs, out = x[:m], torch.empty(m, dtype=torch.complex64, device="cuda")
    f = lambda: plan(xs, out=out)
for _ in range(ncalls):
    f()
torch.cuda.synchronize()
