"""Workload for tools/pair_pmc.sh: the 65536-point transform (gather pass + in-place pass as two kernels) over 2^28 samples and the
65536-point overlap-save (8193 taps), a few launches each after a warm-up, in the build REDIO_BUILD_DIR selects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, libredio_amd as R
n = 1 << 28
x = R.synth_iq(0x5EED0005, 0, n)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
plan = R.Fft(65536); out = torch.empty_like(x)
for _ in range(reps): plan(x, out=out)
torch.cuda.synchronize()
ov = R.OverlapSave(R.dsputils.lpf_corrected(8193, 0.08), 65536)
o2 = torch.empty(ov.nout(n), dtype=torch.complex64, device="cuda")
for _ in range(reps): ov(x, out=o2)
torch.cuda.synchronize()
