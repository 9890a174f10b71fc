"""Run the batched resampler once per mode for profiling: python tools/src_probe.py [nchan] [log2 frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, libredio_amd as R
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
x = torch.stack([R.synth_f32(100 + c, 0, frames) for c in range(nch)])
plan = R.Src(nch, 1)
for _ in range(3):
    out, used = plan.process(x, 0.02)
torch.cuda.synchronize()
print(out.shape, used)
