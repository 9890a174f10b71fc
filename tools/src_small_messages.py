import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, libredio_amd as R
nch = 256
for frames in (1024, 4096, 16384, 65536):
    x = torch.stack([R.synth_f32(100 + c, 0, frames) for c in range(nch)])
    for name, mode in (("exact", R.Src.EXACT), ("fast", R.Src.FAST)):
        plan = R.Src(nch, 1, mode=mode)
        for _ in range(20): plan.process(x, 0.02)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        t0 = time.perf_counter(); e0.record()
        for _ in range(n): plan.process(x, 0.02)
        e1.record(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n
        print(f"resample 1/50 x{nch} ch, {frames} frames per message ({name}): {e0.elapsed_time(e1)/n*1e3:.1f} us per call on the stream, {wall*1e6:.1f} us wall, {nch*frames/wall/1e9:.2f} GS/s", flush=True)
