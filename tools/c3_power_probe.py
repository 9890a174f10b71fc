"""Is C3 (256 channels, ratio 1/50) held back by its kernels' structure or by the chip's power / clock management?  Same binaries, same launches,
steady state (many calls back to back): hash-generated random samples against all-zero samples and random x 1e-3, for the exact (f64) and the FAST
(f32 polyphase) forms, with rocm-smi sampled mid-burst.  On zeros the multiply-adds toggle almost nothing, the chip draws less and holds a higher
clock; instruction stream, LDS traffic and HBM bytes are identical.  usage: python tools/c3_power_probe.py [frames_log2] [calls]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, libredio_amd as R

nch, frames = 256, 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
x = torch.stack([R.synth_f32(100 + c, 0, frames) for c in range(nch)])
z = torch.zeros_like(x)
small = x * 1e-3


def smi(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5)
            out.append(r.stdout.strip().replace("\n", " | "))
        except Exception as e:
            out.append("rocm-smi unavailable: %r" % (e,)); return
        time.sleep(0.25)


for name, mode in (("FAST f32 polyphase", R.Src.FAST), ("exact f64", R.Src.EXACT)):
    for dname, data in (("random", x), ("zeros", z), ("random*1e-3", small), ("random", x)):
        plan = R.Src(nch, 1, mode=mode)
        plan.process(data, 0.02)
        stop, log = threading.Event(), []
        th = threading.Thread(target=smi, args=(stop, log)); th.start()
        torch.cuda.synchronize(); ts = []
        for _ in range(calls):
            t0 = time.perf_counter(); plan.process(data, 0.02); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        stop.set(); th.join()
        tail = sorted(ts[calls // 3:])
        print(f"{name:20s} data={dname:12s} calls={calls} ms per call: median {tail[len(tail)//2]*1e3:.3f} min {tail[0]*1e3:.3f}", flush=True)
        if log: print("    rocm-smi mid-burst:", log[len(log) // 2][:260], flush=True)
