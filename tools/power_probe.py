"""Is the chain kernel held back by its structure or by the chip's power/clock management?
Same binary, same launch, steady state (>= 2 s of back-to-back launches each): hash-generated random IQ against
all-zero IQ.  On zeros every multiply-add toggles almost nothing, the chip draws less and holds a higher clock
(MI355X_MICROARCH.md, "DVFS give-back"); the instruction stream, the LDS traffic and the HBM bytes are identical.
Prints per case: mean / median / min kernel time of the last two thirds of the burst (HIP events), the in-kernel
clock from per-wave s_memtime / s_memrealtime stamps of one extra stamped launch, and the roofline fraction."""
import ctypes as C, os, sys, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, libredio_amd as R

lib = R.lib(); n = 1 << 28
taps = R.dsputils.lpf_corrected(127, 0.08)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
x = R.synth_iq(0x5EED0002, 0, n)
z = torch.zeros_like(x)
small = (x * 1e-3)            # same sign/mantissa activity, small exponent: separates data toggling from value range


def smi(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5)
            out.append(r.stdout.strip().replace("\n", " | "))
        except Exception as e:  # not installed / not permitted: the stamps below are the evidence then
            out.append("rocm-smi unavailable: %r" % (e,)); return
        time.sleep(0.5)


for fused in ((True,) if 'fused' in sys.argv else (True, False)):
    chain = R.Chain(taps, 5, 1024, fused=fused)
    out = torch.empty((chain.nblocks(n), 1024), dtype=torch.complex64, device="cuda")
    used = chain.nblocks(n) * 5120
    for name, data in ((("random", x), ("zeros", z)) if 'fused' in sys.argv else (("random", x), ("zeros", z), ("random*1e-3", small), ("random", x))):
        st = R.current_stream()
        evs = []
        for _ in range(steps + 1):
            e = C.c_void_p(); lib.redio_event_create(C.byref(e)); evs.append(e)
        stop, log = threading.Event(), []
        th = threading.Thread(target=smi, args=(stop, log)); th.start()
        torch.cuda.synchronize()
        lib.redio_event_record(evs[0], st)
        for k in range(steps):
            chain(data, out); lib.redio_event_record(evs[k + 1], st)
        dbg = torch.zeros(4 * chain.launch_waves(chain.nblocks(n)), dtype=torch.int64, device="cuda")  # one record per wavefront of the launch
        chain.set_debug_stamps(dbg); chain(data, out); torch.cuda.synchronize(); chain.set_debug_stamps(None)
        stop.set(); th.join()
        ms = []
        for k in range(steps):
            m = C.c_float(); lib.redio_event_elapsed_ms(evs[k], evs[k + 1], C.byref(m)); ms.append(m.value)
        for e in evs: lib.redio_event_destroy(e)
        tail = np.array(ms[steps // 3:])
        d = dbg.cpu().numpy().reshape(-1, 4); d = d[d[:, 1] > 0]
        clk = d[:, 0] / d[:, 1] * 100e6
        frac = 9.6 * used / (tail.mean() * 1e-3) / 8e12
        print(f"fir_rounding={'fmaf' if fused else 'mul+add'} data={name:12s} launches={steps} kernel_ms mean {tail.mean():.4f} median {np.median(tail):.4f} "
              f"min {tail.min():.4f}  frac_of_8TBps {frac:.3f}  in-kernel clock median {np.median(clk)/1e9:.3f} GHz  wave life median {np.median(d[:,1])/100:.0f} us")
        if log: print("    rocm-smi mid-burst:", log[len(log) // 2][:300])
