"""Per-launch duration series of the chain kernel (HIP events), to see clock ramp/throttle effects."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, libredio_amd as R
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = R.lib(); n = 1 << 28
taps = R.dsputils.lpf_corrected(127, 0.08)
chain = R.Chain(taps, 5, 1024, fused=True)
x = R.synth_iq(0x5EED0002, 0, n); out = torch.empty((chain.nblocks(n), 1024), dtype=torch.complex64, device="cuda")
st = R.current_stream(); evs = []
for _ in range(steps + 1):
    e = C.c_void_p(); lib.redio_event_create(C.byref(e)); evs.append(e)
torch.cuda.synchronize(); time.sleep(0.5)
lib.redio_event_record(evs[0], st)
for k in range(steps):
    chain(x, out); lib.redio_event_record(evs[k + 1], st)
torch.cuda.synchronize()
ms = []
for k in range(steps):
    m = C.c_float(); lib.redio_event_elapsed_ms(evs[k], evs[k + 1], C.byref(m)); ms.append(m.value)
print("first10", [round(v, 3) for v in ms[:10]])
print("every 20th", [round(v, 3) for v in ms[::20]])
print("mean", sum(ms) / len(ms), "min", min(ms), "max", max(ms))
