"""In-kernel clock of the chain kernel under steady load (diagnostic build path: stamps go to their own
buffer; the stamped launches' run times are not quoted).  Clock = d(s_memtime)/d(s_memrealtime) x 100 MHz."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, libredio_amd as R
lib = R.lib()
n = 1 << 28
chain = R.Chain(R.dsputils.lpf_corrected(127, 0.08), 5, 1024, fused=True)
x = R.synth_iq(0x5EED0002, 0, n); out = torch.empty((chain.nblocks(n), 1024), dtype=torch.complex64, device="cuda")
dbg = torch.zeros(4 * chain.launch_waves(chain.nblocks(n)), dtype=torch.int64, device="cuda")  # one record per wavefront of the launch
for burst in (5, 50, 500):
    for _ in range(burst): chain(x, out)        # load the chip
    chain.set_debug_stamps(dbg)
    chain(x, out); torch.cuda.synchronize()
    chain.set_debug_stamps(None)
    d = dbg.cpu().numpy().reshape(-1, 4); d = d[d[:, 1] > 0]
    clk = d[:, 0] / d[:, 1] * 100e6
    import numpy as np
    st = d[:, 2] - d[:, 2].min(); en = st + d[:, 1]
    print(f"   starts: median {np.median(st)/100:.1f} us, max {st.max()/100:.1f} us; ends: min {en.min()/100:.1f} median {np.median(en)/100:.1f} max {en.max()/100:.1f} us; late starters (>50us): {(st>5000).sum()}")
    print(f"after {burst:4d} launches: waves={len(d)} wave life {np.median(d[:,1])/100:.1f} us, clock median {np.median(clk)/1e9:.3f} GHz (min {clk.min()/1e9:.3f}, max {clk.max()/1e9:.3f})")
