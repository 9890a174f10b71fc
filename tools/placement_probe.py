"""Where do the chain kernel's waves land (XCD / SE / CU / SIMD) and how long does each live?"""
import ctypes as C, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, libredio_amd as R
lib = R.lib()
n = 1 << 28
chain = R.Chain(R.dsputils.lpf_corrected(127, 0.08), 5, 1024, fused=True)
x = R.synth_iq(0x5EED0002, 0, n); out = torch.empty((chain.nblocks(n), 1024), dtype=torch.complex64, device="cuda")
dbg = torch.zeros(4 * chain.launch_waves(chain.nblocks(n)), dtype=torch.int64, device="cuda")  # one record per wavefront of the launch
for _ in range(300): chain(x, out)
chain.set_debug_stamps(dbg); chain(x, out); torch.cuda.synchronize(); chain.set_debug_stamps(None)
d = dbg.cpu().numpy().reshape(-1, 4); d = d[d[:, 1] > 0]
life = d[:, 1] / 100.0
xcc = (d[:, 3] >> 32) & 0xF; hw = d[:, 3] & 0xFFFFFFFF
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = xcc * 1000 + se * 100 + sh * 20 + cu
print("blocks per wavefront", chain.blocks_per_wave(chain.nblocks(n)), "(a residency slot hosts several wavefronts one after another when the launch has more wavefronts than slots: the per-CU counts below are wavefronts hosted over the launch, not concurrent ones)")
print("waves", len(d), "distinct CUs", len(set(cuid)), "life us: min %.0f med %.0f max %.0f" % (life.min(), np.median(life), life.max()))
cnt = collections.Counter(cuid)
print("waves per CU histogram:", sorted(collections.Counter(cnt.values()).items()))
for k in sorted(set(cnt.values())):
    sel = np.array([cnt[c] == k for c in cuid])
    print(f"  CUs hosting {k:2d} waves: wave life median {np.median(life[sel]):.0f} us, max {life[sel].max():.0f}")
print("per XCD: count, median life:", [(int(i), int((xcc == i).sum()), int(np.median(life[xcc == i]))) for i in sorted(set(xcc))])
sc = collections.Counter(zip(cuid, simd))
print("waves per SIMD histogram:", sorted(collections.Counter(sc.values()).items()))
for k in sorted(set(sc.values())):
    sel = np.array([sc[(c, s)] == k for c, s in zip(cuid, simd)])
    print(f"  SIMDs hosting {k} waves: wave life median {np.median(life[sel]):.0f} us")
