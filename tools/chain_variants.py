"""The north-star chain (2^28 samples, 127 taps / 5, 1024-point FFT) with REDIO_CHAIN_BPW (blocks per wavefront: how compact the
set of addresses in flight is) swept, 300 launches after 300 warm-up launches each, in ONE process; run it once per build
directory (REDIO_BUILD_DIR) to compare load / store cache policies.  usage: python tools/chain_variants.py [bpw ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the knobs below exist only in the measurement build (make -C libredio_amd/csrc measure -> libredio_amd/_build_measure)
os.environ.setdefault("REDIO_BUILD_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libredio_amd", "_build_measure"))
import torch, libredio_amd as R

n = 1 << 28
taps = R.dsputils.lpf_corrected(127, 0.08)
x = R.synth_iq(0x5EED0002, 0, n)
plan = R.Chain(taps, 5, 1024, fused=True)
out = torch.empty((plan.nblocks(n), 1024), dtype=torch.complex64, device="cuda")
ref = None
for bpw in (sys.argv[1:] or ["0", "16", "8", "4", "2", "1"]):
    if bpw == "0": os.environ.pop("REDIO_CHAIN_BPW", None)
    else: os.environ["REDIO_CHAIN_BPW"] = bpw
    for _ in range(300): plan(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): plan(x, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 300
    same = ""
    if ref is None: ref = out.clone()
    else: same = "  same bits as the first: %s" % torch.equal(out.view(torch.int32), ref.view(torch.int32))
    print(f"build {os.environ.get('REDIO_BUILD_DIR', 'product')}: blocks per wavefront {bpw if bpw != '0' else 'default (1/2048 of the stream)'}: "
          f"{ms:.4f} ms  {9.6 * plan.nblocks(n) * 5120 / ms / 1e6:.0f} GB/s = {9.6 * plan.nblocks(n) * 5120 / ms / 1e6 / 80:.1f} % of 8 TB/s{same}", flush=True)
