"""A/B of the headline chain kernel against its two-deep prefetch form (chain_v4_kernel<..., PF2 = true>: two sub-tiles of loads in flight, the
stream read through a range-checked buffer descriptor, the transform's middle-stage twiddles in LDS), interleaved in ONE process on one box:
the measurement build (make -C libredio_amd/csrc measure) selects the form per launch from REDIO_CHAIN_PF2.  Bit comparison on 2^28 samples
and on ragged sizes (runs of 1-4 blocks, a last run shorter than the others).  usage: python3 tools/chain_pf2_ab.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("REDIO_BUILD_DIR", os.path.join(ROOT, "libredio_amd", "_build_measure"))
import torch, libredio_amd as R

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
taps = R.dsputils.lpf_corrected(127, 0.08)
plan = R.Chain(taps, 5, 1024, fused=True)
plan_ref = R.Chain(taps, 5, 1024, fused=False)


def run(p, x, out, pf2):
    if pf2: os.environ["REDIO_CHAIN_PF2"] = "1"
    else: os.environ.pop("REDIO_CHAIN_PF2", None)
    p(x, out=out)


# bits first: ragged sizes, both roundings
for n in (5120 + 126, 2 * 5120 + 126 + 17, 5 * 5120 + 126, 13 * 5120 + 200, 4099 * 5120 + 126, (1 << 24) + 12345):
    x = R.synth_iq(0x5EED0002, 7, n)
    for p in (plan, plan_ref):
        nb = p.nblocks(n)
        a = torch.zeros((nb, 1024), dtype=torch.complex64, device="cuda"); b = torch.zeros_like(a)
        run(p, x, a, False); run(p, x, b, True)
        torch.cuda.synchronize()
        same = torch.equal(a.view(torch.int32), b.view(torch.int32))
        print(f"n = {n} ({nb} blocks), {'fmaf' if p is plan else 'reference rounding'}: two-deep form has the same bits: {same}", flush=True)
        assert same

n = 1 << 28
x = R.synth_iq(0x5EED0002, 0, n)
out = torch.empty((plan.nblocks(n), 1024), dtype=torch.complex64, device="cuda")
alg = 9.6 * plan.nblocks(n) * 5120
ref = None
for r in range(rounds):
    for pf2 in (False, True):
        for _ in range(300): run(plan, x, out, pf2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): run(plan, x, out, pf2)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 300
        if ref is None: ref = out.clone()
        same = torch.equal(out.view(torch.int32), ref.view(torch.int32))
        print(f"round {r}: {'two sub-tiles in flight (PF2)' if pf2 else 'product form (one sub-tile)      '}: {ms:.4f} ms  {alg / ms / 1e6:.0f} GB/s = {alg / ms / 1e6 / 80:.2f} % of 8 TB/s  same bits: {same}", flush=True)
for pf2 in (False, True):  # the reference-rounding build
    for _ in range(100): run(plan_ref, x, out, pf2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run(plan_ref, x, out, pf2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print(f"reference rounding, {'PF2' if pf2 else 'product form'}: {ms:.4f} ms = {alg / ms / 1e6 / 80:.2f} % of 8 TB/s", flush=True)
