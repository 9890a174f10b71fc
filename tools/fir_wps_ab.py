"""Three wavefronts per SIMD for the stand-alone FIR in the chain's form (chain_v4_kernel<FIR_ONLY>, 94-145 registers): A/B against the
shipped two, interleaved rounds in ONE process, outputs compared bit for bit.  Measurement build only (REDIO_FIR_WPS2 (the two-wave instantiation, kept in measurement builds only) is read by
-DREDIO_MEASURE builds: make -C libredio_amd/csrc measure).  usage: python tools/fir_wps_ab.py [log2 samples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("REDIO_BUILD_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libredio_amd", "_build_measure"))
import torch, libredio_amd as R

m = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
x = R.synth_iq(1, 0, m)


def timed(f, reps=40):
    for _ in range(60): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (k, d) in ((127, 5), (63, 5), (63, 1)):
    taps = R.dsputils.lpf_corrected(k, 0.4 / d if d > 1 else 0.2)
    for fused in (True, False):
        plan = R.Fir(taps, d, fused=fused)
        n_in = (m - k + 1) // (1024 * d) * (1024 * d) + k - 1          # whole 1024-output blocks: the chain-form kernel serves the call
        xv = x[:n_in]
        outs, t = {}, {2: [], 3: []}
        for w in (2, 3):
            os.environ.pop("REDIO_FIR_WPS2", None)
            if w == 2: os.environ["REDIO_FIR_WPS2"] = "1"
            outs[w] = plan(xv).clone()
        for rnd in range(3):
            for w in (2, 3):
                os.environ.pop("REDIO_FIR_WPS2", None)
                if w == 2: os.environ["REDIO_FIR_WPS2"] = "1"
                o = torch.empty_like(outs[2])
                t[w].append(timed(lambda: plan(xv, out=o)))
        os.environ.pop("REDIO_FIR_WPS2", None)
        same = torch.equal(outs[2].view(torch.int32), outs[3].view(torch.int32))
        b = 8.0 * (1 + 1.0 / d) * n_in
        a2, a3 = min(t[2]), min(t[3])
        print(f"FIR {k} taps /{d} {'fmaf' if fused else 'mul+add'} 2^{m.bit_length()-1} samples: two waves per SIMD {a2:.4f} ms ({b/a2/1e6/8000:.1%} of 8 TB/s), "
              f"three {a3:.4f} ms ({b/a3/1e6/8000:.1%}), ratio {a3/a2:.3f}, bit-identical: {same}", flush=True)
