"""A/B of the real-sample 63-tap FIR (BASELINE.json configs[0]'s shape): the round-5 tiled kernel (scalar lane program), the pair-image tile
(fir_core.h fir_lane_pairs: both copies of the tile in LDS, packed multiply-adds only) in several tile shapes, and the wave-private run form
(fir_run.hip), interleaved in ONE process (measurement build).
usage: python3 tools/fir_run_ab.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("REDIO_BUILD_DIR", os.path.join(ROOT, "libredio_amd", "_build_measure"))
import torch, libredio_amd as R

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
taps = R.dsputils.lpf_corrected(63, 0.1)
KNOBS = ("REDIO_FIR_RUN", "REDIO_FIR_RUN_WPS", "REDIO_FIR_RUN_SPW", "REDIO_FIR_PAIRS")


def setenv(**kw):
    for k in KNOBS: os.environ.pop(k, None)
    for k, v in kw.items(): os.environ[k] = str(v)


variants = [("tiled kernel, scalar lane program (product)", {}),
            ("pair images, 16 outputs/lane, 128 threads", dict(REDIO_FIR_PAIRS=16128)),
            ("pair images, 8 outputs/lane, 256 threads", dict(REDIO_FIR_PAIRS=8256)),
            ("pair images, 16 outputs/lane, 256 threads", dict(REDIO_FIR_PAIRS=16256)),
            ("pair images, 8 outputs/lane, 128 threads", dict(REDIO_FIR_PAIRS=8128)),
            ("pair images, 16 outputs/lane, 64 threads", dict(REDIO_FIR_PAIRS=16064)),
            ("run form (wave-private), 3 waves/SIMD", dict(REDIO_FIR_RUN=1, REDIO_FIR_RUN_WPS=3, REDIO_FIR_RUN_SPW=4)),
            ("run form (wave-private), 4 waves/SIMD", dict(REDIO_FIR_RUN=1, REDIO_FIR_RUN_SPW=4))]
for log2n in (26, 28):
    n = 1 << log2n
    x = R.synth_f32(1, 0, n)
    for fused in (True, False):
        plan = R.Fir(taps, 1, complex_input=False, fused=fused)
        out = torch.empty(plan.nout(n), dtype=torch.float32, device="cuda")
        ref = None
        for r in range(rounds):
            for name, env in variants:
                setenv(**env)
                for _ in range(60): plan(x, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100): plan(x, out=out)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 100
                if ref is None: ref = out.clone()
                same = torch.equal(out.view(torch.int32), ref.view(torch.int32))
                print(f"2^{log2n} f32 samples, 63 taps / 1, {'fmaf' if fused else 'mul+add'}, round {r}: {name:52s} {ms:.4f} ms  {n / ms / 1e6:.1f} GS/s  "
                      f"{8 * n / ms / 1e6 / 80:.1f} % of 8 TB/s  {126 * n / ms / 1e9:.1f} TFLOP/s  same bits: {same}", flush=True)
                assert same
