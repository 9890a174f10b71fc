"""C3 (256 channels, ratio 1/50) through the single-launch uniform-phase path with each register blocking of
src_window_rb_kernel (REDIO_SRC_RB = 0: one output per lane, 2 / 4: that many outputs per lane-wing), interleaved rounds in
ONE process, outputs compared bit for bit.  usage: python tools/c3_variants.py [frames_log2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the knobs below exist only in the measurement build (make -C libredio_amd/csrc measure -> libredio_amd/_build_measure)
os.environ.setdefault("REDIO_BUILD_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libredio_amd", "_build_measure"))
import torch, libredio_amd as R

nch, frames = 256, 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
x = torch.stack([R.synth_f32(100 + c, 0, frames) for c in range(nch)])
variants = [v for v in os.environ.get("C3_VARIANTS", "0,2,2w,2t,4").split(",")]  # w: one 640-output tile per CU; t: separate tables per wing


def select(v):
    os.environ["REDIO_SRC_RB"] = v[0]
    os.environ["REDIO_SRC_RB_WIDE"] = "1" if "w" in v else "0"
    if "t" in v: os.environ["REDIO_SRC_TWO_TABLES"] = "1"
    else: os.environ.pop("REDIO_SRC_TWO_TABLES", None)


plans, outs, best = {}, {}, {}
for v in variants:
    select(v)
    plans[v] = R.Src(nch, 1, mode=R.Src.EXACT)
    out, used = plans[v].process(x, 0.02)
    outs[v] = out.clone()
    best[v] = []
for rnd in range(6):
    for v in variants:
        select(v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, used = plans[v].process(x, 0.02)
        torch.cuda.synchronize(); best[v].append(time.perf_counter() - t0)
ref = outs[variants[0]]
for v in variants:
    same = torch.equal(outs[v].view(torch.int32), ref.view(torch.int32)) and outs[v].shape == ref.shape
    t = sorted(best[v])
    print(f"REDIO_SRC_RB={v}: min {t[0]*1e3:.3f} ms  median {t[len(t)//2]*1e3:.3f} ms  {nch*frames/t[0]/1e9:.1f} GS/s in   bit-identical to RB={variants[0]}: {same}", flush=True)
