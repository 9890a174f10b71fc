"""The chain's two-kernel path (FIR -> plan-owned intermediate -> FFT; every shape chain_supported() does not list) on 2^28 samples: one call
over the whole message against the same work cut into calls whose intermediate is at most 32 / 64 / 128 MiB (what a chunked enqueue would do:
the transform then reads what the FIR wrote from the last-level cache).  Emulated through the Python plan: out and in advance by whole blocks."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, libredio_amd as R
n = 1 << 28
x = R.synth_iq(0x5EED0002, 0, n)
def timed(f, reps=10):
    for _ in range(12): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for K, D, N, force in ((127, 5, 1024, True), (127, 5, 4096, False), (63, 1, 512, False), (255, 10, 1024, False), (31, 2, 2048, False), (101, 3, 256, False), (63, 1, 1024, True)):
    plan = R.Chain(R.dsputils.lpf_corrected(K, 0.4 / D), D, N)
    if force: plan.set_unfused(True)
    assert not plan.is_fused
    nb = plan.nblocks(n)
    out = torch.empty((nb, N), dtype=torch.complex64, device="cuda")
    plan.reserve(n)
    whole = min(timed(lambda: plan(x, out=out)) for _ in range(3))
    ref = out.clone()
    alg = (8.0 + 8.0 / D) * n
    line = f"K={K} D={D} nfft={N}: whole {whole:.3f} ms ({alg/whole/1e6/8000:.1%})"
    for mib in (32, 64, 128):
        cb = max(1, (mib << 20) // (8 * N))
        def chunked():
            b = 0
            while b < nb:
                c = min(cb, nb - b)
                xin = x[b * N * D: (b + c - 1) * N * D + (N - 1) * D + K]
                plan(xin, out=out[b:b + c])
                b += c
        out.zero_()
        t = min(timed(chunked) for _ in range(3))
        ok = torch.equal(out.view(torch.int32), ref.view(torch.int32))
        line += f" | {mib} MiB: {t:.3f} ms ({alg/t/1e6/8000:.1%}, {whole/t:.3f}x{'' if ok else ' MISMATCH'})"
    print(line, flush=True)
    del out, ref, plan
