"""What the ring's byte budget really buys (profiles/r06_kpn_ring_bytes.txt): the same two launches per message (1024-point transforms of 2^k samples, then the
checksum of the output), bare from one thread, with the outputs in 1, 2 or 4 buffers in rotation and the reader 0 ... 3 messages behind the writer.
Separates 'the reader finds the message in the cache' from 'the writer finds its recycled buffer in the cache'."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, libredio_amd as R
from libredio_amd.kpn_dev import checksum_u32
def run(k, nbuf, lag, reps=600, do_sum=True):
    n = 1 << k
    x = R.synth_iq(0x5EED0004, 0, 4 * n)
    plan = R.Fft(1024)
    outs = [torch.empty(n, dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
    acc = torch.zeros(1, dtype=torch.int64, device="cuda")
    def burst(m):
        for i in range(m):
            plan(x[(i % 4) * n:(i % 4 + 1) * n], out=outs[i % nbuf])
            if do_sum and i >= lag: checksum_u32(outs[(i - lag) % nbuf], acc)
    burst(100); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); burst(reps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for k in (22, 23, 24, 25):
    print(f"2^{k}-sample messages ({8 << k >> 20} MiB out), us per message:")
    print("  transform only, outputs in 1 / 2 / 4 buffers:", " / ".join(f"{run(k, b, 0, do_sum=False):.1f}" for b in (1, 2, 4)))
    for nbuf in (1, 2, 4):
        lags = [l for l in (0, 1, 3) if l < nbuf]
        print(f"  transform + checksum, {nbuf} output buffer(s): " + ", ".join(f"reader {l} behind: {run(k, nbuf, l):.1f}" for l in lags))
