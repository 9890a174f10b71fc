#!/usr/bin/env python3
"""Host-resident input end to end: pinned host buffers -> H2D copy on one HIP stream while the previous chunk
runs data_to_samples -> FIR/5 -> FFT-1024 on another (double buffered; from bytes that is ONE kernel per window).  The chain runs as a STREAM
(redio_chain_stream_*): the 126-sample FIR seam and the partial block at the end of every chunk are carried on the
device, so the spectra are those of the uninterrupted stream -- nothing is lost at chunk boundaries.  Reports the
PCIe-inclusive rate, which is what a host that hands over host buffers gets; never bench.py's `value`.

    python tools/stream_pipeline.py [u8|cf32] [chunks] [log2 samples per chunk]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libredio_amd as R
from libredio_amd import bitfount as B

fmt = sys.argv[1] if len(sys.argv) > 1 else "u8"
nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 26)
taps = R.dsputils.lpf_corrected(127, 0.08)
chain = R.Chain(taps, 5, 1024, fused=True)
stream = R.Stream(chain, u8=(fmt == "u8"))   # bytes: carried as bytes, every window one kernel (redio_chain_stream_create_u8)
nblk = chain.nblocks(n) + 1          # a chunk can complete one block more than it holds (the carried partial block)
bytes_per_sample = 2 if fmt == "u8" else 8
host = [torch.randint(0, 256, (n * bytes_per_sample,), dtype=torch.uint8).pin_memory() for _ in range(2)]
dev_raw = [torch.empty(n * bytes_per_sample, dtype=torch.uint8, device="cuda") for _ in range(2)]
dev_out = [torch.empty((nblk, 1024), dtype=torch.complex64, device="cuda") for _ in range(2)]
copy_s, comp_s = torch.cuda.Stream(), torch.cuda.Stream()
copied = [torch.cuda.Event() for _ in range(2)]
done = [torch.cuda.Event() for _ in range(2)]


def run(chunks):
    for i in range(chunks):
        b = i & 1
        with torch.cuda.stream(copy_s):
            copy_s.wait_event(done[b])                 # the buffer's previous consumer has finished
            dev_raw[b].copy_(host[b], non_blocking=True)
            copied[b].record(copy_s)
        with torch.cuda.stream(comp_s):
            comp_s.wait_event(copied[b])
            x = dev_raw[b] if fmt == "u8" else dev_raw[b].view(torch.complex64)
            stream(x, out=dev_out[b].view(-1))   # history carried: [tail | chunk] without copying the chunk
            done[b].record(comp_s)
    torch.cuda.synchronize()


for b in range(2):
    done[b].record(comp_s)
run(4)
t0 = time.perf_counter()
run(nchunks)
dt = time.perf_counter() - t0
print(f"{fmt} host stream -> chain, {nchunks} chunks of 2^{n.bit_length() - 1} samples, copy and compute overlapped: "
      f"{nchunks * n / dt / 1e9:.2f} GS/s  ({nchunks * n * bytes_per_sample / dt / 1e9:.1f} GB/s over PCIe)")
