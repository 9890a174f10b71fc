#!/usr/bin/env python3
"""BASELINE.json configs[2] and configs[4] on N GPUs of one node: independent shards, NO collective in the data path
 -- C3: 256 resampler channels dealt in contiguous groups (sharding.channel_shard); C5: overlap-save blocks of
65536 points dealt contiguously, every shard reading the input span of its own blocks (sharding.overlap_save_shard).

    python tools/bench_shards.py c5                                                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_shards.py c5 --gpus 8
    ... tools/bench_shards.py c3 --gpus 8

Weak scaling for C5 (every rank owns 2^28 samples of one ever-longer stream), strong for C3 (256 channels in all).  Prints one
JSON line on rank 0: whole-job rate over the slowest rank's time.  The correctness half of this launch path (every rank's shard
against the CPU restatement's result for the whole stream / all channels) lives with the tests: tests/rank_checks.py c3 | c5.  bench.py stays the headline."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["c3", "c5"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backend", default="nccl")
    a = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import libredio_amd as R
    from libredio_amd import sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(a.backend, rank=rank, world_size=world)

    def sync():
        dist.barrier(); torch.cuda.synchronize()

    def timed(f):
        for _ in range(a.warmup): f()
        sync(); t0 = time.perf_counter()
        for _ in range(a.steps): f()
        sync(); dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / a.steps

    def all_ok(ok):
        flag = torch.tensor([int(ok)], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    if a.config == "c5":
        nfft, k = 65536, 8193
        taps = R.dsputils.lpf_corrected(k, 0.02)
        hop = nfft - k + 1
        plan = R.OverlapSave(taps, nfft)
        n = 1 << 28
        nblk = (n - nfft) // hop + 1
        x = R.synth_iq(0x5EED0005, rank * nblk * hop, n)          # consecutive shards of one ever-longer stream
        out = torch.empty(plan.nout(n), dtype=torch.complex64, device="cuda")
        t = timed(lambda: plan(x, out=out))
        if rank == 0:
            b = 8 * nfft / hop + 8
            print(json.dumps({"workload": "BASELINE.json configs[4]: overlap-save FFT convolution, 65536-pt blocks, 8193 taps, blocks dealt over the GPUs",
                              "n_gpus": world, "samples_per_gpu": n, "steps": a.steps, "ms_per_step": t * 1e3, "scaling": "weak", "collective": None,
                              "value": world * out.numel() / t / 1e6, "unit": "MSamples/s of output",
                              "frac_of_hbm_roofline_per_gpu": b * out.numel() / t / 8e12}))
    else:
        nch_all, ratio = 256, 0.02
        first_ch, nch = sharding.channel_shard(rank, world, nch_all)
        frames = 1 << 20
        x = torch.stack([R.synth_f32(0x5EED0003 + c, 0, frames) for c in range(first_ch, first_ch + nch)])
        plan = R.Src(nch, 1)
        t = timed(lambda: plan.process(x, ratio))
        if rank == 0:
            print(json.dumps({"workload": "BASELINE.json configs[2]: samplerate 2.4 MS/s -> 48 kS/s, 256 channels dealt over the GPUs, 2^20 frames per channel per step",
                              "n_gpus": world, "channels_per_gpu": nch, "steps": a.steps, "ms_per_step": t * 1e3, "scaling": "strong", "collective": None,
                              "value": nch_all * frames / t / 1e6, "unit": "MSamples/s of input"}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
