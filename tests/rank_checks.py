#!/usr/bin/env python3
"""Correctness halves of the multi-GPU launch paths, run by tests/test_gpu_channelizer.py under torch.distributed.run with one
rank per visible device -- the launch line the 8-GPU node uses for tools/bench_c4.py and tools/bench_shards.py.  Test
infrastructure: this is where the oracle is consulted; the tools themselves only measure.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tests/rank_checks.py c4|c3|c5 [--exchange cabi|torch]

c4: BASELINE.json configs[3] -- every rank channelizes its time shard, the exchange (redio_pfb_exchange over RCCL, or
all_to_all_single) regroups, every rank's [all rows][its channels] must be the oracle's channelizer of the WHOLE stream.
c3 / c5: the independent shards of configs[2] / configs[4] (sharding.channel_shard / overlap_save_shard) against the oracle's result
for all channels / the whole stream, restricted to the shard.  Prints {"ok": true} on rank 0 and exits 0, else 1."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["c4", "c3", "c5"])
    ap.add_argument("--exchange", default="cabi", choices=["cabi", "torch"])
    ap.add_argument("--backend", default="nccl")
    a = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import libredio_amd as R
    import oracle as O
    from libredio_amd import sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29566")
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(a.backend, rank=rank, world_size=world)

    def all_ok(ok):
        flag = torch.tensor([int(ok)], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    if a.config == "c4":
        M, P = 64, 16
        h = R.dsputils.lpf_corrected(M * P, 0.45 / M)
        plan = R.Channelizer(h)
        cpg = sharding.channelizer_exchange_layout(world, M)
        use_cabi = a.exchange == "cabi" and a.backend == "nccl"
        comm = R.Comm.from_torch_distributed() if use_cabi else None
        total_rows = 2000 + 5 * world
        xs = O.synth_iq(0x5EED0004, 0, M * total_rows)
        want = O.pfb_channelizer(xs, h, M, P, True)
        first, nout, nin = sharding.channelizer_time_shard(rank, world, total_rows, P)
        g = plan(torch.from_numpy(xs[M * first: M * (first + nin)]).cuda(), ngroups=world).reshape(world, nout, cpg)
        rows = [sharding.channelizer_time_shard(q, world, total_rows, P)[1] for q in range(world)]
        got = (comm.exchange(g, rows) if use_cabi else R.channelizer_all_to_all(g)).cpu().numpy()
        ok = np.array_equal(got.view(np.uint32), np.ascontiguousarray(want[:, rank * cpg:(rank + 1) * cpg]).view(np.uint32))
        what = "channelizer exchange vs oracle"
    elif a.config == "c5":
        nfft, k = 65536, 8193
        taps = R.dsputils.lpf_corrected(k, 0.02)
        hop = nfft - k + 1
        plan = R.OverlapSave(taps, nfft)
        total = nfft + hop * (3 * world + 1) + 777
        first, n, first_out, n_out = sharding.overlap_save_shard(rank, world, total, k, nfft)
        got = plan(R.synth_iq(0x5EED0005, first, n)).cpu().numpy() if n else np.zeros(0, np.complex64)
        ok = len(got) == n_out
        for b in range(n_out // hop):   # every block of this shard against the oracle on its own window of the WHOLE stream
            want = O.overlap_save(O.synth_iq(0x5EED0005, first_out + hop * b, nfft), taps, nfft)
            ok = ok and np.array_equal(got[hop * b: hop * (b + 1)].view(np.uint32), want.view(np.uint32))
        what = "overlap-save shards vs oracle"
    else:
        nch_all, ratio, n = 256, 0.02, 60000
        first_ch, nch = sharding.channel_shard(rank, world, nch_all)
        x = np.stack([O.synth_f32(0x5EED0003 + c, 0, n) for c in range(first_ch, first_ch + nch)]) if nch else np.zeros((0, n), np.float32)
        ok = True
        if nch:
            plan = R.Src(nch, 1)
            got = np.concatenate([plan.process(torch.from_numpy(x[:, lo:hi]).contiguous().cuda(), ratio)[0].cpu().numpy()
                                  for lo, hi in ((0, 25001), (25001, n))], axis=1)
            for c in range(0, nch, max(1, nch // 4)):
                ref = O.Resampler(1)
                want = np.concatenate([ref.block(x[c, lo:hi], ratio) for lo, hi in ((0, 25001), (25001, n))])
                ok = ok and np.array_equal(got[c].view(np.uint32), want.view(np.uint32))
        what = "resampler channel shards vs oracle"
    ok = all_ok(ok)
    if rank == 0:
        print(json.dumps({"check": what, "n_gpus": world, "ok": ok}))
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
