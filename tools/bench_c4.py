#!/usr/bin/env python3
"""BASELINE.json configs[3]: 64-channel polyphase channelizer, time-sharded over N GPUs, then ONE all-to-all
(RCCL over xGMI) that regroups [time shard][all channels] into [all time][channels of this rank].

    python tools/bench_c4.py                                   # 1 GPU (the exchange degenerates to a copy)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_c4.py --gpus 8

Prints one JSON line on rank 0: samples/s of the analysis alone and of analysis + exchange (MAX over ranks),
and the exchange's egress rate per GPU.  Not the headline bench (bench.py is); same timing discipline."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--log2-samples", type=int, default=28, help="input samples per GPU")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--exchange", default="cabi", choices=["cabi", "torch"],
                    help="cabi = redio_pfb_exchange (RCCL send/recv group through the C ABI); torch = all_to_all_single")
    ap.add_argument("--check", action="store_true", help="compare this rank's regrouped rows with the oracle on a short stream and exit")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import libredio_amd as R
    from libredio_amd import sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    M, P = 64, 16
    h = R.dsputils.lpf_corrected(M * P, 0.45 / M)
    plan = R.Channelizer(h)
    n = 1 << a.log2_samples
    rows_in = n // M                                   # this rank's input rows, look-ahead included (weak scaling)
    first_row = rank * (rows_in - (P - 1))             # consecutive shards of one ever-longer stream
    x = R.synth_iq(0x5EED0004, M * first_row, n)
    nrows = plan.nrows(n)
    cpg = sharding.channelizer_exchange_layout(world, M)
    grouped = torch.empty((world, nrows, cpg), dtype=torch.complex64, device="cuda")

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

    use_cabi = a.exchange == "cabi" and a.backend == "nccl"
    comm = R.Comm.from_torch_distributed() if use_cabi else None
    mine_buf = torch.empty((world * nrows, cpg), dtype=torch.complex64, device="cuda")
    exchange = (lambda g: comm.exchange(g, [nrows] * world, out=mine_buf)) if use_cabi else R.channelizer_all_to_all
    if a.check:   # correctness half: a short stream, every rank against the oracle's channelizer of the WHOLE stream
        import numpy as np
        import oracle as O
        total_rows = 2000 + 5 * world
        xs = O.synth_iq(0x5EED0004, 0, M * total_rows)
        want = O.pfb_channelizer(xs, h, M, P, True)
        first, nout, nin = sharding.channelizer_time_shard(rank, world, total_rows, P)
        g = plan(torch.from_numpy(xs[M * first: M * (first + nin)]).cuda(), ngroups=world).reshape(world, nout, cpg)
        rows = [sharding.channelizer_time_shard(q, world, total_rows, P)[1] for q in range(world)]
        got = (comm.exchange(g, rows) if use_cabi else R.channelizer_all_to_all(g)).cpu().numpy()
        ok = np.array_equal(got.view(np.uint32), np.ascontiguousarray(want[:, rank * cpg:(rank + 1) * cpg]).view(np.uint32))
        flag = torch.tensor([int(ok)], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            print(json.dumps({"check": "channelizer exchange vs oracle", "n_gpus": world, "exchange": a.exchange, "ok": bool(flag.item())}))
        dist.destroy_process_group()
        sys.exit(0 if flag.item() else 1)
    t_analysis = timed(lambda: plan(x, ngroups=world, out=grouped))
    t_both = timed(lambda: exchange(plan(x, ngroups=world, out=grouped)))
    mine = exchange(grouped)
    assert mine.shape == (world * nrows, cpg)
    if rank == 0:
        egress = nrows * (M - cpg) * 8                  # bytes this GPU sends to its peers per step
        print(json.dumps({"workload": "BASELINE.json configs[3]: 64-channel polyphase channelizer, P=16, channels sharded over the GPUs",
                          "n_gpus": world, "samples_per_gpu": n, "steps": a.steps,
                          "analysis_GSps": world * n / t_analysis / 1e9, "analysis_plus_exchange_GSps": world * n / t_both / 1e9,
                          "ms_analysis": t_analysis * 1e3, "ms_analysis_plus_exchange": t_both * 1e3,
                          "exchange_egress_GBps_per_gpu": (egress / max(t_both - t_analysis, 1e-9) / 1e9) if world > 1 else None,
                          "scaling": "weak",
                          "collective": "redio_pfb_exchange: RCCL ncclSend/ncclRecv group (C ABI)" if use_cabi else "all_to_all_single (%s)" % a.backend}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
