#!/usr/bin/env python3
"""BASELINE.json configs[3]: 64-channel polyphase channelizer, time-sharded over N GPUs, then ONE all-to-all
(RCCL over xGMI) that regroups [time shard][all channels] into [all time][channels of this rank].

    python tools/bench_c4.py                                   # 1 GPU (the exchange degenerates to a copy)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_c4.py --gpus 8

Prints one JSON line on rank 0: samples/s of the analysis alone and of analysis + exchange (MAX over ranks),
and the exchange's egress rate per GPU.  Not the headline bench (bench.py is); same timing discipline.  The correctness half
of this launch path lives with the tests: tests/rank_checks.py c4."""
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
    ap.add_argument("--pieces", type=int, default=4, help="analyse the slice in this many pieces; the exchange of piece i runs on a second HIP stream "
                                                          "beside the analysis of piece i + 1 (C-ABI exchange only)")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import libredio_amd as R
    from libredio_amd import sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if a.backend == "nccl" and int(os.environ.get("LOCAL_RANK", "0")) >= torch.cuda.device_count():
        sys.exit(f"bench_c4.py rank {rank}: LOCAL_RANK {os.environ.get('LOCAL_RANK')} but only {torch.cuda.device_count()} HIP devices are visible "
                 f"(ranks never share a GPU under the nccl backend)")
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    # who takes part: the world size torch.distributed reports and every rank's device.  Under the nccl backend ranks that share a
    # device would give a curve that is not a scaling curve: every rank exits non-zero before anything is timed.
    pr = torch.cuda.get_device_properties(local)
    me = {"rank": rank, "device": local, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
          "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xFF, getattr(pr, "pci_device_id", 0))}
    seen = [None] * dist.get_world_size()
    dist.all_gather_object(seen, me)
    distinct = len({(d["pci"], d["uuid"]) for d in seen})
    if a.backend == "nccl" and distinct != world:
        if rank == 0:
            print(f"bench_c4.py: {world} ranks on {distinct} distinct devices ({[d['pci'] for d in seen]}); refusing to time ranks that share a GPU", file=sys.stderr)
        dist.destroy_process_group()
        sys.exit(3)
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
    # piece-wise pipeline: analysis of piece i + 1 on the compute stream while piece i is exchanged on the communication stream
    def pipelined(xs_all, rows_all, out_all, pieces):
        comm_s = pipelined.stream
        rp = -(-rows_all // pieces)
        for i in range(pieces):
            r0, r1 = i * rp, min((i + 1) * rp, rows_all)
            if r1 <= r0:
                break
            if pipelined.free[i % 2] is not None:      # this piece buffer was last sent two pieces ago
                torch.cuda.current_stream().wait_event(pipelined.free[i % 2])
            g = pipelined.gbuf[i % 2].view(-1)[: world * (r1 - r0) * cpg].view(world, r1 - r0, cpg)
            plan(xs_all[M * r0: M * (r1 + P - 1)], ngroups=world, out=g)
            ev = torch.cuda.Event(); ev.record()
            comm_s.wait_event(ev)
            comm.exchange_at(g, [r1 - r0] * world, out_all, [q * rows_all + r0 for q in range(world)], stream=comm_s)
            done = torch.cuda.Event(); done.record(comm_s)
            pipelined.free[i % 2] = done
        torch.cuda.current_stream().wait_stream(comm_s)
        return out_all
    if use_cabi:
        pipelined.stream = torch.cuda.Stream()
        rp_max = -(-nrows // max(a.pieces, 1))
        pipelined.gbuf = [torch.empty((world, rp_max, cpg), dtype=torch.complex64, device="cuda") for _ in range(2)]
        pipelined.free = [None, None]
    t_analysis = timed(lambda: plan(x, ngroups=world, out=grouped))
    t_both = timed(lambda: exchange(plan(x, ngroups=world, out=grouped)))
    mine = exchange(grouped)
    assert mine.shape == (world * nrows, cpg)
    t_pipe = None
    if use_cabi and a.pieces > 1:
        pipe_buf = torch.empty_like(mine_buf)
        t_pipe = timed(lambda: pipelined(x, nrows, pipe_buf, a.pieces))
        torch.cuda.synchronize()
        if not torch.equal(pipe_buf, mine):
            bad = (torch.view_as_real(pipe_buf) != torch.view_as_real(mine)).any(dim=2).any(dim=1).nonzero().flatten()
            raise AssertionError(f"piece-wise pipeline differs from the one-shot exchange: {bad.numel()} rows, first {bad[:8].tolist()}, last {bad[-4:].tolist()} of {mine.shape[0]}")
    if rank == 0:
        egress = nrows * (M - cpg) * 8                  # bytes this GPU sends to its peers per step
        print(json.dumps({"workload": "BASELINE.json configs[3]: 64-channel polyphase channelizer, P=16, channels sharded over the GPUs",
                          "n_gpus": world, "samples_per_gpu": n, "steps": a.steps,
                          "analysis_GSps": world * n / t_analysis / 1e9, "analysis_plus_exchange_GSps": world * n / t_both / 1e9,
                          "ms_analysis": t_analysis * 1e3, "ms_analysis_plus_exchange": t_both * 1e3,
                          "ms_pipelined": None if t_pipe is None else t_pipe * 1e3, "pieces": a.pieces,
                          "pipelined_GSps": None if t_pipe is None else world * n / t_pipe / 1e9,
                          "exchange_egress_GBps_per_gpu": (egress / max(t_both - t_analysis, 1e-9) / 1e9) if world > 1 else None,
                          # xGMI is point to point: a rank's world - 1 transfers of a step each have a link of their own
                          "exchange_egress_GBps_per_link": (egress / (world - 1) / max(t_both - t_analysis, 1e-9) / 1e9) if world > 1 else None,
                          "ranks_seen": {"world_size": dist.get_world_size(), "devices": seen, "distinct_devices": distinct},
                          "scaling": "weak",
                          "collective": "redio_pfb_exchange: RCCL ncclSend/ncclRecv group (C ABI)" if use_cabi else "all_to_all_single (%s)" % a.backend}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
