"""The N>1 path on CPU: world_size-2 gloo process groups exercising exactly the slicing arithmetic
bench.py and the channelizer use (libredio_amd/sharding.py).  No GPU: the per-slice compute is done by
the oracle here purely as a stand-in so that the *sharding* (halo, block ownership, exchange layout,
max-over-ranks timing reduction) is what is being tested."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _chain_worker(rank, world, port, total, q):
    import oracle as O
    from libredio_amd import sharding
    _init(rank, world, port)
    taps = O.lpf_corrected(127, 0.08)
    first, n, fb, nb = sharding.chain_slice(rank, world, total, 127, 5, 1024)
    x = O.synth_iq(0x5EED0002, first, n)                       # each rank generates only its slice
    mine = O.chain_fir_fft(x, taps, 5, 1024, fused=True)
    assert mine.shape[0] == nb
    # gather block counts and spectra on rank 0 (a consumer that wants the whole output; the data
    # path itself needs no collective)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([nb]))
    mx = max(int(c) for c in counts)
    pad = np.zeros((mx, 1024), np.complex64); pad[:nb] = mine
    bufs = [torch.zeros((mx, 1024, 2)) for _ in range(world)]
    dist.all_gather(bufs, torch.from_numpy(pad.view(np.float32).reshape(mx, 1024, 2)))
    # the timing reduction bench.py uses: MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(world)
    if rank == 0:
        whole = np.concatenate([b.numpy().reshape(mx, 2048).view(np.complex64)[: int(c)] for b, c in zip(bufs, counts)])
        ref = O.chain_fir_fft(O.synth_iq(0x5EED0002, 0, total), taps, 5, 1024, fused=True)
        q.put(bool(whole.shape == ref.shape and np.array_equal(whole.view(np.uint32), ref.view(np.uint32))))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [7 * 5120 + 126, 8 * 5120 + 126 + 1000])
def test_chain_time_slices_concatenate_to_the_whole_stream(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_chain_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_slice_arithmetic():
    from libredio_amd import sharding as S
    # blocks are dealt contiguously and completely
    for total in (0, 100, 126 + 5120, 126 + 5120 * 9 + 77):
        for world in (1, 2, 3, 8):
            ny = 0 if total < 127 else (total - 127) // 5 + 1
            blocks = [S.chain_slice(r, world, total, 127, 5, 1024) for r in range(world)]
            assert sum(b[3] for b in blocks) == ny // 1024
            nxt = 0
            for first, n, fb, nb in blocks:
                assert fb == nxt
                nxt += nb
                if nb:
                    assert first == fb * 5120 and n == (nb * 1024 - 1) * 5 + 127 and first + n <= total
    assert S.weak_slice(3, 10, 127, 5, 1024) == (3 * 10 * 5120, (10 * 1024 - 1) * 5 + 127)
    assert [S.channel_shard(r, 8, 256) for r in (0, 7)] == [(0, 32), (224, 32)]
    assert sum(S.channel_shard(r, 3, 256)[1] for r in range(3)) == 256
    rows = [S.channelizer_time_shard(r, 8, 1000, 16) for r in range(8)]
    assert sum(r[1] for r in rows) == 985 and all(r[2] == r[1] + 15 for r in rows)
    assert S.channelizer_exchange_layout(8, 64) == 8


def _a2a_worker(rank, world, port, q):
    import oracle as O
    from libredio_amd import sharding
    from libredio_amd.plans import channelizer_all_to_all
    _init(rank, world, port)
    P, M = 16, 64
    h = O.lpf_corrected(M * P, 0.45 / M)
    total_rows = 333                                   # input rows; ragged over two ranks
    first, nout, nin = sharding.channelizer_time_shard(rank, world, total_rows, P)
    x = O.synth_iq(0x5EED0004, M * first, M * nin)     # this rank's rows plus P-1 rows of look-ahead
    y = O.pfb_channelizer(x, h, M, P, True)            # [nout][64] (stand-in for the GPU kernel)
    assert y.shape[0] == nout
    cpg = sharding.channelizer_exchange_layout(world, M)
    grouped = torch.from_numpy(np.ascontiguousarray(y.reshape(nout, world, cpg).transpose(1, 0, 2)))  # [g][row][cpg]
    mine = channelizer_all_to_all(grouped).numpy()     # [all rows][my channels]
    whole = O.pfb_channelizer(O.synth_iq(0x5EED0004, 0, M * total_rows), h, M, P, True)
    ok = mine.shape == (total_rows - P + 1, cpg) and np.array_equal(
        mine.view(np.uint32), np.ascontiguousarray(whole[:, rank * cpg:(rank + 1) * cpg]).view(np.uint32))
    oks = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(oks, torch.tensor([int(ok)]))
    if rank == 0:
        q.put(all(int(o) == 1 for o in oks))
    dist.destroy_process_group()


def test_channelizer_all_to_all_regroups_time_shards_into_channel_shards():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_a2a_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_overlap_save_and_channel_shards_concatenate():
    """C5 / C3 partitions (no collective): the shards' outputs, concatenated in rank order,
    are the unsharded result, bit for bit (oracle on both sides: this checks the partition arithmetic)."""
    import oracle as O
    from libredio_amd import sharding as S
    nfft, k = 256, 33
    taps = O.lpf_corrected(k, 0.1)
    x = O.synth_iq(5, 0, 256 * 40 + 17)
    whole = O.overlap_save(x, taps, nfft)
    for world in (1, 2, 3, 8, 64):
        parts, nxt = [], 0
        for r in range(world):
            first, n, first_out, n_out = S.overlap_save_shard(r, world, len(x), k, nfft)
            assert first_out == nxt
            nxt += n_out
            if n:
                y = O.overlap_save(x[first:first + n], taps, nfft)
                assert len(y) == n_out
                parts.append(y)
        got = np.concatenate(parts)
        assert np.array_equal(got.view(np.uint32), whole.view(np.uint32)), world
    for world in (1, 2, 3, 8):
        shards = [S.channel_shard(r, world, 256) for r in range(world)]
        assert sum(n for _, n in shards) == 256 and shards[0][0] == 0
        assert all(shards[i][0] + shards[i][1] == shards[i + 1][0] for i in range(world - 1))
