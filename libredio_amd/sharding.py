"""How the per-sample paths shard across GPUs -- pure host arithmetic, no device code.

* C2 chain / FIR / FFT / resampler channels: independent slices, no data-path collective.  A stream
  is time-sliced on decimated-block boundaries; every slice carries its own (ntaps - 1)-sample halo,
  so the concatenation of the slices' spectra is the spectra of the whole stream.
* C4 channelizer: time-sharded analysis, then ONE all-to-all that regroups [time-shard][all channels]
  into [all time][channel group] (channelizer_exchange_layout).
"""


def chain_slice(rank, world, total_samples, ntaps, decim, nfft):
    """Slice of a `total_samples`-long stream that rank `rank` of `world` feeds to the chain.

    Returns (first_sample, n_samples, first_block, n_blocks).  Blocks are dealt contiguously, the
    remainder to the lowest ranks; a slice is exactly the input span of its blocks (halo included).
    """
    ny = 0 if total_samples < ntaps else (total_samples - ntaps) // decim + 1
    nblk = ny // nfft
    base, extra = divmod(nblk, world)
    mine = base + (1 if rank < extra else 0)
    first_block = rank * base + min(rank, extra)
    if mine == 0:
        return 0, 0, first_block, 0
    first_sample = first_block * nfft * decim
    n_samples = (mine * nfft - 1) * decim + ntaps
    return first_sample, n_samples, first_block, mine


def weak_slice(rank, blocks_per_rank, ntaps, decim, nfft):
    """bench.py's weak-scaling slice: every rank owns `blocks_per_rank` consecutive blocks of one
    ever-longer stream.  Returns (first_sample, n_samples)."""
    first_sample = rank * blocks_per_rank * nfft * decim
    return first_sample, (blocks_per_rank * nfft - 1) * decim + ntaps


def channel_shard(rank, world, nchan):
    """Independent channels (C3): contiguous groups.  Returns (first_channel, n_channels)."""
    base, extra = divmod(nchan, world)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


def channelizer_time_shard(rank, world, total_rows, ntaps_per_branch):
    """C4: output rows (time instants) are dealt contiguously; a shard reads its rows plus
    ntaps_per_branch - 1 rows of look-ahead.  Returns (first_row, n_out_rows, n_in_rows)."""
    nout = max(total_rows - ntaps_per_branch + 1, 0)
    base, extra = divmod(nout, world)
    mine = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, mine, (mine + ntaps_per_branch - 1 if mine else 0)


def channelizer_exchange_layout(world, nchan):
    """Channels per destination rank for the all-to-all (equal groups; nchan % world == 0)."""
    assert nchan % world == 0, "channel count must divide evenly across ranks"
    return nchan // world


def overlap_save_shard(rank, world, n_in, ntaps, nfft):
    """C5: overlap-save blocks are independent given their ntaps - 1 samples of input overlap, so the
    blocks are dealt contiguously and a shard reads exactly the input span of its blocks.
    Returns (first_sample, n_samples, first_out, n_out)."""
    hop = nfft - ntaps + 1
    nblk = 0 if n_in < nfft else (n_in - nfft) // hop + 1
    base, extra = divmod(nblk, world)
    mine = base + (1 if rank < extra else 0)
    first_block = rank * base + min(rank, extra)
    if mine == 0:
        return 0, 0, first_block * hop, 0
    return first_block * hop, (mine - 1) * hop + nfft, first_block * hop, mine * hop
