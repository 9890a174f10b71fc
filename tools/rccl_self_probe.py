"""A single ncclSend / ncclRecv pair above 1 GiB delivers only its first gigabyte (RCCL 2.26.6, send to self on one MI355X): the reason
comm.hip cuts every transfer into pieces of at most 512 MiB.  Runs the C-ABI exchange (which now cuts) on messages below and above the
limit; with REDIO_COMM_PIECE unset every size must come back intact."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libredio_amd as R
comm = R.Comm.single()
for rows in (1 << 20, (1 << 21) - 8, (1 << 21) + 8, 1 << 22):
    g = torch.view_as_complex(torch.randn((1, rows, 64, 2), device="cuda"))
    out = comm.exchange(g, [rows])
    torch.cuda.synchronize()
    bad = (torch.view_as_real(out) != torch.view_as_real(g[0])).any(dim=2).any(dim=1).nonzero().flatten()
    print(f"{rows} rows = {rows * 512 / 2**30:.4f} GiB: mismatching rows {bad.numel()} {bad[:3].tolist()}")
