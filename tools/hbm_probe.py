"""Probe: what does this MI355X box sustain for plain streaming? (torch copy / read-only sum)"""
import torch, time
x = torch.empty(1 << 28, dtype=torch.complex64, device="cuda").normal_()
y = torch.empty_like(x)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: y.copy_(x)); print(f"copy 2GiB->2GiB: {ms:.3f} ms  {2*x.numel()*8/ms/1e9:.2f} TB/s... ", 2 * x.numel() * 8 / ms / 1e6, "GB/s")
xr = torch.view_as_real(x)
ms = t(lambda: xr.sum()); print(f"read-only sum 2GiB: {ms:.3f} ms  {x.numel()*8/ms/1e6:.0f} GB/s")
z = torch.empty(x.numel() // 5, dtype=torch.complex64, device="cuda")
ms = t(lambda: z.copy_(x[: z.numel()])); print(f"copy 0.4GiB: {ms:.3f} ms {2*z.numel()*8/ms/1e6:.0f} GB/s")
