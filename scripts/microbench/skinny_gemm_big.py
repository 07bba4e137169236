"""A^T B for the whole-unroll row count (T * N rows): chunk size of the batched form vs time."""
import sys, torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = 16384 * T
A = torch.randn(N, 82, device="cuda"); B = torch.randn(N, 161, device="cuda")
def bench(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print("%-40s %8.1f us  %6.2f TB/s" % (name, us, N * 243 * 4 / us / 1e6))
    return out
ref = bench(lambda: A.t() @ B, "A.t() @ B")
for ch in (512, 1024, 2048, 4096, 8192, 16384, 65536):
    Ac = A.view(N // ch, ch, 82); Bc = B.view(N // ch, ch, 161)
    out = bench(lambda: torch.bmm(Ac.transpose(1, 2), Bc).sum(0), "bmm chunks of %d + sum" % ch)
    print("   max abs diff vs plain", float((out - ref).abs().max()))
