"""How fast can torch (rocBLAS / hipBLASLt) form A^T B for tall-skinny A [N, 82], B [N, 161]?"""
import torch, time
N = 16384
A = torch.randn(N, 82, device="cuda"); B = torch.randn(N, 161, device="cuda")
def bench(fn, name):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): out = fn()
    e1.record(); torch.cuda.synchronize()
    print("%-40s %8.1f us" % (name, e0.elapsed_time(e1) * 1e3 / 50))
    return out
ref = bench(lambda: A.t() @ B, "A.t() @ B")
for ch in (64, 128, 256, 512, 1024):
    Ac = A.view(N // ch, ch, 82); Bc = B.view(N // ch, ch, 161)
    out = bench(lambda: torch.bmm(Ac.transpose(1, 2), Bc).sum(0), "bmm chunks of %d + sum" % ch)
    print("   max abs diff vs plain", float((out - ref).abs().max()))
At = A.t().contiguous()
bench(lambda: At @ B, "A^T pre-transposed contiguous @ B")
bench(lambda: torch.einsum("nk,nj->kj", A, B), "einsum")
