"""A^T B when the BPTT kernel stores A and Bm TRANSPOSED ([K][R], the contraction index contiguous): time of
the chunked batched GEMM on strided views vs the row-major form."""
import sys, torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
R = 16384 * T
KA, KB = 82, 161
A = torch.randn(R, KA, device="cuda"); B = torch.randn(R, KB, device="cuda")
AT = A.t().contiguous(); BT = B.t().contiguous()
def bench(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print("%-50s %8.1f us  %6.2f TB/s" % (name, us, R * (KA + KB) * 4 / us / 1e6))
    return out
ref = bench(lambda: torch.bmm(A.view(R // 8192, 8192, KA).transpose(1, 2), B.view(R // 8192, 8192, KB)).sum(0), "row-major, chunks of 8192")
for ch in (2048, 4096, 8192, 16384, 32768):
    n = R // ch
    a = AT.view(KA, n, ch).permute(1, 0, 2)          # [n, KA, ch], strides (ch, R, 1)
    b = BT.view(KB, n, ch).permute(1, 2, 0)          # [n, ch, KB], strides (ch, 1, R)
    out = bench(lambda: torch.bmm(a, b).sum(0), "transposed storage, chunks of %d" % ch)
    print("   max abs diff", float((out - ref).abs().max()))
out = bench(lambda: AT @ BT.t(), "transposed storage, plain AT @ BT.t()")
