#!/usr/bin/env python
"""l2o_atb (split-K A^T B, fp32 MFMA) vs the library path it replaced (one chunked torch.bmm + sum):
the shapes of one meta-training step at config-2 size (R = T x 16384 rows)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from open_l2o_amd import _abi
from open_l2o_amd._engine import HipEngine, NetSpec
eng = HipEngine()
SPEC = {82: NetSpec(_abi.NET_CW, _abi.PRE_IDENTITY, (20, 20), 1.0, False),
        103: NetSpec(_abi.NET_RNNPROP, _abi.PRE_FC_ELU, (20, 20), 0.01, True)}


def lib_atb(A, B, chunk):
    R = A.shape[0]
    n = R // chunk
    out = torch.bmm(A[:n * chunk].view(n, chunk, -1).transpose(1, 2), B[:n * chunk].view(n, chunk, -1)).sum(0)
    if n * chunk < R:
        out = out + A[n * chunk:].t() @ B[n * chunk:]
    return out


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for T, KA, KB in ((20, 82, 161), (100, 82, 161), (20, 103, 181), (100, 103, 181)):
    R = T * 16384
    A = torch.randn(R, KA, device=eng.device)
    B = torch.randn(R, KB, device=eng.device)
    chunk = 4096 if R < (1 << 20) else 8192
    t_new = timeit(lambda: eng.atb(A, B))
    t_blk = timeit(lambda: eng.wgrad(SPEC[KA], A, B))            # only the weight-gradient tiles (l2o_cwlstm_wgrad)
    t_lib = timeit(lambda: lib_atb(A, B, chunk))
    err = float((eng.atb(A, B) - lib_atb(A, B, chunk)).abs().max())
    gb = 4.0 * R * (KA + KB) / 1e9
    print("T=%3d rows=%8d %3dx%3d  l2o_atb %7.1f us (%.2f TB/s, %.1f TFLOP/s)   l2o_cwlstm_wgrad %7.1f us (%.2f TB/s)   "
          "library bmm %7.1f us   max |diff| %.2g"
          % (T, R, KA, KB, t_new, gb / t_new * 1e3, 2.0 * R * KA * KB / t_new / 1e6, t_blk, gb / t_blk * 1e3, t_lib, err))
