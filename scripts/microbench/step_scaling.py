"""Time l2o_cwlstm_step against the tile count (fixed launch cost vs per-tile cost).
Run on the GPU box:  python scripts/microbench/step_scaling.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from open_l2o_amd import _abi, networks
from open_l2o_amd._engine import HipEngine, NetSpec

eng = HipEngine()
NETS = (("dm", "CoordinateWiseDeepLSTM", {"layers": (20, 20)}, NetSpec(_abi.NET_CW, _abi.PRE_IDENTITY, (20, 20), 1.0, False)),
        ("rnnprop", "RNNprop", {"layers": (20, 20), "preprocess_name": "fc", "preprocess_options": {"dim": 20},
                                "scale": 0.01, "tanh_output": True},
         NetSpec(_abi.NET_RNNPROP, _abi.PRE_FC_ELU, (20, 20), 0.01, True)))
for name, cls, opts, spec in NETS:
    net = networks.factory(cls, opts)
    wpack = eng.pack_weights(spec, {m: {v: np.array(a) for v, a in d.items()} for m, d in net.variables.items()})
    for B, D in ((1, 16), (64, 16), (256, 64), (256, 128), (256, 256), (256, 512), (1024, 512)):
        g = eng.tensor((np.random.default_rng(0).standard_normal((B, D)) * 0.1).astype(np.float32))
        m, v, x = eng.zeros(B, D), eng.zeros(B, D), eng.zeros(B, D)
        st = eng.state_alloc(B, D)
        for _ in range(5):
            eng.lstm_step(spec, wpack, g, m, v, 0.95, 0.95, st, x, B, D)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            eng.lstm_step(spec, wpack, g, m, v, 0.95, 0.95, st, x, B, D)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        tiles = B * ((D + 15) // 16)
        print("%-8s B=%5d D=%4d tiles=%6d  %8.2f us/launch  %7.3f us per tile-per-SIMD" % (name, B, D, tiles, us, us / max(1.0, tiles / 1024)))
