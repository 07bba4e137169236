"""Time l2o_cwlstm_step against the tile count (fixed launch cost vs per-tile cost).
Run on the GPU box:  python scripts/microbench/step_scaling.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import oracle as O
from open_l2o_amd._engine import HipEngine
from tests.helpers import spec_of, make_params

eng = HipEngine()
for name, cfg in (("dm", O.DM_IDENTITY), ("rnnprop", O.RNNPROP)):
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, make_params(cfg, 1, trained_like=True))
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
