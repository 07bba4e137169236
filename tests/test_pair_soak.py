"""Soak test of the two-CU unroll's exchange (csrc/l2o_unroll_pair.h).

The halves of a problem swap partial residuals through self-validating 8-byte granules.  Two
publish forms exist: agent-scope (sc1, write-through) stores -- inside the HIP memory model --
and, for partners that CONFIRMED via the XCC_ID handshake that they share an XCD, plain stores
that stay in that XCD's L2 where the partner's L1-bypassing poll reads them (the way agent scope
is implemented on single-L2 parts; L2O_OPT_PAIR_PLAIN_STORES, +6 %).  This test launches the pair
kernel 10 000 times while a second stream keeps the CUs unevenly busy, and checks on the device,
launch by launch, that every output word equals the agent-scope reference bit for bit and that
the sticky status word never reports a partner timeout.  A stale or torn granule, a granule of an
earlier launch taken for the partner's, or a partner that was not co-resident for longer than the
spin bound would each show up here."""
import numpy as np
import pytest
import torch

import oracle as O
from helpers import device_problem, lib_option, make_params, make_problem, rel_err, spec_of
from open_l2o_amd import _abi

pytestmark = pytest.mark.gpu

N_LAUNCHES = int(__import__("os").environ.get("L2O_SOAK_LAUNCHES", "10000"))   # (profiles/archive_r01_r03/r02n: one run with 200 000)


@pytest.mark.parametrize("name,kind,D,B", [("dm", "quadratic", 128, 128), ("rnnprop", "rastrigin", 100, 128)])
def test_pair_exchange_soak(name, kind, D, B):
    from open_l2o_amd._engine import HipEngine
    _soak(HipEngine(), name, kind, D, B)


def _soak(eng, name, kind, D, B):
    cfg = {"dm": O.DM_IDENTITY, "rnnprop": O.RNNPROP}[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=11, trained_like=True)
    T = 6
    prob, x0, arrays = make_problem(kind, B, D, seed=12)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D)
    x0d = eng.tensor(x0.reshape(B, D))
    x, st = x0d.clone(), eng.state_alloc(B, D)
    m, v = eng.zeros(B, D), eng.zeros(B, D)
    fx_part = eng.zeros((T + 1) * B)

    def launch(step0):
        x.copy_(x0d); st.zero_(); m.zero_(); v.zero_()
        eng.unroll(spec, wpack, pd, x, st, m, v, T, step0, fx_part)

    # references: agent-scope stores (bit reference) and the oracle (tolerance)
    with lib_option(_abi.OPT_PAIR_PLAIN_STORES, 0):
        launch(1)
        ref = [t.clone() for t in (fx_part, x, st, m, v)]
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    fx_ref = eng.to_numpy(ref[0]).reshape(T + 1, B).sum(1) / B
    assert rel_err(fx_ref, res.fx) < 1e-5

    # the hammer: a second stream that keeps a varying part of the chip busy (uneven load)
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=eng.device)
    small = torch.randn(512, 4096, device=eng.device)
    sink = torch.empty(4096, 4096, device=eng.device)
    sink_s = torch.empty(512, 4096, device=eng.device)
    mism = torch.zeros((), dtype=torch.int64, device=eng.device)
    outs = (fx_part, x, st, m, v)
    for i in range(N_LAUNCHES):
        if i % 40 == 0:
            with torch.cuda.stream(side):
                torch.mm(big, big, out=sink)                 # ~1.5 ms on every CU
                for _ in range(4):
                    torch.mm(small, big, out=sink_s)         # short kernels on a few CUs
        launch(1)
        for o, r in zip(outs, ref):
            mism += (o != r).sum()
        if i % 2500 == 2499:
            assert int(mism.item()) == 0, "a launch <= %d differs from the agent-scope reference" % i
            eng.check_unroll_status()
    torch.cuda.synchronize()
    assert int(mism.item()) == 0
    eng.check_unroll_status()
    hdr = eng._last_ws[:8].cpu().numpy().view(np.uint32)
    assert hdr[0] == 0 and hdr[1] >= N_LAUNCHES              # status clean, launch sequence advanced by the kernels
    print("pair soak %s/%s: %d launches under a concurrent stream, 0 mismatching words, no timeout" % (name, kind, N_LAUNCHES))
