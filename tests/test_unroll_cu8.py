"""k_unroll_cu8 (round 4; L2O_OPT_UNROLL_CU = 3 / 4 / 5 force it with 4 / 3 / 2 register-resident state tiles per wave): the
streaming fused unroll with eight waves per workgroup -- two per SIMD --, the bf16x3 fragments (PACKED, for RNNProp too) in
LDS and the LSTM state in registers (+ LDS slots for the tiles beyond KR).  Round 5: the DM nets' input-weight rows are
read from LDS, and the default (1) runs it for EVERY net and for the recording unroll wherever its LDS image fits (KR = 4
RNNProp / 3 DM plain, 2 recording).  Same contract as k_unroll_cu (csrc/l2o_unroll_cu.h): against the oracle and against
the four-wave kernel, plain and recording."""
import numpy as np
import pytest

import oracle as O
from helpers import ORACLE_CFGS, device_problem, lib_option, make_params, make_problem, max_abs, rel_err, spec_of
from open_l2o_amd import _abi
from test_hip_kernels import _run_fused

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from open_l2o_amd._engine import HipEngine
    return HipEngine()


@pytest.mark.parametrize("form", [3, 4, 5, 1])
@pytest.mark.parametrize("name,kind,B,D,M", [("rnnprop", "lasso", 3, 512, 256), ("rnnprop", "lasso", 4, 300, 100),
                                             ("dm", "lasso", 3, 512, 64), ("dm_logsign", "quadratic", 2, 256, None),
                                             ("dm_logsign", "lasso", 2, 512, 256), ("dm", "rastrigin", 2, 400, None),
                                             ("rnnprop", "rastrigin", 2, 200, None),
                                             # few tiles, many rows (the LDS-resident forms cannot hold M > 16 tiles' rows):
                                             # waves 4..7 own no tile, the stream still splits over all eight
                                             ("rnnprop", "lasso", 5, 64, 200), ("rnnprop", "lasso", 3, 20, 90)])
def test_cu8_vs_oracle_and_four_wave_kernel(eng, name, kind, B, D, M, form):
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=6, trained_like=True)
    prob, x0, arrays = make_problem(kind, B, D, seed=7, M=M)
    T = 8
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, step0=2)
    out = {}
    for f in (form, 2):
        with lib_option(_abi.OPT_UNROLL_CU, f):
            out[f] = _run_fused(eng, cfg, params, arrays, x0, B, D, T, step0=2)
    fx, x, st, m, v = out[form]
    if form == 1:                                            # the default IS the eight-wave kernel for every net now
        with lib_option(_abi.OPT_UNROLL_CU, 1):
            _run_fused(eng, cfg, params, arrays, x0, B, D, 1)
            assert eng.last_unroll_form()[0] == "k_unroll_cu8"
    assert rel_err(fx, res.fx) < 1e-5
    assert max_abs(x, res.x.reshape(B, D)) < 1e-5 * max(1.0, float(np.abs(res.x).max()))
    tol = 2e-4 if kind == "rastrigin" else 1e-5
    for l in range(2):
        for i in range(2):
            assert max_abs(st[l][i], res.state[l][i]) < tol * max(1.0, float(np.abs(res.state[l][i]).max()))
    if cfg.kind == "rnnprop":
        assert max_abs(m, res.m.reshape(B, D)) < 2e-6 * max(1.0, float(np.abs(res.m).max()))
    # the four-wave kernel on the same inputs: another summation order of the partial gradients, the packed instead of the
    # 6-product gate GEMM for RNNProp
    assert rel_err(fx, out[2][0]) < 2e-6


@pytest.mark.parametrize("name,D,form", [("rnnprop", 512, 3), ("rnnprop", 256, 1), ("dm", 512, 1), ("dm_logsign", 512, 1),
                                         ("dm", 256, 5), ("dm", 512, 4), ("rnnprop", 512, 1)])
def test_cu8_recording_equals_plain(eng, name, D, form):
    """The recording instantiations (l2o_unroll_record at streaming sizes): the default (form 1) records on the eight-wave
    kernel with two register-resident tiles per wave where that LDS image fits (D <= 256), the DM nets at D = 512 with
    three, RNNProp at D = 512 on the four-wave kernel; recorded gradients against the oracle's, x / fx against the plain
    unroll."""
    cfg = ORACLE_CFGS[name]
    rp = cfg.kind == "rnnprop"
    params = make_params(cfg, seed=61, trained_like=True)
    B, M, T = 3, 128, 4
    prob, x0, arrays = make_problem("lasso", B, D, seed=62, M=M)
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D)
    N = B * D

    def run(hist):
        x, st = eng.tensor(x0.reshape(B, D)), eng.state_alloc(B, D)
        m, v = eng.zeros(B, D), eng.zeros(B, D)
        fxp = eng.zeros((T + 1) * B)
        with lib_option(_abi.OPT_UNROLL_CU, form):
            eng.unroll(spec, wpack, pd, x, st, m if rp else None, v if rp else None, T, 2, fxp, hist=hist)
        return eng.to_numpy(x), eng.to_numpy(fxp)

    hist = {"st": eng.zeros(T, eng.state_floats(B, D)), "g": eng.zeros(T, N), "g_final": eng.zeros(N)}
    if rp:
        hist.update(m=eng.zeros(T, N), v=eng.zeros(T, N))
    x_rec, fx_rec = run(hist)
    want = "k_unroll_cu" if (form == 1 and rp and D > 256) else "k_unroll_cu8"
    assert eng.last_unroll_form()[0] == want, eng.last_unroll_form()
    x_pl, fx_pl = run(None)
    np.testing.assert_allclose(x_rec, x_pl, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(fx_rec, fx_pl, rtol=1e-6)
    g0 = prob.grad(x0).reshape(-1)
    assert max_abs(eng.to_numpy(hist["g"][0]), g0) < 2e-6 * float(np.abs(g0).max())
    gT = prob.grad(x_pl.reshape(x0.shape)).reshape(-1)
    assert max_abs(eng.to_numpy(hist["g_final"]), gT) < 2e-5 * float(np.abs(gT).max())
    eng.check_unroll_status()
