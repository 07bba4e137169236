"""Parity in the regime the reference actually runs in: a TRAINED optimizer whose loss FALLS.

The weights are the committed `.l2l` files of tests/golden/trained/ (meta-trained on the MI355X with the repo's own
drivers; command lines in the README there).  Every test runs the full-size BASELINE configuration through the C ABI
and compares with the fp32 C oracle (and, for the long horizons, the float64 NumPy oracle) on the same inputs:

  * configs 2 / 4-shard, T = 100: the whole fx[0..T] at 1e-5 relative (the north_star bar), x_T and the LSTM state, on
    both fused forms -- the two-CU kernel (the reference's arithmetic: r = Wx - y, g = W^T r) and the one-CU kernel
    (until round 4 also an opt-in normal-matrix form, H x - q: removed with ABI v12, its measurements are
    profiles/archive_r01_r03/r03a_trained_parity_probe.txt);
  * the per-step GRADIENT of the two-CU form against the float64 gradient at the kernel's own iterates, bounded by
    the error of the reference's own fp32 arithmetic (NumPy two-pass);
  * T = 1000 and T = 10 000 (DM/train_dm.py:66, DM/evaluate_dm.py:43) against the float64 oracle, bounded by 3 x the
    drift of the fp32 oracle itself -- the default (bf16x3 gates, reference gradient arithmetic), the exact-gates option
    and the one-CU kernel;
  * config 3 (RNNProp on Lasso, T = 200): the l1 term's sign(x) makes the CONVERGED trajectory chaotic -- the fp32
    oracle started one ulp away from x_0 drifts by 2e-3 in fx after ~100 steps -- so the trajectory is held to 1e-5 on
    the prefix where that sensitivity is below 1e-6, to 3 x the oracle's own one-ulp sensitivity beyond, and the
    converged regime is checked by RE-SYNCHRONISED segments (20 steps from the oracle's x / state / moments at
    t = 100 and 160) at 1e-5.
"""
import contextlib
import os

import dill
import numpy as np
import pytest

import oracle as O
from helpers import device_problem, lib_option, make_problem, max_abs, rel_err, spec_of
from open_l2o_amd import _abi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAINED = os.path.join(ROOT, "tests", "golden", "trained")

FORMS = {"two_pass": {}, "one_cu": {_abi.OPT_PAIR: 0},
         # L2O_OPT_EXACT_GATES: the fp32-MFMA gate GEMM (bit-equal to an fmaf chain) instead of the bf16x3 split
         "two_pass_exact": {_abi.OPT_EXACT_GATES: 1},
         "one_cu_exact": {_abi.OPT_PAIR: 0, _abi.OPT_EXACT_GATES: 1}}
# Long horizons.  Every form with the reference's gradient arithmetic -- bf16x3 gates (the default) or exact gates -- is
# held to 3 x the fp32 oracles' own drift from float64.  (Until round 3 the bf16x3 kernels drifted 1.1e-5 .. 1.6e-5 at
# T = 1000: the gate bias rode in two K-slots of the gate GEMM and v_mfma_f32_16x16x32_bf16 truncates every product of
# an 8-slot group at 2^-24 of the group's largest -- the O(1) bias; as the accumulator init it is outside those sums:
# 2.9e-6.  profiles/archive_r01_r03/r03c_mfma_round_probe.txt, r03c_drift_forms.txt -> r03g_drift_forms_bias_as_acc_init.txt.)


@pytest.fixture(scope="module")
def eng():
    from open_l2o_amd._engine import HipEngine
    return HipEngine()


def load_l2l(name, key):
    with open(os.path.join(TRAINED, name, "%s.l2l-0" % key), "rb") as f:
        d = dill.load(f)
    return {k: {v: np.asarray(a, np.float32) for v, a in m.items()} for k, m in d.items()}


@contextlib.contextmanager
def form(name):
    with contextlib.ExitStack() as es:
        for o, v in FORMS[name].items():
            es.enter_context(lib_option(o, v))
        yield


def fused(eng, cfg, params, arrays, x0, B, D, T, Bg=None, state0=None, m0=None, v0=None, step0=1, hist=False):
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D, B_global=Bg)
    x = eng.tensor(np.asarray(x0, np.float32).reshape(B, D))
    st = eng.state_alloc(B, D) if state0 is None else eng.state_pack(*[eng.tensor(a) for hc in state0 for a in hc], B, D)
    m = eng.zeros(B, D) if m0 is None else eng.tensor(m0.reshape(B, D))
    v = eng.zeros(B, D) if v0 is None else eng.tensor(v0.reshape(B, D))
    fx_part, fx = eng.zeros((T + 1) * B), eng.zeros(T + 1)
    h = None
    if hist:
        h = dict(st=eng.zeros(T, eng.state_floats(B, D)), g=eng.zeros(T, B * D), g_final=eng.zeros(B * D))
    eng.unroll(spec, wpack, pd, x, st, m, v, T, step0, fx_part, hist=h)
    eng.reduce_fx(fx_part, T + 1, B, Bg or B, fx)
    eng.synchronize()
    eng.check_unroll_status()
    state = [eng.to_numpy(t).reshape(-1, 20) for t in eng.state_unpack(st, B, D)]
    return eng.to_numpy(fx), eng.to_numpy(x), state, eng.to_numpy(m), eng.to_numpy(v), h


def one_ulp(x0):
    return np.nextafter(x0, np.float32(np.inf)).astype(np.float32)


CASES = {"c2": ("quadratic", "dm_quadratic_d128", 128, 128, None, 14),
         "c4shard": ("rastrigin", "dm_rastrigin_d100", 128, 100, 1024, 16)}


@pytest.mark.parametrize("which", ["two_pass", "one_cu"])
@pytest.mark.parametrize("case", ["c2", "c4shard"])
def test_trained_full_size_trajectory(eng, case, which):
    """Configs 2 and 4 (this GPU's shard of the 1024 problems), T = 100, the trained L2O-DM optimizer: the loss FALLS
    by more than 5x and the whole trajectory matches the C oracle at 1e-5; x_T and the LSTM state within the oracle's
    own sensitivity to a one-ulp change of x_0 (x 3; never tighter than 1e-5 of the largest entry)."""
    from oracle.c_oracle import c_unroll
    kind, wname, B, D, Bg, seed = CASES[case]
    cfg = O.DM_IDENTITY
    params = load_l2l(wname, "cw")
    prob, x0, arrays = make_problem(kind, B, D, seed=seed)
    T = 100
    fx_ref, x_ref, st_ref = c_unroll(kind, cfg, params, arrays, x0, T, B_global=Bg)[:3]
    fx_p, x_p, st_p = c_unroll(kind, cfg, params, arrays, one_ulp(x0), T, B_global=Bg)[:3]
    assert fx_ref[-1] < fx_ref[0] / 5, (fx_ref[0], fx_ref[-1])
    with form(which):
        fx, x, st, _, _, _ = fused(eng, cfg, params, arrays, x0, B, D, T, Bg=Bg)
    e = rel_err(fx, fx_ref)
    xs = max(1.0, float(np.abs(x_ref).max()))
    ex, env_x = max_abs(x, x_ref) / xs, max_abs(x_p, x_ref) / xs
    st_ref = [st_ref[0][0], st_ref[0][1], st_ref[1][0], st_ref[1][1]]
    st_p = [st_p[0][0], st_p[0][1], st_p[1][0], st_p[1][1]]
    es = max(max_abs(a, b) for a, b in zip(st, st_ref))
    env_s = max(max_abs(a, b) for a, b in zip(st_p, st_ref))
    print("%s %s: fx %.5g -> %.5g, rel fx %.3g, x_T %.3g (oracle one-ulp sensitivity %.3g), state %.3g (%.3g)"
          % (case, which, fx_ref[0], fx_ref[-1], e, ex, env_x, es, env_s))
    assert np.all(np.isfinite(fx)) and e < 1e-5
    assert fx[-1] < fx[0] / 5
    assert ex < max(1e-5, 3 * env_x)
    assert es < max(1e-5, 3 * env_s)


def grad64(kind, prob, x, Bg):
    x = x.astype(np.float64)
    if kind == "quadratic":
        W, y = prob.w.astype(np.float64), prob.y.astype(np.float64)
        r = np.einsum("bmd,bd->bm", W, x) - y
        return 2.0 / Bg * np.einsum("bmd,bm->bd", W, r)
    A, Bv, Cv = prob.A.astype(np.float64), prob.B.astype(np.float64)[..., 0], prob.C.astype(np.float64)[..., 0]
    r = np.einsum("bmd,bd->bm", A, x) - Bv
    return (np.einsum("bmd,bm->bd", A, r) + 2 * np.pi * prob.alpha * Cv * np.sin(2 * np.pi * x)) / Bg


def grad32_reference_form(kind, prob, x, Bg):
    """The reference's own fp32 arithmetic (DM/problems.py:98-99 + autodiff): r = Wx - y, then W^T r."""
    f = np.float32
    if kind == "quadratic":
        r = np.einsum("bmd,bd->bm", prob.w, x).astype(f) - prob.y
        return (f(2.0) / f(Bg) * np.einsum("bmd,bm->bd", prob.w, r)).astype(f)
    A, Bv, Cv = prob.A, prob.B[..., 0], prob.C[..., 0]
    r = np.einsum("bmd,bd->bm", A, x).astype(f) - Bv
    return ((np.einsum("bmd,bm->bd", A, r) + f(2 * np.pi * prob.alpha) * Cv * np.sin(f(2 * np.pi) * x)) / f(Bg)).astype(f)


@pytest.mark.parametrize("case", ["c2", "c4shard"])
def test_gradient_error_in_the_converged_regime(eng, case):
    """The recorded per-step gradient (hist["g"][t] of l2o_unroll_record) of the two-CU form against the float64
    gradient at the kernel's OWN iterate x_t, t = 0 .. 99: within 2 x the error of a NumPy fp32 evaluation of the same
    formula (r = Wx - y, g = W^T r -- an error that shrinks with the residual)."""
    kind, wname, B, D, Bg, seed = CASES[case]
    cfg = O.DM_IDENTITY
    params = load_l2l(wname, "cw")
    prob, x0, arrays = make_problem(kind, B, D, seed=seed)
    x0 = x0.reshape(B, D)
    T = 100
    out = {}
    for which in ("two_pass",):
        with form(which):
            h = fused(eng, cfg, params, arrays, x0, B, D, T, Bg=Bg, hist=True)[5]
            hg = eng.to_numpy(h["g"]).reshape(T, B, D)
            rows = []
            for t in (0, 5, 20, 50, 99):
                xt = fused(eng, cfg, params, arrays, x0, B, D, t, Bg=Bg)[1] if t else x0
                g64 = grad64(kind, prob, xt, Bg or B)
                g32 = grad32_reference_form(kind, prob, xt.astype(np.float32), Bg or B)
                n = float(np.linalg.norm(g64))
                rows.append((t, n, float(np.linalg.norm(hg[t] - g64)) / n, float(np.linalg.norm(g32 - g64)) / n))
        out[which] = rows
        for r in rows:
            print("%s %-8s t=%3d |g|=%9.4g  rel err HIP %.3g   NumPy fp32 (reference form) %.3g" % ((case, which) + r))
    assert out["two_pass"][-1][1] < out["two_pass"][0][1] / 20          # the gradient did shrink
    for t, n, e_hip, e_np in out["two_pass"]:
        assert e_hip < 2 * e_np + 1e-7, (t, e_hip, e_np)


@pytest.mark.parametrize("which", ["two_pass", "one_cu", "two_pass_exact", "one_cu_exact"])
@pytest.mark.parametrize("case,T,Bt", [("c2", 1000, 16), ("c2", 10000, 4), ("c4shard", 1000, 16), ("c4shard", 10000, 4)])
def test_trained_long_horizon(eng, case, T, Bt, which):
    """T = 1000 (the curriculum's horizon, DM/train_dm.py:66) and T = 10 000 (DM/evaluate_dm.py:43) in ONE launch, the
    trained optimizer, a slice of the batch with the 1/B of the full batch: the loss keeps falling.  HIP vs the float64
    oracle within 3 x the worst drift of two fp32 evaluations of the same unroll from their float64 twin (the C oracle,
    the C oracle started one ulp away) -- the default bf16x3 gates and the exact gates alike; the first 101 steps of
    every form hold the 1e-5 of the short tests."""
    from oracle.c_oracle import c_unroll
    kind, wname, B, D, Bg, seed = CASES[case]
    Bg = Bg or B
    cfg = O.DM_IDENTITY
    params = load_l2l(wname, "cw")
    prob, x0, arrays = make_problem(kind, B, D, seed=seed)
    arr = {k: (v[:Bt] if isinstance(v, np.ndarray) else v) for k, v in arrays.items()}
    x0 = x0.reshape(B, -1)[:Bt]
    p64 = {k: {v: a.astype(np.float64) for v, a in d.items()} for k, d in params.items()}
    if kind == "quadratic":
        pr = O.Quadratic(prob.w[:Bt].astype(np.float64), prob.y[:Bt].astype(np.float64), batch_global=Bg)
        x64 = x0.astype(np.float64)
    else:
        pr = O.Rastrigin(prob.A[:Bt].astype(np.float64), prob.B[:Bt].astype(np.float64), prob.C[:Bt].astype(np.float64),
                         alpha=prob.alpha, batch_global=Bg)
        x64 = x0.astype(np.float64).reshape(Bt, D, 1)
    r64 = O.unroll(pr, cfg, p64, x64, O.net_initial_state(cfg, Bt * D, np.float64), T)
    fx32 = c_unroll(kind, cfg, params, arr, x0, T, B_global=Bg)[0]
    fx32p = c_unroll(kind, cfg, params, arr, one_ulp(x0), T, B_global=Bg)[0]
    env = max(rel_err(fx32, r64.fx), rel_err(fx32p, r64.fx))
    with form(which):
        fx = fused(eng, cfg, params, arr, x0, Bt, D, T, Bg=Bg)[0]
    e64 = rel_err(fx, r64.fx)
    print("%s T=%d %s: fx %.4g -> %.4g; HIP vs float64 %.3g (fp32 oracles' drift %.3g), first 101 steps %.3g"
          % (case, T, which, r64.fx[0], r64.fx[-1], e64, env, rel_err(fx[:101], r64.fx[:101])))
    assert r64.fx[-1] < r64.fx[100] < r64.fx[0]
    assert rel_err(fx[:101], r64.fx[:101]) < 1e-5
    assert e64 < 3 * env, (e64, env)


def test_c3_trained_lasso_rnnprop(eng):
    """Config 3 with the trained RNNProp optimizer (f: 44.7 -> 1.6 in 200 steps).  See the module docstring: the
    converged trajectory is chaotic (sign(x) of the l1 term + a +-0.01 tanh step), so parity is (a) 1e-5 on the prefix
    where the oracle's own one-ulp sensitivity is below 1e-6, (b) within 3 x that sensitivity beyond, (c) 1e-5 on
    re-synchronised 20-step segments started from the oracle's x / LSTM state / moments at t = 100 and t = 160."""
    from oracle.c_oracle import c_unroll
    cfg = O.RNNPROP
    params = load_l2l("rnnprop_lasso_256x512", "rp")
    B, D, M, T = 256, 512, 256, 200
    prob, x0, arrays = make_problem("lasso", B, D, seed=18, M=M)
    fx_ref = c_unroll("lasso", cfg, params, arrays, x0, T)[0]
    fx_p = c_unroll("lasso", cfg, params, arrays, one_ulp(x0), T)[0]
    assert fx_ref[-1] < fx_ref[0] / 5
    sens = np.abs(fx_p.astype(np.float64) - fx_ref) / np.abs(fx_ref)
    fx = fused(eng, cfg, params, arrays, x0, B, D, T)[0]
    err = np.abs(fx.astype(np.float64) - fx_ref) / np.abs(fx_ref)
    stable = int(np.argmax(np.maximum.accumulate(sens) > 1e-6)) if np.any(sens > 1e-6) else T + 1
    print("C3 trained: fx %.4g -> %.4g; one-ulp sensitivity of the oracle exceeds 1e-6 from step %d (max %.3g); HIP: "
          "prefix %.3g, whole trajectory %.3g" % (fx_ref[0], fx_ref[-1], stable, sens.max(), err[:stable].max(), err.max()))
    assert stable >= 40
    assert err[:stable].max() < 1e-5
    assert err.max() < max(1e-5, 3 * sens.max())
    assert fx[-1] < fx[0] / 5
    # (c) re-synchronised segments in the converged regime
    for t0 in (100, 160):
        _, x_t, st_t, m_t, v_t, _ = c_unroll("lasso", cfg, params, arrays, x0, t0)
        seg_ref, xs_ref = c_unroll("lasso", cfg, params, arrays, x_t, 20, state0=st_t, m0=m_t, v0=v_t, step0=1 + t0)[:2]
        seg, xs, _, _, _, _ = fused(eng, cfg, params, arrays, x_t, B, D, 20, state0=st_t, m0=m_t, v0=v_t, step0=1 + t0)
        seg_p = c_unroll("lasso", cfg, params, arrays, one_ulp(x_t), 20, state0=st_t, m0=m_t, v0=v_t, step0=1 + t0)[0]
        e, ex = rel_err(seg, seg_ref), max_abs(xs, xs_ref) / max(1.0, float(np.abs(xs_ref).max()))
        print("   segment from the oracle's state at t=%d: rel fx %.3g (oracle one-ulp sensitivity %.3g), x %.3g (f %.4g -> %.4g)"
              % (t0, e, rel_err(seg_p, seg_ref), ex, seg_ref[0], seg_ref[-1]))
        assert e < max(1e-5, 3 * rel_err(seg_p, seg_ref))


def test_c3_teacher_forced_every_step(eng):
    """VERDICT r03 "the one place the 1e-5 bar is not asserted": the CONVERGED regime of trained config 3 (RNNProp on
    Lasso 256 x 512), where whole-trajectory parity can only be stated relative to the oracle's own chaos.  Here NO
    sensitivity escape hatch: at EVERY step t of the trained T = 200 trajectory the HIP path starts from the C oracle's
    own (x_t, LSTM state_t, m_t, v_t, step 1 + t) and takes ONE step -- the streaming fused kernel (k_unroll_cu, T = 1)
    and the step-granular kernels (l2o_problem_fg + l2o_cwlstm_step) -- and is compared with the oracle's step:
    f(x_t), f(x_{t+1}) 1e-6 relative, x_{t+1} 1e-6 of max |x|, the moments 2e-6 of their largest entry, the new LSTM
    state 3e-6 absolute (measured: 6.6e-7 / 1.5e-8 / 1.4e-6 / 2.4e-6).  The state and moment figures are above 1e-6
    because RNNProp's inputs g / (sqrt(v^) + 1e-8) turn the fp32 summation-order difference of the 256-row GEMV (1e-6
    of max |g|) into an O(1e-6) difference of an O(1) network input wherever |g| is small; that this is the arithmetic's
    own floor and not a kernel defect is asserted on every 8th step against the FLOAT64 oracle's step from the same
    inputs: the HIP step is within 2 x the fp32 C oracle's own distance from float64 (+ 2e-7) in every quantity.
    The oracle's chained single steps are checked to BE its T = 200 unroll, bit for bit.
    DM/meta_rnnprop_train.py:371-395, DM/problems.py:103-131."""
    from oracle.c_oracle import c_unroll
    cfg = O.RNNPROP
    params = load_l2l("rnnprop_lasso_256x512", "rp")
    p64 = {k: {v: a.astype(np.float64) for v, a in d.items()} for k, d in params.items()}
    B, D, M, T = 256, 512, 256, 200
    prob, x0, arrays = make_problem("lasso", B, D, seed=18, M=M)
    prob64 = O.Lasso(prob.w.astype(np.float64), prob.y.astype(np.float64), l=prob.l)
    fx_whole, x_whole = c_unroll("lasso", cfg, params, arrays, x0, T)[:2]
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D)
    b95 = float(np.float32(0.95))
    fx_part, fx = eng.zeros(2 * B), eng.zeros(2)
    f_s, g_s = eng.zeros(B), eng.zeros(B, D)
    x_t = np.asarray(x0, np.float32).reshape(B, D)
    st_t = tuple((np.zeros((B * D, 20), np.float32), np.zeros((B * D, 20), np.float32)) for _ in range(2))
    m_t, v_t = np.zeros((B, D), np.float32), np.zeros((B, D), np.float32)
    keys = [a + "_" + b for a in ("fused", "step") for b in ("fx", "x", "state", "mv")]
    worst, where = dict.fromkeys(keys, 0.0), dict.fromkeys(keys, -1)
    worst64 = {k: (0.0, 0.0) for k in keys}                 # (HIP vs float64, C oracle vs float64) at the worst ratio
    chained = []

    def errs(r, fx_r, x_r, st_r, m_r, v_r):
        xs = max(1.0, float(np.abs(x_r).max()))
        return dict(fx=rel_err(r["fx"], fx_r), x=max_abs(r["x"], x_r) / xs,
                    state=max(max_abs(a, b) for a, b in zip(r["st"], st_r)),
                    mv=max(max_abs(r["m"], m_r) / max(1e-30, float(np.abs(m_r).max())),
                           max_abs(r["v"], v_r) / max(1e-30, float(np.abs(v_r).max()))))

    for t in range(T):
        seg, x_n, st_n, m_n, v_n, _ = c_unroll("lasso", cfg, params, arrays, x_t, 1, state0=st_t, m0=m_t, v0=v_t,
                                              step0=1 + t)
        chained.append(seg[0])
        st_ref = [st_n[0][0], st_n[0][1], st_n[1][0], st_n[1][1]]
        xd0 = eng.tensor(x_t)
        std0 = eng.state_pack(*[eng.tensor(a) for hc in st_t for a in hc], B, D)
        md0, vd0 = eng.tensor(m_t), eng.tensor(v_t)
        # (a) the streaming fused kernel, one step
        xd, std, md, vd = xd0.clone(), std0.clone(), md0.clone(), vd0.clone()
        eng.unroll(spec, wpack, pd, xd, std, md, vd, 1, 1 + t, fx_part)
        eng.reduce_fx(fx_part, 2, B, B, fx)
        got = dict(fx=eng.to_numpy(fx).copy(), x=eng.to_numpy(xd), m=eng.to_numpy(md), v=eng.to_numpy(vd),
                   st=[eng.to_numpy(a).reshape(-1, 20) for a in eng.state_unpack(std, B, D)])
        # (b) the step-granular kernels, one step
        xd, std, md, vd = xd0, std0, md0, vd0
        eng.problem_fg(pd, xd, f_s, g_s)
        eng.reduce_fx(f_s, 1, B, B, fx[0:1])
        eng.lstm_step(spec, wpack, g_s, md, vd, b95 ** (1 + t), b95 ** (1 + t), std, xd, B, D)
        eng.problem_fg(pd, xd, f_s, None)
        eng.reduce_fx(f_s, 1, B, B, fx[1:2])
        got2 = dict(fx=eng.to_numpy(fx).copy(), x=eng.to_numpy(xd), m=eng.to_numpy(md), v=eng.to_numpy(vd),
                    st=[eng.to_numpy(a).reshape(-1, 20) for a in eng.state_unpack(std, B, D)])
        r64 = None
        if t % 8 == 0 or t == T - 1:                        # the same step in float64 from the same fp32 inputs
            s64 = tuple((h.astype(np.float64), c.astype(np.float64)) for h, c in st_t)
            u = O.unroll(prob64, cfg, p64, x_t.astype(np.float64), s64, 1, m0=m_t.astype(np.float64),
                         v0=v_t.astype(np.float64), step0=1 + t)
            r64 = (u.fx, u.x.reshape(B, D), [u.state[0][0], u.state[0][1], u.state[1][0], u.state[1][1]],
                   u.m.reshape(B, D), u.v.reshape(B, D))
            c64 = errs(dict(fx=seg, x=x_n, st=st_ref, m=m_n, v=v_n), *r64)
        for tag, r in (("fused", got), ("step", got2)):
            e = errs(r, seg, x_n, st_ref, m_n, v_n)
            for k, val in e.items():
                if val > worst[tag + "_" + k]:
                    worst[tag + "_" + k], where[tag + "_" + k] = val, t
            if r64 is not None:
                e64 = errs(r, *r64)
                for k in e64:
                    assert e64[k] <= 2 * c64[k] + 2e-7, (tag, k, t, e64[k], c64[k])
                    if e64[k] - 2 * c64[k] > worst64[tag + "_" + k][0] - 2 * worst64[tag + "_" + k][1] or worst64[tag + "_" + k] == (0.0, 0.0):
                        worst64[tag + "_" + k] = (e64[k], c64[k])
        x_t, st_t, m_t, v_t = x_n, st_n, m_n, v_n
    chained.append(seg[1])
    eng.check_unroll_status()
    print("C3 trained, teacher-forced at every t of %d (f %.4g -> %.4g): worst vs the C oracle's step: " % (T, fx_whole[0], fx_whole[-1]) +
          ", ".join("%s %.3g (t=%d)" % (k, v, where[k]) for k, v in sorted(worst.items())))
    print("   vs the float64 step (every 8th t; HIP / fp32 C oracle): " +
          ", ".join("%s %.3g / %.3g" % ((k,) + v) for k, v in sorted(worst64.items())))
    assert np.array_equal(np.asarray(chained, np.float32), fx_whole) and np.array_equal(x_t, x_whole)
    assert fx_whole[-1] < fx_whole[0] / 5
    bound = dict(fx=1e-6, x=1e-6, state=3e-6, mv=2e-6)
    for k, v in worst.items():
        assert v < bound[k.split("_")[1]], (k, v, where[k])


def _c5_hip(hip, data, params, batch, T, idx, start=None, step0=1):
    """T steps of the trained RNNProp optimizer on the MLP optimizee through the product API (ONE l2o_mlp_unroll
    launch); start = None (the problem's own initial weights, zero state) or the oracle's (variables, LSTM states,
    m, v) to continue from.  Returns (initial variables, fx[0..T], final variables)."""
    from open_l2o_amd import meta, meta_rnnprop_eval, problems
    from open_l2o_amd.session import Session
    cfg = O.RNNPROP
    calls = {"n": 0}

    def sampler(n_evals, b, n_data):
        out = idx[calls["n"]:calls["n"] + n_evals]
        calls["n"] += n_evals
        return out

    meta.set_random_seed(19)
    problem = problems.mnist(layers=(20,), batch_size=batch, data=data, sampler=sampler)
    opts = {"layers": cfg.layers, "initializer": params, "preprocess_name": "fc", "preprocess_options": {"dim": 20},
            "scale": cfg.scale, "tanh_output": True}
    opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, rp={"net": "RNNprop", "net_options": opts})
    ml, _, _, step = opt.meta_loss(problem, T)
    g = opt.graph
    with Session() as sess:
        sess.run(ml.reset)
        if start is not None:
            vs0, sts0, ms0, vv0 = start
            for var, a in zip(g.x, vs0):
                var.load(a)
            for sl in g.slots:
                j = sl.var_index
                sl.state.load(sts0[j])
                sl.m, sl.v = hip.tensor(ms0[j].reshape(1, -1)), hip.tensor(vv0[j].reshape(1, -1))
        v0 = [v.eval() for v in g.x]
        sess.run([ml.fx, ml.update], feed_dict={step: step0})
        fx = hip.to_numpy(g._fx_cache[T]["bufs"][0]).copy()
        xT = [v.eval() for v in g.x]
    assert g.last_path == "mlp_unroll"
    return v0, fx, xT


def test_c5_trained_rnnprop_on_the_mlp_optimizee():
    """Config 5 in the CONVERGING regime (VERDICT r03 missing item 3): the committed RNNProp optimizer meta-trained on
    the 784-20-10 MLP optimizee (tests/golden/trained/rnnprop_mnist_mlp, scripts/train_rnnprop.py --problem mnist on
    the label-noised synthetic digits bench.py uses), minibatch 64, T = 200, ONE persistent launch (l2o_mlp_unroll)
    through the product API, against the oracle's multi-variable RNNProp unroll (O.unroll_multi: one pair of moments
    and one LSTM state per variable, DM/meta_rnnprop_train.py:371-395, DM/problems.py:254-288) on the same minibatch
    sequence.  The loss FALLS (2.30 -> ~0.35).  RNNProp's inputs g / sqrt(v) amplify rounding differences of near-zero
    gradients, so -- as for config 3 -- the whole trajectory is chaotic past ~50 steps (the oracle started one ulp
    away drifts by 3e-4); asserted: (a) 1e-5 on the prefix where that sensitivity is below 1e-6, (b) 1e-5 (or 3 x the
    segment's own one-ulp sensitivity) on TEN re-synchronised 20-step segments that start from the oracle's variables /
    LSTM states / moments at t = 0, 20, ..., 180 and together cover the whole converged regime."""
    from open_l2o_amd import _engine, problems
    hip = _engine.HipEngine()
    old = _engine._default_engine
    _engine.set_default_engine(hip)
    try:
        data = problems.synthetic_mnist(4096, seed=5, label_noise=0.1)      # (bench.py's config-5 data)
        T, batch, L = 200, 64, 20
        idx = np.random.default_rng(170).integers(0, 4096, size=(T + 1, batch))
        cfg = O.RNNPROP
        params = load_l2l("rnnprop_mnist_mlp", "rp")
        ref = O.MnistMLP(data["images"], data["labels"].astype(np.int32), "sigmoid")
        v0, fx, xT = _c5_hip(hip, data, params, batch, T, idx)
        fg = lambda vs, t, wg: ref.fg(vs, idx[t], wg)
        states = [O.net_initial_state(cfg, a.size) for a in v0]
        fx_ref = O.unroll_multi(fg, cfg, params, v0, states, T)[0]
        fx_p = O.unroll_multi(fg, cfg, params, [one_ulp(a) for a in v0], states, T)[0]
        sens = np.abs(fx_p.astype(np.float64) - fx_ref) / np.abs(fx_ref)
        err = np.abs(fx.astype(np.float64) - fx_ref) / np.abs(fx_ref)
        env = np.maximum.accumulate(sens)
        stable = int(np.argmax(env > 1e-6)) if np.any(env > 1e-6) else T + 1
        print("C5 trained: fx %.4g -> %.4g (min %.4g); oracle one-ulp sensitivity > 1e-6 from step %d (max %.3g); HIP: "
              "prefix %.3g, whole trajectory %.3g" % (fx_ref[0], fx_ref[-1], fx_ref.min(), stable, sens.max(),
                                                       err[:stable].max() if stable else 0.0, err.max()))
        assert fx_ref[-1] < 0.6 * fx_ref[0] and fx[-1] < 0.6 * fx[0]      # the trained optimizer does optimize
        assert stable >= 20 and err[:stable].max() < 1e-5
        # (b) re-synchronised segments: the oracle's chained 20-step unrolls ARE its 200-step unroll (checked)
        cur = ([a.copy() for a in v0], states, None, None)
        chained, worst = [], 0.0
        for k in range(T // L):
            t0 = k * L
            fgk = lambda vs, t, wg, _t0=t0: ref.fg(vs, idx[_t0 + t], wg)
            seg_ref, v_n, st_n, m_n, vv_n = O.unroll_multi(fgk, cfg, params, cur[0], cur[1], L, ms=cur[2], vs=cur[3],
                                                           step0=1 + t0, return_moments=True)
            seg_p = O.unroll_multi(fgk, cfg, params, [one_ulp(a) for a in cur[0]], cur[1], L, ms=cur[2], vs=cur[3],
                                   step0=1 + t0)[0]
            start = None if k == 0 else (cur[0], cur[1], cur[2], cur[3])
            _, seg, xs = _c5_hip(hip, data, params, batch, L, idx[t0:t0 + L + 1], start=start, step0=1 + t0)
            e, es = rel_err(seg, seg_ref), rel_err(seg_p, seg_ref)
            ex = max(max_abs(a.reshape(b.shape), b) for a, b in zip(xs, v_n))
            worst = max(worst, e)
            print("   segment from the oracle's state at t=%3d: rel fx %.3g (oracle one-ulp sensitivity %.3g), weights %.3g "
                  "(f %.4g -> %.4g)" % (t0, e, es, ex, seg_ref[0], seg_ref[-1]))
            assert e < max(1e-5, 3 * es), (t0, e, es)
            chained.extend(seg_ref[:L])
            cur = (v_n, st_n, m_n, vv_n)
        chained.append(seg_ref[L])
        assert np.array_equal(np.asarray(chained, np.float32), fx_ref)
        print("   worst segment: %.3g" % worst)
    finally:
        _engine.set_default_engine(old)
