"""CPU-side checks of the C-ABI library (no GPU compute is launched):
 * it loads and exports every symbol include/l2o_abi.h declares;
 * the host weight packer (l2o_wpack_host) + the documented MFMA operand layout
   reproduce the oracle's net_apply when emulated lane by lane in NumPy;
 * size queries and argument validation.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as O
from helpers import ORACLE_CFGS, make_params, random_state, spec_of
import mfma_emulator as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from open_l2o_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _abi.lib()


def test_header_symbols_exported(lib):
    from open_l2o_amd import _abi
    hdr = open(os.path.join(ROOT, "include", "l2o_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(l2o_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_abi.SYMBOLS), declared ^ set(_abi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.l2o_abi_version() == _abi.L2O_ABI_VERSION


def test_struct_sizes_match_header():
    from open_l2o_amd import _abi
    # 6 int32 + 4 double + the options word ; 6 int32 + 2 double + 4 pointers ; 8 int32 + 2 pointers
    assert C.sizeof(_abi.NetCfg) == 6 * 4 + 4 * 8 + 8
    assert C.sizeof(_abi.Problem) == 6 * 4 + 2 * 8 + 4 * 8
    assert C.sizeof(_abi.Mlp) == 8 * 4 + 2 * 8


def test_options_are_caller_owned():
    """ABI v9: the library has no option state.  The binding keeps the caller's switches and encodes them into every
    l2o_net_cfg (L2O_OPTW words), l2o_problem.flags and l2o_mlp.flags."""
    from open_l2o_amd import _abi
    from open_l2o_amd._engine import NetSpec
    spec = NetSpec(_abi.NET_CW, _abi.PRE_IDENTITY, (20, 20))
    assert _abi.options_word() == 0 and spec.to_c().options == 0
    assert _abi.get_option(_abi.OPT_PAIR) == 1 and _abi.get_option(_abi.OPT_EXACT_GATES) == 0
    old = _abi.set_option(_abi.OPT_PAIR, 0)
    try:
        assert old == 1 and spec.to_c().options == (8 | 0) << (4 * _abi.OPT_PAIR)
        _abi.set_option(_abi.OPT_EXACT_GATES, 1)
        assert spec.to_c().options == ((8 | 0) << (4 * _abi.OPT_PAIR)) | ((8 | 1) << (4 * _abi.OPT_EXACT_GATES))
        _abi.set_option(_abi.OPT_BWD_BLOCKS, 300)
        assert (spec.to_c().options >> 48) == 300
        # option 12 must not land in the count's bits 48-63 (it did for a while in round 4): its field is the one that
        # option 5 -- the count, which has its own 16 bits -- leaves unused (include/l2o_abi.h: L2O_OPT_FIELD_)
        w0 = spec.to_c().options
        _abi.set_option(_abi.OPT_ONE_LDS, 2)
        w1 = spec.to_c().options
        assert (w1 >> 48) == 300 and (w1 ^ w0) == (8 | 2) << (4 * _abi.OPT_BWD_BLOCKS)
    finally:
        _abi.set_option(_abi.OPT_PAIR, 1)
        _abi.set_option(_abi.OPT_EXACT_GATES, 0)
        _abi.set_option(_abi.OPT_BWD_BLOCKS, 0)
        _abi.set_option(_abi.OPT_ONE_LDS, _abi.OPT_DEFAULTS[_abi.OPT_ONE_LDS])
    assert _abi.options_word() == 0
    with pytest.raises(ValueError):
        _abi.set_option(1000, 1)
    assert _abi.get_option(-3) == -1


def test_size_queries(lib):
    assert lib.l2o_state_floats(1, 16) == 1280
    assert lib.l2o_state_floats(3, 17) == 3 * 2 * 1280
    assert lib.l2o_state_floats(128, 128) == 128 * 8 * 1280
    assert lib.l2o_state_floats(0, 5) == 0
    for name, cfg in ORACLE_CFGS.items():
        cc = spec_of(cfg).to_c()
        assert lib.l2o_wpack_floats(C.byref(cc)) == (E.wp_rows(cc.preprocess)["total"] * 64 + E.bx_words(cc.preprocess)
                                                    + E.bxb_words(cc.preprocess))


def test_unsupported_layers_are_reported(lib):
    from open_l2o_amd import _abi
    from open_l2o_amd._engine import pack_weights_host
    cfg = O.NetConfig("cw", (1,), "identity", None, 1.0, False)
    params = O.init_net_params(cfg, np.random.default_rng(0))
    with pytest.raises(_abi.L2OUnsupported):
        pack_weights_host(lib, spec_of(cfg), params)


def test_bad_arguments_return_error_codes(lib):
    from open_l2o_amd import _abi
    p = _abi.Problem()
    p.kind, p.B_local, p.B_global, p.D = _abi.PROB_QUADRATIC, 0, 0, 4
    rc = lib.l2o_problem_fg(C.byref(p), None, None, None, None)
    assert rc == _abi.L2O_ERR_ARG
    assert b"bad problem sizes" in lib.l2o_last_error()
    with pytest.raises(ValueError):
        _abi.check(rc)
    cc = spec_of(O.DM_IDENTITY).to_c()
    p.B_local = p.B_global = 2
    p.M = 4
    p.kind = _abi.PROB_SQUARE_COS
    assert lib.l2o_unroll_supported(C.byref(cc), C.byref(p)) == 1
    p.kind = _abi.PROB_MLP                            # neural optimizee: step-granular path only
    assert lib.l2o_unroll_supported(C.byref(cc), C.byref(p)) == 0
    p.kind = _abi.PROB_QUADRATIC
    p.D = p.M = 128
    assert lib.l2o_unroll_supported(C.byref(cc), C.byref(p)) == 1
    assert lib.l2o_unroll_record_supported(C.byref(cc), C.byref(p)) == 1
    p.D = p.M = 512                                    # > 8 tiles: the streaming form (records its history too, ABI v6)
    assert lib.l2o_unroll_supported(C.byref(cc), C.byref(p)) == 1
    assert lib.l2o_unroll_record_supported(C.byref(cc), C.byref(p)) == 1
    assert lib.l2o_unroll_workspace_bytes(C.byref(cc), C.byref(p), 10) == 0
    p.M = 37                                           # any row count
    assert lib.l2o_unroll_supported(C.byref(cc), C.byref(p)) == 1
    p.D, p.M = 16, 40                                  # few columns, more rows than the LDS forms take
    assert lib.l2o_unroll_supported(C.byref(cc), C.byref(p)) == 1
    assert lib.l2o_unroll_record_supported(C.byref(cc), C.byref(p)) == 1
    for D in (516, 1024, 130, 129):                    # too large for the LDS-resident state / not float4 rows
        p.D = p.M = D
        assert lib.l2o_unroll_supported(C.byref(cc), C.byref(p)) == 0


@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
def test_packed_weights_reproduce_oracle_under_mfma_layout(lib, name):
    from open_l2o_amd._engine import pack_weights_host
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=21)
    wpack = pack_weights_host(lib, spec, params)
    rng = np.random.default_rng(22)
    n = 16
    state = random_state(cfg, n, seed=23)
    g = rng.standard_normal((n,)).astype(np.float32)
    if cfg.kind == "rnnprop":
        mt = rng.standard_normal((n,)).astype(np.float32)
        inputs = (mt, g)
        in0, in1 = mt, g
    elif cfg.preprocess_name == "LogAndSign":
        inputs = g
        ls = O.log_and_sign(g[:, None].astype(np.float64), 5)
        in0, in1 = ls[:, 0], ls[:, 1]
    else:
        inputs = g
        in0, in1 = g, np.zeros_like(g)
    f64 = lambda t: tuple((h.astype(np.float64), c.astype(np.float64)) for h, c in t)
    p64 = {k: {v: a.astype(np.float64) for v, a in d.items()} for k, d in params.items()}
    in64 = tuple(a.astype(np.float64) for a in inputs) if isinstance(inputs, tuple) else inputs.astype(np.float64)
    delta, st_ref = O.net_apply(cfg, p64, in64, f64(state))
    coords = list(range(16))
    lanes = np.arange(64)
    h1, c1, h2, c2 = [E.ref_to_lanes(a, coords) for a in (state[0][0], state[0][1], state[1][0], state[1][1])]
    # fp32 MFMA section (exact weights: float64 emulation agrees to 1e-9) and the bf16x3
    # section (weights and activations carried as three bf16 terms = 24 bits: fp32-level error)
    # (both forms of it: 4 packed MFMAs per M-tile -- the DM nets' default -- and one MFMA per product)
    # (round 4: RNNProp's wpack carries the packed section too -- k_unroll_lds reads it from LDS -- so both forms are
    #  emulated for every net, whatever its register-resident kernels default to)
    bx_levels = lambda *a: E.tile_step_bx3(*a, packed=False)
    bx_packed = lambda *a: E.tile_step_bx3(*a, packed=True)
    for step_fn, rtol, atol in ((E.tile_step, 1e-9, 1e-12), (E.tile_step_bx3, 2e-6, 2e-7), (bx_levels, 2e-6, 2e-7),
                                (bx_packed, 2e-6, 2e-7)):
        d, h1n, c1n, h2n, c2n = step_fn(wpack, spec.preprocess, h1, c1, h2, c2,
                                        np.asarray(in0, np.float64)[lanes & 15],
                                        np.asarray(in1, np.float64)[lanes & 15])
        d = np.tanh(d) if cfg.tanh_output else d
        np.testing.assert_allclose(d[:16] * cfg.scale, delta, rtol=rtol, atol=atol)
        for q in range(1, 4):                              # the four q lanes agree
            np.testing.assert_allclose(d[16 * q:16 * q + 16], d[:16], rtol=1e-12)
        for lane_arr, ref in ((h1n, st_ref[0][0]), (c1n, st_ref[0][1]), (h2n, st_ref[1][0]), (c2n, st_ref[1][1])):
            back = np.zeros((16, 20))
            E.lanes_to_ref(lane_arr, back, coords)
            np.testing.assert_allclose(back, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
def test_wpack_transposed_fragments_emulated(lib, name):
    """The BPTT section of wpack (bf16x3 fragments of W itself, csrc/l2o_bwd_mfma.h), pushed through the
    MFMA emulator with the kernel's operand construction, gives d[in | h(t-1)] = W dz of the LSTM
    backward (the transposed gate products of tf.gradients through snt.LSTM, DM/networks.py:192-200)."""
    from open_l2o_amd._engine import pack_weights_host
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=31)
    wpack = pack_weights_host(lib, spec, params)
    rng = np.random.default_rng(32)
    lanes = np.arange(64)
    c, q = lanes & 15, lanes >> 4
    P = cfg.in_dim
    for layer, W in ((2, params["lstm_2"]["w_gates"]), (1, params["lstm_1"]["w_gates"])):
        dz = (rng.standard_normal((16, 80)) * 0.1).astype(np.float32)
        lane_dz = [np.stack([dz[c, r * 20 + 4 * i + q] for i in range(5)], 1).astype(np.float64) for r in range(4)]
        first, second = E.tgemm_bx3(wpack, spec.preprocess, layer, lane_dz)
        ref = dz.astype(np.float64) @ W.astype(np.float64).T          # [16, rows of W]
        two = layer == 2 or cfg.kind == "rnnprop"
        sec0 = 20 if two else P
        for i in range(5):
            np.testing.assert_allclose(second[:, i], ref[c, sec0 + 4 * i + q], rtol=1e-6, atol=1e-7)
            if two:
                np.testing.assert_allclose(first[:, i], ref[c, 4 * i + q], rtol=1e-6, atol=1e-7)
        assert (first is None) == (not two)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from open_l2o_amd._engine import HipEngine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipEngine()


def test_bench_cli_presets_parse():
    """bench.py's command line (the driver's contract: --gpus / --steps / --warmup, plus the BASELINE presets)
    parses without a GPU; the run itself needs one."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--config", "--no-cpu-baseline", "--scaling"):
        assert flag in out.stdout
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--config", "4", "--gpus", "8"])
    assert (a.problem, a.dims, a.batch, a.unroll, a.scaling) == ("rastrigin", 100, 1024, 100, "strong")
    a = bench.parse_args([])
    assert (a.gpus, a.problem, a.net, a.dims, a.batch, a.unroll, a.scaling) == (1, "quadratic", "dm", 128, 128, 100, "weak")


def test_bench_cpu_baseline_legs_run():
    """The cpu_baseline legs of bench.py (the oracle, timed) on tiny inputs: they are the only place outside tests/
    and smoke() that may touch oracle/."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from helpers import make_params
    w = make_params(O.RNNPROP, seed=1, trained_like=True)
    out = bench.cpu_baseline_mnist(w, batch=8, T=2, max_seconds=5.0)
    assert out["value"] > 0 and out["kind"] == "port" and "2 steps" in out["sample"]
    rng = np.random.default_rng(0)
    arrays = {"W": rng.random((4, 10, 10), dtype=np.float32), "y": rng.random((4, 10), dtype=np.float32), "l1": 0.1, "alpha": 10.0}
    out = bench.cpu_baseline("quadratic", "dm", arrays, make_params(O.DM_IDENTITY, seed=2, trained_like=True),
                             (rng.standard_normal((4, 10)) * 0.01).astype(np.float32), 5, max_seconds=1.0)
    assert out["value"] > 0 and out["cores"] >= 1 and np.isfinite(out["fx_T"])


def test_adam_and_device_pack_reject_bad_arguments(lib):
    from open_l2o_amd import _abi
    assert lib.l2o_adam_step(None, None, None, None, 4, 0.1, 0.9, 0.999, 1e-8, None) == _abi.L2O_ERR_ARG
    assert lib.l2o_adam_step_guarded(None, None, None, None, 4, 0.1, 0.9, 0.999, 1e-8, None, None) == _abi.L2O_ERR_ARG
    cc = spec_of(O.DM_IDENTITY).to_c()
    assert lib.l2o_wpack_device(C.byref(cc), None, None, None) == _abi.L2O_ERR_ARG
