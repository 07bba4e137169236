"""bench.py's roofline arithmetic on CPU (no GPU): the work-based figure of round 4 (stated minimal instruction counts x
measured issue costs / measured cycles per tile-step) and the same-lease counters lookup, on a synthetic counters file."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("l2o_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_work_model_counts():
    b = _bench()
    dm = b.work_model("quadratic", "dm", 128, 128)
    assert dm == {"valu_plain": 2 * 32 + 26 + 10 + 60 + 54 + 27, "transcendental": 80, "mfma": 60}
    ls = b.work_model("quadratic", "dm_logsign", 128, 128)
    assert ls["valu_plain"] == dm["valu_plain"] + 10 and ls["mfma"] == 60
    rp = b.work_model("lasso", "rnnprop", 512, 256)
    assert rp["mfma"] == 120 and rp["transcendental"] == 102 and rp["valu_plain"] > dm["valu_plain"]
    # the streamed matrix slice of a wave: M * D / (tiles * 64) FMAs per pass
    assert b.work_model("lasso", "dm", 512, 256)["valu_plain"] - b.work_model("lasso", "dm", 512, 128)["valu_plain"] == 2 * 32


def test_work_block_and_same_lease_lookup(tmp_path, monkeypatch):
    b = _bench()
    counters = {"workload": ["quadratic", "dm", 128, 128, 100], "kernel": "void k_unroll_pair<0, 1, 8, false, false>(UnrollPairArgs)",
                "per_launch": {"SQ_INSTS_VALU": 455.0 * 1024 * 100.3, "SQ_WAVES": 1024.0, "SQ_ACTIVE_INST_VALU": 5.5e7,
                               "SQ_VALU_MFMA_BUSY_CYCLES": 9.9e7, "SQ_INSTS_MFMA": 6.2e6, "FETCH_SIZE_KiB": 4800.0,
                               "WRITE_SIZE_KiB": 6100.0},
                "kernel_ns_profiled": {"p": 183000.0}, "clock_hz_profiled": 2.43e9, "one_wave_per_simd": False}
    (tmp_path / "counters_c2.json").write_text(json.dumps(counters))
    monkeypatch.setenv("L2O_COUNTERS_DIR", str(tmp_path))
    found = b.counters_for(["quadratic", "dm", 128, 128, 100], "k_unroll_pair")
    assert found is not None and found[0].startswith(str(tmp_path))          # the lease's file wins over profiles/
    case = {"kern_ms": 0.1834, "kern_ms_min": 0.183, "kernel": "k_unroll_pair", "alg_bytes": 2.746e9, "bpc": 1676.0,
            "flops": 1.69e10, "hbm_bound": False, "hbm_model_bytes": 0.0, "fused": True, "D": 128, "Mrows": 128, "T": 100,
            "dispatches": 1}

    class A:
        problem, net = "quadratic", "dm"
    roof = b.roofline_block(case, A, found)
    assert roof["bound"] == "valu_issue" and 0.4 < roof["frac"] < 0.6
    cyc = 0.1834e-3 * 2.43e9 / 100.3
    assert abs(roof["cycles_per_tile_step"] - cyc) < 1e-6 * cyc
    floor = 241 * b.ISSUE_COST["valu"] + 80 * b.ISSUE_COST["trans"] + 60 * b.ISSUE_COST["mfma"]
    assert abs(roof["issue_floor_cycles_per_tile_step"] - floor) < 1e-9
    assert abs(roof["frac_work"] - floor / cyc) < 1e-9 and 0.45 < roof["frac_work"] < 0.56
    assert abs(roof["valu_insts_per_tile_step"] - 455.0) < 1e-6
    assert abs(roof["traffic"] - (2 * 4800.0 + 6100.0) * 1024.0) < 1.0        # FETCH_SIZE doubled on gfx950
    # a streaming (HBM-bound) kernel carries no work block
    case3 = dict(case, hbm_bound=True, kernel="k_unroll_cu")
    assert "frac_work" not in b.roofline_block(case3, A, found)
    # a two-waves-per-SIMD kernel (k_unroll_lds): the floor is the SIMD's PIPE time for two tile-steps per step
    case4 = dict(case, kernel="k_unroll_lds (one problem per CU, two waves per SIMD, fragments in LDS)", kern_ms=1.2263,
                 D=100, Mrows=100, B=1024, n_cus=256)

    class A4:
        problem, net = "rastrigin", "dm"
    roof4 = b.roofline_block(case4, A4, found)
    wm = b.work_model("rastrigin", "dm", 100, 100)
    floor4 = 2 * (wm["valu_plain"] * b.PIPE_COST["valu"] + wm["transcendental"] * b.PIPE_COST["trans"])
    cyc4 = 1.2263e-3 * 2.43e9 / (4 * 100.3)                                   # four rounds of problems per CU
    assert abs(roof4["cycles_per_step"] - cyc4) < 1e-6 * cyc4 and roof4["tiles_per_simd"] == 2
    assert abs(roof4["pipe_floor_cycles_per_step"] - floor4) < 1e-9 and abs(roof4["frac_work"] - floor4 / cyc4) < 1e-9
    assert 0.3 < roof4["frac_work"] < 0.45 and "cycles_per_tile_step" not in roof4
