"""bench.py's roofline arithmetic on CPU (no GPU): the work / peak figure (stated minimal instruction counts x measured
PIPE rates / measured cycles per step) and the counters lookup by build id, on synthetic counters files."""
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


def _counters(build_id=None, collected=None, insts=455.0):
    c = {"workload": ["quadratic", "dm", 128, 128, 100], "kernel": "void k_unroll_pair<0, 1, 8, false, false>(UnrollPairArgs)",
         "per_launch": {"SQ_INSTS_VALU": insts * 1024 * 100.3, "SQ_WAVES": 1024.0, "SQ_ACTIVE_INST_VALU": 5.5e7,
                        "SQ_VALU_MFMA_BUSY_CYCLES": 9.9e7, "SQ_INSTS_MFMA": 6.2e6, "FETCH_SIZE_KiB": 4800.0,
                        "WRITE_SIZE_KiB": 6100.0, "SQ_WAVE_CYCLES": 1.0e8, "SQ_ACTIVE_INST_ANY": 0.58e8,
                        "SQ_WAIT_INST_ANY": 0.27e8, "SQ_WAIT_ANY": 0.15e8},
         "kernel_ns_profiled": {"p": 183000.0}, "clock_hz_profiled": 2.43e9, "one_wave_per_simd": False}
    if build_id:
        c["build_id"] = build_id
    if collected:
        c["collected_unix"] = collected
    return c


def test_counters_are_chosen_by_build_id_not_by_file_name(tmp_path, monkeypatch):
    """`sorted(glob)` picked profiles/archive_r04/r04z_* over the newer r04av_*.  The selection reads what the file says
    about itself: the build it was collected on, then when."""
    b = _bench()
    wl = ["quadratic", "dm", 128, 128, 100]
    # names sort z > a: the OLD build's file would win a filename sort
    (tmp_path / "counters_zz_old.json").write_text(json.dumps(_counters("aaaaaaaaaaaaaaaa", 1000.0, insts=455.0)))
    (tmp_path / "counters_aa_new.json").write_text(json.dumps(_counters("bbbbbbbbbbbbbbbb", 2000.0, insts=423.0)))
    (tmp_path / "counters_mm_legacy.json").write_text(json.dumps(_counters()))            # (no id, no timestamp)
    monkeypatch.setenv("L2O_COUNTERS_DIR", str(tmp_path))
    monkeypatch.setattr(b, "ROOT", str(tmp_path / "no_profiles_here"))
    path, c, status = b.counters_for(wl, "k_unroll_pair", "bbbbbbbbbbbbbbbb")
    assert status == "same_build" and path.endswith("counters_aa_new.json")
    path, c, status = b.counters_for(wl, "k_unroll_pair", "aaaaaaaaaaaaaaaa")
    assert status == "same_build" and path.endswith("counters_zz_old.json")
    path, c, status = b.counters_for(wl, "k_unroll_pair", "cccccccccccccccc")            # a build nobody profiled
    assert status == "stale" and path.endswith("counters_aa_new.json")                  # ... the most recently collected
    assert b.counters_for(wl, "k_unroll_lds", "bbbbbbbbbbbbbbbb") is None                  # another kernel's counters
    assert b.counters_for(["lasso", "rnnprop", 512, 256, 200, 256], "", "bbbbbbbbbbbbbbbb") is None


def test_roofline_frac_is_work_over_the_guide_peak(tmp_path):
    """frac = stated minimal instruction counts x the hardware guide's pipe rates (2 cycles plain, 8 transcendental:
    MI355X_MICROARCH.md) / measured cycles (config 2: 1 122 / 4 331 = 0.26); the same counts at the pipe
    rates MEASURED on this part ride along as frac_measured_pipe (0.32), like the utilisation, stall and issue-cost
    figures under their own names; stale counters are marked and never enter frac."""
    b = _bench()
    case = {"kern_ms": 0.1777, "kern_ms_min": 0.177, "kernel": "k_unroll_pair (every problem on two CUs, one wave per SIMD)",
            "alg_bytes": 2.746e9, "bpc": 1676.0, "flops": 1.69e10, "hbm_bound": False, "hbm_model_bytes": 0.0, "fused": True,
            "D": 128, "Mrows": 128, "T": 100, "B": 128, "dispatches": 1, "loop_ticks": (3944.0 * 100.3, 4343.0 * 100.3)}

    class A:
        problem, net = "quadratic", "dm"
    measured_floor = 241 * b.PIPE_COST["valu"] + 80 * b.PIPE_COST["trans"]
    assert abs(measured_floor - 1377.3) < 0.5
    pipe_floor = 241 * 2.0 + 80 * 8.0                                          # the guide's peak: 1 122 cycles
    assert b.GUIDE_COST == {"valu": 2.0, "trans": 8.0}
    for counters, status in (((str(tmp_path / "c.json"), _counters("x" * 16), "same_build"), "same_build"),
                             ((str(tmp_path / "c.json"), _counters("y" * 16), "stale"), "stale"), (None, "none")):
        roof = b.roofline_block(case, A, counters)
        assert roof["bound"] == "valu_pipe" and roof["counters"] == status
        clock = 2.43e9 if counters is not None else 2.4e9                      # (the PMC passes' clock, else nominal)
        cyc = 0.1777e-3 * clock / 100.3                                        # live kernel time x clock: the denominator of frac
        assert abs(roof["cycles_per_step"] - cyc) < 1e-6 * cyc and roof["cycles_source"].startswith("kernel_ms_avg")
        assert abs(roof["frac"] - pipe_floor / cyc) < 1e-9 and 0.25 < roof["frac"] < 0.27
        assert roof["frac_guide_peak"] == roof["frac"] and "MI355X_MICROARCH" in roof["peak_source"]
        assert abs(roof["frac_measured_pipe"] - measured_floor / cyc) < 1e-9 and 0.31 < roof["frac_measured_pipe"] < 0.33
        assert abs(roof["cycles_per_step_in_kernel"] - 4343.0) < 1e-6 and "s_memtime" in roof["in_kernel_cycles_source"]
        assert abs(roof["frac_in_kernel_cycles"] - pipe_floor / 4343.0) < 1e-9
        assert abs(roof["cycles_per_step_loop"] - 3944.0) < 1e-6 and abs(roof["frac_step_loop"] - pipe_floor / 3944.0) < 1e-9
        assert abs(roof["achieved"] / roof["peak"] - roof["frac"]) < 1e-12
        issue_floor = 241 * b.ISSUE_COST["valu"] + 80 * b.ISSUE_COST["trans"] + 60 * b.ISSUE_COST["mfma"]
        assert abs(roof["issue_cost_frac"] - issue_floor / cyc) < 1e-9
        if counters is None:
            assert "valu_active_frac" not in roof and roof["traffic"] is None and "wait_inst_frac" not in roof
        else:
            assert abs(roof["wait_inst_frac"] - 0.27) < 1e-9 and abs(roof["wait_any_frac"] - 0.15) < 1e-9
            assert abs(roof["active_frac"] - 0.58) < 1e-9
            assert 0.4 < roof["valu_active_frac"] < 0.6 and abs(roof["valu_insts_per_tile_step"] - 455.0) < 1e-6
            assert abs(roof["traffic"] - (2 * 4800.0 + 6100.0) * 1024.0) < 1.0    # FETCH_SIZE doubled on gfx950
    # without the in-kernel count: the primary figure is unchanged, the in-kernel secondaries are absent
    case_t = dict(case, loop_ticks=None)
    roof = b.roofline_block(case_t, A, None)
    cyc = 0.1777e-3 * 2.4e9 / 100.3
    assert abs(roof["cycles_per_step"] - cyc) < 1e-6 * cyc and "frac_in_kernel_cycles" not in roof
    # a streaming (HBM-bound) kernel: bytes of the form / time / 8 TB/s, no work block
    case3 = dict(case, hbm_bound=True, kernel="k_unroll_cu8", hbm_model_bytes=27.06e9, kern_ms=4.608)
    r3 = b.roofline_block(case3, A, None)
    assert r3["bound"] == "hbm" and abs(r3["frac"] - 27.06e9 / 4.608e-3 / 8e12) < 1e-9 and "frac_work" not in r3
    # two waves per SIMD (k_unroll_lds): two tile-steps per SIMD and step; 1024 problems = four rounds per CU when
    # the cycles come from the kernel time
    case4 = dict(case, kernel="k_unroll_lds (one problem per CU, two waves per SIMD, fragments in LDS)", kern_ms=1.2263,
                 D=100, Mrows=100, B=1024, n_cus=256, loop_ticks=None)

    class A4:
        problem, net = "rastrigin", "dm"
    roof4 = b.roofline_block(case4, A4, None)
    wm = b.work_model("rastrigin", "dm", 100, 100)
    floor4 = 2 * (wm["valu_plain"] * 2.0 + wm["transcendental"] * 8.0)
    cyc4 = 1.2263e-3 * 2.4e9 / (4 * 100.3)
    assert abs(roof4["cycles_per_step"] - cyc4) < 1e-6 * cyc4 and roof4["tiles_per_simd"] == 2
    assert abs(roof4["frac"] - floor4 / cyc4) < 1e-9 and 0.25 < roof4["frac"] < 0.4 and "issue_cost_frac" not in roof4
