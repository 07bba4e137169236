"""ctypes binding of the C ABI declared in ``include/l2o_abi.h``.

This is the ONLY way the Python host reaches the compute path: there is no
PyTorch / NumPy fallback.  If ``libl2o_hip.so`` is missing (not built) every
entry point raises ``RuntimeError`` -- the product path must fail loudly when
the HIP extension is absent.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# L2O_HIP_LIB: alternative build of the SAME library (timing ablations, scripts/ablate.sh)
LIB_PATH = os.environ.get("L2O_HIP_LIB") or os.path.join(_HERE, "libl2o_hip.so")

L2O_ABI_VERSION = 13
L2O_OK, L2O_ERR_ARG, L2O_ERR_UNSUPPORTED, L2O_ERR_HIP, L2O_ERR_TIMEOUT = 0, -1, -2, -3, -4

NET_CW, NET_RNNPROP = 0, 1
PRE_IDENTITY, PRE_LOGSIGN, PRE_FC_ELU = 0, 1, 2
PROB_SIMPLE, PROB_QUADRATIC, PROB_LASSO, PROB_RASTRIGIN, PROB_SQUARE_COS, PROB_MLP = 0, 1, 2, 3, 4, 5

# every symbol include/l2o_abi.h declares (tests check the library exports all of them)
SYMBOLS = (
    "l2o_abi_version", "l2o_last_error", "l2o_build_id", "l2o_last_unroll_form", "l2o_coresident_workgroups", "l2o_wpack_floats", "l2o_wpack_host",
    "l2o_state_floats", "l2o_state_pack", "l2o_state_unpack", "l2o_problem_fg", "l2o_problem_hvp", "l2o_mlp_fg",
    "l2o_mlp_scratch_floats", "l2o_mlp_unroll", "l2o_mlp_unroll_record", "l2o_mlp_unroll_supported", "l2o_mlp_unroll_workspace_bytes",
    "l2o_mlp_unroll_multi", "l2o_mlp_unroll_multi_supported", "l2o_mlp_unroll_multi_workspace_bytes",
    "l2o_mlp_deep_fg", "l2o_mlp_deep_scratch_floats",
    "l2o_cwlstm_step", "l2o_cwlstm_step_multi", "l2o_cwlstm_step_generic", "l2o_cwlstm_bwd_step_generic", "l2o_gen_state_floats", "l2o_cwlstm_bwd_step", "l2o_cwlstm_bwd_multi", "l2o_cwlstm_bwd_unroll", "l2o_cwlstm_bwd_unroll_compact", "l2o_cwlstm_wgrad_compact", "l2o_unroll", "l2o_unroll_record", "l2o_unroll_reduce", "l2o_unroll_workspace_init", "l2o_unroll_workspace_layout", "l2o_cwlstm_wgrad", "l2o_cwlstm_wgrad_dims", "l2o_unroll_supported", "l2o_unroll_record_supported", "l2o_adam_step", "l2o_adam_step_guarded", "l2o_adam_step_gather", "l2o_wpack_device", "l2o_unroll_workspace_bytes",
    "l2o_unroll_status", "l2o_reduce_fx", "l2o_atb", "l2o_atb_workspace_bytes",
    "l2o_suffix_sums", "l2o_colsum", "l2o_colsum_scratch_floats", "l2o_lincomb", "l2o_rnnprop_input_adjoint",
)


# option ids of include/l2o_abi.h ("options": per call, caller-owned -- the LIBRARY keeps no option state since ABI v9).
# This binding keeps the caller's side of it: one process-wide dict of non-default values (applied from the L2O_*
# environment variables once at import, changed by set_option) that NetSpec.to_c() / the problem and MLP descriptors
# encode into every struct they hand to the library.
# (id 8 was OPT_PAIR_NORMAL until ABI v11; since v13 it is OPT_MLP_XCD_WAVES)
OPT_PAIR, OPT_PAIR_PLAIN_STORES, OPT_UNROLL_CU, OPT_FG_TWO_PASS, OPT_MLP_GENERIC, OPT_BWD_BLOCKS, OPT_BWD_KERNEL, \
    OPT_MLP_UNROLL, OPT_MLP_XCD_WAVES, OPT_EXACT_GATES, OPT_WPACK_NO_CLEAR, OPT_MLP_HIER, OPT_ONE_LDS = range(13)
OPT_DEFAULTS = {OPT_PAIR: 1, OPT_PAIR_PLAIN_STORES: 1, OPT_UNROLL_CU: 1, OPT_FG_TWO_PASS: 0, OPT_MLP_GENERIC: 0,
                OPT_BWD_BLOCKS: 0, OPT_BWD_KERNEL: 0, OPT_MLP_UNROLL: 1, OPT_MLP_XCD_WAVES: 0, OPT_EXACT_GATES: 0,
                OPT_WPACK_NO_CLEAR: 0, OPT_MLP_HIER: 1, OPT_ONE_LDS: 1}
# l2o_last_unroll_form(): which kernel a fused launch ran (include/l2o_abi.h L2O_FORM_*)
FORM_NAMES = {1: "k_unroll", 2: "k_unroll_pair", 5: "k_unroll_lds", 6: "k_unroll_cu", 7: "k_unroll_cu8",
              8: "k_mlp_unroll (flat all-reduce)", 9: "k_mlp_unroll (XCD-hierarchical all-reduce)", 10: "k_mlp_unroll (generic loops)",
              11: "k_mlp_xcd (one optimizee instance per XCD)"}
FORMS_WITH_EXCHANGE = (2, 8, 9, 10, 11)    # workgroups wait for partner workgroups: can end in L2OPartnerTimeout
PROB_FG_TWO_PASS = 2      # l2o_problem.flags
MLP_GENERIC = 1           # l2o_mlp.flags
_options = {}
# The library never reads the environment; this binding applies two variables ONCE, when it is imported:
#   L2O_EXACT_GATES=1          the fmaf-chain-equal fp32 MFMA gate GEMM (an accuracy choice a user may want: DESIGN.md 4)
#   L2O_OPTIONS=name=v,...     any option by name (pair, one_lds, unroll_cu, mlp_hier, ...): the A/B scripts' escape hatch.
# (Until round 5 every option had its own variable -- thirteen of them; tests and callers use set_option / option_scope.)
OPTION_NAMES = {"pair": OPT_PAIR, "pair_plain_stores": OPT_PAIR_PLAIN_STORES, "unroll_cu": OPT_UNROLL_CU,
                "fg_two_pass": OPT_FG_TWO_PASS, "mlp_generic": OPT_MLP_GENERIC, "bwd_blocks": OPT_BWD_BLOCKS,
                "bwd_kernel": OPT_BWD_KERNEL, "mlp_unroll": OPT_MLP_UNROLL, "exact_gates": OPT_EXACT_GATES,
                "mlp_hier": OPT_MLP_HIER, "one_lds": OPT_ONE_LDS, "mlp_xcd_waves": OPT_MLP_XCD_WAVES}
if os.environ.get("L2O_EXACT_GATES"):
    _options[OPT_EXACT_GATES] = 1
for _item in filter(None, os.environ.get("L2O_OPTIONS", "").split(",")):
    _name, _, _val = _item.partition("=")
    if _name.strip() not in OPTION_NAMES:
        raise ValueError("L2O_OPTIONS: unknown option %r (known: %s)" % (_name, ", ".join(sorted(OPTION_NAMES))))
    _options[OPTION_NAMES[_name.strip()]] = int(_val)


def set_option(opt, value):
    """Change one kernel A/B switch for every later call of this process; returns the previous value."""
    if opt not in OPT_DEFAULTS:
        raise ValueError("unknown option %r" % (opt,))
    old = get_option(opt)
    value = int(value)
    if value == OPT_DEFAULTS[opt]:
        _options.pop(opt, None)
    else:
        if opt != OPT_BWD_BLOCKS and not 0 <= value <= 7:
            raise ValueError("option %d: value %d out of range" % (opt, value))
        _options[opt] = value
    return old


class option_scope(object):
    """with option_scope({OPT_PAIR: 0}): ... -- the settings for the calls inside, restored afterwards."""

    def __init__(self, settings):
        self.settings = dict(settings)

    def __enter__(self):
        self.old = {o: set_option(o, v) for o, v in self.settings.items()}
        return self

    def __exit__(self, *exc):
        for o, v in self.old.items():
            set_option(o, v)
        return False


def get_option(opt):
    if opt not in OPT_DEFAULTS:
        return -1
    return _options.get(opt, OPT_DEFAULTS[opt])


def options_word():
    """l2o_net_cfg.options of the current settings: L2O_OPTW(o, v) words OR-ed together (0 = every default)."""
    w = 0
    for o, v in _options.items():
        if o == OPT_BWD_BLOCKS:
            w |= (v & 0xffff) << 48
        else:
            # (include/l2o_abi.h L2O_OPT_FIELD_: bits 48-63 are the BWD_BLOCKS count, option 12 uses that option's unused field)
            w |= (8 | (v & 7)) << (4 * (OPT_BWD_BLOCKS if o == OPT_ONE_LDS else o))
    return w


class NetCfg(C.Structure):
    """struct l2o_net_cfg"""
    _fields_ = [
        ("kind", C.c_int32), ("preprocess", C.c_int32), ("n_layers", C.c_int32),
        ("hidden", C.c_int32), ("tanh_output", C.c_int32), ("reserved", C.c_int32),
        ("scale", C.c_double), ("logsign_k", C.c_double),
        ("beta1", C.c_double), ("beta2", C.c_double), ("options", C.c_uint64),
    ]


class Problem(C.Structure):
    """struct l2o_problem"""
    _fields_ = [
        ("kind", C.c_int32), ("B_local", C.c_int32), ("B_global", C.c_int32),
        ("D", C.c_int32), ("M", C.c_int32), ("flags", C.c_int32),
        ("l1", C.c_double), ("alpha", C.c_double),
        ("W", C.c_void_p), ("y", C.c_void_p), ("C", C.c_void_p), ("x_scale", C.c_void_p),
    ]


class Mlp(C.Structure):
    """struct l2o_mlp"""
    _fields_ = [
        ("n_in", C.c_int32), ("n_hidden", C.c_int32), ("n_out", C.c_int32), ("batch", C.c_int32),
        ("activation", C.c_int32), ("n_data", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
        ("images", C.c_void_p), ("labels", C.c_void_p),
    ]


class MlpHist(C.Structure):            # l2o_mlp_hist: per variable (w1, b1, w2, b2) history of l2o_mlp_unroll_record
    _fields_ = [("st", C.c_void_p * 4), ("g", C.c_void_p * 4), ("m", C.c_void_p * 4), ("v", C.c_void_p * 4)]


class MlpDeep(C.Structure):
    """struct l2o_mlp_deep"""
    _fields_ = [("n_in", C.c_int32), ("n_out", C.c_int32), ("batch", C.c_int32), ("activation", C.c_int32),
                ("n_data", C.c_int32), ("n_hidden_layers", C.c_int32), ("hidden", C.c_int32 * 3), ("reserved", C.c_int32),
                ("images", C.c_void_p), ("labels", C.c_void_p)]


class MlpInstance(C.Structure):        # l2o_mlp_instance: one replica of l2o_mlp_unroll_multi
    _fields_ = [("indices", C.c_void_p), ("x", C.c_void_p * 4), ("st", C.c_void_p * 4), ("m", C.c_void_p * 4),
                ("v", C.c_void_p * 4), ("x_scale", C.c_void_p * 4), ("fx", C.c_void_p)]


class GenNet(C.Structure):
    """struct l2o_gen_net"""
    _fields_ = [("n_layers", C.c_int32), ("hidden", C.c_int32 * 3), ("in_dim", C.c_int32), ("direct_inputs", C.c_int32),
                ("w_gates", C.c_void_p * 3), ("b_gates", C.c_void_p * 3), ("w_lin", C.c_void_p), ("b_lin", C.c_void_p),
                ("w_fc", C.c_void_p), ("b_fc", C.c_void_p)]


class GenBwdIO(C.Structure):
    """struct l2o_gen_bwd_io"""
    _fields_ = [(n, C.c_void_p) for n in ("g", "m", "v", "st_prev", "dx_next", "carry_in", "carry_out")] + \
               [("act", C.c_void_p * 3), ("dz", C.c_void_p * 3)] + \
               [(n, C.c_void_p) for n in ("tc", "h_last", "dd", "feats", "du", "dg")]


class NetWeights(C.Structure):
    """struct l2o_net_weights"""
    _fields_ = [(n, C.c_void_p) for n in ("w_gates1", "b_gates1", "w_gates2", "b_gates2", "w_lin", "b_lin",
                                          "w_fc", "b_fc", "wpack")]


class BwdIO(C.Structure):
    """struct l2o_bwd_io"""
    _fields_ = [(n, C.c_void_p) for n in ("g", "m", "v", "st_prev", "dx_next", "carry_in", "carry_out", "act1",
                                          "dz1", "act2", "dz2", "h2", "dd", "feats", "du")] + \
               [("a_stride", C.c_int64), ("b_stride", C.c_int64), ("dg", C.c_void_p)]


UNROLL_ZERO_STATE = 1     # l2o_unroll_reduce flags
PROB_W_SHARED = 1     # l2o_problem.flags: W is one [M, D] matrix for every problem


class StepSeg(C.Structure):
    """struct l2o_step_seg"""
    _fields_ = [("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("st", C.c_void_p), ("x", C.c_void_p),
                ("B", C.c_int64), ("D", C.c_int64), ("st_out", C.c_void_p), ("m_out", C.c_void_p), ("v_out", C.c_void_p)]


class BwdSeg(C.Structure):
    """struct l2o_bwd_seg"""
    _fields_ = [(n, C.c_void_p) for n in ("g", "m", "v", "st_prev", "dx_next")] + [("B", C.c_int64), ("D", C.c_int64)]


class BwdUnrollSeg(C.Structure):
    """struct l2o_bwd_unroll_seg"""
    _fields_ = [("B", C.c_int64), ("D", C.c_int64), ("g_final", C.c_void_p)]


class UnrollHist(C.Structure):
    """struct l2o_unroll_hist"""
    _fields_ = [(n, C.c_void_p) for n in ("st", "g", "m", "v", "g_final")]


class L2OError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libl2o_hip error %d: %s" % (code, msg))
        self.code = code


class L2OUnsupported(L2OError):
    """L2O_ERR_UNSUPPORTED: no fused kernel for this configuration."""


class L2OPartnerTimeout(L2OError):
    """L2O_ERR_TIMEOUT (l2o_unroll_status): a workgroup of a kernel that exchanges data with partner workgroups gave up
    waiting -- that launch's outputs are invalid.  Recoverable: the host re-runs the unroll on an exchange-free form."""


def source_build_id():
    """What l2o_build_id() of a library built from the sources in this tree returns (csrc/Makefile: sha256 over
    the two .hip translation units, the headers and the Makefile in make's $(sort) order), or None without the sources."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    names = ["l2o_kernels.hip", "l2o_kernels_ilp.hip", "Makefile", "../../include/l2o_abi.h"] + [
        os.path.basename(p) for p in glob.glob(os.path.join(csrc, "*.h"))]
    h = hashlib.sha256()
    try:
        for n in sorted(set(names)):
            with open(os.path.join(csrc, n), "rb") as f:
                h.update(f.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


_lib = None


def build_id():
    """l2o_build_id() of the loaded library (16 hex digits)."""
    return lib().l2o_build_id().decode("ascii", "replace")


def last_unroll_form():
    """(name, dispatches) of the kernel the last fused unroll call of this thread launched, or (None, 0)."""
    w = int(lib().l2o_last_unroll_form())
    return FORM_NAMES.get(w & 0xff), w >> 8


def lib():
    """Load (once) and return the C-ABI library; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "open_l2o_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C open_l2o_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    fp, vp, i32, i64, dbl = C.POINTER(C.c_float), C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.l2o_abi_version.restype = C.c_int
    L.l2o_abi_version.argtypes = []
    L.l2o_last_error.restype = C.c_char_p
    L.l2o_last_error.argtypes = []
    L.l2o_build_id.restype = C.c_char_p
    L.l2o_build_id.argtypes = []
    L.l2o_last_unroll_form.restype = C.c_int
    L.l2o_last_unroll_form.argtypes = []
    L.l2o_coresident_workgroups.restype = C.c_int32
    L.l2o_coresident_workgroups.argtypes = [C.c_void_p, C.c_void_p]
    L.l2o_wpack_floats.restype = C.c_size_t
    L.l2o_wpack_floats.argtypes = [C.POINTER(NetCfg)]
    L.l2o_wpack_host.restype = C.c_int
    L.l2o_wpack_host.argtypes = [C.POINTER(NetCfg)] + [vp] * 9
    L.l2o_state_floats.restype = C.c_size_t
    L.l2o_state_floats.argtypes = [i64, i64]
    L.l2o_state_pack.restype = C.c_int
    L.l2o_state_pack.argtypes = [vp, vp, vp, vp, vp, i64, i64, vp]
    L.l2o_state_unpack.restype = C.c_int
    L.l2o_state_unpack.argtypes = [vp, vp, vp, vp, vp, i64, i64, vp]
    L.l2o_problem_fg.restype = C.c_int
    L.l2o_problem_fg.argtypes = [C.POINTER(Problem), vp, vp, vp, vp]
    L.l2o_problem_hvp.restype = C.c_int
    L.l2o_problem_hvp.argtypes = [C.POINTER(Problem), vp, vp, vp, vp, vp]
    L.l2o_mlp_fg.restype = C.c_int
    L.l2o_mlp_fg.argtypes = [C.POINTER(Mlp)] + [vp] * 12
    L.l2o_mlp_scratch_floats.restype = C.c_size_t
    L.l2o_mlp_scratch_floats.argtypes = [C.POINTER(Mlp)]
    L.l2o_mlp_unroll_supported.restype = C.c_int
    L.l2o_mlp_unroll_supported.argtypes = [C.POINTER(NetCfg), C.POINTER(Mlp), vp]
    L.l2o_mlp_unroll_workspace_bytes.restype = C.c_size_t
    L.l2o_mlp_unroll_workspace_bytes.argtypes = [C.POINTER(Mlp)]
    L.l2o_mlp_unroll.restype = C.c_int
    L.l2o_mlp_unroll.argtypes = [C.POINTER(NetCfg), vp, C.POINTER(Mlp), vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.l2o_mlp_unroll_record.restype = C.c_int
    L.l2o_mlp_unroll_record.argtypes = [C.POINTER(NetCfg), vp, C.POINTER(Mlp), vp, vp, vp, vp, vp, vp, i32, i32, vp,
                                        C.POINTER(MlpHist), vp, vp]
    L.l2o_mlp_deep_scratch_floats.restype = C.c_size_t
    L.l2o_mlp_deep_scratch_floats.argtypes = [C.POINTER(MlpDeep)]
    L.l2o_mlp_deep_fg.restype = C.c_int
    L.l2o_mlp_deep_fg.argtypes = [C.POINTER(MlpDeep), vp, vp, vp, vp, vp, vp]
    L.l2o_mlp_unroll_multi_supported.restype = C.c_int
    L.l2o_mlp_unroll_multi_supported.argtypes = [C.POINTER(NetCfg), C.POINTER(Mlp), i32, vp]
    L.l2o_mlp_unroll_multi_workspace_bytes.restype = C.c_size_t
    L.l2o_mlp_unroll_multi_workspace_bytes.argtypes = [C.POINTER(Mlp), i32]
    L.l2o_mlp_unroll_multi.restype = C.c_int
    L.l2o_mlp_unroll_multi.argtypes = [C.POINTER(NetCfg), vp, C.POINTER(Mlp), C.POINTER(MlpInstance), i32, i32, i32, vp, vp]
    L.l2o_cwlstm_step.restype = C.c_int
    L.l2o_cwlstm_step.argtypes = [C.POINTER(NetCfg), vp, vp, vp, vp, dbl, dbl, vp, vp, i64, i64, vp]
    L.l2o_gen_state_floats.restype = C.c_size_t
    L.l2o_gen_state_floats.argtypes = [C.POINTER(GenNet), i64]
    L.l2o_cwlstm_step_generic.restype = C.c_int
    L.l2o_cwlstm_step_generic.argtypes = [C.POINTER(NetCfg), C.POINTER(GenNet), vp, vp, vp, vp, dbl, dbl, vp, vp, i64, vp]
    L.l2o_cwlstm_bwd_step_generic.restype = C.c_int
    L.l2o_cwlstm_bwd_step_generic.argtypes = [C.POINTER(NetCfg), C.POINTER(GenNet), C.POINTER(GenBwdIO), dbl, dbl, i64, vp]
    L.l2o_cwlstm_step_multi.restype = C.c_int
    L.l2o_cwlstm_step_multi.argtypes = [C.POINTER(NetCfg), vp, C.POINTER(StepSeg), C.c_int32, dbl, dbl, vp]
    L.l2o_cwlstm_bwd_step.restype = C.c_int
    L.l2o_cwlstm_bwd_step.argtypes = [C.POINTER(NetCfg), C.POINTER(NetWeights), C.POINTER(BwdIO), dbl, dbl, i64,
                                      i64, vp]
    L.l2o_cwlstm_bwd_multi.restype = C.c_int
    L.l2o_cwlstm_bwd_multi.argtypes = [C.POINTER(NetCfg), C.POINTER(NetWeights), C.POINTER(BwdSeg), C.c_int32, vp, vp,
                                       vp, vp, dbl, dbl, vp]
    L.l2o_cwlstm_bwd_unroll.restype = C.c_int
    L.l2o_cwlstm_bwd_unroll.argtypes = [C.POINTER(NetCfg), C.POINTER(NetWeights), C.POINTER(BwdUnrollSeg), C.c_int32, vp,
                                        C.c_int32, i64, vp, vp, vp, vp, vp]
    L.l2o_cwlstm_bwd_unroll_compact.restype = C.c_int
    L.l2o_cwlstm_bwd_unroll_compact.argtypes = L.l2o_cwlstm_bwd_unroll.argtypes
    L.l2o_adam_step.restype = C.c_int
    L.l2o_adam_step.argtypes = [vp, vp, vp, vp, i64, C.c_float, dbl, dbl, dbl, vp]
    L.l2o_adam_step_guarded.restype = C.c_int
    L.l2o_adam_step_guarded.argtypes = [vp, vp, vp, vp, i64, C.c_float, dbl, dbl, dbl, vp, vp]
    L.l2o_adam_step_gather.restype = C.c_int
    L.l2o_adam_step_gather.argtypes = [vp, vp, vp, vp, vp, i64, C.c_float, dbl, dbl, dbl, vp, vp]
    L.l2o_wpack_device.restype = C.c_int
    L.l2o_wpack_device.argtypes = [C.POINTER(NetCfg), C.POINTER(NetWeights), vp, vp]
    L.l2o_unroll.restype = C.c_int
    L.l2o_unroll.argtypes = [C.POINTER(NetCfg), vp, C.POINTER(Problem), vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.l2o_unroll_record.restype = C.c_int
    L.l2o_unroll_record.argtypes = [C.POINTER(NetCfg), vp, C.POINTER(Problem), vp, vp, vp, vp, i32, i32, vp, vp,
                                    C.POINTER(UnrollHist), vp]
    L.l2o_unroll_reduce.restype = C.c_int
    L.l2o_unroll_reduce.argtypes = [C.POINTER(NetCfg), vp, C.POINTER(Problem), vp, vp, vp, vp, vp, i32, i32, i32, vp, vp,
                                    vp, C.POINTER(UnrollHist), vp]
    L.l2o_unroll_workspace_init.restype = C.c_int
    L.l2o_unroll_workspace_init.argtypes = [vp, C.c_size_t, vp]
    L.l2o_cwlstm_wgrad_dims.restype = C.c_int32
    L.l2o_cwlstm_wgrad_dims.argtypes = [C.POINTER(NetCfg), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.l2o_cwlstm_wgrad.restype = C.c_int
    L.l2o_cwlstm_wgrad.argtypes = [C.POINTER(NetCfg), vp, vp, C.c_int64, vp, vp, vp]
    L.l2o_cwlstm_wgrad_compact.restype = C.c_int
    L.l2o_cwlstm_wgrad_compact.argtypes = [C.POINTER(NetCfg), vp, vp, C.c_int32, C.c_int64, vp, vp, vp]
    L.l2o_unroll_workspace_layout.restype = C.c_int64
    L.l2o_unroll_workspace_layout.argtypes = [C.POINTER(NetCfg), C.POINTER(Problem)]
    L.l2o_unroll_workspace_bytes.restype = C.c_size_t
    L.l2o_unroll_workspace_bytes.argtypes = [C.POINTER(NetCfg), C.POINTER(Problem), i32]
    L.l2o_unroll_status.restype = C.c_int
    L.l2o_unroll_status.argtypes = [vp]
    L.l2o_unroll_supported.restype = C.c_int
    L.l2o_unroll_supported.argtypes = [C.POINTER(NetCfg), C.POINTER(Problem)]
    L.l2o_unroll_record_supported.restype = C.c_int
    L.l2o_unroll_record_supported.argtypes = [C.POINTER(NetCfg), C.POINTER(Problem)]
    L.l2o_atb_workspace_bytes.restype = C.c_size_t
    L.l2o_atb_workspace_bytes.argtypes = [i64, i32, i32]
    L.l2o_atb.restype = C.c_int
    L.l2o_atb.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp]
    L.l2o_reduce_fx.restype = C.c_int
    L.l2o_reduce_fx.argtypes = [vp, i32, i32, i32, vp, vp]
    L.l2o_suffix_sums.restype = C.c_int
    L.l2o_suffix_sums.argtypes = [vp, vp, vp, i64, i32, vp]
    L.l2o_colsum_scratch_floats.restype = C.c_size_t
    L.l2o_colsum_scratch_floats.argtypes = [i64, i32]
    L.l2o_colsum.restype = C.c_int
    L.l2o_colsum.argtypes = [vp, i64, i64, i32, vp, i32, vp, vp]
    L.l2o_lincomb.restype = C.c_int
    L.l2o_lincomb.argtypes = [vp, vp, C.c_float, vp, C.c_float, vp, C.c_float, i64, vp]
    L.l2o_rnnprop_input_adjoint.restype = C.c_int
    L.l2o_rnnprop_input_adjoint.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, dbl, dbl, dbl, dbl, vp, vp, vp, i64, vp]
    if L.l2o_abi_version() != L2O_ABI_VERSION:
        raise RuntimeError("libl2o_hip.so ABI version %d != binding version %d"
                           % (L.l2o_abi_version(), L2O_ABI_VERSION))
    _lib = L
    return L


def check(rc):
    """Turn a C-ABI return code into the Python exception the host layer raises."""
    if rc == L2O_OK:
        return
    msg = lib().l2o_last_error().decode("utf-8", "replace")
    if rc == L2O_ERR_UNSUPPORTED:
        raise L2OUnsupported(rc, msg)
    if rc == L2O_ERR_TIMEOUT:
        raise L2OPartnerTimeout(rc, msg)
    if rc == L2O_ERR_ARG:
        raise ValueError("libl2o_hip: " + msg)
    raise L2OError(rc, msg)
