"""Debug probe for k_mlp_xcd: which (net, T, instances) combinations complete."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["L2O_NO_RECOVERY"] = "1"
import torch
import oracle as O
from helpers import make_params
from open_l2o_amd import _engine, meta, meta_rnnprop_eval, problems
from open_l2o_amd.replicas import Replicas
from test_meta_api import _net_config

eng = _engine.HipEngine()
_engine.set_default_engine(eng)
_orig_check = eng.check_unroll_status
def _check():
    ws = eng._last_ws
    if ws is not None:
        hdr = ws[:64].view(torch.int32).cpu().numpy()
        if hdr[0]:
            print("   header: status %d seq %d fault %d | first timeout: step %d phase %d, instance %d member %d | team %s" % (
                hdr[0], hdr[1], hdr[2], hdr[8] // 16, hdr[8] % 16, hdr[9] // 32, hdr[9] % 32,
                ws[192:224].view(torch.int32).cpu().numpy().tolist()), flush=True)
    _orig_check()
eng.check_unroll_status = _check
data = problems.synthetic_mnist(512, seed=3)
for netname in (sys.argv[1:] or ["rnnprop", "dm_logsign", "dm"]):
    cfg = {"rnnprop": O.RNNPROP, "dm_logsign": O.DM_LOGSIGN, "dm": O.DM_IDENTITY}[netname]
    params = make_params(cfg, seed=71, trained_like=True)
    for T in ((5, 50, 200) if os.environ.get('XCD_FULL') else (5,)):
        for n in ((1, 2, 3, 8) if os.environ.get('XCD_FULL') else (1, 2)):
            meta.set_random_seed(9)
            probs = [problems.mnist(layers=(20,), batch_size=64, data=data) for _ in range(n)]
            opt = (meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp")) if cfg.kind == "rnnprop"
                   else meta.MetaOptimizer(**_net_config(cfg, params)))
            reps = Replicas(opt, probs, T)
            reps.reset()
            feed = {reps.step: 1} if cfg.kind == "rnnprop" else None
            t0 = time.time()
            try:
                fx = reps.run(feed, form="xcd")
                torch.cuda.synchronize()
                msg = "ok fx=%s" % np.round(fx[:3], 4)
            except Exception as e:
                msg = "FAIL %s" % str(e)[:60]
            print("%-10s T=%3d n=%d: %s (%.3f s)" % (netname, T, n, msg, time.time() - t0), flush=True)
