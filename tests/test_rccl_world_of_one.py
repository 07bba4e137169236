"""RCCL executes (round 6).  A 1-GPU box cannot host two RCCL ranks, but it can host ONE: a real
`nccl` process group of world size 1 on the MI355X, through which a shard of a sharded job (rank 0 of an emulated world of 8:
contiguous batch slice, 1/B_global = 1/1024) issues every collective of the N > 1 path -- the asynchronous loss all-reduce on
torch's collective stream, the MAX-reduce of the status word and the MIN-reduce of the recovery decision on the compute
stream.  The communicator, the stream hand-over and the enqueue path are the real ones; the results must equal the same
shard run without a process group.  Each case runs in its own process (a process group is process-global state)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import json, os, socket, sys, warnings
import numpy as np
import torch
import torch.distributed as dist
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
from helpers import ORACLE_CFGS, make_params, make_problem
from open_l2o_amd import _engine, _graph_core, meta, meta_rnnprop_eval, problems
from open_l2o_amd.session import Session
from test_meta_api import _net_config

real, netname, fault = %(real)r, %(net)r, %(fault)r
eng = _engine.HipEngine("cuda:0")
_engine.set_default_engine(eng)
info = {}
if real:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    info["backend"] = dist.get_backend()
_graph_core.emulate_world(0, 8, collectives=real)
cfg = ORACLE_CFGS[netname]
params = make_params(cfg, seed=70, trained_like=True)
B, D, T = 1024, 100, 20
prob, x0, _ = make_problem("rastrigin", B, D, seed=71)
problem = problems.rastrigin(B, D, data={"A": prob.A, "B": prob.B, "C": prob.C, "x": x0})
feed = {}
if cfg.kind == "rnnprop":
    opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
    ml, _, _, step = opt.meta_loss(problem, T)
else:
    opt = meta.MetaOptimizer(**_net_config(cfg, params))
    ml = opt.meta_loss(problem, T)
g = opt._graph
assert g.sharded and g.shard == (0, 128)
out, warned = [], 0
with Session() as sess:
    sess.run(ml.reset)
    for i in range(3):
        if cfg.kind == "rnnprop":
            feed[step] = 1 + i * T
        if fault and i == 1:
            eng.inject_unroll_fault()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            loss, fx, _ = sess.run([ml.loss, ml.fx, ml.update], feed_dict=feed)
        warned += sum(issubclass(x.category, RuntimeWarning) for x in w)
        out.append((float(loss), float(fx)))
info.update(out=out, form=eng.last_unroll_form()[0], warned=warned, recoveries=getattr(g, "recoveries", 0))
print("RESULT " + json.dumps(info))
if real:
    dist.destroy_process_group()
'''


def _run(real, net="dm", fault=False):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT, "real": real, "net": net, "fault": fault}], env=env,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


@pytest.mark.parametrize("net", ["dm", "rnnprop"])
def test_shard_through_a_real_rccl_group_equals_the_shard_without_one(net):
    plain = _run(False, net)
    rccl = _run(True, net)
    assert rccl["backend"] == "nccl"
    assert rccl["out"] == plain["out"], (rccl["out"], plain["out"])          # a one-rank sum: bit-identical
    assert rccl["form"] == plain["form"] == "k_unroll_pair"                  # (config 4's shard of 8: the exchanging kernel)


def test_partner_timeout_recovers_through_the_real_collectives():
    """The sharded evaluation recovery of ADVICE r05 on the GPU: the status MAX-reduce and the recovery-decision MIN-reduce
    run through RCCL on the compute stream; the re-run on the exchange-free kernel returns the healthy run's losses."""
    good = _run(True, "dm", fault=False)
    bad = _run(True, "dm", fault=True)
    assert bad["recoveries"] == 1 and bad["warned"] == 1 and good["recoveries"] == 0
    assert bad["form"] == "k_unroll_pair"                                    # (the third unroll is back on the two-CU kernel)
    for (l0, f0), (l1, f1) in zip(good["out"], bad["out"]):
        assert l1 == pytest.approx(l0, rel=1e-5) and f1 == pytest.approx(f0, rel=1e-5)
