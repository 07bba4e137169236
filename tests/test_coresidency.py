"""Co-residency is sized, not assumed.  The two-CU unroll exchanges data between workgroups that must
be resident together; the library sizes every launch against the CUs the stream can really use -- the device's CU count
(reflects ROC_GLOBAL_CU_MASK), the stream's CU mask, and the capacity MEASURED by l2o_coresident_workgroups (what HIP
cannot see: HSA_CU_MASK).  Under a mask of HALF the CUs the BASELINE config-2 batch (128 problems = 256 workgroups) must
still give the oracle's trajectory -- as two launches of 64 problems -- not a partner timeout (L2O_ERR_HIP)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import oracle as O
from helpers import device_problem, make_params, make_problem, rel_err, spec_of
from oracle.c_oracle import c_unroll
from open_l2o_amd._engine import HipEngine
eng = HipEngine()
cfg = O.DM_IDENTITY
params = make_params(cfg, seed=13, trained_like=True)
B, D, T = 128, 128, 20
prob, x0, arrays = make_problem("quadratic", B, D, seed=14)
fx_ref = c_unroll("quadratic", cfg, params, arrays, x0, T)[0]
spec = spec_of(cfg)
wpack = eng.pack_weights(spec, params)
pd = device_problem(eng, arrays, B, D)
out = {"cus": eng.coresident_cus}
for rep in range(3):
    x, st = eng.tensor(x0), eng.state_alloc(B, D)
    fx_part, fx = eng.zeros((T + 1) * B), eng.zeros(T + 1)
    eng.unroll(spec, wpack, pd, x, st, None, None, T, 1, fx_part, fx=fx)
    eng.synchronize()
    eng.check_unroll_status()                      # raises on a partner timeout
    out["err"] = max(out.get("err", 0.0), rel_err(eng.to_numpy(fx), fx_ref))
print("RESULT " + json.dumps(out))
"""


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_unmasked_device_is_fully_coresident():
    r = _run({})
    assert r["cus"] >= 64 and r["err"] < 1e-5, r


@pytest.mark.parametrize("env", [{"HSA_CU_MASK": "0:0-127"}, {"ROC_GLOBAL_CU_MASK": "0x" + "f" * 32}],
                         ids=["HSA_CU_MASK", "ROC_GLOBAL_CU_MASK"])
def test_two_cu_unroll_under_a_half_cu_mask(env):
    full = _run({})["cus"]
    if full < 256:
        pytest.skip("needs the whole 256-CU device for a half mask of 128")
    r = _run(env)
    print("mask %s: %d CUs co-resident (of %d), rel fx %.3g" % (env, r["cus"], full, r["err"]))
    assert r["cus"] <= 160, r                      # the restriction was SEEN (128 of 256; the probe may count a few late arrivals)
    assert r["err"] < 1e-5, r                      # ... and the batch ran correctly in smaller launches
