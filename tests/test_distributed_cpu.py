"""world_size-2 test of the multi-GPU path on CPU (gloo): the problem batch is sharded
by contiguous slices, 1/B stays the global batch, the per-step partial losses are
all-reduced -- and the result equals the single-process run.  The arithmetic is the
oracle-backed test engine (tests/oracle_engine.py); what is under test is
open_l2o_amd.meta's sharding / collective logic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(kind, netname, B, D, T, sharded_world):
    """Build + run two chained unrolls; returns (fx_T list, loss list, local x)."""
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle as O
    from helpers import ORACLE_CFGS, make_params, make_problem
    from open_l2o_amd import _engine, meta, meta_rnnprop_eval, problems
    from open_l2o_amd.session import Session
    from oracle_engine import OracleEngine
    from test_meta_api import _net_config

    _engine.set_default_engine(OracleEngine())
    cfg = ORACLE_CFGS[netname]
    params = make_params(cfg, seed=50, trained_like=True)
    prob, x0, _ = make_problem(kind, B, D, seed=51)
    if kind == "quadratic":
        problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
    else:
        problem = problems.rastrigin(B, D, data={"A": prob.A, "B": prob.B, "C": prob.C, "x": x0})
    feed = {}
    if cfg.kind == "rnnprop":
        opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
        ml, _, _, step = opt.meta_loss(problem, T)
        feed = {step: 1}
    else:
        opt = meta.MetaOptimizer(**_net_config(cfg, params))
        ml = opt.meta_loss(problem, T)
    out = []
    with Session() as sess:
        sess.run(ml.reset)
        for i in range(2):
            if feed:
                feed[step] = 1 + i * T
            loss, fx, x, _ = sess.run([ml.loss, ml.fx, ml.x, ml.update], feed_dict=feed)
            out.append((float(loss), float(fx)))
    assert opt._graph.sharded == (sharded_world > 1)
    return out, x[0]


def _worker(rank, world, port, kind, netname, B, D, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out, x = _run(kind, netname, B, D, T, world)
        q.put((rank, out, x))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,netname", [("quadratic", "dm"), ("rastrigin", "rnnprop")])
def test_two_rank_sharding_matches_single_process(kind, netname):
    B, D, T = 8, 10, 5
    ref_out, ref_x = _run(kind, netname, B, D, T, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, netname, B, D, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        rank, out, x = q.get(timeout=240)
        results[rank] = (out, x)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        out, x = results[rank]
        # every rank sees the GLOBAL loss (all-reduce), equal to the single-process value
        np.testing.assert_allclose(np.array(out), np.array(ref_out), rtol=2e-6)
        np.testing.assert_allclose(x, ref_x[rank * 4:(rank + 1) * 4], rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------------
# sharded meta-TRAINING: the weight gradients are summed over ranks (one flat all-reduce per
# network) and every replica takes the same Adam step as the single-process run
# ---------------------------------------------------------------------------
def _train(netname, B, D, T, nsteps, if_scale=False):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import ORACLE_CFGS, make_params, make_problem
    from open_l2o_amd import _engine, meta, meta_dm_train, meta_rnnprop_eval, problems, util
    from open_l2o_amd.session import Session
    from oracle_engine import OracleEngine
    from test_meta_api import _net_config

    _engine.set_default_engine(OracleEngine())
    cfg = ORACLE_CFGS[netname]
    params = make_params(cfg, seed=60, trained_like=True)
    prob, x0, _ = make_problem("quadratic", B, D, seed=61)
    problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
    np.random.seed(7 + (dist.get_rank() if dist.is_initialized() else 0))     # ranks start UNSYNCHRONISED on purpose
    if if_scale:
        opt = meta_dm_train.MetaOptimizer(0, **_net_config(cfg, params))
        out = opt.meta_minimize(problem, T, learning_rate=1e-2)
        ms, scale, var_x = out[0], out[1], out[2]
        costs = []
        with Session() as sess:
            for _ in range(nsteps):
                _, cost = util.run_epoch(sess, ms.fx, [ms.update, ms.step], ms.reset, 2, scale=scale, rd_scale=True,
                                         rd_scale_bound=1.0, var_x=var_x,
                                         assign_func=lambda vals: [v.load(a) for v, a in zip(var_x, vals)])
                costs.append(float(cost))
        return opt.save(), costs
    feed = {}
    if cfg.kind == "rnnprop":
        opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
        out = opt.meta_minimize(problem, T, learning_rate=1e-2)
        ms, step = out[0], out[3]
        feed = {step: 1}
    else:
        opt = meta.MetaOptimizer(**_net_config(cfg, params))
        ms = opt.meta_minimize(problem, T, learning_rate=1e-2)
    costs = []
    with Session() as sess:
        sess.run(ms.reset)
        for i in range(nsteps):
            if feed:
                feed[step] = 1 + i * T
            costs.append(float(sess.run([ms.fx, ms.update, ms.step], feed_dict=feed)[0]))
    return opt.save(), costs


def _train_worker(rank, world, port, args, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _train(*args))
    finally:
        dist.destroy_process_group()


def _two_ranks(args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, args, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        rank, weights, costs = q.get(timeout=300)
        results[rank] = (weights, costs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def _assert_same_weights(a, b, tol):
    assert a.keys() == b.keys()
    for net in a:
        for mod in a[net]:
            for var in a[net][mod]:
                ref = np.asarray(a[net][mod][var])
                np.testing.assert_allclose(np.asarray(b[net][mod][var]), ref, rtol=0, atol=tol * max(1.0, float(np.abs(ref).max())),
                                           err_msg="%s/%s/%s" % (net, mod, var))


@pytest.mark.parametrize("netname", ["dm", "rnnprop"])
def test_two_rank_meta_training_matches_single_process(netname):
    """the per-block weight gradients are non-contiguous views of A^T Bm; they must be
    reduced as one contiguous buffer.  lr = 1e-2: a sign flip of one gradient entry moves a weight by
    2e-2, four orders of magnitude above the tolerance."""
    args = (netname, 8, 16, 4, 2)
    ref_w, ref_c = _train(*args)
    results = _two_ranks(args)
    for rank in (0, 1):
        w, c = results[rank]
        np.testing.assert_allclose(c, ref_c, rtol=5e-6)
        _assert_same_weights(ref_w, w, 2e-6)
    _assert_same_weights(results[0][0], results[1][0], 0.0)      # the replicas stay bit-identical


def test_two_rank_random_scaling_epoch():
    """util.run_epoch(rd_scale=True) under batch sharding: the scale factors are drawn with the global
    shape on rank 0 and broadcast, each rank rescales its own shard; the replicas agree bit for bit."""
    args = ("dm", 8, 16, 3, 2, True)
    results = _two_ranks(args)
    np.testing.assert_allclose(results[0][1], results[1][1], rtol=0)
    _assert_same_weights(results[0][0], results[1][0], 0.0)


# ---------------------------------------------------------------------------
# a partner timeout on ONE rank (round 5): the status words are MAX-reduced ahead of the guarded meta-step, so
# BOTH ranks skip that update and both raise -- the replicas cannot drift apart
# ---------------------------------------------------------------------------
def _train_with_fault(faulty_rank):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import ORACLE_CFGS, make_params, make_problem
    from open_l2o_amd import _abi, _engine, meta, problems
    from open_l2o_amd.session import Session
    from oracle_engine import OracleEngine
    from test_meta_api import _net_config

    eng = OracleEngine()
    _engine.set_default_engine(eng)
    cfg = ORACLE_CFGS["dm"]
    params = make_params(cfg, seed=60, trained_like=True)
    prob, x0, _ = make_problem("quadratic", 8, 16, seed=61)
    problem = problems.quadratic(8, 16, data={"w": prob.w, "y": prob.y, "x": x0})
    opt = meta.MetaOptimizer(**_net_config(cfg, params))
    ms = opt.meta_minimize(problem, 4, learning_rate=1e-2)
    raised = False
    with Session() as sess:
        sess.run(ms.reset)
        sess.run([ms.fx, ms.update, ms.step])                       # a good step on both ranks
        after_good = opt.save()
        if dist.is_initialized() and dist.get_rank() == faulty_rank:
            eng.inject_unroll_fault()
        try:
            sess.run([ms.fx, ms.update, ms.step])
        except _abi.L2OPartnerTimeout:
            raised = True
    return after_good, opt.save(), raised, list(eng.calls), opt.graph._adam["t"]


def _fault_worker(rank, world, port, faulty_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _train_with_fault(faulty_rank))
    finally:
        dist.destroy_process_group()


def test_two_rank_partner_timeout_skips_the_update_on_both_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fault_worker, args=(r, 2, port, 1, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        item = q.get(timeout=300)
        results[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        good, final, raised, calls, t = results[rank]
        assert raised, "rank %d did not learn of rank 1's timeout" % rank
        assert "adam_step_skipped" in calls and calls.count("adam_step") == 1, calls
        assert t == 1                                               # (the skipped step's count was taken back)
        _assert_same_weights(good, final, 0.0)                      # the failed step changed nothing ...
    assert "unroll_timeout" in results[1][3] and "unroll_timeout" not in results[0][3]
    _assert_same_weights(results[0][1], results[1][1], 0.0)         # ... and the replicas are still identical


# ---------------------------------------------------------------------------
# a partner timeout on ONE rank of a sharded EVALUATION (round 6): the status word is MAX-reduced ahead of the
# host check, so BOTH ranks re-run the unroll together on the exchange-free kernels -- the loss all-reduces stay paired,
# nobody keeps the contaminated fx, and the four runs give the single-process answer
# ---------------------------------------------------------------------------
def _eval_with_fault(faulty_rank, fault_before_run=1, runs=4):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import warnings
    from helpers import ORACLE_CFGS, make_params, make_problem
    from open_l2o_amd import _engine, meta, problems
    from open_l2o_amd.session import Session
    from oracle_engine import OracleEngine
    from test_meta_api import _net_config

    eng = OracleEngine()
    _engine.set_default_engine(eng)
    cfg = ORACLE_CFGS["dm"]
    params = make_params(cfg, seed=60, trained_like=True)
    prob, x0, _ = make_problem("quadratic", 8, 16, seed=61)
    problem = problems.quadratic(8, 16, data={"w": prob.w, "y": prob.y, "x": x0})
    opt = meta.MetaOptimizer(**_net_config(cfg, params))
    ml = opt.meta_loss(problem, 4)
    out, warned = [], 0
    with Session() as sess:
        sess.run(ml.reset)
        for i in range(runs):
            if i == fault_before_run and dist.is_initialized() and dist.get_rank() == faulty_rank:
                eng.inject_unroll_fault()
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                fx, _ = sess.run([ml.fx, ml.update])
            warned += sum(issubclass(x.category, RuntimeWarning) for x in w)
            out.append(float(fx))
    return out, warned, list(eng.calls), getattr(opt._graph, "recoveries", 0)


def _eval_fault_worker(rank, world, port, faulty_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _eval_with_fault(faulty_rank))
    finally:
        dist.destroy_process_group()


def test_two_rank_partner_timeout_in_evaluation_recovers_on_both_ranks():
    ref, _, _, _ = _eval_with_fault(faulty_rank=-1)                 # single process, the global batch, no fault
    assert np.all(np.isfinite(ref))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_fault_worker, args=(r, 2, port, 1, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        item = q.get(timeout=300)
        results[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        out, warned, calls, recoveries = results[rank]
        assert np.all(np.isfinite(out)), "rank %d kept a contaminated loss: %r" % (rank, out)
        np.testing.assert_allclose(out, ref, rtol=2e-6)             # (the all-reduce adds two partial means)
        assert recoveries == 1 and warned == 1, (rank, recoveries, warned)
    assert "unroll_timeout" in results[1][2] and "unroll_timeout" not in results[0][2]
    assert results[0][0] == results[1][0]                            # same all-reduced losses on both ranks, all four runs


# ---------------------------------------------------------------------------
# the loss all-reduce is deferred and coalesced (round 6): launches that nobody reads cost one collective per FX_RING - 1
# unrolls, and what is read afterwards is the all-reduced loss of the right unroll
# ---------------------------------------------------------------------------
def _launch_many(n_launch, count):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import ORACLE_CFGS, make_params, make_problem
    from open_l2o_amd import _engine, _graph_core, meta, problems
    from oracle_engine import OracleEngine
    from test_meta_api import _net_config

    _engine.set_default_engine(OracleEngine())
    cfg = ORACLE_CFGS["dm"]
    params = make_params(cfg, seed=62, trained_like=True)
    prob, x0, _ = make_problem("quadratic", 8, 16, seed=63)
    problem = problems.quadratic(8, 16, data={"w": prob.w, "y": prob.y, "x": x0})
    opt = meta.MetaOptimizer(**_net_config(cfg, params))
    opt.meta_loss(problem, 3)
    g = opt.graph
    g.reset()
    calls = []
    if count:
        real = meta._all_reduce

        def counting(t, async_op=False, op=None):
            calls.append(int(t.numel()))
            return real(t, async_op=async_op, op=op)
        meta._all_reduce = counting
    last = None
    for _ in range(n_launch):
        last, _ = g.launch({}, commit=True)
    g.wait_fx()
    return [float(v) for v in last], calls, g.FX_RING if g.sharded else 1


def _launch_many_worker(rank, world, port, n_launch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _launch_many(n_launch, True))
    finally:
        dist.destroy_process_group()


def test_two_rank_loss_all_reduce_is_deferred_and_coalesced():
    n_launch = 150
    ref, _, _ = _launch_many(n_launch, False)                          # single process, the global batch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_launch_many_worker, args=(r, 2, port, n_launch, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        item = q.get(timeout=300)
        results[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        fx, calls, ring = results[rank]
        np.testing.assert_allclose(fx, ref, rtol=2e-6)
        assert ring == 64
        # 150 unrolls nobody read: a collective when the ring is about to wrap (every 63 launches; a wrapping run is two
        # contiguous pieces) + the one wait_fx() issues -- not 150
        assert len(calls) <= 6, calls
        assert sum(calls) == n_launch * 4, calls                       # every unroll's T + 1 = 4 losses reduced exactly once
    assert results[0][0] == results[1][0]
