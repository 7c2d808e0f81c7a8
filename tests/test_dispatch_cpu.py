"""CPU checks of the host-side dispatch logic added in round 5 (no kernel runs): the exact-zero routing rule of ops.wino_ok and
the hook bookkeeping of the sink route."""
import gc
import os
import sys

import pytest

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))


def test_wino_ok_forward_launches_follow_the_exact_zero_rule():
    """F(4x4) for a forward 3x3 launch only with a dense bias or inside dense_inputs(); exact_zero_inputs() overrides both; data
    gradients (fwd=False) are never affected.  A bias whose flags cannot be evaluated (here: a CPU tensor) reads as not dense."""
    from bmc_hip import ops
    B, H, W, C = 8, 180, 240, 128
    assert ops.wino_ok(B, H, W, C, 9) == 4                                  # data gradient
    assert ops.wino_ok(B, H, W, C, 9, fwd=True) == 2                        # forward, no bias known
    assert ops.wino_ok(B, H, W, C, 9, fwd=True, rule=torch.ones(C)) == 2    # flags unknown -> the exact kernel
    with ops.dense_inputs():
        assert ops.wino_ok(B, H, W, C, 9, fwd=True) == 4
        with ops.exact_zero_inputs():
            assert ops.wino_ok(B, H, W, C, 9, fwd=True) == 2
            assert ops.wino_ok(B, H, W, C, 9) == 4
    assert ops.wino_ok(B, H, W, C, 9, fwd=True) == 2
    # a known-dense bias (cache entry as prime_bias_dense would write it)
    b = torch.full((C,), 0.01)
    import weakref
    ops._DENSE[id(b)] = (weakref.ref(b), b._version, 0.01, 0.01)
    assert ops.bias_dense(b) and ops.bias_positive(b) and ops.wino_ok(B, H, W, C, 9, fwd=True, rule=b) == 4
    assert not ops.bias_dense((b, None)) and not ops.bias_dense(None)
    b.add_(1.0)                                                               # version bump: the entry is stale
    assert not ops.bias_dense(b)
    # sizes / strides
    assert ops.wino_ok(1, 45, 80, C, 9) == 0 or ops.wino_ok(1, 45, 80, C, 9) == 2
    assert ops.wino_ok(B, H, W, C, 9, stride=1 << 14) == 0                    # offsets beyond 32 bits: direct kernel
    assert ops.wino_ok(B, H, W, 32, 9) == 0 and ops.wino_ok(B, H, W, C, 1) == 0


def test_sink_route_recognises_reducer_hooks_by_id():
    """is_sink must know WHOSE hook sits on a parameter.  A reducer's own hooks are fine; a foreign hook, a second
    reducer that was detached, or the hook that takes the place of a reducer whose hooks were removed are not."""
    from bmc_hip import ops
    from bmc_hip.parallel import GradAllReducer
    net = torch.nn.Linear(4, 4)
    p = net.weight
    assert ops.is_sink(p)
    red = GradAllReducer(net)
    assert ops.is_sink(p) and len(p._bmc_sink_hooks) == 1
    red2 = GradAllReducer(net)                       # two reducers on one model: both known
    assert ops.is_sink(p) and len(p._bmc_sink_hooks) == 2
    red2.detach()
    assert ops.is_sink(p) and len(p._bmc_sink_hooks) == 1
    h = p.register_post_accumulate_grad_hook(lambda t: None)
    assert not ops.is_sink(p)                        # a foreign hook beside the reducer's
    h.remove()
    assert ops.is_sink(p)
    red.detach()
    assert ops.is_sink(p) and not p._bmc_sink_hooks
    # a reducer whose hooks were taken off by hand and that was then dropped (no detach(); round 6: the reducer has no __del__ --
    # its hooks keep it alive, tests/test_distributed_gloo.py::test_reducer_lives_with_the_model_until_detach): the ids it
    # left behind vouch for nothing, because hook ids are never re-used -- the next hook on the parameter is a foreign one
    red3 = GradAllReducer(net)
    ids = set(p._bmc_sink_hooks)
    for hd in red3._handles:
        hd.remove()
    del red3
    gc.collect()
    assert ops.is_sink(p)                            # no hook at all: a sink by the basic rule
    h2 = p.register_post_accumulate_grad_hook(lambda t: None)
    assert h2.id not in ids and not ops.is_sink(p)


def test_wino_rows_rule_matches_the_host_side_tile_count():
    """csrc/wino.hip::wino_rows (exported as bmc_conv_wino_rows, a host-only function): 4-row tiles where the 8-row tiling
    needs at most two rounds of the 256 CUs and the 4-row one 0.6 x the rounds fewer; ops.wino_tiles counts with it."""
    from bmc_hip import lib, ops
    ops._WINO_ROWS.clear()
    for (B, H, W), th in {(8, 31, 56): 4, (16, 31, 56): 8, (4, 45, 80): 4, (8, 45, 80): 8, (8, 64, 96): 4, (8, 90, 120): 8,
                          (8, 180, 240): 8, (2, 80, 80): 4, (4, 72, 80): 8}.items():
        assert lib._wino_rows(B, H, W, 128, 0) == th, (B, H, W)
        assert ops.wino_tiles(B, H, W, 128) == (B * ((H + th - 1) // th) * ((W + 15) // 16), th)
    assert lib._wino_rows(8, 31, 56, 256, 0) == 8              # two channel tiles double the count: one round of 8-row tiles


def test_bench_ranks_pin_to_disjoint_runs_of_physical_cores():
    """bench.pin_rank_to_cores (round 6): every rank of `bench.py --gpus N` binds its host threads to its own run of physical cores
    (SMT siblings included) BEFORE its first GPU call; the runs of different ranks are disjoint, a single rank is left alone, and
    BMC_BENCH_PIN=0 switches it off.  Run in child processes: the call changes the caller's affinity."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib.util, json, os, sys\n"
            "spec = importlib.util.spec_from_file_location('bench', os.path.join(%r, 'bench.py')); b = importlib.util.module_from_spec(spec)\n"
            "sys.argv = ['bench.py']; spec.loader.exec_module(b)\n"
            "before = sorted(os.sched_getaffinity(0))\n"
            "info = b.pin_rank_to_cores(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(os.environ['W']))\n"
            "print(json.dumps({'info': info, 'before': before, 'after': sorted(os.sched_getaffinity(0))}))\n") % root

    def run(rank, world, pin="1"):
        env = dict(os.environ, W=str(world), BMC_BENCH_PIN=pin)
        r = subprocess.run([sys.executable, "-c", code.replace("sys.argv[1]) if len(sys.argv) > 1 else 0", "%d) if True else 0" % rank)],
                           env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-1000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    one = run(0, 1)
    assert one["info"] is None and one["after"] == one["before"]
    if len(one["before"]) < 2:
        pytest.skip("one CPU: nothing to share out")
    a, b = run(0, 2), run(1, 2)
    assert a["info"] and b["info"] and a["info"]["physical_cores"] == b["info"]["physical_cores"] >= 1
    assert set(a["after"]).isdisjoint(b["after"]) and set(a["after"]) | set(b["after"]) <= set(one["before"])
    assert a["after"] == list(range(a["info"]["first"], a["info"]["last"] + 1)) or len(a["after"]) == a["info"]["logical_cpus"]
    off = run(1, 2, pin="0")
    assert off["info"] is None and off["after"] == off["before"]
