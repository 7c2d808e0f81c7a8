"""N > 1 path on CPU: two gloo ranks, each with half of the sequence batch; the bucketed gradient all-reduce
(bmc_hip.parallel.GradAllReducer) must reproduce the single-process full-batch gradient and keep the
`loss.backward(); optimizer.step()` loop shape.  Gradients come from the CPU oracle (test infrastructure)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleNet(torch.nn.Module):
    """Plain-PyTorch stand-in with shared parameters (same aliasing pattern as BMCNet: one tensor used many times)."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(7)
        self.a = torch.nn.Parameter(torch.randn(8, 8, generator=g) * 0.3)
        self.b = torch.nn.Parameter(torch.randn(8, generator=g) * 0.1)
        self.unused = torch.nn.Parameter(torch.zeros(3))
        self.c = torch.nn.Parameter(torch.randn(8, 4, generator=g) * 0.3)

    def forward(self, x):
        h = x
        for _ in range(3):                      # weight shared across "blocks" and "windows"
            h = torch.tanh(h @ self.a + self.b)
        return h @ self.c


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmc_hip.parallel import GradAllReducer, reduce_tensor
    torch.manual_seed(0)
    net = OracleNet()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, weight_decay=1e-5, amsgrad=True)
    GradAllReducer(net, opt, bucket_mb=1e-4)          # tiny buckets -> several all-reduces in flight
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 8, generator=g); y = torch.randn(8, 4, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(2):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(net(xs), ys)
        loss.backward()
        opt.step()
    rl = reduce_tensor(loss)
    q.put((rank, [p.detach().numpy().copy() for p in net.parameters()], float(rl)))
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_full_batch():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full batch
    torch.manual_seed(0)
    net = OracleNet()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, weight_decay=1e-5, amsgrad=True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 8, generator=g); y = torch.randn(8, 4, generator=g)
    for _ in range(2):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(net(x), y)     # mean over the full batch = mean of the per-rank means
        loss.backward()
        opt.step()
    ref = [p.detach().numpy() for p in net.parameters()]
    for r in range(2):
        for a, b in zip(res[r][1], ref):
            assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert np.allclose(res[0][1][0], res[1][1][0])          # ranks stay in lock-step
    assert abs(res[0][2] - res[1][2]) < 1e-7


def test_single_process_reducer_is_transparent():
    sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
    from bmc_hip.parallel import GradAllReducer
    net = OracleNet()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    GradAllReducer(net, opt)
    x = torch.randn(4, 8)
    net(x).sum().backward()
    g0 = net.a.grad.clone()
    opt.step()
    assert torch.equal(net.a.grad, g0) and net.unused.grad is None


# ---------------------------------------------------------------------------------------------------------------------
# The real training step, sharded: models.BMCNet's own parameter containers (real names, real aliasing), the real
# train_step.bptt_step + shard_sequences + GradAllReducer; only the arithmetic of a window comes from the CPU oracle
# (there is no CPU path in the product, and no GPU here).
class OracleBackedBMCNet(torch.nn.Module):
    def __init__(self, scale, n_c, n_b):
        super().__init__()
        sys.path[:0] = [ROOT, os.path.join(ROOT, "bmcnet-esr_amd")]
        from models.BMCNet import BMCNet
        self.net = BMCNet(scale, n_c, n_b)           # parameters only; its forward needs the MI355X
        self.scale = scale

    def forward(self, x, h, hp, hn, o, init):
        from oracle import bmc_oracle as O
        params = {k: v for k, v in self.net.state_dict(keep_vars=True).items()}
        return O.bmcnet_forward(params, x, h, hp, hn, o, init, self.scale)


def _bmc_worker(rank, world, port, q, accumulate):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bmcnet-esr_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmc_hip.parallel import GradAllReducer
    from train_step import bptt_step, shard_sequences
    scale, n_c, n_b, B, L, H, W = 4, 16, 1, max(4, world), 3, 6, 8
    torch.manual_seed(3)
    net = OracleBackedBMCNet(scale, n_c, n_b)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, amsgrad=True)
    GradAllReducer(net, opt, bucket_mb=0.02)
    g = torch.Generator().manual_seed(4)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.4), generator=g)
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.4), generator=g)
    si, sg = shard_sequences(inp, gt, rank, world)
    if accumulate:      # two backward() calls before one step(): the reducer must reduce the ACCUMULATED gradient once
        opt.zero_grad()
        from oracle import bmc_oracle as O
        for half in (slice(0, 1), slice(1, 2)):
            z = lambda c: torch.zeros(1, c, H, W)
            h, hp, hn, pred = net(si[half, 0:2].transpose(1, 2), z(n_c), z(n_c), z(n_c), z(2 * scale * scale), True)
            torch.nn.functional.mse_loss(pred, sg[half, 1]).backward()
        opt.step()
    else:
        for _ in range(2):
            bptt_step(net, opt, si, sg, n_c, scale)
    q.put((rank, [p.detach().numpy().copy() for p in net.parameters()]))
    dist.destroy_process_group()


def _run_bmc(accumulate, world=2):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bmc_worker, args=(r, world, port, q, accumulate)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_rank_sharded_bmcnet_bptt_matches_full_batch():
    res = _run_bmc(False)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bmcnet-esr_amd")]
    from train_step import bptt_step
    scale, n_c, n_b, B, L, H, W = 4, 16, 1, 4, 3, 6, 8
    torch.manual_seed(3)
    net = OracleBackedBMCNet(scale, n_c, n_b)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, amsgrad=True)
    g = torch.Generator().manual_seed(4)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.4), generator=g)
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.4), generator=g)
    for _ in range(2):
        bptt_step(net, opt, inp, gt, n_c, scale)          # full batch, one process
    ref = [p.detach().numpy() for p in net.parameters()]
    assert len(ref) == 54
    for r in range(2):
        for a, b in zip(res[r][1], ref):
            assert np.allclose(a, b, rtol=2e-4, atol=2e-6)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)                       # ranks in lock-step, bit for bit


@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_rank_sharded_bmcnet_bptt_matches_full_batch(world):
    """The bucket order / finish() staging of bmc_hip.parallel at the world sizes the scaling runs use (only ever
    seen at 2): 4 and 8 gloo ranks on the real bptt_step, one sequence (world 8) or two (world 4) per rank, the parameter the
    loss does not reach (the last block's v2) included; == the single-process full batch, ranks bit-equal."""
    res = _run_bmc(False, world)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bmcnet-esr_amd")]
    from train_step import bptt_step
    scale, n_c, n_b, B, L, H, W = 4, 16, 1, max(4, world), 3, 6, 8
    torch.manual_seed(3)
    net = OracleBackedBMCNet(scale, n_c, n_b)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, amsgrad=True)
    g = torch.Generator().manual_seed(4)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.4), generator=g)
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.4), generator=g)
    for _ in range(2):
        bptt_step(net, opt, inp, gt, n_c, scale)
    ref = [p.detach().numpy() for p in net.parameters()]
    # (two Adam steps: an element whose gradient is rounding noise moves by +-lr whichever way the summation order tips it, so
    #  the comparison with the single-process run is a norm per tensor; between ranks it is bit for bit)
    for r in range(world):
        for a, b in zip(res[r][1], ref):
            assert np.linalg.norm(a.astype(np.float64) - b) <= 2e-3 * max(np.linalg.norm(b), 1e-12), (r, a.shape)
    for r in range(1, world):
        for a, b in zip(res[0][1], res[r][1]):
            assert np.array_equal(a, b)


def test_two_rank_gradient_accumulation_is_reduced_once():
    """Two backward() calls before step(): hooks fire twice per parameter; the reducer must fall back to one
    reduction of the accumulated gradients at step time instead of racing / double-launching."""
    res = _run_bmc(True)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bmcnet-esr_amd")]
    scale, n_c, n_b, B, L, H, W = 4, 16, 1, 4, 3, 6, 8
    torch.manual_seed(3)
    net = OracleBackedBMCNet(scale, n_c, n_b)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, amsgrad=True)
    g = torch.Generator().manual_seed(4)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.4), generator=g)
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.4), generator=g)
    opt.zero_grad()
    z = lambda c: torch.zeros(1, c, H, W)
    total = 0
    for b in range(4):       # gradient of the mean over ranks of (sum over each rank's two single-sequence losses)
        h, hp, hn, pred = net(inp[b:b + 1, 0:2].transpose(1, 2), z(n_c), z(n_c), z(n_c), z(2 * scale * scale), True)
        total = total + torch.nn.functional.mse_loss(pred, gt[b:b + 1, 1])
    (total / 2).backward()
    opt.step()
    ref = [p.detach().numpy() for p in net.parameters()]
    for r in range(2):
        for a, b in zip(res[r][1], ref):
            assert np.allclose(a, b, rtol=2e-4, atol=2e-6)


# ---------------------------------------------------------------------------------------------------------------------
# The sink route (bmc_hip.ops.sink_group): the HIP kernels add weight gradients straight into .grad and hand autograd
# None, so no post-accumulate hook fires for those parameters.  Emulated here on CPU with a Function that follows the
# same protocol, including a parameter that takes BOTH routes in one backward (nothing enforces that no
# parameter does).
class SinkMatmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.param = w
        return x @ w

    @staticmethod
    def backward(ctx, g):
        from bmc_hip import ops
        x, w = ctx.saved_tensors
        gw = x.t() @ g
        sg = ops.sink_group([ctx.param])
        if sg is None:
            return g @ w.t(), gw
        (dst,), acc = sg
        dst.add_(gw) if acc else dst.copy_(gw)
        return g @ w.t(), None


class SinkNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.a = torch.nn.Parameter(torch.randn(8, 8, generator=g) * 0.3)       # sink route, shared by three uses
        self.b = torch.nn.Parameter(torch.randn(8, generator=g) * 0.1)          # autograd route
        self.mixed = torch.nn.Parameter(torch.randn(8, 8, generator=g) * 0.3)   # sink route early, autograd route late
        self.c = torch.nn.Parameter(torch.randn(8, 4, generator=g) * 0.3)       # sink route

    def forward(self, x, sink=True):
        mm = SinkMatmul.apply if sink else torch.matmul
        h = mm(x, self.mixed)
        for _ in range(3):
            h = torch.tanh(mm(h, self.a) + self.b)
        h = h @ self.mixed                    # the late autograd use: its AccumulateGrad runs BEFORE the early sink add
        return mm(h, self.c)


def _sink_worker(rank, world, port, q, accum):
    sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmc_hip import ops
    from bmc_hip.parallel import GradAllReducer
    ops.set_accumulate_param_grads(accum)
    net = SinkNet()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, weight_decay=1e-5, amsgrad=True)
    red = GradAllReducer(net, opt, bucket_mb=1e-4)          # one parameter per bucket: hooks launch whatever they complete
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 8, generator=g); y = torch.randn(8, 4, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(xs), ys).backward()
        opt.step()
    # one more step with .grad left aliasing the buckets (set_to_none=False): legal -- a Function that sinks into a
    # parameter has it as an input, so autograd runs the parameter's AccumulateGrad (and our hook) only after that
    # Function's backward: a hook never fires before a sink add of the same backward
    opt.zero_grad(set_to_none=False)
    torch.nn.functional.mse_loss(net(xs), ys).backward()
    opt.step()
    q.put((rank, [p.detach().numpy().copy() for p in net.parameters()]))
    dist.destroy_process_group()


@pytest.mark.parametrize("accum", [True, False])
def test_two_rank_sink_route_and_mixed_route_parameters(accum):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sink_worker, args=(r, 2, port, q, accum)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    net = SinkNet()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, weight_decay=1e-5, amsgrad=True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 8, generator=g); y = torch.randn(8, 4, generator=g)
    for _ in range(4):
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(x, sink=False), y).backward()        # full batch, plain autograd
        opt.step()
    ref = [p.detach().numpy() for p in net.parameters()]
    for r in range(2):
        for a, b in zip(res[r][1], ref):
            assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)


def test_reducer_refuses_a_sink_add_behind_a_launched_aliased_bucket():
    """The one order the reducer cannot repair (and that no Function of this repo can produce, see above): a bucket that its
    hooks completed and launched, then a kernel adding into a .grad that IS the bucket.  finish() must raise, not average
    garbage.  Driven by hand in one process with a pretended world size."""
    sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
    from bmc_hip import ops
    from bmc_hip.parallel import GradAllReducer
    ops.set_accumulate_param_grads(True)
    net = SinkNet()
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    red = GradAllReducer(net, opt, bucket_mb=1e-4)
    red.world = 2
    x = torch.randn(4, 8)
    net(x).sum().backward()
    opt.step()                                   # .grad are views of the buckets from here on
    opt.zero_grad(set_to_none=False)
    p = net.c
    assert p.grad.data_ptr() == red.flat[red.slot[p][0]][red.slot[p][1]:].data_ptr()
    red._on_grad(p)                              # "autograd finished this parameter": its bucket is complete and launched
    assert red.pending[red.slot[p][0]] == 0
    (dst,), acc = ops.sink_group([p])            # ... and then a kernel adds into it
    dst.add_(1.0)
    with pytest.raises(RuntimeError, match="aliases the reduction bucket"):
        red.finish()


def test_hooked_parameters_keep_the_autograd_route():
    """bmc_hip.ops.is_sink: a parameter with somebody else's tensor hook or post-accumulate hook must get its gradient
    through autograd (the hook would never fire otherwise); the reducer's own hooks do not count."""
    sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
    from bmc_hip import ops
    from bmc_hip.parallel import GradAllReducer
    ops.set_accumulate_param_grads(True)
    net = SinkNet()
    assert ops.is_sink(net.a) and ops.is_sink(net.c)
    fired = []
    net.c.register_post_accumulate_grad_hook(lambda p: fired.append("post"))
    net.a.register_hook(lambda g: fired.append("tensor") or g)
    assert not ops.is_sink(net.c) and not ops.is_sink(net.a)
    net(torch.randn(4, 8)).sum().backward()
    assert "post" in fired and "tensor" in fired and net.c.grad is not None
    net2 = SinkNet()
    red2 = GradAllReducer(net2)
    assert all(ops.is_sink(p) for p in net2.parameters())
    ops.set_accumulate_param_grads(False)
    assert not ops.is_sink(net2.a)
    ops.set_accumulate_param_grads(True)
    # somebody else's hook registered BESIDE the reducer's (a clipping / logging hook added later): autograd route again
    h = net2.a.register_post_accumulate_grad_hook(lambda p: None)
    assert not ops.is_sink(net2.a) and ops.is_sink(net2.c)
    h.remove()
    assert ops.is_sink(net2.a)
    # ... or BEFORE it: the reducer does not claim such a parameter
    net3 = SinkNet()
    net3.a.register_hook(lambda g: g)
    red3 = GradAllReducer(net3)
    assert not ops.is_sink(net3.a) and ops.is_sink(net3.c)
    # detach(): the model outlives its reducer -- hooks gone, parameters plain again, later hooks are respected
    red2.detach()
    assert all(ops.is_sink(p) for p in net2.parameters())            # (no hooks at all: a sink by the basic rule)
    net2.c.register_post_accumulate_grad_hook(lambda p: None)
    assert not ops.is_sink(net2.c)
    red3.detach()


def test_reducer_lives_with_the_model_until_detach():
    """round 5's `__del__` claimed to clean up after a reducer dropped without detach(); it never ran, because the
    parameters hold the reducer through its hooks.  That lifetime is the contract now (a bare `GradAllReducer(model, opt)` statement
    is a complete set-up -- _worker above relies on it): dropping the caller's reference changes nothing, detach() ends it."""
    import gc
    import weakref
    sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
    from bmc_hip import ops
    from bmc_hip.parallel import GradAllReducer
    ops.set_accumulate_param_grads(True)
    net = SinkNet()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    red = GradAllReducer(net, opt)
    assert not hasattr(GradAllReducer, "__del__")
    wr = weakref.ref(red)
    del red
    gc.collect()
    assert wr() is not None                                           # the model's hooks keep it
    assert all(len(p._bmc_sink_hooks) == 1 and len(p._post_accumulate_grad_hooks) == 1 for p in net.parameters())
    net(torch.randn(4, 8)).sum().backward()
    opt.step()                                                        # finish() ran from the pre-step hook
    assert not wr().seen
    wr().detach()
    gc.collect()
    assert wr() is None
    assert all(not p._bmc_sink_hooks and not p._post_accumulate_grad_hooks for p in net.parameters())


class DeferredSinkMatmul(torch.autograd.Function):
    """The protocol of a MERGED weight gradient (bmc_hip.ops.wgrad_wino / wgrad_pgemm with BMC_WGRAD_MERGE > 1): autograd gets None for
    the weight, and the add into .grad is only QUEUED -- it runs from an engine callback at the end of the backward pass."""
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.param = w
        return x @ w

    @staticmethod
    def backward(ctx, g):
        from bmc_hip import ops
        x, w = ctx.saved_tensors
        gw = x.t() @ g
        p = ctx.param

        def flush():
            (dst,), acc = ops.sink_group([p])
            dst.add_(gw) if acc else dst.copy_(gw)
        torch.autograd.Variable._execution_engine.queue_callback(flush)
        return g @ w.t(), None


def test_reducer_hook_with_an_undefined_gradient_and_a_queued_sink_add():
    """`bench.py --gpus 2` at 31x56 (tests/test_gpu_r6.py) found it: autograd runs a parameter's post-accumulate hook even when every
    use handed it None -- with p.grad still None when the kernels' adds are queued (round 5's merged weight gradients) rather than
    launched.  The hook leaves such a parameter to finish(), which stages the bucket from .grad after the pass's flush."""
    sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
    from bmc_hip import ops
    from bmc_hip.parallel import GradAllReducer
    ops.set_accumulate_param_grads(True)
    g = torch.Generator().manual_seed(5)
    w = torch.nn.Parameter(torch.randn(8, 8, generator=g) * 0.3)
    b = torch.nn.Parameter(torch.randn(8, generator=g) * 0.1)
    x = torch.randn(4, 8, generator=g)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w, self.b = w, b

        def forward(self, x, deferred=True):
            mm = DeferredSinkMatmul.apply if deferred else torch.matmul
            return torch.tanh(mm(torch.tanh(mm(x, self.w) + self.b), self.w))

    net = Net()
    net(x, deferred=False).sum().backward()
    ref = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    red = GradAllReducer(net, opt, bucket_mb=1e-4)
    fired = []
    h = red._on_grad
    red._on_grad = lambda p: (fired.append(p.grad is None), h(p))[1]
    for p, hd in zip(red.params, red._handles):       # (re-register so that the wrapped hook is the one autograd calls)
        p._bmc_sink_hooks.discard(hd.id)
        hd.remove()
    red._handles = []
    for p in red.params:
        hd = p.register_post_accumulate_grad_hook(red._on_grad)
        red._handles.append(hd)
        p._bmc_sink_hooks.add(hd.id)
    net(x).sum().backward()
    assert True in fired                              # the hook did fire with p.grad still None ...
    opt.step()                                        # ... finish() staged the gradient that the end-of-pass flush produced
    for p, r in zip(net.parameters(), ref):
        assert torch.allclose(p.grad, r, rtol=1e-6, atol=1e-7)
    red.detach()
