"""N > 1 path on CPU: two gloo ranks, each with half of the sequence batch; the bucketed gradient all-reduce
(bmc_hip.parallel.GradAllReducer) must reproduce the single-process full-batch gradient and keep the
`loss.backward(); optimizer.step()` loop shape.  Gradients come from the CPU oracle (test infrastructure)."""
import os
import socket
import sys

import numpy as np
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
