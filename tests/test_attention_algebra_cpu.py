"""The algebra behind the BIE attention without its value tensor (bmc_hip/bie.py), checked in float64 on the
CPU against autograd of the explicit form (reference models/submodules.py:63-73: v = conv1x1(x); att = scale * bmm(center, v^T);
out = bmm(softmax(att), v)) -- independent of any kernel: every formula the launch list uses is written out here once."""
import torch


def test_value_free_attention_matches_the_explicit_form_and_its_autograd():
    torch.manual_seed(0)
    B, P, C = 3, 37, 8                                   # samples, pixels, channels
    d = torch.float64
    x = torch.randn(B, P, C, dtype=d, requires_grad=True)             # value input
    c = torch.randn(B, P, C, dtype=d, requires_grad=True)             # centres
    W = torch.randn(C, C, dtype=d, requires_grad=True)                # value weights [j, k]
    b = torch.randn(C, dtype=d, requires_grad=True)
    scale = 0.37
    g_o = torch.randn(B, P, C, dtype=d)
    # ---- explicit form
    v = x @ W.t() + b                                                 # [B, P, j]
    att = scale * c.transpose(1, 2) @ v                               # [B, i, j]
    Pm = torch.softmax(att, -1)
    out = v @ Pm.transpose(1, 2)                                      # out[px, i] = sum_j P[i, j] v[px, j]
    gx, gc, gW, gb = torch.autograd.grad(out, (x, c, W, b), g_o)
    # ---- without v: forward
    xd, cd, Wd, bd = x.detach(), c.detach(), W.detach(), b.detach()
    G0 = cd.transpose(1, 2) @ xd                                      # center^T x            [B, i, k]
    s = cd.sum(1)                                                     # column sums of center [B, i]
    att2 = scale * (G0 @ Wd.t() + s[:, :, None] * bd[None, None, :])
    P2 = torch.softmax(att2, -1)
    out2 = xd @ (P2 @ Wd).transpose(1, 2) + (P2 @ bd)[:, None, :]     # (P W) x + P b
    assert torch.allclose(att2, att.detach(), rtol=1e-12, atol=1e-12)
    assert torch.allclose(out2, out.detach(), rtol=1e-12, atol=1e-12)
    # ---- without v: backward
    dM = g_o.transpose(1, 2) @ xd                                     # g_o^T x               [B, i, k]
    t = g_o.sum(1)                                                    # column sums of g_o    [B, i]
    dP = dM @ Wd.t() + t[:, :, None] * bd[None, None, :]
    da = scale * P2 * (dP - (dP * P2).sum(-1, keepdim=True))          # softmax backward, times scale
    dG0 = da @ Wd                                                     # [B, i, k]
    ds = da @ bd                                                      # [B, i]
    dx = g_o @ (P2 @ Wd) + cd @ dG0                                   # (P W)^T g_o + dG0^T center   (as row-vector products)
    dc = xd @ dG0.transpose(1, 2) + ds[:, None, :]                    # dG0 x + da b
    dW = (P2.transpose(1, 2) @ dM + da.transpose(1, 2) @ G0).sum(0)   # P^T dM + da^T G0
    db = (P2.transpose(1, 2) @ t[:, :, None] + da.transpose(1, 2) @ s[:, :, None]).sum(0)[:, 0]
    for name, a_, r_ in (("dx", dx, gx), ("dcenter", dc, gc), ("dW_v", dW, gW), ("db_v", db, gb)):
        assert torch.allclose(a_, r_, rtol=1e-10, atol=1e-10), name
