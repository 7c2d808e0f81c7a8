"""CPU checks of the oracle's bf16 operand-rounding mode (oracle.operand_rounding / bmcnet_forward(operand_round="bf16")):
the contract of the kernels' BMC_MATH_BF16 arithmetic (BASELINE configs[3]) -- every contraction rounds both operands to
bf16 and accumulates exactly, in forward, data gradient and weight gradient alike.  Validated at layer level against hand
rounding written out with plain numpy / float64 einsums (no autocast, no custom Function), and against its own invariants."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import bmc_oracle as O  # noqa: E402


def np_round_bf16(a):
    """RNE to bf16 by integer arithmetic on the float32 bits (finite inputs), independent of torch's cast."""
    u = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).astype(np.float64)


def test_round_bf16_is_rne_of_the_float32_value():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(20000, generator=g, dtype=torch.float64) * torch.logspace(-6, 6, 20000, dtype=torch.float64)
    ties = torch.tensor([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, -(1.0 + 2.0 ** -8), 2.0 ** -130, 0.0, 3.0], dtype=torch.float64)
    x = torch.cat([x, ties])
    got = O.round_bf16(x).numpy()
    assert np.array_equal(got, np_round_bf16(x.numpy()))
    assert got[-6] == 1.0 and got[-5] == 1.0 + 2.0 ** -6          # ties go to the even neighbour
    assert O.round_bf16(x.float()).dtype == torch.float32 and O.round_bf16(x).dtype == torch.float64


@pytest.mark.parametrize("k,cin,cout", [(3, 5, 7), (1, 6, 4)])
def test_rounded_conv_matches_hand_rounding(k, cin, cout):
    g = torch.Generator().manual_seed(k)
    B, H, W = 2, 6, 5
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64) * 0.3).requires_grad_()
    b = torch.randn(cout, generator=g, dtype=torch.float64, requires_grad=True)
    go = torch.randn(B, cout, H, W, generator=g, dtype=torch.float64)
    with O.operand_rounding("bf16"):
        y = O.conv2d(x, w, b)
    y.backward(go)
    # hand rounding, written out: zero-padded input, explicit sums
    xr, wr, gr = np_round_bf16(x.detach().numpy()), np_round_bf16(w.detach().numpy()), np_round_bf16(go.numpy())
    p = k // 2
    xp = np.pad(xr, ((0, 0), (0, 0), (p, p), (p, p)))
    yr = np.zeros((B, cout, H, W))
    dw = np.zeros_like(wr)
    dxp = np.zeros_like(xp)
    for dy in range(k):
        for dx in range(k):
            win = xp[:, :, dy:dy + H, dx:dx + W]
            yr += np.einsum("bchw,oc->bohw", win, wr[:, :, dy, dx])
            dw[:, :, dy, dx] = np.einsum("bohw,bchw->oc", gr, win)
            dxp[:, :, dy:dy + H, dx:dx + W] += np.einsum("bohw,oc->bchw", gr, wr[:, :, dy, dx])
    yr += b.detach().numpy()[None, :, None, None]
    dxr = dxp[:, :, p:p + H, p:p + W] if p else dxp
    assert np.allclose(y.detach().numpy(), yr, rtol=0, atol=1e-12)
    assert np.allclose(w.grad.numpy(), dw, rtol=0, atol=1e-12)
    assert np.allclose(x.grad.numpy(), dxr, rtol=0, atol=1e-12)
    assert np.allclose(b.grad.numpy(), go.numpy().sum((0, 2, 3)), rtol=0, atol=1e-12)       # the bias gradient sums the UNROUNDED g
    # and it differs from the unrounded convolution by bf16-level amounts, not by nothing
    y0 = F.conv2d(x.detach(), w.detach(), b.detach(), padding=p)
    rel = float((y.detach() - y0).norm() / y0.norm())
    assert 1e-4 < rel < 2e-2, rel


def test_rounded_bmm_matches_hand_rounding():
    g = torch.Generator().manual_seed(5)
    a = torch.randn(3, 4, 9, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(3, 9, 5, generator=g, dtype=torch.float64, requires_grad=True)
    go = torch.randn(3, 4, 5, generator=g, dtype=torch.float64)
    with O.operand_rounding("bf16"):
        c = O.bmm(a, b)
    c.backward(go)
    ar, br, gr = np_round_bf16(a.detach().numpy()), np_round_bf16(b.detach().numpy()), np_round_bf16(go.numpy())
    assert np.allclose(c.detach().numpy(), np.einsum("bik,bkj->bij", ar, br), rtol=0, atol=1e-13)
    assert np.allclose(a.grad.numpy(), np.einsum("bij,bkj->bik", gr, br), rtol=0, atol=1e-13)
    assert np.allclose(b.grad.numpy(), np.einsum("bik,bij->bkj", ar, gr), rtol=0, atol=1e-13)


def _tiny_params(n_c=16, n_b=1, scale=2, seed=3, dtype=torch.float64):
    """A BMCNet-shaped parameter dict without importing the product package (CPU tests must not need the .so)."""
    g = torch.Generator().manual_seed(seed)
    s2 = scale * scale
    r = 3
    p = {}

    def cv(name, cin, cout, k):
        p[name + ".weight"] = (torch.randn(cout, cin, k, k, generator=g, dtype=dtype) * (0.5 / (cin * k * k) ** 0.5)).requires_grad_()
        p[name + ".bias"] = (torch.randn(cout, generator=g, dtype=dtype) * 0.05).requires_grad_()

    def alias(dst, src):
        for k in [k for k in p if k.startswith(src + ".")]:
            p[dst + k[len(src):]] = p[k]

    def res(name):
        cv(name + ".conv1", n_c, n_c, 3)
        cv(name + ".conv2", n_c, n_c, 3)

    def bie(name):
        res(name + ".conv1"); alias(name + ".conv2", name + ".conv1")
        cv(name + ".convf1", 2 * n_c, n_c, 1); alias(name + ".convf2", name + ".convf1")
        p[name + ".norm_s.weight"] = (1 + 0.1 * torch.randn(n_c, generator=g, dtype=dtype)).requires_grad_()
        p[name + ".norm_s.bias"] = (0.1 * torch.randn(n_c, generator=g, dtype=dtype)).requires_grad_()
        cv(name + ".clustering", n_c, n_c, 1)
        cv(name + ".unclustering", 2 * n_c, n_c, 1)
        cv(name + ".v1", n_c, n_c, 1)
        cv(name + ".v2", n_c, n_c, 1)

    cv("neuro.conv_fpst", s2 + n_c + 2 * r, n_c, 3); alias("neuro.conv_fnst", "neuro.conv_fpst")
    cv("neuro.conv_fps", r + n_c, n_c, 3); alias("neuro.conv_fns", "neuro.conv_fps")
    cv("neuro.conv_fs", 2 * s2 + 3 * n_c, n_c, 3)
    blk = "neuro.para_reschunk.0"
    res(blk + ".conv1"); alias(blk + ".conv2", blk + ".conv1")
    res(blk + ".conv1_st"); alias(blk + ".conv2_st", blk + ".conv1_st")
    bie(blk + ".lBIE"); bie(blk + ".gBIE")
    for i in range(1, n_b):
        alias("neuro.para_reschunk.%d" % i, blk)
    for nm in ("hs", "hp", "hn"):
        cv("neuro.conv_" + nm, n_c, n_c, 3)
    cv("neuro.conv_o", 2 * n_c, 2 * s2, 3)
    return p


def _run(p, operand_round, H=6, W=7, B=2, scale=2, n_c=16, nwin=2, seed=8, quantise=None):
    g = torch.Generator().manual_seed(seed)
    dt = next(iter(p.values())).dtype
    frames = torch.poisson(torch.full((B, nwin + 1, 2, H, W), 0.6), generator=g).to(dt)
    gts = torch.poisson(torch.full((B, nwin + 1, 2, scale * H, scale * W), 0.6), generator=g).to(dt)
    xs = [frames[:, i:i + 2].transpose(1, 2) for i in range(nwin)]
    for v in set(p.values()):
        v.grad = None
    loss, preds, states = O.bptt_loss(p, xs, [gts[:, i + 1] for i in range(nwin)], n_c, scale, operand_round=operand_round)
    loss.backward()
    return loss.item(), [q.detach() for q in preds], {k: v.grad.clone() for k, v in p.items() if v.grad is not None}


def test_full_model_rounding_mode_is_a_bf16_sized_perturbation_and_leaves_the_default_alone():
    p = _tiny_params()
    l0, pr0, g0 = _run(p, None)
    l0b, pr0b, g0b = _run(p, None)
    assert l0 == l0b and all(torch.equal(a, b) for a, b in zip(pr0, pr0b))        # the switch leaves no state behind
    l1, pr1, g1 = _run(p, "bf16")
    assert O._OPERAND_ROUND is None
    rel = [float((a - b).norm() / b.norm()) for a, b in zip(pr1, pr0)]
    assert all(1e-5 < r < 3e-2 for r in rel), rel                                  # bf16: 2^-9 per operand, a few layers deep
    gk = [k for k in g0 if g0[k].norm() > 0]
    grel = max(float((g1[k] - g0[k]).norm() / g0[k].norm()) for k in gk)
    assert 1e-4 < grel < 0.2, grel
    assert set(g1) == set(g0)                                                      # every parameter still receives a gradient


def test_rounding_mode_is_the_identity_on_bf16_representable_contractions():
    """One 1x1 convolution whose operands and upstream gradient are already bf16 values: rounded == unrounded, exactly."""
    g = torch.Generator().manual_seed(1)
    q = lambda t: O.round_bf16(t)
    x = q(torch.randn(2, 8, 4, 4, generator=g, dtype=torch.float64)).requires_grad_()
    w = q(torch.randn(8, 8, 1, 1, generator=g, dtype=torch.float64)).requires_grad_()
    go = q(torch.randn(2, 8, 4, 4, generator=g, dtype=torch.float64))
    with O.operand_rounding("bf16"):
        y = O.conv2d(x, w, None)
    (dx, dw) = torch.autograd.grad(y, (x, w), go)
    y0 = F.conv2d(x, w)
    (dx0, dw0) = torch.autograd.grad(y0, (x, w), go)
    assert torch.equal(y, y0) and torch.equal(dx, dx0) and torch.equal(dw, dw0)


def test_unknown_mode_raises():
    with pytest.raises(ValueError):
        O.operand_rounding("fp8")
