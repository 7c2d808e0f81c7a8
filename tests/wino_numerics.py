#!/usr/bin/env python3
"""CPU numerics experiment: is Winograd F(4x4,3x3) in fp32 inside the parity budget of this network?

Emulates, with torch-CPU float32 tensors, what a kernel would compute: the input transform V = B^T d B, the transformed
weights U = G g G^T (made in float64 and rounded once, as a packing kernel can), the per-position contraction over input
channels in float32, the output transform A^T m A -- for the forward, the data gradient (same operator on the mirrored,
transposed weights) and the weight gradient (dU = sum_tiles (A dY A^T)(B^T d B), dW = G^T dU G).  Every dense 3x3
convolution of the oracle's network is replaced by that emulation and compared with the float64 oracle, beside the
direct float32 convolution and F(2x2,3x3) (what round 3 ships).

    python tests/wino_numerics.py conv                  # one 128->128 convolution, fwd / dgrad / wgrad
    python tests/wino_numerics.py c2    [--gain 2.0]     # test_c2_full_size_window_forward_backward_vs_oracle's problem
    python tests/wino_numerics.py rec   [--windows 8]    # 45x80 recurrence, 8 windows, forward + gradients
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (lives under tests/: it uses the oracle as its checker)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
from oracle import bmc_oracle as O  # noqa: E402  (test infrastructure: this file lives under tests/, the only place beside smoke() and the bench cpu_baseline leg that may touch oracle/)


def cook_toom(points, m, r=3):
    """Winograd / Cook-Toom matrices for F(m, r) on the given finite points (+ the point at infinity), in float64.
    Returns AT [m, n], G [n, r], BT [n, n] with n = m + r - 1, such that y = AT ((G g) * (BT d))."""
    n = m + r - 1
    pts = np.asarray(points, dtype=np.float64)
    assert len(pts) == n - 1
    # evaluation matrices (Vandermonde), last row = infinity
    def vander(k):
        V = np.zeros((n, k))
        for i, p in enumerate(pts):
            V[i] = p ** np.arange(k)
        V[n - 1, k - 1] = 1.0
        return V
    AT = vander(m).T                       # [m, n]
    G = vander(r)                          # [n, r]
    # B^T from the exactness condition: for all g, d: AT((G g)*(BT d)) == correlation(d, g).  Solve column by column.
    # y_i = sum_j d_{i+j} g_j  ->  for each (i, j) : sum_k AT[i,k] G[k,j] BT[k,:] = e_{i+j}
    M = np.zeros((m * r, n))
    rhs = np.zeros((m * r, n))
    for i in range(m):
        for j in range(r):
            M[i * r + j] = AT[i] * G[:, j]
            rhs[i * r + j, i + j] = 1.0
    BT = np.linalg.lstsq(M, rhs, rcond=None)[0]
    assert np.abs(M @ BT - rhs).max() < 1e-9
    return AT, G, BT


def lavin(m):
    """The standard matrices (Lavin & Gray 2015), scaled as published."""
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
    elif m == 4:
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                       [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                      [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
    else:
        raise ValueError(m)
    return AT, G, BT


class Scheme:
    def __init__(self, name, m, mats):
        self.name, self.m = name, m
        AT, G, BT = mats
        self.n = m + 2
        self.AT64, self.G64, self.BT64 = AT, G, BT
        self.AT, self.G, self.BT = (torch.tensor(a, dtype=torch.float32) for a in (AT, G, BT))
        # sanity: exactness in float64 on one random tile
        rng = np.random.default_rng(0)
        d, g = rng.standard_normal((self.n, self.n)), rng.standard_normal((3, 3))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        ref = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(m)] for i in range(m)])
        assert np.abs(y - ref).max() < 1e-9, (name, np.abs(y - ref).max())


def tiles_of(x, m):
    """x [B,C,H,W] -> overlapping (m+2)x(m+2) input tiles of the zero-padded image: [B,C,th,tw,n,n]."""
    B, C, H, W = x.shape
    th, tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, tw * m - W + 1, 1, th * m - H + 1))
    return xp.unfold(2, m + 2, m).unfold(3, m + 2, m), th, tw


def out_tiles_of(g, m):
    B, C, H, W = g.shape
    th, tw = -(-H // m), -(-W // m)
    gp = F.pad(g, (0, tw * m - W, 0, th * m - H))
    return gp.unfold(2, m, m).unfold(3, m, m), th, tw


def wino_fwd(x, w, s):
    """float32 Winograd correlation (no bias)."""
    B, C, H, W = x.shape
    K = w.shape[0]
    t, th, tw = tiles_of(x, s.m)                                            # [B,C,th,tw,n,n]
    V = torch.einsum("ai,bcyxij,dj->adbyxc", s.BT, t, s.BT)                 # [n,n,B,th,tw,C]
    U = torch.tensor(np.einsum("ai,kcij,bj->abkc", s.G64, w.double().numpy(), s.G64), dtype=torch.float32)
    n = s.n
    Mt = torch.bmm(V.reshape(n * n, -1, C), U.reshape(n * n, K, C).transpose(1, 2))       # [nn, B*th*tw, K]
    Mt = Mt.reshape(n, n, B, th, tw, K)
    Y = torch.einsum("ia,abnyxk,jb->nkyixj", s.AT, Mt, s.AT)                # [B,K,th,m,tw,m]
    return Y.reshape(B, K, th * s.m, tw * s.m)[:, :, :H, :W]


def wino_wgrad(x, g, s):
    """float32 Winograd weight gradient."""
    B, C, H, W = x.shape
    K = g.shape[1]
    t, th, tw = tiles_of(x, s.m)
    V = torch.einsum("ai,bcyxij,dj->adbyxc", s.BT, t, s.BT)                 # [n,n,B,th,tw,C]
    gt, _, _ = out_tiles_of(g, s.m)                                         # [B,K,th,tw,m,m]
    dM = torch.einsum("ia,bkyxij,jd->adbyxk", s.AT, gt, s.AT)               # A dY A^T : [n,n,B,th,tw,K]
    n = s.n
    dU = torch.bmm(dM.reshape(n * n, -1, K).transpose(1, 2), V.reshape(n * n, -1, C)).reshape(n, n, K, C)
    return torch.einsum("ai,abkc,bj->kcij", s.G, dU, s.G)


class WinoConv(torch.autograd.Function):
    scheme = None          # data gradient, weight gradient, and the forward of every convolution not named in `safe`
    safe_scheme = None     # forward of the convolutions in `safe` (None: the direct convolution)
    safe = ()              # name prefixes (oracle parameter names) -- see --policy
    current = ""           # name of the convolution being evaluated (set by patched_conv)

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        if any(WinoConv.current.startswith(q) for q in WinoConv.safe):
            y = wino_fwd(x, w, WinoConv.safe_scheme) if WinoConv.safe_scheme is not None else F.conv2d(x, w, None, padding=1)
        else:
            y = wino_fwd(x, w, WinoConv.scheme)
        return y + b.view(1, -1, 1, 1) if b is not None else y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        s = WinoConv.scheme
        g = g.contiguous()
        dx = wino_fwd(g, w.flip(2, 3).transpose(0, 1).contiguous(), s) if ctx.needs_input_grad[0] else None
        dw = wino_wgrad(x, g, s) if ctx.needs_input_grad[1] else None
        db = g.sum(dim=(0, 2, 3)) if ctx.needs_input_grad[2] else None
        return dx, dw, db


_orig_conv2d = O.conv2d
_orig_conv = O.conv


def patched_conv(p, name, x):
    WinoConv.current = name
    try:
        return _orig_conv(p, name, x)
    finally:
        WinoConv.current = ""


# Exact zeros (found by this experiment): with zero biases and a zero recurrent state (window 0 at
# initialisation) the reference's direct convolution gives EXACTLY 0 wherever a pixel's receptive field holds no event, and
# relu'(0) = 0 gates the gradient there.  A Winograd tile computes such a pixel from a patch that also holds its neighbours'
# events, through rounded transformed weights: +-1e-8 instead of 0, i.e. a coin flip of the ReLU mask, which the bias
# gradients of the first layers see (F(4x4): 6x6 patches around 3x3 fields; F(2x2) is nearly exact on integer counts).
POLICIES = {
    "all": (),
    "first": ("neuro.conv_fpst", "neuro.conv_fnst", "neuro.conv_fps", "neuro.conv_fns", "neuro.conv_fs"),
    "first+blk0": ("neuro.conv_fpst", "neuro.conv_fnst", "neuro.conv_fps", "neuro.conv_fns", "neuro.conv_fs",
                   "neuro.para_reschunk.0.conv1.", "neuro.para_reschunk.0.conv2.", "neuro.para_reschunk.0.conv1_st.",
                   "neuro.para_reschunk.0.conv2_st."),
}


def patched_conv2d(x, w, b):
    """The kernels' rule (ops.wino_ok): dense 3x3 with 128-granular output channels and >=128-channel input go through the
    transform; narrow inputs / outputs (conv_o, 16-channel sources) stay direct.  Here: every 3x3 with Cin,Cout >= 16."""
    if WinoConv.scheme is not None and w.shape[-1] == 3 and x.dtype == torch.float32 and min(w.shape[0], w.shape[1]) >= 16:
        return WinoConv.apply(x, w, b)
    return _orig_conv2d(x, w, b)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


SCHEMES = None


def schemes():
    global SCHEMES
    if SCHEMES is None:
        SCHEMES = [
            None,
            Scheme("F(2x2,3x3)", 2, lavin(2)),
            Scheme("F(4x4,3x3) points 0,+-1,+-2 (Lavin)", 4, lavin(4)),
            Scheme("F(4x4,3x3) points 0,+-1,+-1/2", 4, cook_toom([0, 1, -1, .5, -.5], 4)),
            Scheme("F(4x4,3x3) points 0,+-1,1/2,-2", 4, cook_toom([0, 1, -1, .5, -2], 4)),
            Scheme("F(3x3,3x3) points 0,+-1,2", 3, cook_toom([0, 1, -1, 2], 3)),
            Scheme("F(3x3,3x3) points 0,+-1,1/2", 3, cook_toom([0, 1, -1, .5], 3)),
        ]
    return SCHEMES


def run_conv(args):
    torch.manual_seed(0)
    B, C, K, H, W = 2, 128, 128, 60, 80
    x = torch.relu(torch.randn(B, C, H, W))
    w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    g = torch.randn(B, K, H, W)
    y64 = F.conv2d(x.double(), w.double(), None, padding=1)
    dx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), g.double(), padding=1)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, g.double(), padding=1)
    print("one 3x3 %d->%d convolution, %dx%dx%d, rel-L2 vs float64 (fwd / dgrad / wgrad)" % (C, K, B, H, W))
    for s in schemes():
        if s is None:
            y = F.conv2d(x, w, None, padding=1)
            dx = torch.nn.grad.conv2d_input(x.shape, w, g, padding=1)
            dw = torch.nn.grad.conv2d_weight(x, w.shape, g, padding=1)
            name = "direct fp32 (ATen)"
        else:
            y = wino_fwd(x, w, s)
            dx = wino_fwd(g, w.flip(2, 3).transpose(0, 1).contiguous(), s)
            dw = wino_wgrad(x, g, s)
            name = s.name
        print("  %-44s %.2e  %.2e  %.2e" % (name, rel(y, y64), rel(dx, dx64), rel(dw, dw64)))


def network_problem(H, W, B, nwin, gain, seed=71):
    from models.BMCNet import BMCNet
    scale, n_c, n_b = 4, 128, 5
    torch.manual_seed(seed)
    m = BMCNet(scale, n_c, n_b)
    with torch.no_grad():                   # tests/test_gpu_r2.py::scaled_init
        for q in m.parameters():
            q.mul_(gain)
    params = m.state_dict()
    g = torch.Generator().manual_seed(seed + 1)
    frames = torch.poisson(torch.full((B, nwin + 1, 2, H, W), 0.284), generator=g)
    gts = torch.poisson(torch.full((B, nwin + 1, 2, scale * H, scale * W), 0.284), generator=g)
    xs = [frames[:, i:i + 2].transpose(1, 2) for i in range(nwin)]
    gt = [gts[:, i + 1] for i in range(nwin)]
    return params, xs, gt, n_c, scale


def run_net(args, H, W, B, nwin):
    params, xs, gt, n_c, scale = network_problem(H, W, B, nwin, args.gain)
    torch.set_num_threads(os.cpu_count() or 1)

    def run(dtype, scheme):
        WinoConv.scheme = scheme
        WinoConv.safe = POLICIES[args.policy] if scheme is not None and scheme.m > 2 else ()
        WinoConv.safe_scheme = schemes()[1] if args.safe == "f2" else None
        O.conv2d = patched_conv2d
        O.conv = patched_conv
        try:
            p, seen = {}, {}               # alias keys share one leaf (tests/test_gpu_r2.py::oracle_params)
            for k, v in params.items():
                p[k] = seen.setdefault(v.data_ptr(), v.detach().to(dtype).clone().requires_grad_())
            t0 = time.time()
            loss, preds, _ = O.bptt_loss(p, [x.to(dtype) for x in xs], [g_.to(dtype) for g_ in gt], n_c, scale)
            loss.backward()
            grads, done = {}, set()
            for k, v in p.items():
                if v.grad is not None and id(v) not in done:
                    done.add(id(v))
                    grads[k] = v.grad
            return loss.item(), [q.detach() for q in preds], grads, time.time() - t0
        finally:
            O.conv2d = _orig_conv2d
            O.conv = _orig_conv
            WinoConv.scheme = None

    l64, p64, g64, t = run(torch.float64, None)
    print("BMCNet(4,128,5) %dx%d B=%d, %d windows, weight gain %.2f, policy %s (safe forward: %s); float64 oracle: loss %.6f (%.0f s)" % (
        H, W, B, nwin, args.gain, args.policy, args.safe, l64, t))
    sys.stdout.flush()
    print("  %-44s %-10s %-s" % ("3x3 convolutions in float32 as", "loss err", "SR rel-L2 per window | whole gradient | worst 3 parameter gradients"))
    want = args.schemes.split(",") if args.schemes else None
    for i, s in enumerate(schemes()):
        if want is not None and str(i) not in want:
            continue
        l, p, g, t = run(torch.float32, s)
        sr = ["%.1e" % rel(a, b) for a, b in zip(p, p64)]
        errs = sorted(((rel(g[k], g64[k]), k) for k in g64), reverse=True)
        whole = rel(torch.cat([g[k].flatten() for k in g64]), torch.cat([g64[k].flatten() for k in g64]))
        print("  %-44s %.1e    %s | %.1e | %s  (%.0f s)" % ("direct fp32 (ATen)" if s is None else s.name, abs(l - l64) / abs(l64),
              " ".join(sr), whole, ", ".join("%s %.1e" % (k.replace(".weight", ".w").replace(".bias", ".b"), e) for e, k in errs[:3]), t))
        sys.stdout.flush()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["conv", "c2", "rec"])
    ap.add_argument("--gain", type=float, default=2.0)
    ap.add_argument("--windows", type=int, default=8)
    ap.add_argument("--policy", default="all", choices=sorted(POLICIES), help="which forward convolutions stay off F(m>2)")
    ap.add_argument("--safe", default="f2", choices=["f2", "direct"], help="what those run instead")
    ap.add_argument("--schemes", default=None, help="comma list of scheme indices (0 = direct)")
    a = ap.parse_args()
    if a.what == "conv":
        run_conv(a)
    elif a.what == "c2":
        run_net(a, 180, 240, 1, 2)
    else:
        run_net(a, 45, 80, 2, a.windows)
