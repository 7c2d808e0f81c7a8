"""GPU parity tests added in round 4 (-m gpu), all through the C ABI of libbmc_hip.so: the Winograd F(4x4, 3x3) convolution
kernel (csrc/wino4.hip) against float64, against the direct kernel with every epilogue, over ragged geometries; the routing
rule for exact-zero inputs (ops.exact_zero_inputs)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from test_gpu_r2 import _gpu, oracle_params, rel_l2, scaled_init  # noqa: E402,F401


@pytest.fixture
def force_wino4():
    """Send every eligible 3x3 launch through the F(4x4) kernel, whatever its size (default: only launches that fill the chip)."""
    from bmc_hip import ops
    old = ops.WINO_MIN_TILES, ops.WINO4_MIN_TILES, ops.WINO4
    ops.WINO_MIN_TILES, ops.WINO4_MIN_TILES, ops.WINO4 = 0, 0, True
    yield ops
    ops.WINO_MIN_TILES, ops.WINO4_MIN_TILES, ops.WINO4 = old


@pytest.mark.parametrize("B,H,W,cins,cout,relu,res", [
    (1, 4, 64, [128], 128, True, False),              # exactly one workgroup tile: one row of 16 tiles
    (2, 9, 19, [128], 128, False, False),             # 5 x 3 tiles: the strip wraps twice, ragged in both directions
    (2, 19, 37, [128], 128, True, True),              # 10 x 5 tiles: 4 workgroup tiles per image, the last one partial
    (2, 13, 21, [16, 128, 16], 128, True, False),     # multi-source with narrow sources
    (1, 11, 18, [16, 128, 16, 16, 32], 128, True, False),
    (2, 5, 40, [256], 256, False, True),              # two channel tiles, 16 chunks
    (3, 17, 33, [32], 128, False, False),
    (1, 45, 80, [128], 128, True, True),              # the reference's own LR frame size
])
def test_winograd4_conv_fwd_bwd_vs_float64(force_wino4, B, H, W, cins, cout, relu, res):
    """Forward and data gradient through wino4.hip (weight gradient: the F(2x2) / pixel-reduction kernels), against float64.
    One F(4x4) convolution measures 1.9e-6 (forward) / 2.3e-6 (data gradient) in the CPU emulation (profiles/r04_wino_numerics.txt)."""
    dev = _gpu()
    ops = force_wino4
    from bmc_hip.ops import ConvSpec, View
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    r = torch.randn(B, cout, H, W, generator=g) if res else None
    go = torch.randn(B, cout, H, W, generator=g)
    xs_c = [x.double().requires_grad_() for x in xs]
    w_c, b_c = w.double().requires_grad_(), b.double().requires_grad_()
    r_c = r.double().requires_grad_() if res else None
    y = F.conv2d(torch.cat(xs_c, 1), w_c, b_c, padding=1)
    if res:
        y = y + r_c
    if relu:
        y = torch.relu(y)
    y.backward(go.double())
    xs_g = [nhwc(x).to(dev).requires_grad_() for x in xs]
    w_g, b_g = w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    # (a forward launch takes F(4x4) only with a bias that is dense -- the exact-zero rule above ops.wino_ok -- as this one is)
    assert ops.wino_ok(B, H, W, cout, 9, fwd=True, rule=b_g) == 4 and ops.wino_ok(B, H, W, cout, 9, fwd=True) == 2
    r_g = nhwc(r).to(dev).requires_grad_() if res else None
    ops.PROFILE = []
    yg = ops.conv([View(x) for x in xs_g], w_g, b_g, ConvSpec.dense(*cins), relu=relu, residual=View(r_g) if res else None)
    yg.backward(nhwc(go).to(dev))
    names = [p[0] for p in ops.PROFILE]
    ops.PROFILE = None
    assert names.count("wino4_conv<9,128>") >= 1 + sum(1 for c in cins if c % 128 == 0), names      # the kernel under test really ran
    assert rel_l2(yg.permute(0, 3, 1, 2), y) < 1e-5
    for xg, xc in zip(xs_g, xs_c):
        assert rel_l2(xg.grad.permute(0, 3, 1, 2), xc.grad) < 1e-5
    assert rel_l2(w_g.grad, w_c.grad) < 2e-5
    assert rel_l2(b_g.grad, b_c.grad) < 2e-5
    if res:
        assert rel_l2(r_g.grad.permute(0, 3, 1, 2), r_c.grad) < 2e-5


def test_winograd4_matches_direct_kernel_with_every_epilogue(force_wino4):
    """The same launches through the direct fp32 kernel and through the F(4x4) kernel: per-group bias, residual with a batch
    rotation, ReLU, ReLU mask, accumulate, per-group weights, output written into a channel window of a wider tensor."""
    dev = _gpu()
    from bmc_hip.ops import _packed_weight, _src, conv_raw, coutpad, ConvSpec
    torch.manual_seed(12)
    B, H, W, Cn = 4, 21, 35, 128
    spec = ConvSpec.dense(Cn)
    x = torch.randn(B, H, W, Cn, device=dev)
    res = torch.randn(B, H, W, Cn, device=dev)
    msk = torch.randn(B, H, W, Cn, device=dev)
    w = torch.randn(2, Cn, Cn, 9, device=dev) * 0.03
    bias = torch.randn(2, Cn, device=dev)
    cp = coutpad(Cn)

    def run(wino, G, relu, use_res, use_mask, accumulate, wide):
        w4 = w[:G].contiguous()
        wp = _packed_weight(w4, spec, None, wino=wino)
        Co = 2 * Cn if wide else Cn
        out = torch.full((B, H, W, Co), 0.25, device=dev)
        conv_raw([_src(x, 0, Cn, 0, None, 0, B)], wp, spec.kpad * 9 * cp, bias[:G].contiguous(), Cn,
                 out.data_ptr() + (4 * Cn if wide else 0), H * W * Co, Co, B, H, W, Cn, 9, relu=relu,
                 residual=_src(res, 0, Cn, 2, B, 0, B) if use_res else None, bpg=B // G, accumulate=accumulate,
                 mask=_src(msk, 0, Cn, 0, None, 0, B) if use_mask else None, wino=wino)
        return out

    for G, relu, use_res, use_mask, accumulate, wide in [(1, False, False, False, False, False), (1, True, True, False, False, False),
                                                         (2, False, False, True, False, False), (2, True, True, False, True, True),
                                                         (1, False, True, True, True, False)]:
        a, b = run(4, G, relu, use_res, use_mask, accumulate, wide), run(0, G, relu, use_res, use_mask, accumulate, wide)
        assert rel_l2(a, b) < 1e-5, (G, relu, use_res, use_mask, accumulate, wide)
        if wide:
            assert torch.equal(a[..., :Cn], torch.full_like(a[..., :Cn], 0.25))      # the other channel window is untouched


def test_winograd4_geometry_fuzz_vs_direct_kernel(force_wino4):
    """Random image sizes (every tiles-per-row count from 5 up, strips that wrap 0-3 times, partial last workgroup tiles, more
    workgroup tiles than CUs so that the persistent walk and the streams' hand-over between tiles run), against the direct
    kernel; every operand ends at the end of its own allocation."""
    dev = _gpu()
    from bmc_hip.ops import _packed_weight, _src, conv_raw, coutpad, ConvSpec
    g = torch.Generator().manual_seed(404)
    Cn = 128
    spec = ConvSpec.dense(Cn)
    cp = coutpad(Cn)
    w = (torch.randn(1, Cn, Cn, 9, generator=g) * 0.03).to(dev)
    bias = torch.randn(1, Cn, generator=g).to(dev)
    shapes = [(1, 4, 17), (1, 5, 20), (3, 7, 23), (2, 16, 24), (1, 33, 28), (2, 30, 61), (1, 64, 64), (5, 31, 56), (16, 64, 96), (2, 180, 240)]
    for B, H, W in shapes + [(int(torch.randint(1, 4, (1,), generator=g)), int(torch.randint(1, 70, (1,), generator=g)),
                              int(torch.randint(17, 90, (1,), generator=g))) for _ in range(12)]:
        x = torch.randn(B, H, W, Cn, generator=g).to(dev)
        outs = []
        for wino in (4, 0):
            wp = _packed_weight(w, spec, None, wino=wino)
            out = torch.full((B, H, W, Cn), 7.0, device=dev)
            conv_raw([_src(x, 0, Cn, 0, None, 0, B)], wp, spec.kpad * 9 * cp, bias, Cn, out.data_ptr(), H * W * Cn, Cn, B, H, W, Cn, 9,
                     relu=False, bpg=B, wino=wino)
            outs.append(out)
        assert rel_l2(outs[0], outs[1]) < 1e-5, (B, H, W, rel_l2(outs[0], outs[1]))


def test_exact_zero_rule_keeps_reference_relu_gates(force_wino4):
    """Sparse event counts: where a pixel's receptive field holds no event the reference gives EXACTLY its bias.  With a zero
    bias (as `initialize_weights` leaves it) that is exactly 0 and relu'(0) = 0 gates the gradient: the forward launch must
    keep a kernel that is exact there (F(2x2): outputs are combinations of products of their own 3x3 field only).  With a
    dense bias the launch takes F(4x4): its +-1e-8 residue on empty fields cannot move a gate.  Round 5: the rule reads the
    bias (ops.bias_dense), not the caller's `init` flag; the last part forces F(4x4) onto the zero-bias launch to show what the
    rule prevents."""
    dev = _gpu()
    ops = force_wino4
    from bmc_hip.ops import ConvSpec, View
    g = torch.Generator().manual_seed(5)
    B, H, W, Cn = 2, 40, 64, 128
    x = torch.poisson(torch.full((B, H, W, Cn), 0.02), generator=g)
    x = x * (torch.rand(B, H, W, 1, generator=g) < 0.15)                 # most pixels hold no event at all
    w = (torch.randn(Cn, Cn, 3, 3, generator=g) * 0.05).to(dev)
    xg = x.to(dev)
    occupied = F.max_pool2d(x.abs().sum(-1, keepdim=True).permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1) > 0
    empty = (~occupied).expand(B, H, W, Cn).to(dev)
    assert empty.float().mean() > 0.1

    def run(bias):
        ops.PROFILE = []
        try:
            y = ops.conv([View(xg)], w, bias, ConvSpec.dense(Cn), relu=True)
            torch.cuda.synchronize()
            return y, [r[0] for r in ops.PROFILE]
        finally:
            ops.PROFILE = None

    b_zero = torch.zeros(Cn, device=dev)
    b_dense = ((torch.rand(Cn, generator=g) - 0.5) * 0.2).to(dev)
    b_dense[b_dense.abs() < 1e-3] = 0.05
    assert not ops.bias_dense(b_zero) and ops.bias_dense(b_dense) and not ops.bias_dense(None)
    y_zero, k_zero = run(b_zero)
    assert k_zero == ["wino_conv<9,128>"] and torch.count_nonzero(y_zero[empty]) == 0         # exact zeros on F(2x2)
    y_none, k_none = run(None)
    assert k_none == ["wino_conv<9,128>"] and torch.equal(y_none, y_zero)
    y_dense, k_dense = run(b_dense)
    assert k_dense == ["wino4_conv<9,128>"]
    want = torch.relu(b_dense).expand(B, H, W, Cn)
    resid = float((y_dense - want)[empty].abs().max())
    print("F(4x4) residue on empty receptive fields beside a dense bias: %.1e (DENSE_FLOOR %.0e)" % (resid, ops.DENSE_FLOOR))
    assert resid < ops.DENSE_FLOOR / 3 and bool(((y_dense > 0) == (want > 0))[empty].all())   # the same gates
    with ops.exact_zero_inputs():                                        # the context forces the exact kernels whatever the bias
        assert run(b_dense)[1] == ["wino_conv<9,128>"]
    with ops.dense_inputs():                                             # the caller vouches for inputs without empty fields
        assert run(b_zero)[1] == ["wino4_conv<9,128>"]
    b_pos = torch.zeros(Cn, device=dev)
    b_pos[7] = 0.01
    assert ops.bias_positive(b_dense) and ops.bias_positive(b_pos) and not ops.bias_dense(b_pos) and not ops.bias_positive(b_zero)
    # what the rule prevents: F(4x4) on the zero-bias launch
    old = ops.DENSE_FLOOR
    ops.DENSE_FLOOR = 0.0
    ops._DENSE.clear()
    try:
        y_f4, k_f4 = run(b_zero)
    finally:
        ops.DENSE_FLOOR = old
        ops._DENSE.clear()
    assert k_f4 == ["wino4_conv<9,128>"] and rel_l2(y_f4, y_zero) < 1e-5
    print("F(4x4) residue on empty receptive fields: %d of %d outputs non-zero" % (torch.count_nonzero(y_f4[empty]), int(empty.sum())))
    assert torch.count_nonzero(y_f4[empty]) > 0


# ------------------------------------------------------------------ weight gradients on the side stream at every size (fp32 default)
@pytest.mark.parametrize("keep_limit", [0, 1 << 30])
def test_side_stream_weight_gradients_bit_identical_with_winograd_kernels(force_wino4, keep_limit):
    """Round 4: in the fp32 mode the weight-gradient kernels (pixel-reduction GEMM AND the Winograd weight gradient, with their
    reductions) run on a second stream by default (bmc_hip.ops.wgrad_side; from 2^14 pixels per launch -- forced here).  The whole step -- loss, every
    parameter gradient, the parameters after three Adam steps -- must be bit-identical to the one-stream run, repeatedly, both
    when the operands are kept referenced until the join (keep_limit large) and when they are handed to the caching allocator
    with record_stream (keep_limit 0: the route of the large frames, where a use-after-reuse would show up here)."""
    dev = _gpu()
    from bmc_hip import ops
    from models.BMCNet import BMCNet
    from train_step import bptt_step
    ops.set_math("fp32")
    scale, n_c, n_b, B, L, H, W = 4, 128, 1, 2, 4, 72, 96         # (force_wino4: the F(4x4) kernel runs at this size too)
    g = torch.Generator().manual_seed(191)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.5), generator=g).to(dev)
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.5), generator=g).to(dev)

    def run(mode):
        old, old_lim, old_min = ops.WGRAD_SIDE, ops.WGRAD_KEEP_MAX_PIXELS, ops.WGRAD_SIDE_MIN_PIXELS
        ops.WGRAD_SIDE, ops.WGRAD_KEEP_MAX_PIXELS, ops.WGRAD_SIDE_MIN_PIXELS = mode, keep_limit, 0     # (auto: at this test's size too)
        try:
            torch.manual_seed(192)
            m = BMCNet(scale, n_c, n_b).to(dev)
            scaled_init(m, 2.0)
            opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5, amsgrad=True)
            losses = []
            ops.PROFILE, ops.PROFILE_WINO[:] = [], [0, 0]
            for _ in range(3):
                loss, _ = bptt_step(m, opt, inp, gt, n_c, scale)
                losses.append(loss.item())
            torch.cuda.synchronize()
            kinds = {r[0] for r in ops.PROFILE}
            ops.PROFILE = None
            return losses, [p.grad.clone() for p in m.parameters() if p.grad is not None], [p.detach().clone() for p in m.parameters()], kinds
        finally:
            ops.WGRAD_SIDE, ops.WGRAD_KEEP_MAX_PIXELS, ops.WGRAD_SIDE_MIN_PIXELS = old, old_lim, old_min
            ops.PROFILE = None

    assert ops.WGRAD_SIDE == "auto"                  # the shipped default
    l0, g0, p0, kinds = run("0")
    assert {"wino4_conv<9,128>", "wgrad_wino<9>", "pgemm_kernel<1>"} <= kinds, kinds
    for rep in range(2):
        l1, g1, p1, _ = run("auto")
        assert l0 == l1
        assert len(g0) == len(g1) and all(torch.equal(a, b) for a, b in zip(g0, g1))
        assert all(torch.equal(a, b) for a, b in zip(p0, p1))
    st = next(iter(ops._SIDE.values()))
    assert st.side and not st.armed and not st.keep


# ------------------------------------------------------------------ batched C x C products (csrc/smallmm.hip)
@pytest.mark.parametrize("M,N,K,nb,bpg", [(128, 128, 128, 16, 8), (48, 80, 40, 6, 2), (16, 16, 16, 3, 1), (33, 65, 17, 4, 4)])
def test_small_mm_vs_float64(M, N, K, nb, bpg):
    """bmc_small_mm: two terms, per-batch and per-group operands, transposed operands, the rank-1 term, the vector result, alpha,
    accumulate, and a result written transposed into a column block of a wider matrix -- against float64 einsums."""
    dev = _gpu()
    from bmc_hip import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    ng = nb // bpg
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    A1, B1g, w1g = r(nb, M, K), r(ng, N, K), r(ng, K)          # term 1: A1[b] (B1[g])^T, w1[g]
    A2t, B2, w2 = r(nb, K, M), r(nb, K, N), r(nb, K)           # term 2: (A2t[b])^T B2[b], w2[b]
    u, v = r(nb, M), r(ng, N)
    gi = torch.arange(nb, device=dev) // bpg
    d = lambda t: t.double()
    ref = torch.einsum("bik,bjk->bij", d(A1), d(B1g)[gi]) + torch.einsum("bki,bkj->bij", d(A2t), d(B2)) + d(u)[:, :, None] * d(v)[gi][:, None, :]
    refv = torch.einsum("bik,bk->bi", d(A1), d(w1g)[gi]) + torch.einsum("bki,bk->bi", d(A2t), d(w2))
    alpha = 0.37
    terms = [((A1, M * K, 0, K, 1), (B1g, 0, N * K, 1, K), (w1g, 0, K, 1)), ((A2t, K * M, 0, 1, M), (B2, K * N, 0, N, 1), (w2, K, 0, 1))]
    # (1) plain result + vector
    c, vec = torch.empty(nb, M, N, device=dev), torch.empty(nb, M, device=dev)
    ops.small_mm(terms, nb, bpg, M, N, K, c=(c, M * N, 0, N, 1), alpha=alpha, uv=((u, M, 0), (v, 0, N)), vec_out=(vec, M, 0))
    assert rel_l2(c, alpha * ref) < 2e-6 and rel_l2(vec, alpha * refv) < 2e-6
    # (2) accumulate on top of it
    ops.small_mm(terms, nb, bpg, M, N, K, c=(c, M * N, 0, N, 1), alpha=alpha, uv=((u, M, 0), (v, 0, N)), vec_out=(vec, M, 0), accumulate=True)
    assert rel_l2(c, 2 * alpha * ref) < 2e-6 and rel_l2(vec, 2 * alpha * refv) < 2e-6
    # (3) transposed into columns [5, 5 + M) of a wider matrix [nb, N, M + 9]; the rest stays untouched
    wide = torch.full((nb, N, M + 9), 7.0, device=dev)
    ops.small_mm(terms[:1], nb, bpg, M, N, K, c=(wide[:, :, 5:], N * (M + 9), 0, 1, M + 9))
    ref1 = torch.einsum("bik,bjk->bij", d(A1), d(B1g)[gi])
    assert rel_l2(wide[:, :, 5:5 + M], ref1.transpose(1, 2)) < 2e-6
    assert bool((wide[:, :, :5] == 7).all()) and bool((wide[:, :, 5 + M:] == 7).all())


def test_attention_without_value_tensor_matches_explicit_form():
    """BIETwinFn / BIEFirstFn with and without the value tensor (bie.VFREE): the same function, another order of summation --
    outputs and every gradient agree to fp32 rounding."""
    dev = _gpu()
    from bmc_hip import bie, ops
    from models.submodules import BIE
    ops.set_math("fp32")
    torch.manual_seed(5)
    Cn, n, H, W = 128, 2, 24, 40
    m = BIE(Cn).to(dev)
    scaled_init(m, 3.0)
    x12 = torch.randn(2 * n, H, W, Cn, device=dev)
    xs = torch.randn(n, H, W, Cn, device=dev)
    go, gx = torch.randn(2 * n, H, W, Cn, device=dev), torch.randn(n, H, W, Cn, device=dev)

    def run(fn, vfree, nout):
        old, oldacc, oldmin = bie.VFREE, ops.ACCUM_PARAM_GRADS, bie.VFREE_MIN_PIXELS
        bie.VFREE, bie.VFREE_MIN_PIXELS = vfree, 0    # (the form without v at this test's size too)
        ops.ACCUM_PARAM_GRADS = False                 # every gradient through autograd: comparable tensors
        try:
            for p_ in m.parameters():
                p_.grad = None
            a, b = x12.clone().requires_grad_(), xs.clone().requires_grad_()
            o, xn = fn(m, a, b)
            torch.autograd.backward([o, xn], [go[:nout], gx])
            return [o.detach(), xn.detach(), a.grad, b.grad] + [p_.grad.clone() for p_ in m.parameters() if p_.grad is not None]
        finally:
            bie.VFREE, ops.ACCUM_PARAM_GRADS, bie.VFREE_MIN_PIXELS = old, oldacc, oldmin

    for fn, nout in ((bie.bie_twin, 2 * n), (bie.bie_first, n)):
        ref, new = run(fn, False, nout), run(fn, True, nout)
        assert len(ref) == len(new)
        worst = max(rel_l2(a, b) for a, b in zip(new, ref))
        print("%s: worst rel-L2 between the two forms %.2e" % (fn.__name__, worst))
        assert worst < 5e-6


@pytest.fixture
def force_vfree():
    """The BIE attention without the value tensor at every launch size (default: from 2^16 pixels per launch)."""
    from bmc_hip import bie
    old = bie.VFREE, bie.VFREE_MIN_PIXELS
    bie.VFREE, bie.VFREE_MIN_PIXELS = True, 0
    yield bie
    bie.VFREE, bie.VFREE_MIN_PIXELS = old


def test_reference_goldens_with_attention_without_value_tensor(force_vfree):
    """The reference's own golden vectors (BIE, ParallelBlk, the 3-window BPTT of the full models in both fp32-class modes) and
    the quarter-frame oracle gradients, with the value-free attention forced at these small sizes (C = 16 / 32 / 128 matrices
    through bmc_small_mm, value-parameter gradients straight into .grad): the same bars as the explicit form."""
    import test_gpu_parity as tp
    tp.test_bie_golden()
    tp.test_parallel_blk_golden()
    tp.test_full_model_bptt_golden("bmcnet_nc16", False, "fp32")
    tp.test_full_model_bptt_golden("bmcnet_nc32", False, "bf16x6")
    tp.test_full_model_bptt_golden("bmcnet_nc32", False, "fp32")
    tp.test_quarter_frame_window_gradients_vs_oracle()
    tp.test_fused_twin_bie_matches_unfused_autograd_path()
    tp.test_fused_first_output_bie_matches_unfused_autograd_path()


def test_weight_gradient_stream_recovers_from_a_backward_that_raised():
    """A backward pass that raises after its first weight-gradient launch never runs the join callback it queued.  The next pass
    must notice (another autograd graph task), join the stale work and queue a join of its own -- otherwise the optimizer could
    read a .grad the side stream is still adding to."""
    dev = _gpu()
    from bmc_hip import ops
    from bmc_hip.ops import ConvSpec, View

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    torch.manual_seed(3)
    w = (torch.randn(128, 128, 3, 3, device=dev) * 0.05).requires_grad_()
    b = torch.zeros(128, device=dev, requires_grad=True)
    x = torch.randn(2, 24, 32, 128, device=dev, requires_grad=True)
    spec = ConvSpec.dense(128)
    old, old_merge = ops.WGRAD_SIDE, ops.WGRAD_MERGE
    ops.WGRAD_SIDE, ops.WGRAD_MERGE = "1", 1          # (no queue: the weight gradient of the raising pass is LAUNCHED before it raises)
    try:
        with pytest.raises(RuntimeError, match="boom"):
            ops.conv([View(Boom.apply(x))], w, b, spec).sum().backward()
        st = ops._SIDE[torch.cuda.current_device()]
        assert st.armed                                   # the callback of the failed pass never ran
        w.grad = b.grad = None
        ops.conv([View(x)], w, b, spec).sum().backward()
        assert not st.armed and not st.keep               # this pass joined (its own callback ran)
        torch.cuda.synchronize()
        got = w.grad.clone()
        ops.WGRAD_SIDE = "0"
        w.grad = b.grad = None
        ops.conv([View(x)], w, b, spec).sum().backward()
        assert torch.equal(got, w.grad)
    finally:
        ops.WGRAD_SIDE, ops.WGRAD_MERGE = old, old_merge
