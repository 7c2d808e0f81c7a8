"""GPU parity tests added in round 2 (-m gpu), all through the C ABI of libbmc_hip.so:
bicubic resize + the size-mismatch branch of the training loop, the collate layout through the GPU sequence encoder,
BASELINE configs[3] (EventZoom 31x56, bf16) and configs[4] (RGB 180x190, T=16, per-window recompute) shapes, the
pretrained plain checkpoint pushed through the kernels, checkpoint/resume of a HIP-path trajectory, 2-rank RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, GOLDEN)


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(autouse=True)
def _restore_math_mode():
    yield
    from bmc_hip import ops
    ops.set_math(os.environ.get("BMC_MATH", "fp32"))


def rel_l2(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def oracle_params(model):
    """{state-dict key: leaf tensor} for the CPU oracle with the module's aliasing (clones of the CPU parameters)."""
    params, seen = {}, {}
    for k, v in model.state_dict().items():
        params[k] = seen.setdefault(v.data_ptr(), v.detach().cpu().clone().requires_grad_())
    return params


def scaled_init(model, gain):
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(gain)


# ------------------------------------------------------------------ bicubic resize (train.py:227-231)
@pytest.mark.parametrize("tag", ["ez", "odd", "up", "down", "same"])
def test_bicubic_resize_golden(tag):
    dev = _gpu()
    from bmc_hip import ops
    z = load("bicubic.npz")
    x = torch.tensor(z[tag + "/x"], device=dev).requires_grad_()
    y = ops.bicubic_resize(x, z[tag + "/y"].shape[-2:])
    assert tuple(y.shape) == z[tag + "/y"].shape
    y.backward(torch.tensor(z[tag + "/go"], device=dev))
    assert rel_l2(y, z[tag + "/y"]) < 2e-6          # same index arithmetic as ATen (fma in float32): weight rounding only
    assert rel_l2(x.grad, z[tag + "/gx"]) < 2e-6


def test_bicubic_backward_is_the_transpose_at_eventzoom_size():
    """<R x, g> == <x, R^T g> for the full EventZoom HR frame batch, and the backward is deterministic (gather)."""
    dev = _gpu()
    from bmc_hip import ops
    torch.manual_seed(3)
    x = torch.randn(4, 2, 124, 224, device=dev, dtype=torch.float32).requires_grad_()
    g = torch.randn(4, 2, 124, 222, device=dev)
    y = ops.bicubic_resize(x, (124, 222))
    (gx,) = torch.autograd.grad(y, x, g)
    (gx2,) = torch.autograd.grad(ops.bicubic_resize(x, (124, 222)), x, g)
    assert torch.equal(gx, gx2)
    lhs = (y.double() * g.double()).sum().item()
    rhs = (x.detach().double() * gx.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), 1.0)
    ref = F.interpolate(x.detach().cpu(), size=(124, 222), mode="bicubic", align_corners=False)
    assert rel_l2(y, ref) < 2e-6


@pytest.mark.parametrize("math", ["fp32", "bf16x6"])
def test_bptt_with_resize_branch_golden(math):
    """The reference's loop body with the size-mismatch branch taken (golden bmcnet_resize.npz): predictions, resized
    predictions, loss and every parameter gradient, through train_step.bptt_step's own code path."""
    dev = _gpu()
    from bmc_hip import ops
    ops.set_math(math)
    from models.BMCNet import BMCNet
    from test_gpu_parity import _check_grads, _load_sd
    z = load("bmcnet_resize.npz")
    scale, n_c, n_b, B, H, W, nwin, gh, gw = (int(v) for v in z["meta"])
    m = BMCNet(scale, n_c, n_b)
    _load_sd(m, z); m.to(dev)
    frames, gts = torch.tensor(z["frames"]).to(dev), torch.tensor(z["gts"]).to(dev)
    zz = lambda c: torch.zeros(B, c, H, W, device=dev)
    h, hp, hn, pred = zz(n_c), zz(n_c), zz(n_c), zz(2 * scale * scale)
    loss = 0
    for i in range(nwin):
        h, hp, hn, pred = m(frames[:, i:i + 2].transpose(1, 2), h, hp, hn, pred, i == 0)
        assert rel_l2(pred, z["pred%d" % i]) < 1e-4
        sp = ops.bicubic_resize(pred, (gh, gw))
        assert rel_l2(sp, z["spred%d" % i]) < 1e-4
        loss = loss + F.mse_loss(sp, gts[:, i + 1])
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * abs(float(z["loss"]))
    loss.backward()
    assert _check_grads(m, z, 1e-3) >= 40
    # and the packaged step does the same thing
    from train_step import bptt_step
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    l2, _ = bptt_step(m, opt, frames, gts, n_c, scale)
    assert abs(l2.item() - float(z["loss"])) < 1e-4 * abs(float(z["loss"]))


# ------------------------------------------------------------------ a-2: collate layout through the GPU sequence encoder
def test_collate_layout_from_gpu_sequence_encoder():
    """bmc_encode_raw_events on the raw int16/float64 columns of the stub recordings (all frames of the batch in two
    launches) + bptt_step's [B,L] -> window slicing == the window list the reference's
    SequenceDataset -> custom_collate -> concat_dict chain produced (golden collate.npz), bit for bit."""
    dev = _gpu()
    from bmc_hip.encodings import augment_flags, raw_events_to_channels_batch
    from test_oracle_golden_r2 import collate_case
    z, files = collate_case()
    B, L = z["lr_ranges"].shape[:2]
    seqn = int(z["seqn"])
    Hl, Wl = (int(v) for v in z["inp_res"])
    Hg, Wg = (int(v) for v in z["gt_res"])
    mech = tuple(str(s) for s in z["augment"])
    probs = tuple(float(p) for p in z["augment_prob"])

    def encode(prex, ranges, H, W):
        xs, ys, ps, off, fl = [], [], [], [0], []
        for b, f in enumerate(files):
            flags = augment_flags(int(z["item_seeds"][b]), mech, probs)
            for j in range(L):
                i0, i1 = (int(v) for v in ranges[b, j])
                xs.append(f[prex + "_events/xs"][i0:i1]); ys.append(f[prex + "_events/ys"][i0:i1])
                ps.append(f[prex + "_events/ps"][i0:i1]); off.append(off[-1] + i1 - i0); fl.append(flags)
        t = lambda a, dt: torch.tensor(np.concatenate(a), dtype=dt, device=dev)
        out = raw_events_to_channels_batch(t(xs, torch.int16), t(ys, torch.int16), t(ps, torch.float64),
                                           torch.tensor(off, dtype=torch.int64, device=dev),
                                           torch.tensor(fl, dtype=torch.uint8, device=dev), (H, W))
        return out.view(B, L, 2, H, W)

    inp, gt = encode("down8", z["lr_ranges"], Hl, Wl), encode("down2", z["gt_ranges"], Hg, Wg)
    for i in range(L - seqn + 1):               # the slicing of train_step.bptt_step
        assert np.array_equal(inp[:, i:i + seqn].cpu().numpy(), z["w%d/inp_cnt" % i].astype(np.float32)), i
        assert np.array_equal(gt[:, i:i + seqn].cpu().numpy(), z["w%d/gt_cnt" % i].astype(np.float32)), i
        # what the loop body then feeds the model / the loss with (train.py:211,213)
        x = inp[:, i:i + seqn].transpose(1, 2)
        assert tuple(x.shape) == (B, 2, seqn, Hl, Wl) and tuple(gt[:, i + 1].shape) == (B, 2, Hg, Wg)


# ------------------------------------------------------------------ BASELINE configs[3]: EventZoom 31x56 -> 124x224 (GT 124x222), bs=4
def _config3_data(B, L, H, W, gh, gw, seed=11):
    g = torch.Generator().manual_seed(seed)
    lam = 1024.0 / (H * W) / 2            # WINDOW 1024 events per LR frame (config/train_EventZoom.yml), per polarity
    inp = torch.poisson(torch.full((B, L, 2, H, W), lam), generator=g)
    gt = torch.poisson(torch.full((B, L, 2, gh, gw), lam), generator=g)
    return inp, gt


def _config3_run(m, mode, xs, gts, preds_ref, states_ref, params, B, H, W, n_c, scale, gh, gw, dev):
    """8-window BPTT of configs[3] through the HIP path in arithmetic `mode`; -> (loss, SR rel-L2 per window + final states,
    {parameter name: gradient rel-L2}, whole-gradient rel-L2)."""
    from bmc_hip import ops
    ops.set_math(mode)
    m.zero_grad(set_to_none=True)
    z = lambda c: torch.zeros(B, c, H, W, device=dev)
    st = (z(n_c), z(n_c), z(n_c), z(2 * scale * scale))
    loss, errs = 0, []
    L1 = len(xs)
    for i in range(L1):
        st = m(xs[i].to(dev), *st, i == 0)
        errs.append(rel_l2(st[-1], preds_ref[i]))
        if i == L1 - 1:
            errs.append(max(rel_l2(a, b) for a, b in zip(st[:3], states_ref)))       # final hidden states (no bilinear base in them)
        loss = loss + F.mse_loss(ops.bicubic_resize(st[-1], (gh, gw)), gts[i].to(dev))
    loss.backward()
    named = [(n, p) for n, p in m.named_parameters() if params[n].grad is not None]
    gerr = {n: rel_l2(p.grad, params[n].grad) for n, p in named}
    flat = lambda ts: torch.cat([t.detach().double().cpu().reshape(-1) for t in ts])
    ga, gb = flat([p.grad for _, p in named]), flat([params[n].grad for n, _ in named])
    return loss.item(), errs, gerr, ((ga - gb).norm() / gb.norm()).item()


def test_config3_eventzoom_shape_fp32():
    """configs[3]: BMCNet(4,128,5), LR 31x56 (W not a multiple of 16, ~110 tiles: the small-frame dispatcher paths),
    bs=4, SEQL=9 -> 8 windows BPTT, prediction 124x224 bicubic-resized to the 124x222 ground truth.
    fp32 arithmetic: SR tensors <= 1e-4, loss and every parameter gradient vs the CPU oracle."""
    dev = _gpu()
    from models.BMCNet import BMCNet
    from oracle import bmc_oracle as O
    scale, n_c, n_b, B, L, H, W = 4, 128, 5, 4, 9, 31, 56
    gh, gw = 124, 222
    torch.manual_seed(33)
    m = BMCNet(scale, n_c, n_b)
    scaled_init(m, 2.5)      # network term ~60 % of the SR tensor's norm, recurrence still well conditioned (x3 is chaotic)
    params = oracle_params(m)
    inp, gt = _config3_data(B, L, H, W, gh, gw)
    xs = [inp[:, i:i + 2].transpose(1, 2) for i in range(L - 1)]
    gts = [gt[:, i + 1] for i in range(L - 1)]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    loss_ref, preds_ref, states_ref = O.bptt_loss(params, xs, gts, n_c, scale)
    loss_ref.backward()
    m.to(dev)
    l32, e32, g32, gall = _config3_run(m, "fp32", xs, gts, preds_ref, states_ref, params, B, H, W, n_c, scale, gh, gw, dev)
    print("config3 fp32: SR rel-L2 per window (+ final states)", ["%.1e" % e for e in e32], "grad (worst tensor, whole vector)",
          max(g32.values()), gall)
    assert max(e32) < 1e-4, e32
    assert abs(l32 - loss_ref.item()) < 1e-4 * abs(loss_ref.item())
    assert max(g32.values()) < 1e-3, g32


def test_config3_eventzoom_bf16_vs_operand_rounding_oracle():
    """configs[3] in ITS arithmetic (bf16 operands, fp32 accumulate: ops.set_math("bf16") = BMC_MATH_BF16), same shape and
    step as above, against the oracle evaluated under the same contract -- oracle.bptt_loss(operand_round="bf16"): both
    operands of every convolution / 1x1 / bmm rounded to bf16 in forward, data gradient and weight gradient, exact (float64)
    accumulation (tests/test_oracle_bf16.py pins that mode against hand rounding).

    What a correct implementation of the contract can reach is NOT fp32-level agreement: rounding is discontinuous, and an
    operand that differs by one fp32 ulp between two implementations rounds to the other bf16 neighbour with probability
    ~1e-4, a 2^-8 jump that the 8-window recurrence then amplifies like any other perturbation.  That floor is measured in
    the test itself: the SAME rounded oracle evaluated with float32 accumulation (ATen's CPU kernels: another
    fp32-accumulating implementation of the contract, in another summation order) against the float64 one.  Bounds:
      * first window (no recurrence yet) <= 1e-4, loss <= 1e-3;
      * every window's SR tensor, every parameter gradient and the whole gradient vector within 3x of that floor
        (floors below 2e-4 / 1e-3 count as 2e-4 / 1e-3);
      * and the check resolves the contract: the rounded-vs-unrounded oracle distance is >= 3x the residual in every window.
    (Round 2's bounds -- 0.15 on the SR tensor, 0.8 on the gradient, against the UNROUNDED oracle -- only said "correlated".)"""
    dev = _gpu()
    from models.BMCNet import BMCNet
    from oracle import bmc_oracle as O
    scale, n_c, n_b, B, L, H, W = 4, 128, 5, 4, 9, 31, 56
    gh, gw = 124, 222
    torch.manual_seed(33)
    m = BMCNet(scale, n_c, n_b)
    scaled_init(m, 2.5)
    base = oracle_params(m)

    def cast(dt):       # aliasing kept
        made = {}
        return {k: made.setdefault(id(v), v.detach().to(dt).requires_grad_()) for k, v in base.items()}

    inp, gt = _config3_data(B, L, H, W, gh, gw)
    xs = [inp[:, i:i + 2].transpose(1, 2) for i in range(L - 1)]
    gts = [gt[:, i + 1] for i in range(L - 1)]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    params = cast(torch.float64)
    loss_ref, preds_ref, states_ref = O.bptt_loss(params, [x.double() for x in xs], [g.double() for g in gts], n_c, scale,
                                                  operand_round="bf16")
    loss_ref.backward()
    # the floor: the same contract with float32 accumulation on the CPU
    p32 = cast(torch.float32)
    loss32, preds32, _ = O.bptt_loss(p32, xs, gts, n_c, scale, operand_round="bf16")
    loss32.backward()
    names = [n for n, _ in m.named_parameters() if params[n].grad is not None]
    floor_sr = [rel_l2(a, b) for a, b in zip(preds32, preds_ref)]
    floor_g = {n: rel_l2(p32[n].grad, params[n].grad) for n in names}
    flat = lambda d: torch.cat([d[n].grad.detach().double().reshape(-1) for n in names])
    floor_all = float((flat(p32) - flat(params)).norm() / flat(params).norm())
    # how far the rounding contract itself moves the result: unrounded float64 forward
    with torch.no_grad():
        _, preds_plain, _ = O.bptt_loss(params, [x.double() for x in xs], [g.double() for g in gts], n_c, scale)
    drift = [rel_l2(a, b) for a, b in zip(preds_ref, preds_plain)]
    m.to(dev)
    lbf, ebf, gbf, gall = _config3_run(m, "bf16", xs, gts, preds_ref, states_ref, params, B, H, W, n_c, scale, gh, gw, dev)
    ratio = {n: gbf[n] / max(floor_g[n], 1e-3) for n in names}
    worst = sorted(ratio.items(), key=lambda kv: -kv[1])[:4]
    print("config3 bf16 vs bf16-operand oracle: SR rel-L2 per window (+ final states)", ["%.2e" % e for e in ebf],
          "| floor (fp32-accumulating CPU oracle vs the float64 one)", ["%.2e" % e for e in floor_sr],
          "| rounded-vs-unrounded oracle", ["%.1e" % d for d in drift], "| loss", lbf, "vs", loss_ref.item(), "(cpu fp32 %.6f)" % loss32.item(),
          "| parameter gradients: worst residual/floor", [(n, "%.2e / %.2e" % (gbf[n], floor_g[n])) for n, _ in worst],
          "| whole gradient %.2e (floor %.2e)" % (gall, floor_all))
    assert ebf[0] < 1e-4, ebf
    assert abs(lbf - loss_ref.item()) < 1e-3 * abs(loss_ref.item())
    for i in range(L - 1):
        assert ebf[i] < 3 * max(floor_sr[i], 2e-4), (i, ebf[i], floor_sr[i])
        assert drift[i] > 3 * ebf[i], (i, drift[i], ebf[i])
    assert worst[0][1] < 3, worst
    assert gall < 3 * max(floor_all, 1e-3), (gall, floor_all)


# ------------------------------------------------------------------ BASELINE configs[4]: RGB 180x190, T=16 windows, 8 sequences per GPU
def test_config4_rgb_shape_forward_vs_oracle_b1():
    """configs[4] per-GPU shape cut to B=1 for the oracle: LR 180x190 (W = 11.9 tiles: ragged last tile column), 16
    recurrent windows forward: every SR tensor <= 1e-4 of the CPU oracle's."""
    dev = _gpu()
    from models.BMCNet import BMCNet
    from oracle import bmc_oracle as O
    scale, n_c, n_b, B, L, H, W = 4, 128, 5, 1, 17, 180, 190
    torch.manual_seed(44)
    m = BMCNet(scale, n_c, n_b)
    # x2: at this frame size the channel attention (a sum over 34 200 pixels) saturates earlier than at 31x56 -- x2.5 is
    # already chaotic here (fp32 vs fp64 of the ORACLE diverges after 3 windows), x2 is well conditioned (7e-7)
    scaled_init(m, 2.0)
    params = {k: v.detach() for k, v in oracle_params(m).items()}
    g = torch.Generator().manual_seed(12)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 16384.0 / (H * W) / 2), generator=g)     # WINDOW 16384 (config/train_RGB.yml)
    m.to(dev)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    z = lambda c, d: torch.zeros(B, c, H, W, device=d)
    st = (z(n_c, dev), z(n_c, dev), z(n_c, dev), z(2 * scale * scale, dev))
    rs = (z(n_c, "cpu"), z(n_c, "cpu"), z(n_c, "cpu"), z(2 * scale * scale, "cpu"))
    errs = []
    with torch.no_grad():
        for i in range(L - 1):
            x = inp[:, i:i + 2].transpose(1, 2)
            st = m(x.to(dev), *st, i == 0)
            rs = O.bmcnet_forward(params, x, *rs, i == 0, scale)
            errs.append(max(rel_l2(st[-1], rs[-1]), rel_l2(st[0], rs[0])))          # SR tensor and hidden state
    print("config4 B=1 T=16 SR rel-L2:", ["%.1e" % e for e in errs])
    assert max(errs) < 1e-4, errs


def test_config4_rgb_shape_full_batch_recompute_properties():
    """configs[4] at its full per-GPU size (8 sequences, 16 windows, 180x190) with per-window recompute: fits in HBM,
    and -- size-independent properties -- the step's gradient is the mean of the gradients of its two half-batches
    (MSE is a batch mean; sequences are independent), its loss the mean of theirs, and a 2-sequence slice of it is
    bit-identical to the store-everything path."""
    dev = _gpu()
    from models.BMCNet import BMCNet
    from train_step import bptt_step
    scale, n_c, n_b, B, L, H, W = 4, 128, 5, 8, 17, 180, 190
    torch.manual_seed(45)
    m = BMCNet(scale, n_c, n_b).to(dev)
    scaled_init(m, 2.0)      # well conditioned at this frame size (see the B=1 test)
    g = torch.Generator().manual_seed(13)
    lam = 16384.0 / (H * W) / 2
    inp = torch.poisson(torch.full((B, L, 2, H, W), lam), generator=g).to(dev)
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), lam), generator=g).to(dev)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)

    def step(sl, recompute=True, nwin=L):
        torch.cuda.reset_peak_memory_stats()
        loss, _ = bptt_step(m, opt, inp[sl, :nwin], gt[sl, :nwin], n_c, scale, recompute=recompute)
        grads = [p.grad.clone() for p in m.parameters()]
        return loss.item(), grads, torch.cuda.max_memory_allocated() / 2 ** 30

    l_all, g_all, mem = step(slice(0, 8))
    print("config4 full batch: loss %.6f peak %.1f GiB" % (l_all, mem))
    assert np.isfinite(l_all) and mem < 200.0
    l_a, g_a, _ = step(slice(0, 4))
    l_b, g_b, _ = step(slice(4, 8))
    assert abs(l_all - 0.5 * (l_a + l_b)) < 2e-6 * abs(l_all)
    worst = max(rel_l2(ga, 0.5 * (a + b)) for ga, a, b in zip(g_all, g_a, g_b))
    assert worst < 2e-4, worst
    # recompute == store-everything, bit for bit (2 sequences x 4 windows fits either way)
    l_r, g_r, _ = step(slice(0, 2), True, 5)
    l_s, g_s, _ = step(slice(0, 2), False, 5)
    assert l_r == l_s and all(torch.equal(a, b) for a, b in zip(g_r, g_s))


# ------------------------------------------------------------------ pretrained checkpoint through the kernels
@pytest.mark.parametrize("math", ["fp32", "bf16x6"])
def test_pretrained_plain_checkpoint_on_hip_path(math):
    """The one trained weight set of the reference (pretrain/BMCNet_plain_nfs_x4.pth, n_c=128, n_b=5; its tensors are
    stored as arrays in plain_pretrained_weights.npz) on the HIP path at 45x80: the two recurrent predictions the
    REFERENCE computed with it (plain_pretrained.npz) within 1e-4."""
    dev = _gpu()
    from bmc_hip import ops
    ops.set_math(math)
    from models.BMCNet_plain import BMCNet_plain
    zw, z = load("plain_pretrained_weights.npz"), load("plain_pretrained.npz")
    m = BMCNet_plain(4, 128, 5)
    named = dict(m.named_parameters())
    assert sorted(named) == sorted(k[2:] for k in zw.files)
    with torch.no_grad():
        for k in zw.files:
            named[k[2:]].copy_(torch.tensor(zw[k]))
    assert sorted(m.state_dict().keys()) == [str(k) for k in z["keys"]]
    m.to(dev).eval()
    frames = torch.tensor(z["frames"]).to(dev)
    h, pred = torch.zeros(1, 128, 45, 80, device=dev), torch.zeros(1, 32, 45, 80, device=dev)
    with torch.no_grad():
        for i in range(2):
            h, pred = m(frames[:, i:i + 2].transpose(1, 2), h, pred, i == 0)
            e = rel_l2(pred, z["pred%d" % i])
            assert e < 1e-4, (i, e)
    assert abs(h.abs().mean().item() - float(z["h_mean_abs"])) < 1e-4 * float(z["h_mean_abs"])


# ------------------------------------------------------------------ checkpoint / resume of a HIP-path trajectory (f-4)
def test_checkpoint_resume_hip_trajectory_bit_identical(tmp_path):
    dev = _gpu()
    from checkpoint import resume, save_checkpoint
    from models.BMCNet import BMCNet
    from train_step import bptt_step
    scale, n_c, n_b, B, L, H, W = 4, 16, 1, 2, 4, 12, 20
    g = torch.Generator().manual_seed(8)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.4), generator=g).to(dev)
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.4), generator=g).to(dev)

    def make():
        torch.manual_seed(2)
        m = BMCNet(scale, n_c, n_b).to(dev)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5, amsgrad=True)
        sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.95)
        return m, opt, sch

    m, opt, sch = make()
    for _ in range(2):
        bptt_step(m, opt, inp, gt, n_c, scale)
    sch.step()
    path = str(tmp_path / "checkpoint-iteration2.pth")
    save_checkpoint(path, m, opt, sch, iteration=2)
    for _ in range(2):
        bptt_step(m, opt, inp, gt, n_c, scale)
    want = [p.detach().clone() for p in m.parameters()]
    m2, opt2, sch2 = make()
    scaled_init(m2, 0.5)                         # make sure the restore does the work
    tr = resume(path, m2, opt2, sch2)
    assert tr["iteration"] == 2 and sch2.get_last_lr() == sch.get_last_lr()
    for _ in range(2):
        bptt_step(m2, opt2, inp, gt, n_c, scale)
    for a, b in zip(want, m2.parameters()):
        assert torch.equal(a, b)
    # the bare file is the reference's format, alias keys included, with shared storage written once
    sd = torch.load(path, map_location="cpu")
    assert len(sd) == len(m.state_dict())
    nbytes = sum(p.numel() for p in m.parameters()) * 4
    assert os.path.getsize(path) < 1.5 * nbytes + 200_000, (os.path.getsize(path), nbytes)


# ------------------------------------------------------------------ 2 ranks over RCCL: the real BMCNet step, sharded
def _rank_main(rank, world, port, q):
    sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "bmcnet-esr_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from bmc_hip.parallel import GradAllReducer
    from models.BMCNet import BMCNet
    from train_step import bptt_step, shard_sequences
    scale, n_c, n_b, B, L, H, W = 4, 16, 1, 4, 3, 12, 20
    torch.manual_seed(6)
    m = BMCNet(scale, n_c, n_b).to(dev)
    g = torch.Generator().manual_seed(9)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.4), generator=g).to(dev)
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.4), generator=g).to(dev)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    GradAllReducer(m, opt, bucket_mb=0.05)
    si, sg = shard_sequences(inp, gt, rank, world)
    loss, _ = bptt_step(m, opt, si, sg, n_c, scale)
    q.put((rank, float(loss), [p.grad.detach().cpu().numpy() for p in m.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_rccl_bmcnet_step_matches_full_batch():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    dev = torch.device("cuda:0")
    from models.BMCNet import BMCNet
    from train_step import bptt_step
    scale, n_c, n_b, B, L, H, W = 4, 16, 1, 4, 3, 12, 20
    torch.manual_seed(6)
    m = BMCNet(scale, n_c, n_b).to(dev)
    g = torch.Generator().manual_seed(9)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.4), generator=g).to(dev)
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.4), generator=g).to(dev)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    loss, _ = bptt_step(m, opt, inp, gt, n_c, scale)
    assert abs(0.5 * (res[0][1] + res[1][1]) - loss.item()) < 1e-5 * abs(loss.item())
    for r in range(2):
        for a, p in zip(res[r][2], m.parameters()):
            assert rel_l2(a, p.grad) < 1e-4
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b)              # ranks hold the same averaged gradient


# ------------------------------------------------------------------ fused centre chain (csrc/chain.hip)
@pytest.mark.parametrize("Cn,n,H,W", [(128, 2, 13, 21), (128, 1, 8, 16), (64, 3, 9, 17), (32, 2, 11, 5)])
def test_bie_with_fused_chain_vs_float64_oracle(Cn, n, H, W):
    """BIE twin node with the fused convf -> LayerNorm2d -> clustering launches (forward and backward) against the
    float64 CPU oracle of the whole block: outputs, input gradients and every parameter gradient; odd image sizes
    (ragged tiles), n = 1..3 pairs.  Also checks that the fused path really ran and agrees with the unfused one."""
    dev = _gpu()
    from bmc_hip import bie as B
    from models.submodules import BIE
    from oracle import bmc_oracle as O
    assert B.chain_supported(Cn)
    torch.manual_seed(7 + Cn + n)
    m = BIE(Cn)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() == 4:
                p.copy_(torch.randn_like(p) / (p.shape[1] * p.shape[2] * p.shape[3]) ** 0.5)
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
            else:
                p.copy_(0.1 * torch.randn_like(p))
    params, seen = {}, {}
    for k, v in m.state_dict().items():
        params["b." + k] = seen.setdefault(v.data_ptr(), v.detach().double().requires_grad_())
    x12 = torch.randn(2 * n, H, W, Cn)
    xs = torch.randn(n, H, W, Cn)
    go, gs = torch.randn(2 * n, H, W, Cn), torch.randn(n, H, W, Cn)
    nchw = lambda t: t.permute(0, 3, 1, 2).double()
    x1r, x2r, xsr = (nchw(t).requires_grad_() for t in (x12[:n], x12[n:], xs))
    o1, o2, xn = O.bie(params, "b", x1r, x2r, xsr)
    torch.autograd.backward([o1, o2, xn], [nchw(go[:n]), nchw(go[n:]), nchw(gs)])
    m.to(dev)

    def run():
        m.zero_grad(set_to_none=True)
        a, b = x12.to(dev).requires_grad_(), xs.to(dev).requires_grad_()
        o12, xsn = m.forward_twin(a, b)
        torch.autograd.backward([o12, xsn], [go.to(dev), gs.to(dev)])
        return o12, xsn, a.grad, b.grad, {k: p.grad.clone() for k, p in m.named_parameters()}

    launches = []
    from bmc_hip import ops
    ops.PROFILE = launches
    o12, xsn, ga, gb, gp = run()
    ops.PROFILE = None
    kinds = [r[0] for r in launches]
    assert kinds.count("chain_kernel<fwd>") == 1 and kinds.count("chain_kernel<bwd>") == 1
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1)
    assert rel_l2(o12[:n], nhwc(o1)) < 2e-5 and rel_l2(o12[n:], nhwc(o2)) < 2e-5 and rel_l2(xsn, nhwc(xn)) < 2e-5
    assert rel_l2(ga[:n], nhwc(x1r.grad)) < 5e-5 and rel_l2(ga[n:], nhwc(x2r.grad)) < 5e-5 and rel_l2(gb, nhwc(xsr.grad)) < 5e-5
    for k, g in gp.items():
        assert rel_l2(g, params["b." + k].grad) < 5e-5, k
    # unfused launches (conv, LayerNorm, conv / their separate backward kernels) give the same numbers
    B.FUSE_CHAIN = False
    try:
        o12u, xsnu, gau, gbu, gpu_ = run()
    finally:
        B.FUSE_CHAIN = True
    assert rel_l2(o12, o12u) < 5e-6 and rel_l2(xsn, xsnu) < 5e-6 and rel_l2(ga, gau) < 2e-5 and rel_l2(gb, gbu) < 2e-5
    for k in gp:
        assert rel_l2(gp[k], gpu_[k]) < 2e-5, k


# ------------------------------------------------------------------ remaining encodings of dataloader/encodings.py
@pytest.mark.parametrize("tag", list("abcdefg"))
def test_events_to_stack_polarity_golden(tag):
    dev = _gpu()
    from bmc_hip.encodings import events_to_stack_polarity
    z = load("stack_polarity.npz")
    H, W, bins = (int(v) for v in z[tag + "/meta"])
    xs, ys, ts, ps = (torch.tensor(z[tag + "/" + k], device=dev) for k in ("xs", "ys", "ts", "ps"))
    st = events_to_stack_polarity(xs, ys, ts, ps, bins, sensor_size=(H, W))
    assert tuple(st.shape) == z[tag + "/stack"].shape              # [B,H,W] zeros on the early return, else [2,B,H,W]
    assert np.array_equal(st.cpu().numpy(), z[tag + "/stack"])
    assert np.array_equal(xs.cpu().numpy(), z[tag + "/xs_after"]) and np.array_equal(ys.cpu().numpy(), z[tag + "/ys_after"])
    assert np.array_equal(ps.cpu().numpy(), z[tag + "/ps_after"])


@pytest.mark.parametrize("tag", list("abcd"))
def test_events_to_mask_golden(tag):
    dev = _gpu()
    from bmc_hip.encodings import events_to_mask
    z = load("mask.npz")
    H, W = (int(v) for v in z[tag + "/meta"])
    xs, ys, ps = (torch.tensor(z[tag + "/" + k], device=dev) for k in ("xs", "ys", "ps"))
    mk = events_to_mask(xs, ys, ps, sensor_size=(H, W))
    assert np.array_equal(mk.cpu().numpy(), z[tag + "/mask"])
    for t, k in ((xs, "xs_after"), (ys, "ys_after"), (ps, "ps_after")):
        assert np.array_equal(t.cpu().numpy(), z[tag + "/" + k])


def test_voxel_full_size_is_deterministic_and_conserves_weight():
    """180x240 frames, 24 576 events each, 5 bins, 4 frames in one launch: identical bits on every run, equal to the
    event-order numpy oracle, and (size-independent property) every in-range event's weights over the bins sum to p."""
    dev = _gpu()
    from bmc_hip import ops
    from oracle import bmc_oracle as O
    rng = np.random.default_rng(3)
    H, W, bins, nf, n = 180, 240, 5, 4, 24576
    xs = rng.uniform(-1.0, W + 1.0, nf * n).astype(np.float32)
    ys = rng.uniform(-1.0, H + 1.0, nf * n).astype(np.float32)
    ts = np.concatenate([np.sort(rng.uniform(0, 1, n)) for _ in range(nf)]).astype(np.float32)
    ps = rng.choice([-1.0, 1.0], nf * n).astype(np.float32)
    off = torch.arange(nf + 1, dtype=torch.int64, device=dev) * n
    t = lambda a: torch.tensor(a, device=dev)
    v1 = ops.events_to_voxel_batched(t(xs), t(ys), t(ts), t(ps), off, bins, H, W, mutate=False)
    v2 = ops.events_to_voxel_batched(t(xs), t(ys), t(ts), t(ps), off, bins, H, W, mutate=False)
    assert torch.equal(v1, v2)
    for f in (0, nf - 1):
        sl = slice(f * n, (f + 1) * n)
        ref, _, _ = O.events_to_voxel_np(xs[sl], ys[sl], ts[sl], ps[sl], bins, (H, W))
        assert np.array_equal(v1[f].cpu().numpy(), ref)
    inr = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
    assert abs(v1.double().sum().item() - float(ps[inr].astype(np.float64).sum())) < 20.0      # + the quirk's (H-1, 0) deposits


# ------------------------------------------------------------------ bf16x6 arithmetic on adversarial operands
def _conv_err_vs_f64(x, w, k, math):
    """-> (y, error of the HIP conv / sum_k |x||w| per output, i.e. relative to the condition-independent scale)."""
    from bmc_hip import ops
    from bmc_hip.ops import ConvSpec, View
    ops.set_math(math)
    with torch.no_grad():
        y = ops.conv([View(x)], w, None, ConvSpec.dense(x.shape[-1]))
    xd, wd = x.double().cpu().permute(0, 3, 1, 2), w.double().cpu()
    ref = F.conv2d(xd, wd, None, padding=k // 2).permute(0, 2, 3, 1)
    scale = F.conv2d(xd.abs(), wd.abs(), None, padding=k // 2).permute(0, 2, 3, 1)
    return y, ((y.double().cpu() - ref).abs() / scale.clamp_min(1e-300)).max().item()


@pytest.mark.parametrize("k", [3, 1])
def test_bf16x6_wide_dynamic_range_and_cancellation(k):
    """bf16x6 (three exact bf16 planes per fp32 operand, six plane products, fp32 accumulation) against float64 on
    operands a Gaussian test never produces: magnitudes spread over 2^-40 .. 2^40 in BOTH operands (products over
    2^-80 .. 2^80: the planes' relative weights 1, 2^-8, 2^-16 must hold at every exponent), and exactly cancelling
    channel pairs (x, -x with equal weights: every plane product has an exact opposite).  Error is measured against
    sum |x||w| (the scale any fp32 summation's rounding error is proportional to); the bar is the native fp32 MFMA's own
    error on the same data x 2 -- the mode claims fp32 equivalence, not more."""
    dev = _gpu()
    g = torch.Generator().manual_seed(5 + k)
    B, H, W, Cn = 2, 24, 40, 128
    mant = lambda *s: (1.0 + torch.rand(*s, generator=g)) * (torch.randint(0, 2, s, generator=g) * 2 - 1)
    x = (mant(B, H, W, Cn) * torch.exp2(torch.randint(-40, 41, (B, H, W, Cn), generator=g).float())).to(dev)
    w = (mant(Cn, Cn, k, k) * torch.exp2(torch.randint(-40, 41, (Cn, Cn, k, k), generator=g).float())).to(dev)
    _, e32 = _conv_err_vs_f64(x, w, k, "fp32")
    y6, e6 = _conv_err_vs_f64(x, w, k, "bf16x6")
    print("wide range k=%d: max |err| / sum|x||w|: fp32 %.2e  bf16x6 %.2e" % (k, e32, e6))
    assert torch.isfinite(y6).all()
    assert e32 < 2e-6 and e6 < max(2 * e32, 2.5e-7)
    # exact cancellation: channel 2j+1 = -channel 2j, same weights -> the exact result is 0 everywhere
    xc = torch.randn(B, H, W, Cn, generator=g).to(dev)
    xc[..., 1::2] = -xc[..., 0::2]
    wc = torch.randn(Cn, Cn, k, k, generator=g).to(dev)
    wc[:, 1::2] = wc[:, 0::2]
    yc32, c32 = _conv_err_vs_f64(xc, wc, k, "fp32")
    yc6, c6 = _conv_err_vs_f64(xc, wc, k, "bf16x6")
    print("cancellation k=%d: max |y| / sum|x||w|: fp32 %.2e  bf16x6 %.2e" % (k, c32, c6))
    assert c32 < 2e-6 and c6 < max(2 * c32, 2.5e-7)


def test_bf16x6_denormals_and_nonfinite_semantics():
    """Documented exceptional-value behaviour of the split arithmetic (DESIGN.md, bf16-plane modes):
      * subnormal fp32 operands: each plane is a (sub)normal bf16 the matrix core may flush -- the result may lose them, i.e.
        differ from float64 by at most K * 2^-126 * max|w| (absolute), never by more;
      * a +-Inf operand splits into (Inf, NaN, NaN): outputs that touch it are NaN (native fp32 MFMA: +-Inf or NaN),
        a NaN operand gives NaN in both modes; outputs that do not touch them are unaffected."""
    dev = _gpu()
    from bmc_hip import ops
    from bmc_hip.ops import ConvSpec, View
    g = torch.Generator().manual_seed(9)
    B, H, W, Cn, k = 1, 16, 32, 128, 3
    x = torch.randn(B, H, W, Cn, generator=g)
    x[:, :, :16] *= 1e-41                     # left half of the image: subnormal activations
    w = torch.randn(Cn, Cn, k, k, generator=g) / 34.0
    x, w = x.to(dev), w.to(dev)
    y6, _ = _conv_err_vs_f64(x, w, k, "bf16x6")
    ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), None, padding=1).permute(0, 2, 3, 1)
    aerr = (y6.double().cpu() - ref).abs()
    bound = Cn * k * k * 2.0 ** -126 * w.abs().max().item()
    assert aerr[:, :, :14].max().item() <= bound, (aerr[:, :, :14].max().item(), bound)          # purely subnormal windows
    scale = F.conv2d(x.double().cpu().permute(0, 3, 1, 2).abs(), w.double().cpu().abs(), None, padding=1).permute(0, 2, 3, 1)
    assert (aerr[:, :, 18:] / scale[:, :, 18:]).max().item() < 2.5e-7                             # normal windows: fp32-class error
    # non-finite operands
    x2 = torch.randn(B, H, W, Cn, generator=g).to(dev)
    x2[0, 3, 5, 7] = float("inf")
    x2[0, 10, 20, 9] = float("nan")
    ops.set_math("bf16x6")
    with torch.no_grad():
        y = ops.conv([View(x2)], w, None, ConvSpec.dense(Cn))
    touched = torch.zeros(H, W, dtype=torch.bool)
    touched[2:5, 4:7] = True
    touched[9:12, 19:22] = True
    yc = y[0].cpu()
    assert torch.isnan(yc[touched]).all()
    assert torch.isfinite(yc[~touched]).all()
    ops.set_math("fp32")
    with torch.no_grad():
        y32 = ops.conv([View(x2)], w, None, ConvSpec.dense(Cn))[0].cpu()
    assert (~torch.isfinite(y32[touched])).all() and torch.isfinite(y32[~touched]).all()
    assert rel_l2(yc[~touched], y32[~touched]) < 1e-5


# ------------------------------------------------------------------ conv1.hip: the 16x16x4 / LDS-DMA-ring 1x1 convolution
def test_conv1_kernel_all_epilogues_vs_old_kernel_and_float64():
    """Runs in a child process with BMC_CONV1_MIN_TILES=0 (so that small problems take conv1.hip too) and compares, on
    random 1x1 problems covering every epilogue feature (multi-source, batch maps, per-sample / per-group weights, bias,
    residual with rotation, ReLU, mask, accumulate, narrow outputs, ragged images), conv1.hip against float64 and against
    conv.hip's kernel (BMC_NO_CONV1=1)."""
    _gpu()
    import subprocess
    code = r'''
import os, sys, json
import torch, torch.nn.functional as F
sys.path[:0] = [ROOT, os.path.join(ROOT, "bmcnet-esr_amd")]
from bmc_hip import ops, lib
from bmc_hip.ops import ConvSpec, View, _src, conv_raw, _packed_weight, coutpad
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(os.environ["SEED"]))
out = []
for case in range(14):
    B = int(torch.randint(1, 5, (1,), generator=g)); H = int(torch.randint(3, 30, (1,), generator=g)); W = int(torch.randint(3, 45, (1,), generator=g))
    nsrc = int(torch.randint(1, 4, (1,), generator=g))
    nchs = [16 * int(torch.randint(1, 9, (1,), generator=g)) for _ in range(nsrc)]
    Cout = [128, 128, 32, 64, 16, 256][case % 6]
    G = [1, B, 1, 1][case % 4] if B > 1 else 1
    relu, use_res, use_mask, acc, bias = case % 2 == 0, case % 3 == 0, case % 5 == 1, case % 4 == 2, case % 3 != 1
    xs = [torch.randn(B, H, W, c, generator=g).to(dev) for c in nchs]
    cin = sum(nchs)
    w = (torch.randn(G, Cout, cin, 1, generator=g) / cin ** 0.5).to(dev)
    b = (torch.randn(G, Cout, generator=g) * 0.3).to(dev) if bias else None
    res = torch.randn(B, H, W, Cout, generator=g).to(dev) if use_res else None
    mask = torch.randn(B, H, W, Cout, generator=g).to(dev) if use_mask else None
    base = torch.randn(B, H, W, Cout, generator=g).to(dev)
    spec = ConvSpec.dense(*nchs)
    wp = _packed_weight(w.contiguous(), spec, None)
    y = base.clone()
    shift = 1 if (use_res and B > 1) else 0
    conv_raw([_src(t, 0, c, 0, None, 0, B) for t, c in zip(xs, nchs)], wp, spec.kpad * coutpad(Cout), b, Cout if bias else 0,
             y.data_ptr(), H * W * Cout, Cout, B, H, W, Cout, 1, relu=relu,
             residual=_src(res, 0, Cout, shift, B, 0, B) if use_res else None, bpg=B // G, accumulate=acc,
             mask=_src(mask, 0, Cout, 0, None, 0, B) if use_mask else None)
    xd = torch.cat(xs, -1).double().cpu()
    wd = w.double().cpu()
    ref = torch.stack([xd[i] @ wd[(i // (B // G))][:, :, 0].T for i in range(B)])
    if bias: ref = ref + b.double().cpu()[torch.arange(B) // (B // G)][:, None, None, :]
    if use_res: ref = ref + torch.roll(res.double().cpu(), -shift, 0)
    if relu: ref = ref.clamp_min(0)
    if use_mask: ref = torch.where(mask.double().cpu() > 0, ref, torch.zeros_like(ref))
    if acc: ref = ref + base.double().cpu()
    err = ((y.double().cpu() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    out.append((err, y.cpu().numpy().tobytes().hex()[:64], float(y.double().sum())))
print(json.dumps(out))
'''.replace("ROOT", repr(os.path.dirname(HERE)))
    res = {}
    for tag, env in (("conv1", {"BMC_CONV1_MIN_TILES": "0"}), ("old", {"BMC_NO_CONV1": "1"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, SEED="5", **env))
        assert r.returncode == 0, r.stderr[-3000:]
        import json
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    for (e1, _, s1), (e0, _, s0) in zip(res["conv1"], res["old"]):
        assert e1 < 3e-6 and e0 < 3e-6, (e1, e0)
        assert abs(s1 - s0) <= 1e-3 * max(1.0, abs(s0))


# ------------------------------------------------------------------ streaming inference: HIP-graph replay per window
def test_streaming_inference_graph_replay_matches_eager():
    """infer.StreamingSR(graph=True): from the third window on a window is one HIP-graph replay with the input and the
    recurrent state in static buffers -- same predictions as the eager loop, bit for bit, over 7 windows."""
    dev = _gpu()
    from infer import StreamingSR
    from models.BMCNet import BMCNet
    torch.manual_seed(4)
    scale, n_c, n_b, B, H, W = 4, 32, 2, 1, 45, 80
    m = BMCNet(scale, n_c, n_b).to(dev)
    scaled_init(m, 2.0)
    g = torch.Generator().manual_seed(6)
    frames = torch.poisson(torch.full((B, 8, 2, H, W), 0.284), generator=g).to(dev)
    eager, graph = StreamingSR(m, n_c, scale), StreamingSR(m, n_c, scale, graph=True)
    for i in range(7):
        x = frames[:, i:i + 2].transpose(1, 2)
        pe = eager.step(x).clone()
        pg = graph.step(x).clone()
        assert torch.equal(pe, pg), i
    assert graph._graph is not None


# ------------------------------------------------------------------ no kernel reads past the end of an operand
@pytest.mark.parametrize("math", ["fp32", "bf16x6", "bf16", "fp32-wino", "fp32-c1p"])
def test_no_reads_past_operand_end(math):
    """The conv fuzz shapes (1-5 sources, 3..90 pixel sides, fwd + data + weight gradients) with every operand placed so
    that it ENDS at the end of its own 2 MiB-multiple hipMalloc (caching allocator off, tools/fuzz_repro.py): an interior
    tile's fast path that fetches one halo row too many -- harmless inside the caching allocator's big blocks, a page fault
    when the tensor happens to close a mapping (found that way in pgemm_bf9x3_kernel) -- aborts the child process."""
    _gpu()
    import subprocess
    env = dict(os.environ, PYTORCH_NO_HIP_MEMORY_CACHING="1", PYTORCH_NO_CUDA_MEMORY_CACHING="1", FZ_END="1", FZ_SEEDS="2")
    if math == "fp32-wino":        # the Winograd kernels' LDS-DMA halo / weight streams and row loads under the same guard (round 3)
        math, env["FZ_WINO"] = "fp32", "1"
    if math == "fp32-c1p":         # conv1p.hip's pixel-tile DMA (partial last tiles re-read the image's last pixel)
        math, env["FZ_C1P"], env["BMC_CONV1P_MIN_TILES"] = "fp32", "1", "0"
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "fuzz_repro.py"), math],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("done"), (r.returncode, r.stdout[-500:], r.stderr[-1500:])
