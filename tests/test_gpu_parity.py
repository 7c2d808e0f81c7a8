"""GPU parity tests (-m gpu): the HIP path (through the C ABI of libbmc_hip.so) against the CPU oracle on the
same seeded inputs and against the committed golden vectors generated from the reference.

Tolerances (fp32 everywhere; the MFMA f32 path is an exact fp32 fma chain, only summation order differs):
  * event scatter: bit-exact;
  * single kernels / layers: rel-L2 <= 2e-5;
  * SR tensor of the full recurrent model: rel-L2 <= 1e-4 (the bar BASELINE.json states);
  * parameter gradients of the full BPTT: rel-L2 <= 1e-3.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from parity_bars import *  # noqa: E402,F401,F403

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(autouse=True)
def _restore_math_mode():
    """Tests that switch the convolution arithmetic (bmc_hip.ops.set_math) leave the process default behind."""
    yield
    from bmc_hip import ops
    ops.set_math(os.environ.get("BMC_MATH", "fp32"))


def rel_l2(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# ------------------------------------------------------------------ events
@pytest.mark.parametrize("tag", ["tiny", "nfs_lr", "oob_float", "empty", "hot", "c2_lr"])
def test_events_bit_exact(tag):
    dev = _gpu()
    from bmc_hip.encodings import events_to_channels
    z = load("events.npz")
    xs, ys, ps = (torch.tensor(z[f"{tag}/{k}"], device=dev) for k in ("xs", "ys", "ps"))
    H, W = (int(v) for v in z[f"{tag}/size"])
    img = events_to_channels(xs, ys, ps, (H, W))
    assert np.array_equal(img.cpu().numpy(), z[f"{tag}/img"])
    assert np.array_equal(xs.cpu().numpy(), z[f"{tag}/xs_after"])      # caller's tensors reset in place
    assert np.array_equal(ys.cpu().numpy(), z[f"{tag}/ys_after"])


def test_events_batched_vs_oracle_full_size():
    """C2-size frames (LR 24 576 events, HR 393 216 events) in one batched launch vs the numpy oracle."""
    dev = _gpu()
    from bmc_hip.encodings import events_to_channels_batch
    from oracle.bmc_oracle import events_to_channels_np
    rng = np.random.default_rng(0)
    H, W = 720, 960
    ns = [393216, 0, 100000]
    xs = np.concatenate([rng.uniform(-2, W + 2, n) for n in ns]).astype(np.float32)
    ys = np.concatenate([rng.uniform(-2, H + 2, n) for n in ns]).astype(np.float32)
    ps = np.concatenate([rng.choice([-1.0, 1.0], n) for n in ns]).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    out = events_to_channels_batch(torch.tensor(xs, device=dev), torch.tensor(ys, device=dev),
                                   torch.tensor(ps, device=dev), torch.tensor(off, device=dev), (H, W)).cpu().numpy()
    for f in range(3):
        ref, _, _ = events_to_channels_np(xs[off[f]:off[f + 1]], ys[off[f]:off[f + 1]], ps[off[f]:off[f + 1]], (H, W))
        assert np.array_equal(out[f], ref)
    # size-independent property: total count = in-range events + out-of-range negative events
    oob = (xs >= W) | (xs < 0) | (ys >= H) | (ys < 0)
    assert out.sum() == float(np.sum(~oob | (ps < 0)))


# ------------------------------------------------------------------ conv kernel vs F.conv2d
def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,W,cins,cout,k,relu,res", [
    (2, 9, 7, [16], 16, 3, False, False),
    (1, 8, 16, [128], 128, 3, True, False),
    (2, 13, 21, [16, 32, 16], 128, 3, True, False),
    (3, 17, 33, [32], 32, 3, False, True),
    (2, 10, 12, [128, 128], 128, 1, False, True),
    (1, 20, 35, [48], 16, 1, True, False),
    (1, 11, 18, [16, 128, 16, 16, 32], 128, 3, True, False),
    (2, 5, 40, [256], 256, 3, False, False),
])
def test_conv_fwd_bwd_vs_torch(B, H, W, cins, cout, k, relu, res):
    dev = _gpu()
    from bmc_hip import ops
    from bmc_hip.ops import ConvSpec, View
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    r = torch.randn(B, cout, H, W, generator=g) if res else None
    go = torch.randn(B, cout, H, W, generator=g)
    # oracle (CPU)
    xs_c = [x.clone().requires_grad_() for x in xs]
    w_c, b_c = w.clone().requires_grad_(), b.clone().requires_grad_()
    r_c = r.clone().requires_grad_() if res else None
    y = F.conv2d(torch.cat(xs_c, 1), w_c, b_c, padding=k // 2)
    if res:
        y = y + r_c
    if relu:
        y = torch.relu(y)
    y.backward(go)
    # HIP
    xs_g = [_nhwc(x).to(dev).requires_grad_() for x in xs]
    w_g, b_g = w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    r_g = _nhwc(r).to(dev).requires_grad_() if res else None
    yg = ops.conv([View(x) for x in xs_g], w_g, b_g, ConvSpec.dense(*cins), relu=relu,
                  residual=View(r_g) if res else None)
    yg.backward(_nhwc(go).to(dev))
    assert rel_l2(yg.permute(0, 3, 1, 2), y) < 2e-5
    for xg, xc in zip(xs_g, xs_c):
        assert rel_l2(xg.grad.permute(0, 3, 1, 2), xc.grad) < 2e-5
    assert rel_l2(w_g.grad, w_c.grad) < 2e-5
    assert rel_l2(b_g.grad, b_c.grad) < 2e-5
    if res:
        assert rel_l2(r_g.grad.permute(0, 3, 1, 2), r_c.grad) < 2e-5


def test_conv_batch_views_and_groups():
    """Operands shared across / rotated over the doubled batch, two weight groups (the twin-branch launches)."""
    dev = _gpu()
    from bmc_hip import ops
    from bmc_hip.ops import ConvSpec, View
    g = torch.Generator().manual_seed(5)
    B, H, W, Cn = 2, 9, 11, 16
    x12 = torch.randn(2 * B, Cn, H, W, generator=g)
    xs = torch.randn(B, Cn, H, W, generator=g)
    w = torch.randn(2, Cn, 2 * Cn, 1, 1, generator=g) / (2 * Cn) ** 0.5
    go = torch.randn(2 * B, Cn, H, W, generator=g)
    x12c, xsc, wc = x12.clone().requires_grad_(), xs.clone().requires_grad_(), w.clone().requires_grad_()
    swapped = torch.cat([x12c[B:], x12c[:B]], 0)
    inp = torch.cat([torch.cat([xsc, xsc], 0), swapped], 1)
    y = torch.cat([F.conv2d(inp[:B], wc[0]), F.conv2d(inp[B:], wc[1])], 0)
    y.backward(go)
    x12g, xsg, wg = _nhwc(x12).to(dev).requires_grad_(), _nhwc(xs).to(dev).requires_grad_(), w.to(dev).requires_grad_()
    yg = ops.conv([View(xsg, mod=B), View(x12g, shift=B, mod=2 * B)], wg, None, ConvSpec.dense(Cn, Cn), B=2 * B, G=2,
                  cache=False)
    yg.backward(_nhwc(go).to(dev))
    assert rel_l2(yg.permute(0, 3, 1, 2), y) < 2e-5
    assert rel_l2(x12g.grad.permute(0, 3, 1, 2), x12c.grad) < 2e-5
    assert rel_l2(xsg.grad.permute(0, 3, 1, 2), xsc.grad) < 2e-5
    assert rel_l2(wg.grad, wc.grad) < 2e-5


# ------------------------------------------------------------------ layers vs golden (reference outputs)
def _sub(z, pre):
    class V:
        files = [k[len(pre):] for k in z.files if k.startswith(pre)]
        def __getitem__(self, k): return z[pre + k]
    return V()


def _load_sd(module, z, prefix="sd/"):
    sd = {k[len(prefix):]: torch.tensor(z[k]) for k in z.files if k.startswith(prefix)}
    module.load_state_dict(sd, strict=True)


def _check_grads(module, z, tol):
    named = dict(module.named_parameters())
    n = 0
    for k in z.files:
        if k.startswith("grad/"):
            assert rel_l2(named[k[5:]].grad, z[k]) < tol, k
            n += 1
    return n


def test_resblock_golden():
    dev = _gpu()
    from models.submodules import ResidualBlock_noBN
    z = _sub(load("layers.npz"), "res/")
    m = ResidualBlock_noBN(16); _load_sd(m, z); m.to(dev)
    x = torch.tensor(z["x"], device=dev, requires_grad=True)
    y = m(x)
    assert rel_l2(y, z["y"]) < 2e-5
    y.backward(torch.tensor(z["go"], device=dev))
    assert rel_l2(x.grad, z["gx"]) < 2e-5
    assert _check_grads(m, z, 2e-5) == 4


def test_layernorm_golden():
    dev = _gpu()
    from models.submodules import LayerNorm2d
    z = _sub(load("layers.npz"), "ln/")
    m = LayerNorm2d(16); _load_sd(m, z); m.to(dev)
    x = torch.tensor(z["x"], device=dev, requires_grad=True)
    y = m(x)
    assert rel_l2(y, z["y"]) < 2e-5
    y.backward(torch.tensor(z["go"], device=dev))
    assert rel_l2(x.grad, z["gx"]) < 2e-5
    assert _check_grads(m, z, 2e-5) == 2


def test_bie_golden():
    dev = _gpu()
    from models.submodules import BIE
    z = _sub(load("layers.npz"), "bie/")
    m = BIE(16); _load_sd(m, z); m.to(dev)
    xs = [torch.tensor(z[f"x{i}"], device=dev, requires_grad=True) for i in range(3)]
    ys = m(*xs)
    for i in range(3):
        assert rel_l2(ys[i], z[f"y{i}"]) < 2e-5, i
    torch.autograd.backward(ys, [torch.tensor(z[f"go{i}"], device=dev) for i in range(3)])
    for i in range(3):
        assert rel_l2(xs[i].grad, z[f"gx{i}"]) < 5e-5, i
    assert _check_grads(m, z, 5e-5) >= 14


def test_parallel_blk_golden():
    dev = _gpu()
    from models.BMCNet import ParallelBlk
    z = _sub(load("layers.npz"), "pblk/")
    m = ParallelBlk(16); _load_sd(m, z); m.to(dev)
    xs = [torch.tensor(z[f"x{i}"], device=dev, requires_grad=True) for i in range(7)]
    ys = m(*xs)
    for i in range(7):
        assert rel_l2(ys[i], z[f"y{i}"]) < 2e-5, i
    torch.autograd.backward(ys, [torch.tensor(z[f"go{i}"], device=dev) for i in range(7)])
    for i in range(7):
        assert rel_l2(xs[i].grad, z[f"gx{i}"]) < 5e-5, i
    assert _check_grads(m, z, 5e-5) >= 30


def test_shuffle_head_golden():
    dev = _gpu()
    from models.submodules import pixel_unshuffle
    from bmc_hip import ops
    z = load("layers.npz")
    y = pixel_unshuffle(torch.tensor(z["unshuffle/x"], device=dev), 4)
    assert np.array_equal(y.cpu().numpy(), z["unshuffle/y"])
    xo = torch.tensor(z["head/xo"], device=dev).permute(0, 2, 3, 1).contiguous()
    pred = ops.head(xo, torch.tensor(z["head/f2"], device=dev), 4)
    assert np.abs(pred.cpu().numpy() - z["head/y"]).max() < 1e-6


# ------------------------------------------------------------------ full recurrent models (BPTT) vs golden
@pytest.mark.parametrize("math", ["fp32", "bf16x6"])
@pytest.mark.parametrize("tag,plain", [("bmcnet_nc16", False), ("plain_nc16", True), ("bmcnet_nc32", False)])
def test_full_model_bptt_golden(tag, plain, math):
    """3-window BPTT of the reference (golden fixtures) in both fp32-class arithmetic modes: native fp32 MFMA and the
    bf16x6 split (three exact bf16 planes per operand, six plane products) -- same tolerances."""
    dev = _gpu()
    from bmc_hip import ops
    ops.set_math(math)
    from models.BMCNet import BMCNet
    from models.BMCNet_plain import BMCNet_plain
    z = load(tag + ".npz")
    scale, n_c, n_b, B, H, W, nwin = (int(v) for v in z["meta"])
    m = (BMCNet_plain if plain else BMCNet)(scale, n_c, n_b)
    _load_sd(m, z); m.to(dev)
    frames = torch.tensor(z["frames"]); gts = torch.tensor(z["gts"])
    zz = lambda c: torch.zeros(B, c, H, W, device=dev)
    h, hp, hn, pred = zz(n_c), zz(n_c), zz(n_c), zz(2 * scale * scale)
    loss = 0
    for i in range(nwin):
        x = frames[:, i:i + 2].transpose(1, 2).to(dev)          # as train.py:211,221
        if plain:
            h, pred = m(x, h, pred, i == 0)
        else:
            h, hp, hn, pred = m(x, h, hp, hn, pred, i == 0)
        assert rel_l2(pred, z[f"pred{i}"]) < 1e-4, i
        loss = loss + F.mse_loss(pred, gts[:, i + 1].to(dev))
    assert rel_l2(h, z["h"]) < 1e-4
    if not plain:
        assert rel_l2(hp, z["hp"]) < 1e-4 and rel_l2(hn, z["hn"]) < 1e-4
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * abs(float(z["loss"]))
    loss.backward()
    assert _check_grads(m, z, 1e-3) >= 20


def test_cpu_tensor_fails_loudly():
    _gpu()
    from models.submodules import ResidualBlock_noBN
    m = ResidualBlock_noBN(16)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 16, 4, 4))


def test_rccl_grad_allreducer_single_rank():
    """The RCCL path (init_process_group('nccl'), bucketed async all-reduce on a side stream, optimizer pre-step
    hook) on one GPU / one rank: gradients must equal those of the same step without the reducer."""
    dev = _gpu()
    import socket
    import torch.distributed as dist
    from models.BMCNet_plain import BMCNet_plain
    from bmc_hip.parallel import GradAllReducer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(1)
        m = BMCNet_plain(4, 16, 1).to(dev)
        x = torch.poisson(torch.full((1, 2, 2, 8, 16), 0.3)).to(dev)
        gt = torch.rand(1, 2, 32, 64, device=dev)
        z = lambda c: torch.zeros(1, c, 8, 16, device=dev)

        def grads(with_reducer):
            opt = torch.optim.SGD(m.parameters(), lr=0.0)
            red = GradAllReducer(m, opt, bucket_mb=0.01) if with_reducer else None
            opt.zero_grad()
            h, pred = m(x, z(16), z(32), True)
            h, pred = m(x, h, pred, False)
            F.mse_loss(pred, gt).backward()
            opt.step()
            out = [p.grad.clone() for p in m.parameters()]
            if red is not None:
                for hd in red._handles:
                    hd.remove()
            return out
        g0, g1 = grads(False), grads(True)
        for a, b in zip(g0, g1):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()


def test_bptt_step_recompute_is_bit_identical():
    """Per-window activation recompute (train_step.bptt_step(recompute=True)) must give the same loss and
    gradients as the store-everything path, bit for bit."""
    dev = _gpu()
    from models.BMCNet import BMCNet
    from train_step import bptt_step, encode_sequence, synthetic_events
    torch.manual_seed(3)
    B, L, H, W, scale, n_c = 1, 4, 12, 20, 4, 16
    m = BMCNet(scale, n_c, 1).to(dev)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(4.0)
    ev = synthetic_events(B, L, H, W, scale, 128, dev, seed=1)
    inp, gt = encode_sequence(ev, B, L, H, W, scale)
    assert inp.shape == (B, L, 2, H, W) and gt.shape == (B, L, 2, 4 * H, 4 * W)
    assert float(inp.sum()) == B * L * 128 and float(gt.sum()) == B * L * 128 * 16      # every event counted once
    res = []
    for rc in (False, True):
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        loss, _ = bptt_step(m, opt, inp, gt, n_c, scale, recompute=rc)
        res.append((loss.clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None]))
    assert torch.equal(res[0][0], res[1][0])
    assert len(res[0][1]) == len(res[1][1]) > 20
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)


def test_packed_weight_cache_is_not_fooled_by_recycled_addresses():
    """Two different weight tensors that the caching allocator places at the same address (same shape, version 0)
    must not share a packed-weight cache entry."""
    dev = _gpu()
    from bmc_hip import ops
    from bmc_hip.ops import ConvSpec, View
    spec = ConvSpec.dense(16)
    x = torch.randn(1, 9, 9, 16, device=dev)
    outs, refs = [], []
    for seed in (1, 2, 3):
        g = torch.Generator().manual_seed(seed)
        w = (torch.randn(16, 16, 3, 3, generator=g) * 0.1).to(dev)
        with torch.no_grad():
            outs.append(ops.conv([View(x)], w, None, spec).clone())
        refs.append(F.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1))
        del w
    for o, r in zip(outs, refs):
        assert rel_l2(o, r) < 2e-5


def test_streaming_inference_matches_training_forward():
    """infer.StreamingSR (no_grad, state carried across calls, infer_BMCNet.py:45-64 semantics) must reproduce the
    golden recurrent predictions."""
    dev = _gpu()
    from models.BMCNet import BMCNet
    from infer import StreamingSR
    z = load("bmcnet_nc16.npz")
    scale, n_c, n_b, B, H, W, nwin = (int(v) for v in z["meta"])
    m = BMCNet(scale, n_c, n_b); _load_sd(m, z); m.to(dev)
    sr = StreamingSR(m, n_c=n_c, scale=scale)
    frames = torch.tensor(z["frames"])
    for i in range(nwin):
        pred = sr.step(frames[:, i:i + 2].transpose(1, 2).to(dev))
        assert not pred.requires_grad
        assert rel_l2(pred, z[f"pred{i}"]) < 1e-4
    assert len(sr.times_ms) == nwin and sr.latency_ms() > 0


def test_raw_column_sequence_encoder_bit_exact():
    """bmc_encode_raw_events (int16/int16/float64 columns + flip flags, all frames in one launch) vs the reference's
    CPU chain get_events -> augment_event -> event_formatting -> events_to_channels (golden) -- bit-exact."""
    dev = _gpu()
    from bmc_hip.encodings import augment_flags, raw_events_to_channels_batch
    z = load("events_raw.npz")
    n = int(z["n"])
    by_size = {}
    for i in range(n):
        by_size.setdefault(tuple(int(v) for v in z[f"c{i}/size"]), []).append(i)
    for size, idxs in by_size.items():
        xs = torch.tensor(np.concatenate([z[f"c{i}/xs"] for i in idxs]), device=dev)
        ys = torch.tensor(np.concatenate([z[f"c{i}/ys"] for i in idxs]), device=dev)
        ps = torch.tensor(np.concatenate([z[f"c{i}/ps"] for i in idxs]), device=dev)
        off = torch.tensor(np.concatenate([[0], np.cumsum([len(z[f"c{i}/xs"]) for i in idxs])]), dtype=torch.int64, device=dev)
        flips = torch.tensor([augment_flags(int(z[f"c{i}/seed"])) for i in idxs], dtype=torch.uint8, device=dev)
        out = raw_events_to_channels_batch(xs, ys, ps, off, flips, size).cpu().numpy()
        for j, i in enumerate(idxs):
            assert np.array_equal(out[j], z[f"c{i}/cnt"]), (size, i)
        # no augmentation == flags 0
        out0 = raw_events_to_channels_batch(xs, ys, ps, off, None, size)
        outz = raw_events_to_channels_batch(xs, ys, ps, off, torch.zeros_like(flips), size)
        assert torch.equal(out0, outz)


# ------------------------------------------------------------------ BASELINE.json full sizes (C2: 180x240, n_c=128)
def test_full_size_conv3x3_vs_oracle_and_linearity():
    """The dominant launch shape at full size: 3x3 128->128 over the doubled twin batch (2B = 8, 180x240), against
    the CPU oracle (F.conv2d) on two samples, plus linearity conv(a*x1 + b*x2) = a*conv(x1) + b*conv(x2) on all."""
    dev = _gpu()
    from bmc_hip import ops
    from bmc_hip.ops import ConvSpec, View
    g = torch.Generator().manual_seed(0)
    B, H, W, Cn = 8, 180, 240, 128
    x1 = torch.randn(B, H, W, Cn, generator=g)
    x2 = torch.randn(B, H, W, Cn, generator=g)
    w = torch.randn(Cn, Cn, 3, 3, generator=g) / (9 * Cn) ** 0.5
    b = torch.randn(Cn, generator=g)
    spec = ConvSpec.dense(Cn)
    wg, bg = w.to(dev), b.to(dev)
    with torch.no_grad():
        y1 = ops.conv([View(x1.to(dev))], wg, bg, spec)
        y2 = ops.conv([View(x2.to(dev))], wg, bg, spec)
        y12 = ops.conv([View((0.5 * x1 - 2.0 * x2).to(dev))], wg, None, spec)
        lin = 0.5 * (y1 - bg) - 2.0 * (y2 - bg)
        assert rel_l2(y12, lin) < 1e-5
        for s in (0, 7):
            ref = F.conv2d(x1[s:s + 1].permute(0, 3, 1, 2), w, b, padding=1)
            assert rel_l2(y1[s:s + 1].permute(0, 3, 1, 2), ref) < 1e-5


def test_full_size_window_forward_vs_oracle():
    """Two full-size recurrent windows (B=1, 180x240 -> 720x960, n_c=128, n_b=5) through the HIP path vs the CPU
    oracle evaluated in float64 (so the comparison measures OUR fp32 error, not the sum of two fp32 paths):
    SR tensor within 1e-4 rel-L2 (the bar BASELINE.json states).  Weights are scaled x3 so every branch matters
    (the 0.1-scaled init is nearly linear); that also makes the second, state-carrying window ill-conditioned --
    the float32 oracle itself sits at ~7e-5 from float64 there."""
    dev = _gpu()
    from models.BMCNet import BMCNet
    from oracle import bmc_oracle as O
    torch.manual_seed(3407)
    scale, n_c, n_b, H, W = 4, 128, 5, 180, 240
    m = BMCNet(scale, n_c, n_b)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(3.0)
    params, seen = {}, {}
    for k, v in m.state_dict().items():
        params[k] = seen.setdefault(v.data_ptr(), v.double())
    frames = torch.poisson(torch.full((1, 3, 2, H, W), 0.284))
    z = lambda c: torch.zeros(1, c, H, W, dtype=torch.float64)
    with torch.no_grad():
        h, hp, hn, pred = z(n_c), z(n_c), z(n_c), z(2 * scale * scale)
        ref = []
        for i in range(2):
            h, hp, hn, pred = O.bmcnet_forward(params, frames[:, i:i + 2].transpose(1, 2).double(), h, hp, hn, pred, i == 0, scale)
            ref.append(pred)
        m.to(dev)
        from bmc_hip import ops
        zz = lambda c: torch.zeros(1, c, H, W, device=dev)
        for math in ("fp32", "bf16x6"):
            ops.set_math(math)
            h, hp, hn, pred = zz(n_c), zz(n_c), zz(n_c), zz(2 * scale * scale)
            for i in range(2):
                h, hp, hn, pred = m(frames[:, i:i + 2].transpose(1, 2).to(dev), h, hp, hn, pred, i == 0)
                assert pred.shape == (1, 2, 720, 960)
                err = rel_l2(pred, ref[i])
                print("full-size window %d (%s): SR rel-L2 vs float64 oracle %.2e" % (i, math, err))
                within(err, BAR_FULLSIZE_SR[i], CONTRACT_SR, "full-size window %d (%s) vs float64" % (i, math))


def test_quarter_frame_window_gradients_vs_oracle():
    """Forward + backward of one BMCNet(4,128,5) window at 90x120 (a quarter of the C2 frame; ~10 s of CPU oracle):
    loss and every parameter gradient vs the CPU oracle's autograd -- checks the pixel-split weight-gradient
    reductions at realistic reduction lengths (10 800 pixels per sample)."""
    dev = _gpu()
    from models.BMCNet import BMCNet
    from oracle import bmc_oracle as O
    torch.manual_seed(11)
    scale, n_c, n_b, H, W = 4, 128, 5, 90, 120
    m = BMCNet(scale, n_c, n_b)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(3.0)
    params, seen = {}, {}
    for k, v in m.state_dict().items():
        params[k] = seen.setdefault(v.data_ptr(), v.clone().requires_grad_())
    x = torch.poisson(torch.full((1, 2, 2, H, W), 0.284))
    gt = torch.poisson(torch.full((1, 2, scale * H, scale * W), 0.284))
    z = lambda c: torch.zeros(1, c, H, W)
    _, _, _, pred = O.bmcnet_forward(params, x, z(n_c), z(n_c), z(n_c), z(32), True, scale)
    loss_ref = F.mse_loss(pred, gt)
    loss_ref.backward()
    m.to(dev)
    from bmc_hip import ops
    zz = lambda c: torch.zeros(1, c, H, W, device=dev)
    for math in ("fp32", "bf16x6"):
        ops.set_math(math)
        m.zero_grad(set_to_none=True)
        _, _, _, pg = m(x.to(dev), zz(n_c), zz(n_c), zz(n_c), zz(32), True)
        loss = F.mse_loss(pg, gt.to(dev))
        loss.backward()
        within(rel_l2(pg, pred), BAR_QUARTER_SR, CONTRACT_SR, "quarter frame SR (%s)" % math)
        within(abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()), 2e-6, 1e-5, "quarter frame loss (%s)" % math)
        worst = 0.0
        n = 0
        for name, p in m.named_parameters():
            if p.grad is None:
                assert params[name].grad is None
                continue
            e = rel_l2(p.grad, params[name].grad)
            worst = max(worst, e)
            n += 1
        within(worst, BAR_QUARTER_GRAD, CONTRACT_GRAD, "quarter frame worst parameter gradient (%s)" % math)
        assert n >= 40
        print("quarter-frame (%s): SR rel-L2 %.2e, worst parameter-gradient rel-L2 %.2e over %d tensors"
              % (math, rel_l2(pg, pred), worst, n))


def test_fused_twin_bie_matches_unfused_autograd_path():
    """bmc_hip.bie.BIETwinFn (hand-written forward + backward) against the same block composed from the generic
    autograd Functions: outputs, input gradients and all 16 parameter gradients."""
    dev = _gpu()
    from models.submodules import BIE
    torch.manual_seed(4)
    Cn, B, H, W = 32, 2, 13, 21
    m = BIE(Cn).to(dev)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(4.0).add_(0.05 * torch.randn_like(p))
    x12 = torch.randn(2 * B, H, W, Cn, device=dev)
    xs = torch.randn(B, H, W, Cn, device=dev)
    go, gx = torch.randn(2 * B, H, W, Cn, device=dev), torch.randn(B, H, W, Cn, device=dev)
    res = []
    for fn in (m.forward_twin_unfused, m.forward_twin):
        a, b = x12.clone().requires_grad_(), xs.clone().requires_grad_()
        for p in m.parameters():
            p.grad = None
        o, s = fn(a, b)
        torch.autograd.backward([o, s], [go, gx])
        res.append((o.detach(), s.detach(), a.grad, b.grad, [p.grad.clone() for p in m.parameters()]))
    for i in range(4):
        assert rel_l2(res[1][i], res[0][i]) < 2e-5, i
    assert len(res[0][4]) == 16
    for ga, gb in zip(res[1][4], res[0][4]):
        assert rel_l2(ga, gb) < 5e-5


def test_fused_first_output_bie_matches_unfused_autograd_path():
    """bmc_hip.bie.BIEFirstFn (the last block's local BIE: only the first output) against forward_pair(need_second=False)
    composed from the generic autograd Functions: outputs, input gradients and every parameter gradient that exists
    (v2 gets none in either path)."""
    dev = _gpu()
    from bmc_hip import bie, ops
    from models.submodules import BIE
    ops.set_math("fp32")
    torch.manual_seed(8)
    Cn, B, H, W = 32, 2, 13, 21
    assert bie.chain_supported(Cn)
    m = BIE(Cn).to(dev)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(4.0).add_(0.05 * torch.randn_like(p))
    x12 = torch.randn(2 * B, H, W, Cn, device=dev)
    xs = torch.randn(B, H, W, Cn, device=dev)
    go, gx = torch.randn(B, H, W, Cn, device=dev), torch.randn(B, H, W, Cn, device=dev)

    def unfused(a, b):
        o1, _, s = m.forward_pair(a[:B], a[B:], b, need_second=False)
        return o1, s
    res = []
    for fn in (unfused, m.forward_first):
        a, b = x12.clone().requires_grad_(), xs.clone().requires_grad_()
        for p in m.parameters():
            p.grad = None
        o, s = fn(a, b)
        torch.autograd.backward([o, s], [go, gx])
        res.append((o.detach(), s.detach(), a.grad, b.grad, {k: (p.grad.clone() if p.grad is not None else None)
                                                               for k, p in m.named_parameters()}))
    for i in range(4):
        assert rel_l2(res[1][i], res[0][i]) < 2e-5, i
    n = 0
    for k in res[0][4]:
        ga, gb = res[1][4][k], res[0][4][k]
        assert (ga is None) == (gb is None), k
        if ga is not None:
            assert rel_l2(ga, gb) < 5e-5, k
            n += 1
    assert n == 14 and res[0][4]["v2.weight"] is None


@pytest.mark.parametrize("math", ["fp32", "bf16x6"])
def test_training_reduces_loss_and_matches_oracle_trajectory(math):
    """Three optimizer steps of the reference's recipe (Adam lr=1e-4, wd=1e-5, amsgrad; train.py:647-656) on the HIP
    path vs the same three steps of the CPU oracle + torch.optim.Adam: losses agree step by step (the trajectory, not
    only one gradient), and the loss goes down.  Both fp32-class arithmetic modes."""
    dev = _gpu()
    from bmc_hip import ops
    ops.set_math(math)
    from models.BMCNet_plain import BMCNet_plain
    from oracle import bmc_oracle as O
    from train_step import bptt_step
    torch.manual_seed(5)
    scale, n_c, n_b, B, L, H, W = 4, 16, 2, 2, 3, 10, 12
    m = BMCNet_plain(scale, n_c, n_b)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(3.0)
    # oracle side: same parameters (aliasing rebuilt), same optimizer
    params, seen = {}, {}
    for k, v in m.state_dict().items():
        params[k] = seen.setdefault(v.data_ptr(), v.clone().requires_grad_())
    uniq = list(seen.values())
    opt_ref = torch.optim.Adam(uniq, lr=1e-3, weight_decay=1e-5, amsgrad=True)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.4))
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.4))
    ref_losses = []
    for _ in range(3):
        opt_ref.zero_grad()
        loss, _, _ = O.bptt_loss(params, [inp[:, i:i + 2].transpose(1, 2) for i in range(L - 1)],
                                 [gt[:, i + 1] for i in range(L - 1)], n_c, scale, plain=True)
        loss.backward()
        opt_ref.step()
        ref_losses.append(loss.item())
    m.to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5, amsgrad=True)
    losses = []
    for _ in range(3):
        loss, _ = bptt_step(m, opt, inp.to(dev), gt.to(dev), n_c, scale, plain=True)
        losses.append(loss.item())
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 1e-4 * abs(b), (losses, ref_losses)
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("cls_name,scale,n_c,n_b,H,W", [
    ("BMCNet", 8, 16, 1, 9, 11),          # x8 SR: 64-channel pixel-unshuffle halves, 128-channel conv_o
    ("BMCNet_plain", 4, 64, 1, 12, 20),   # n_c = 64: half-filled 128-channel tiles, LayerNorm over 64 channels
    ("BMCNet", 4, 48, 2, 10, 17),         # n_c = 48 is not a power of two: LayerNorm with idle lanes, 3 chunks of 16
    ("BMCNet", 2, 32, 1, 11, 13),         # x2 SR (config/train_nfs.yml SCALE 2): 4 sub-pixel channels per polarity, padded granules
    ("BMCNet_plain", 2, 16, 2, 9, 14),    # x2 SR, plain model
])
def test_other_model_shapes_vs_oracle(cls_name, scale, n_c, n_b, H, W):
    dev = _gpu()
    import models.BMCNet as MB
    import models.BMCNet_plain as MP
    from oracle import bmc_oracle as O
    plain = cls_name == "BMCNet_plain"
    torch.manual_seed(17)
    m = (MP.BMCNet_plain if plain else MB.BMCNet)(scale, n_c, n_b)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(4.0)
    params, seen = {}, {}
    for k, v in m.state_dict().items():
        params[k] = seen.setdefault(v.data_ptr(), v.clone().requires_grad_())
    B, L = 2, 3
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.4))
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.4))
    xs = [inp[:, i:i + 2].transpose(1, 2) for i in range(L - 1)]
    gts = [gt[:, i + 1] for i in range(L - 1)]
    m.to(dev)
    z = lambda c: torch.zeros(B, c, H, W, device=dev)
    loss_ref, preds_ref, _ = O.bptt_loss(params, xs, gts, n_c, scale, plain)
    loss_ref.backward()
    state = (z(n_c), z(2 * scale * scale)) if plain else (z(n_c), z(n_c), z(n_c), z(2 * scale * scale))
    loss = 0
    for i in range(L - 1):
        state = m(xs[i].to(dev), *state, i == 0)
        assert rel_l2(state[-1], preds_ref[i]) < 1e-4, i
        loss = loss + F.mse_loss(state[-1], gts[i].to(dev))
    loss.backward()
    n = 0
    for name, p in m.named_parameters():
        if params[name].grad is None:
            continue
        assert rel_l2(p.grad, params[name].grad) < 1e-3, name
        n += 1
    assert n >= 15


@pytest.mark.parametrize("math", ["fp32", "bf16x6"])
def test_conv_fuzz_random_shapes_vs_torch(math):
    """(both fp32-class arithmetic modes: the bf16-plane kernels have their own slow paths -- several sources per column
    block, channel counts that are not multiples of the block, border tiles, the tap-split weight gradient)
    40 random convolution problems (1x1 / 3x3, 1-5 sources of 16..64 channels, Cout 16..160, odd image sizes up to
    70x90, batch 1..5, bias / ReLU / residual on or off): forward, data, weight and bias gradients vs F.conv2d.
    Exercises every tile shape of the size-adaptive dispatcher (8x16x128, 4x16x128, 4x16x64, 8x16x32)."""
    dev = _gpu()
    import random
    from bmc_hip import ops
    from bmc_hip.ops import ConvSpec, View
    ops.set_math(math)
    rnd = random.Random(1234)
    g = torch.Generator().manual_seed(99)
    for case in range(40):
        k = rnd.choice([1, 3])
        nsrc = rnd.randint(1, 5)
        cins = [16 * rnd.randint(1, 4) for _ in range(nsrc)]
        cout = 16 * rnd.choice([1, 2, 3, 4, 8, 10])
        B, H, W = rnd.randint(1, 5), rnd.randint(3, 70), rnd.randint(3, 90)
        relu, res, bias = rnd.random() < 0.5, rnd.random() < 0.5, rnd.random() < 0.7
        xs = [torch.randn(B, c, H, W, generator=g) for c in cins]
        cin = sum(cins)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g) if bias else None
        r = torch.randn(B, cout, H, W, generator=g) if res else None
        go = torch.randn(B, cout, H, W, generator=g)
        xs_g = [_nhwc(x).to(dev).requires_grad_() for x in xs]
        w_g = w.to(dev).requires_grad_()
        b_g = b.to(dev).requires_grad_() if bias else None
        yg = ops.conv([View(x) for x in xs_g], w_g, b_g, ConvSpec.dense(*cins), relu=relu,
                      residual=View(_nhwc(r).to(dev)) if res else None)
        yg.backward(_nhwc(go).to(dev))
        xs_c = [x.clone().requires_grad_() for x in xs]
        w_c = w.clone().requires_grad_()
        b_c = b.clone().requires_grad_() if bias else None
        z = F.conv2d(torch.cat(xs_c, 1), w_c, b_c, padding=k // 2)
        if res:
            z = z + r
        y = torch.relu(z) if relu else z
        if relu:
            # back-propagate through the ReLU mask the kernel actually applied: a pre-activation within rounding of
            # zero may legitimately land on either side, and one flipped element would dominate the gradient check
            (z * (yg.detach().permute(0, 3, 1, 2).cpu() > 0)).backward(go)
        else:
            y.backward(go)
        tag = (case, k, cins, cout, B, H, W, relu, res, bias)
        assert rel_l2(yg.permute(0, 3, 1, 2), y) < 2e-5, tag
        for xg, xc in zip(xs_g, xs_c):
            assert rel_l2(xg.grad.permute(0, 3, 1, 2), xc.grad) < 2e-5, tag
        assert rel_l2(w_g.grad, w_c.grad) < 3e-5, tag
        if bias:
            assert rel_l2(b_g.grad, b_c.grad) < 3e-5, tag


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_events_to_voxel_golden(tag):
    """GPU voxel encoder vs the reference's output: float weights summed per pixel in EVENT ORDER (segmented, no float
    atomics), i.e. the order of the reference's sequential index_put_ -- bit-exact, and identical run to run; the
    in-place reset of out-of-range coordinates is exact too."""
    dev = _gpu()
    from bmc_hip.encodings import events_to_voxel
    z = load("voxel.npz")
    H, W, bins = (int(v) for v in z[f"{tag}/meta"])
    xs, ys, ts, ps = (torch.tensor(z[f"{tag}/{k}"], device=dev) for k in ("xs", "ys", "ts", "ps"))
    vox = events_to_voxel(xs, ys, ts, ps, bins, (H, W))
    assert vox.shape == (bins, H, W)
    assert np.array_equal(vox.cpu().numpy(), z[f"{tag}/vox"])
    assert np.array_equal(xs.cpu().numpy(), z[f"{tag}/xs_after"]) and np.array_equal(ys.cpu().numpy(), z[f"{tag}/ys_after"])
    xs2, ys2 = (torch.tensor(z[f"{tag}/{k}"], device=dev) for k in ("xs", "ys"))
    assert torch.equal(events_to_voxel(xs2, ys2, ts, ps, bins, (H, W)), vox)          # deterministic


# ------------------------------------------------------------------ arithmetic modes of the MFMA kernels
@pytest.mark.parametrize("B,H,W,cins,cout,k", [
    (2, 45, 80, [128], 128, 3),          # 4x16x64 tiles
    (3, 37, 53, [16, 128], 128, 3),      # two sources, ragged image
    (2, 45, 80, [128, 128], 128, 1),     # 1x1, K = 256
    (2, 45, 80, [128], 32, 3),           # narrow output (8x16x32 tiles)
    (4, 90, 120, [128], 128, 3),         # 8x16x128 tiles, persistent workgroups
])
def test_math_modes_vs_float64(B, H, W, cins, cout, k):
    """Convolution forward, weight and bias gradient in the three arithmetic modes against float64:
      fp32   (v_mfma_f32_32x32x2_f32)                      : fp32 rounding error only;
      bf16x6 (3 exact bf16 planes per operand, 6 products) : the same bar -- it is an fp32-equivalent arithmetic;
      bf16   (operands rounded to bf16, fp32 accumulate)   : exact w.r.t. a float64 conv of the bf16-rounded operands,
                                                             bf16-level (4e-3) w.r.t. the unrounded one."""
    dev = _gpu()
    from bmc_hip import ops
    from bmc_hip.ops import ConvSpec, View
    g = torch.Generator().manual_seed(7)
    cin = sum(cins)
    xs = [torch.randn(B, H, W, c, generator=g).to(dev) for c in cins]
    w0 = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    b0 = (torch.randn(cout, generator=g) * 0.1).to(dev)
    go = torch.randn(B, H, W, cout, generator=g).to(dev)
    xcat = torch.cat(xs, -1).permute(0, 3, 1, 2)

    def ref(rounded):
        r = (lambda t: t.bfloat16().double()) if rounded else (lambda t: t.double())
        wd, bd = r(w0).requires_grad_(), b0.double().requires_grad_()
        y = F.conv2d(r(xcat), wd, bd, padding=k // 2)
        # the weight gradient contracts the (rounded) output gradient with the (rounded) input
        gr = r(go.permute(0, 3, 1, 2))
        dw = torch.autograd.grad(F.conv2d(r(xcat), wd, None, padding=k // 2), wd, gr)[0]
        return y.permute(0, 2, 3, 1), dw, go.double().sum((0, 1, 2))

    exact, rounded = ref(False), ref(True)
    spec = ConvSpec.dense(*cins)
    for math in ("fp32", "bf16x6", "bf16"):
        ops.set_math(math)
        w = w0.clone().requires_grad_()
        b = b0.clone().requires_grad_()
        y = ops.conv([View(t) for t in xs], w, b, spec)
        y.backward(go)
        target = rounded if math == "bf16" else exact
        e = (rel_l2(y, target[0]), rel_l2(w.grad, target[1]), rel_l2(b.grad, target[2]))
        print("%-6s fwd %.2e  dW %.2e  db %.2e (rel-L2 vs float64)" % (math, *e))
        assert max(e) < 3e-6, (math, e)
        if math == "bf16":
            assert rel_l2(y, exact[0]) < 4e-3 and rel_l2(w.grad, exact[1]) < 4e-3


def test_bf16_mode_full_model_step():
    """BMCNet(4,16,1) 3-window BPTT with bf16 operands / fp32 accumulation (BASELINE configs[3]'s arithmetic) against
    the fp32 golden: bf16-level agreement of the SR tensors, loss and gradients (not the fp32 bar)."""
    dev = _gpu()
    from bmc_hip import ops
    from models.BMCNet import BMCNet
    ops.set_math("bf16")
    z = load("bmcnet_nc16.npz")
    scale, n_c, n_b, B, H, W, nwin = (int(v) for v in z["meta"])
    m = BMCNet(scale, n_c, n_b)
    _load_sd(m, z); m.to(dev)
    frames = torch.tensor(z["frames"]); gts = torch.tensor(z["gts"])
    zz = lambda c: torch.zeros(B, c, H, W, device=dev)
    h, hp, hn, pred = zz(n_c), zz(n_c), zz(n_c), zz(2 * scale * scale)
    loss = 0
    for i in range(nwin):
        h, hp, hn, pred = m(frames[:, i:i + 2].transpose(1, 2).to(dev), h, hp, hn, pred, i == 0)
        e = rel_l2(pred, z[f"pred{i}"])
        assert 1e-6 < e < 3e-2, (i, e)      # visibly bf16 (not silently fp32), and not broken
        loss = loss + F.mse_loss(pred, gts[:, i + 1].to(dev))
    assert abs(loss.item() - float(z["loss"])) < 2e-2 * abs(float(z["loss"]))
    loss.backward()
    assert _check_grads(m, z, 0.15) >= 20


@pytest.mark.parametrize("tag", list("abcdefg"))
def test_events_to_stack_golden(tag):
    """GPU event stack vs the reference's output: bit-exact (+-1 polarities: integer-valued sums), including the
    caller-visible zeroing of out-of-range events."""
    dev = _gpu()
    from bmc_hip.encodings import events_to_stack_no_polarity
    z = load("stack.npz")
    H, W, bins = (int(v) for v in z[f"{tag}/meta"])
    xs, ys, ts, ps = (torch.tensor(z[f"{tag}/{k}"], device=dev) for k in ("xs", "ys", "ts", "ps"))
    st = events_to_stack_no_polarity(xs, ys, ts, ps, bins, sensor_size=(H, W))
    assert st.shape == (bins, H, W)
    assert np.array_equal(st.cpu().numpy(), z[f"{tag}/stack"])
    assert np.array_equal(xs.cpu().numpy(), z[f"{tag}/xs_after"]) and np.array_equal(ys.cpu().numpy(), z[f"{tag}/ys_after"])
    assert np.array_equal(ps.cpu().numpy(), z[f"{tag}/ps_after"])


def test_events_to_stack_full_size_vs_oracle():
    """A C2-sized HR window (393 216 events, 720x960, 5 bins) against the numpy oracle: bit-exact."""
    dev = _gpu()
    from bmc_hip.encodings import events_to_stack_no_polarity
    from oracle import bmc_oracle as O
    rng = np.random.default_rng(3)
    n, H, W, bins = 393216, 720, 960, 5
    xs = rng.integers(-2, W + 2, n).astype(np.float32)
    ys = rng.integers(-2, H + 2, n).astype(np.float32)
    ts = np.sort(rng.uniform(0, 1, n)).astype(np.float32)
    ts = ((ts - ts[0]) / (ts[-1] - ts[0] + np.float32(1e-6))).astype(np.float32)
    ps = rng.choice([-1.0, 1.0], n).astype(np.float32)
    ref, xa, ya, pa = O.events_to_stack_no_polarity_np(xs, ys, ts, ps, bins, (H, W))
    xt, yt, tt, pt = (torch.tensor(a, device=dev) for a in (xs, ys, ts, ps))
    st = events_to_stack_no_polarity(xt, yt, tt, pt, bins, sensor_size=(H, W))
    assert np.array_equal(st.cpu().numpy(), ref)
    assert np.array_equal(xt.cpu().numpy(), xa) and np.array_equal(pt.cpu().numpy(), pa)
