"""Pin the CPU oracle (oracle/bmc_oracle.py) against golden vectors produced by
the reference itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import bmc_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def params_from(z, prefix="sd/", requires_grad=False):
    """Rebuild a state-dict mapping, re-creating the aliasing of shared tensors."""
    p = {}
    cache = {}
    for k in z.files:
        if not k.startswith(prefix):
            continue
        arr = z[k]
        key = arr.tobytes()[:256] + str(arr.shape).encode()
        if key not in cache:
            cache[key] = torch.tensor(arr, requires_grad=requires_grad)
        p[k[len(prefix):]] = cache[key]
    return p


@pytest.mark.parametrize("tag", ["tiny", "nfs_lr", "oob_float", "empty", "hot", "c2_lr"])
def test_events_to_channels_bit_exact(tag):
    z = load("events.npz")
    img, xa, ya = O.events_to_channels_np(z[f"{tag}/xs"], z[f"{tag}/ys"], z[f"{tag}/ps"], tuple(z[f"{tag}/size"]))
    assert np.array_equal(img, z[f"{tag}/img"])
    assert np.array_equal(xa, z[f"{tag}/xs_after"])
    assert np.array_equal(ya, z[f"{tag}/ys_after"])


def test_events_quirk_documented():
    # the out-of-range negative event lands on [H-1, 0] of channel 1
    z = load("events.npz")
    img = z["tiny/img"]
    assert img[1, 5, 0] == 1 and img[0, 5, 0] == 1 and img[0].sum() == 3 and img[1].sum() == 3


def _sub(z, pre):
    class V:  # view of one sub-case
        files = [k[len(pre):] for k in z.files if k.startswith(pre)]
        def __getitem__(self, k): return z[pre + k]
    return V()


def test_resblock():
    z = _sub(load("layers.npz"), "res/")
    p = params_from(z, requires_grad=True)
    x = torch.tensor(z["x"], requires_grad=True)
    y = O.res_block(p, "", x) if False else O.res_block({("m." + k): v for k, v in p.items()}, "m", x)
    assert rel_l2(y.detach(), z["y"]) < 1e-6
    y.backward(torch.tensor(z["go"]))
    assert rel_l2(x.grad, z["gx"]) < 1e-6
    assert rel_l2(p["conv1.weight"].grad, z["grad/conv1.weight"]) < 1e-6
    assert rel_l2(p["conv2.bias"].grad, z["grad/conv2.bias"]) < 1e-6


def test_layernorm():
    z = _sub(load("layers.npz"), "ln/")
    w = torch.tensor(z["sd/weight"], requires_grad=True); b = torch.tensor(z["sd/bias"], requires_grad=True)
    x = torch.tensor(z["x"], requires_grad=True)
    y = O.layer_norm_2d(x, w, b)
    assert rel_l2(y.detach(), z["y"]) < 1e-6
    y.backward(torch.tensor(z["go"]))
    assert rel_l2(x.grad, z["gx"]) < 2e-6
    assert rel_l2(w.grad, z["grad/weight"]) < 1e-6
    assert rel_l2(b.grad, z["grad/bias"]) < 1e-6


def test_bie():
    z = _sub(load("layers.npz"), "bie/")
    p = {("m." + k): v for k, v in params_from(z, requires_grad=True).items()}
    xs = [torch.tensor(z[f"x{i}"], requires_grad=True) for i in range(3)]
    ys = O.bie(p, "m", *xs)
    for i in range(3):
        assert rel_l2(ys[i].detach(), z[f"y{i}"]) < 1e-6
    torch.autograd.backward(ys, [torch.tensor(z[f"go{i}"]) for i in range(3)])
    for i in range(3):
        assert rel_l2(xs[i].grad, z[f"gx{i}"]) < 2e-6
    for k in z.files:
        if k.startswith("grad/"):
            assert rel_l2(p["m." + k[5:]].grad, z[k]) < 2e-6, k


def test_parallel_blk():
    z = _sub(load("layers.npz"), "pblk/")
    p = {("m." + k): v for k, v in params_from(z, requires_grad=True).items()}
    xs = [torch.tensor(z[f"x{i}"], requires_grad=True) for i in range(7)]
    ys = O.parallel_blk(p, "m", *xs)
    for i in range(7):
        assert rel_l2(ys[i].detach(), z[f"y{i}"]) < 1e-6
    torch.autograd.backward(ys, [torch.tensor(z[f"go{i}"]) for i in range(7)])
    for i in range(7):
        assert rel_l2(xs[i].grad, z[f"gx{i}"]) < 2e-6
    for k in z.files:
        if k.startswith("grad/"):
            assert rel_l2(p["m." + k[5:]].grad, z[k]) < 2e-6, k


def test_shuffle_and_head():
    z = load("layers.npz")
    assert np.array_equal(O.pixel_unshuffle(torch.tensor(z["unshuffle/x"]), 4).numpy(), z["unshuffle/y"])
    x = torch.tensor(z["unshuffle/x"])
    assert torch.equal(O.pixel_shuffle(O.pixel_unshuffle(x, 4), 4), x)
    y = O.pixel_shuffle(torch.tensor(z["head/xo"]), 4) + O.bilinear_up(torch.tensor(z["head/f2"]), 4)
    assert np.abs(y.numpy() - z["head/y"]).max() < 5e-7


@pytest.mark.parametrize("tag,plain", [("bmcnet_nc16", False), ("plain_nc16", True), ("bmcnet_nc32", False)])
def test_full_model_bptt(tag, plain):
    z = load(tag + ".npz")
    scale, n_c, n_b, B, H, W, nwin = (int(v) for v in z["meta"])
    p = params_from(z, requires_grad=True)
    frames = torch.tensor(z["frames"]); gts = torch.tensor(z["gts"])
    inp = [frames[:, i:i + 2].transpose(1, 2) for i in range(nwin)]
    gt = [gts[:, i + 1] for i in range(nwin)]
    loss, preds, (h, hp, hn) = O.bptt_loss(p, inp, gt, n_c, scale, plain)
    for i in range(nwin):
        assert rel_l2(preds[i].detach(), z[f"pred{i}"]) < 1e-6, i
    assert rel_l2(h.detach(), z["h"]) < 1e-6
    if not plain:
        assert rel_l2(hp.detach(), z["hp"]) < 1e-6 and rel_l2(hn.detach(), z["hn"]) < 1e-6
    assert abs(loss.item() - float(z["loss"])) < 1e-6 * abs(float(z["loss"]))
    loss.backward()
    n = 0
    for k in z.files:
        if k.startswith("grad/"):
            assert rel_l2(p[k[5:]].grad, z[k]) < 5e-6, k
            n += 1
    assert n >= 20


def test_plain_pretrained_outputs():
    """Needs the reference's pretrained checkpoint (reference data, not shipped):
    skipped where /root/reference is absent (e.g. the GPU box)."""
    ck = "/root/reference/pretrain/BMCNet_plain_nfs_x4.pth"
    if not os.path.exists(ck):
        pytest.skip("reference checkpoint not present")
    z = load("plain_pretrained.npz")
    p = torch.load(ck, map_location="cpu")
    assert sorted(p.keys()) == [str(k) for k in z["keys"]]
    frames = torch.tensor(z["frames"])
    h, pred = torch.zeros(1, 128, 45, 80), torch.zeros(1, 32, 45, 80)
    with torch.no_grad():
        for i in range(2):
            h, pred = O.plain_forward(p, frames[:, i:i + 2].transpose(1, 2), h, pred, i == 0)
            assert rel_l2(pred, z[f"pred{i}"]) < 1e-6


def test_adam_amsgrad():
    z = load("adam.npz")
    ws = [torch.tensor(z["w0"]), torch.tensor(z["w1"])]
    st = {"step": 0, "m": [torch.zeros_like(w) for w in ws], "v": [torch.zeros_like(w) for w in ws],
          "vmax": [torch.zeros_like(w) for w in ws]}
    for step in range(3):
        O.adam_amsgrad_step(ws, [torch.tensor(z[f"g{step}_{i}"]) for i in range(2)], st)
        for i in range(2):
            assert np.abs(ws[i].numpy() - z[f"w_after{step}_{i}"]).max() < 1e-7


def test_raw_column_encoder_and_augmentation_flags():
    """Raw int16/float64 columns + flip augmentation (dataloader/h5dataset.py:261-316 chain) vs the reference."""
    z = load("events_raw.npz")
    n = int(z["n"])
    seen = set()
    for i in range(n):
        k = f"c{i}"
        size = tuple(int(v) for v in z[k + "/size"])
        flags = O.augment_flags(int(z[k + "/seed"]))
        assert flags == int(z[k + "/flags"]), k
        seen.add(flags)
        img = O.encode_raw_frame_np(z[k + "/xs"], z[k + "/ys"], z[k + "/ps"], flags, size)
        assert np.array_equal(img, z[k + "/cnt"]), k
    assert seen == set(range(8))   # every flip combination exercised


def test_product_augment_flags_match_reference_seeding():
    from bmc_hip.encodings import augment_flags
    z = load("events_raw.npz")
    for i in range(int(z["n"])):
        assert augment_flags(int(z[f"c{i}/seed"])) == int(z[f"c{i}/flags"])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_events_to_voxel_oracle(tag):
    """Temporal-bilinear voxel grid (dataloader/encodings.py:272-287) incl. the out-of-range side effect of the first
    bin's call; the golden was produced with one thread (sequential accumulation), which the oracle reproduces bit
    for bit."""
    z = load("voxel.npz")
    H, W, bins = (int(v) for v in z[f"{tag}/meta"])
    vox, xa, ya = O.events_to_voxel_np(z[f"{tag}/xs"], z[f"{tag}/ys"], z[f"{tag}/ts"], z[f"{tag}/ps"], bins, (H, W))
    assert np.array_equal(vox, z[f"{tag}/vox"])
    assert np.array_equal(xa, z[f"{tag}/xs_after"]) and np.array_equal(ya, z[f"{tag}/ys_after"])


@pytest.mark.parametrize("tag", list("abcdefg"))
def test_events_to_stack_oracle(tag):
    """Event stack (dataloader/encodings.py:202-238) incl. the hand-written binary search's quirks (ties, bounds that
    hit a timestamp exactly), the <= 3 events / all-zero-timestamps early return and the in-place masking of the
    caller's xs / ys / ps: bit-exact against outputs of the reference itself."""
    z = load("stack.npz")
    H, W, bins = (int(v) for v in z[f"{tag}/meta"])
    st, xa, ya, pa = O.events_to_stack_no_polarity_np(z[f"{tag}/xs"], z[f"{tag}/ys"], z[f"{tag}/ts"], z[f"{tag}/ps"], bins, (H, W))
    assert np.array_equal(st, z[f"{tag}/stack"])
    assert np.array_equal(xa, z[f"{tag}/xs_after"]) and np.array_equal(ya, z[f"{tag}/ys_after"])
    assert np.array_equal(pa, z[f"{tag}/ps_after"])
