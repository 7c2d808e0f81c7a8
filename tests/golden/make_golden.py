#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE (imported from
/root/reference, build container only) on seeded inputs.

Only data (inputs, weights, expected outputs) is written -- no reference
source travels.  Run:  python tests/golden/make_golden.py
The .npz files it writes are committed; tests never import the reference.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("BMC_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))  # unused import in models/submodules.py:4
sys.path.insert(0, REF)
from models.BMCNet import BMCNet, ParallelBlk          # noqa: E402
from models.BMCNet_plain import BMCNet_plain           # noqa: E402
from models import submodules as sm                    # noqa: E402
from dataloader.encodings import events_to_channels    # noqa: E402

torch.set_num_threads(8)


def randomise(model, g, wscale=1.0):
    """Replace the 0.1-scaled init by O(1)-gain weights so every path matters."""
    seen = set()
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if prm.data_ptr() in seen:
                continue
            seen.add(prm.data_ptr())
            if prm.dim() == 4:
                fan_in = prm.shape[1] * prm.shape[2] * prm.shape[3]
                prm.copy_(torch.randn(prm.shape, generator=g) * (wscale / fan_in ** 0.5))
            elif "norm" in name and name.endswith("weight"):
                prm.copy_(1.0 + 0.3 * torch.randn(prm.shape, generator=g))
            else:
                prm.copy_(0.1 * torch.randn(prm.shape, generator=g))


def sd_np(model):
    return {"sd/" + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


def unique_named_grads(model):
    out = {}
    seen = set()
    for name, prm in model.named_parameters():   # named_parameters() de-duplicates shared tensors
        if prm.data_ptr() in seen:
            continue
        seen.add(prm.data_ptr())
        if prm.grad is None:       # parameter not reached by this loss
            print("  (no grad)", name)
            continue
        out["grad/" + name] = prm.grad.detach().numpy().copy()
    return out


def counts(g, shape, lam=0.284):
    return torch.poisson(torch.full(shape, lam), generator=g)


# ---------------------------------------------------------------- events
def gen_events():
    rng = np.random.default_rng(3407)
    cases = {}

    def run(tag, xs, ys, ps, size):
        xt, yt, pt = (torch.tensor(a, dtype=torch.float32) for a in (xs, ys, ps))
        img = events_to_channels(xt, yt, pt, sensor_size=size)
        cases[f"{tag}/xs"] = np.asarray(xs, np.float32)
        cases[f"{tag}/ys"] = np.asarray(ys, np.float32)
        cases[f"{tag}/ps"] = np.asarray(ps, np.float32)
        cases[f"{tag}/size"] = np.asarray(size, np.int64)
        cases[f"{tag}/img"] = img.numpy()
        cases[f"{tag}/xs_after"] = xt.numpy()
        cases[f"{tag}/ys_after"] = yt.numpy()

    # SURVEY appendix A.1 example
    run("tiny", [0, 7, 3, 3, 8, -1, 2.9], [0, 5, 2, 2, 1, 1, 4.2], [1, -1, 1, -1, 1, -1, 1], (6, 8))
    # in-range integer events, duplicates
    n = 2048
    run("nfs_lr", rng.integers(0, 80, n), rng.integers(0, 45, n), rng.choice([-1, 1], n), (45, 80))
    # out-of-range of both polarities, float coordinates, negatives
    n = 4096
    run("oob_float", rng.uniform(-3, 27, n), rng.uniform(-2, 20, n), rng.choice([-1, 1], n), (17, 23))
    # empty
    run("empty", np.zeros(0), np.zeros(0), np.zeros(0), (5, 7))
    # hot pixel: many duplicates on one location
    n = 3000
    run("hot", np.full(n, 3), np.full(n, 2), rng.choice([-1, 1], n), (4, 6))
    # default sensor size, C2 density
    n = 24576
    run("c2_lr", rng.integers(0, 240, n), rng.integers(0, 180, n), rng.choice([-1, 1], n), (180, 240))
    np.savez_compressed(os.path.join(OUT, "events.npz"), **cases)


# ---------------------------------------------------------------- layers
def gen_layers():
    g = torch.Generator().manual_seed(11)
    out = {}
    C, B, H, W = 16, 2, 9, 7

    def tensor(*shape):
        return torch.randn(*shape, generator=g)

    # residual block
    m = sm.ResidualBlock_noBN(C); randomise(m, g)
    x = tensor(B, C, H, W).requires_grad_()
    y = m(x); go = tensor(*y.shape); y.backward(go)
    out.update({"res/" + k: v for k, v in sd_np(m).items()})
    out.update({"res/x": x.detach().numpy(), "res/y": y.detach().numpy(), "res/go": go.numpy(),
                "res/gx": x.grad.numpy()})
    out.update({"res/" + k: v for k, v in unique_named_grads(m).items()})

    # LayerNorm2d
    m = sm.LayerNorm2d(C); randomise(m, g)
    x = tensor(B, C, H, W).requires_grad_()
    y = m(x); go = tensor(*y.shape); y.backward(go)
    out.update({"ln/" + k: v for k, v in sd_np(m).items()})
    out.update({"ln/x": x.detach().numpy(), "ln/y": y.detach().numpy(), "ln/go": go.numpy(),
                "ln/gx": x.grad.numpy()})
    out.update({"ln/" + k: v for k, v in unique_named_grads(m).items()})

    # BIE
    m = sm.BIE(C); randomise(m, g)
    xs = [tensor(B, C, H, W).requires_grad_() for _ in range(3)]
    ys = m(*xs); gos = [tensor(*t.shape) for t in ys]
    torch.autograd.backward(ys, gos)
    out.update({"bie/" + k: v for k, v in sd_np(m).items()})
    for i in range(3):
        out[f"bie/x{i}"] = xs[i].detach().numpy(); out[f"bie/y{i}"] = ys[i].detach().numpy()
        out[f"bie/go{i}"] = gos[i].numpy(); out[f"bie/gx{i}"] = xs[i].grad.numpy()
    out.update({"bie/" + k: v for k, v in unique_named_grads(m).items()})

    # ParallelBlk
    m = ParallelBlk(C); randomise(m, g)
    xs = [tensor(B, C, H, W).requires_grad_() for _ in range(7)]
    ys = m(*xs); gos = [tensor(*t.shape) for t in ys]
    torch.autograd.backward(ys, gos)
    out.update({"pblk/" + k: v for k, v in sd_np(m).items()})
    for i in range(7):
        out[f"pblk/x{i}"] = xs[i].detach().numpy(); out[f"pblk/y{i}"] = ys[i].detach().numpy()
        out[f"pblk/go{i}"] = gos[i].numpy(); out[f"pblk/gx{i}"] = xs[i].grad.numpy()
    out.update({"pblk/" + k: v for k, v in unique_named_grads(m).items()})

    # pixel unshuffle / head pieces
    x = tensor(2, 2, 12, 20)
    out["unshuffle/x"] = x.numpy(); out["unshuffle/y"] = sm.pixel_unshuffle(x, 4).numpy()
    x = tensor(2, 32, 5, 6); f = counts(g, (2, 2, 5, 6))
    out["head/xo"] = x.numpy(); out["head/f2"] = f.numpy()
    out["head/y"] = (torch.nn.functional.pixel_shuffle(x, 4)
                     + torch.nn.functional.interpolate(f, scale_factor=4, mode="bilinear",
                                                       align_corners=False)).numpy()
    np.savez_compressed(os.path.join(OUT, "layers.npz"), **out)


# ---------------------------------------------------------------- full models
def gen_model(tag, cls, n_c, n_b, B, H, W, nwin, seed, plain, wscale=1.0, scale=4):
    g = torch.Generator().manual_seed(seed)
    m = cls(scale, n_c, n_b); randomise(m, g, wscale)
    out = sd_np(m)
    out["meta"] = np.asarray([scale, n_c, n_b, B, H, W, nwin], np.int64)
    frames = counts(g, (B, nwin + 1, 2, H, W))
    gts = counts(g, (B, nwin + 1, 2, scale * H, scale * W))
    out["frames"] = frames.numpy(); out["gts"] = gts.numpy()
    z = lambda c: torch.zeros(B, c, H, W)
    h, hp, hn, pred = z(n_c), z(n_c), z(n_c), z(2 * scale * scale)
    loss = 0
    for i in range(nwin):
        x = frames[:, i:i + 2].transpose(1, 2)            # [B, 2(pol), 2(T), H, W]  (train.py:211)
        if plain:
            h, pred = m(x, h, pred, i == 0)
        else:
            h, hp, hn, pred = m(x, h, hp, hn, pred, i == 0)
        out[f"pred{i}"] = pred.detach().numpy()
        loss = loss + torch.nn.functional.mse_loss(pred, gts[:, i + 1])
    out["h"] = h.detach().numpy()
    if not plain:
        out["hp"] = hp.detach().numpy(); out["hn"] = hn.detach().numpy()
    loss.backward()
    out["loss"] = np.asarray(loss.item(), np.float64)
    out.update(unique_named_grads(m))
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    return m


def gen_pretrained_plain():
    path = os.path.join(REF, "pretrain", "BMCNet_plain_nfs_x4.pth")
    sd = torch.load(path, map_location="cpu")
    m = BMCNet_plain(4, 128, 5)
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(5)
    B, H, W = 1, 45, 80
    frames = counts(g, (B, 3, 2, H, W), lam=0.284)
    h, pred = torch.zeros(B, 128, H, W), torch.zeros(B, 32, H, W)
    out = {"frames": frames.numpy(), "keys": np.asarray(sorted(sd.keys()))}
    with torch.no_grad():
        for i in range(2):
            h, pred = m(frames[:, i:i + 2].transpose(1, 2), h, pred, i == 0)
            out[f"pred{i}"] = pred.numpy()
    out["h_mean_abs"] = np.asarray(h.abs().mean().item())
    np.savez_compressed(os.path.join(OUT, "plain_pretrained.npz"), **out)


def gen_adam():
    g = torch.Generator().manual_seed(9)
    ws = [torch.randn(5, 3, generator=g), torch.randn(7, generator=g)]
    prm = [torch.nn.Parameter(w.clone()) for w in ws]
    opt = torch.optim.Adam(prm, lr=1e-4, weight_decay=1e-5, amsgrad=True)
    out = {"w0": ws[0].numpy(), "w1": ws[1].numpy()}
    for step in range(3):
        gs = [torch.randn(5, 3, generator=g), torch.randn(7, generator=g)]
        for q, gr in zip(prm, gs):
            q.grad = gr.clone()
        opt.step()
        for i in range(2):
            out[f"g{step}_{i}"] = gs[i].numpy(); out[f"w_after{step}_{i}"] = prm[i].detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "adam.npz"), **out)


# ---------------------------------------------------------------- raw dataset columns + flip augmentation
def gen_raw_events():
    """Runs the reference's per-item CPU chain on raw int16/float64 columns: H5Dataset.augment_event
    (dataloader/h5dataset.py:559-578) -> BaseDataset.event_formatting (base_dataset.py:24-31) ->
    H5Dataset.create_cnt_encoding (h5dataset.py:518-526) -> events_to_channels.  h5py / cv2 are stubbed: they are
    imported by dataloader/h5dataset.py but not touched by these methods."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    for name in ("h5py", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    plt.style.use = lambda *a, **k: None          # 'seaborn-whitegrid' no longer exists (h5dataset.py:15)
    from dataloader.h5dataset import H5Dataset
    rng = np.random.default_rng(77)
    out = {}
    fake = types.SimpleNamespace(config={"data_augment": {"enabled": True,
                                                         "augment": ["Horizontal", "Vertical", "Polarity"],
                                                         "augment_prob": [0.5, 0.5, 0.5]}})
    cases = [("lr", (45, 80), 2048), ("odd", (17, 23), 700), ("hr", (180, 320), 32768)]
    seeds = [0, 1, 2, 3, 4, 5, 7, 22, 2 ** 31 + 5]      # 0..22 cover all 8 flip combinations
    idx = 0
    for tag, (H, W), n in cases:
        for seed in (seeds if n < 10000 else seeds[:2]):
            xs = rng.integers(-2, W + 2, n).astype(np.int16)       # a few out-of-range coordinates
            ys = rng.integers(-2, H + 2, n).astype(np.int16)
            ts = np.sort(rng.uniform(0, 1, n))
            ps = rng.choice([-1.0, 1.0], n).astype(np.float64)
            ev = np.concatenate((xs[None], ys[None], ts[None], ps[None]), axis=0)          # get_events layout (:407-414)
            aug = H5Dataset.augment_event(fake, ev, (H, W), seed)
            fm = H5Dataset.event_formatting(aug)
            cnt = H5Dataset.create_cnt_encoding(fake, fm, (H, W))
            k = f"c{idx}"
            out[k + "/xs"], out[k + "/ys"], out[k + "/ps"] = xs, ys, ps
            out[k + "/size"], out[k + "/seed"] = np.asarray((H, W)), np.asarray(seed, np.int64)
            out[k + "/cnt"] = cnt.numpy()
            fl = 0      # which flips happened, recovered from the augmented arrays (data, not code)
            if not np.array_equal(aug[0], xs.astype(np.float64)): fl |= 1
            if not np.array_equal(aug[1], ys.astype(np.float64)): fl |= 2
            if not np.array_equal(aug[3], ps): fl |= 4
            out[k + "/flags"] = np.asarray(fl, np.int64)
            idx += 1
    out["n"] = np.asarray(idx)
    np.savez_compressed(os.path.join(OUT, "events_raw.npz"), **out)


# ---------------------------------------------------------------- temporal-bilinear voxel grid
def gen_voxel():
    from dataloader.encodings import events_to_voxel
    rng = np.random.default_rng(5)
    out = {}
    cases = [("a", (17, 23), 900, 5), ("b", (45, 80), 2048, 3), ("c", (8, 9), 0, 4), ("d", (12, 10), 300, 1), ("e", (20, 31), 1500, 2)]
    for tag, (H, W), n, bins in cases:
        xs = rng.uniform(-1.5, W + 1.5, n).astype(np.float32)
        ys = rng.uniform(-1.5, H + 1.5, n).astype(np.float32)
        ts = np.sort(rng.uniform(0, 1, n)).astype(np.float32)
        ps = rng.choice([-1.0, 1.0], n).astype(np.float32)
        xt, yt, tt, pt = (torch.tensor(a) for a in (xs, ys, ts, ps))
        torch.set_num_threads(1)                     # sequential index_put_: a fixed summation order for the golden
        vox = events_to_voxel(xt, yt, tt, pt, bins, sensor_size=(H, W))
        torch.set_num_threads(8)
        for k, v in (("xs", xs), ("ys", ys), ("ts", ts), ("ps", ps), ("vox", vox.numpy()), ("xs_after", xt.numpy()),
                     ("ys_after", yt.numpy()), ("meta", np.asarray([H, W, bins]))):
            out[f"{tag}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "voxel.npz"), **out)


# ---------------------------------------------------------------- event stack (temporal bins, no polarity split)
def gen_stack():
    from dataloader.encodings import events_to_stack_no_polarity
    rng = np.random.default_rng(9)
    out = {}
    # (tag, sensor, events, bins, timestamp style)
    cases = [("a", (17, 23), 900, 5, "uniform"), ("b", (45, 80), 2048, 3, "uniform"), ("c", (8, 9), 3, 4, "uniform"),
             ("d", (12, 10), 300, 1, "uniform"), ("e", (20, 31), 1500, 7, "ties"), ("f", (9, 11), 64, 4, "zeros"),
             ("g", (16, 16), 700, 6, "grid")]
    for tag, (H, W), n, bins, style in cases:
        xs = rng.uniform(-1.5, W + 1.5, n).astype(np.float32)
        ys = rng.uniform(-1.5, H + 1.5, n).astype(np.float32)
        if style == "uniform":
            ts = np.sort(rng.uniform(0, 1, n))
        elif style == "ties":                      # many equal timestamps: exercises the first-equal-probe returns
            ts = np.sort(rng.integers(0, 40, n) / 40.0)
        elif style == "grid":                      # timestamps that hit bin boundaries exactly
            ts = np.sort(rng.integers(0, 13, n) / 12.0)
        else:
            ts = np.zeros(n)
        ts = ts.astype(np.float32)
        if style != "zeros":                       # event_formatting's normalisation (dataloader/base_dataset.py:30)
            ts = ((ts - ts[0]) / (ts[-1] - ts[0] + np.float32(1e-6))).astype(np.float32)
        ps = rng.choice([-1.0, 1.0], n).astype(np.float32)
        xt, yt, tt, pt = (torch.tensor(a) for a in (xs, ys, ts, ps))
        st = events_to_stack_no_polarity(xt, yt, tt, pt, bins, sensor_size=(H, W))
        for k, v in (("xs", xs), ("ys", ys), ("ts", ts), ("ps", ps), ("stack", st.numpy()), ("xs_after", xt.numpy()),
                     ("ys_after", yt.numpy()), ("ps_after", pt.numpy()), ("meta", np.asarray([H, W, bins]))):
            out[f"{tag}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "stack.npz"), **out)


if __name__ == "__main__":
    if os.environ.get("BMC_GOLDEN_ONLY") == "stack":
        gen_stack()
        sys.exit(0)
    if os.environ.get("BMC_GOLDEN_ONLY") == "voxel":
        gen_voxel()
        sys.exit(0)
    if os.environ.get("BMC_GOLDEN_ONLY") == "raw":
        gen_raw_events()
        sys.exit(0)
    gen_events()
    gen_raw_events()
    gen_voxel()
    gen_stack()
    gen_layers()
    gen_model("bmcnet_nc16", BMCNet, 16, 2, 2, 10, 12, 3, seed=21, plain=False, wscale=0.6)
    gen_model("plain_nc16", BMCNet_plain, 16, 2, 2, 9, 7, 3, seed=22, plain=True)
    gen_model("bmcnet_nc32", BMCNet, 32, 1, 1, 20, 35, 2, seed=23, plain=False)
    gen_pretrained_plain()
    gen_adam()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
