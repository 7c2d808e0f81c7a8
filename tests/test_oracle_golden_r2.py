"""CPU tests: the oracle against the round-2 golden vectors (generated from the reference by
tests/golden/make_golden_r2.py): collate layout, bicubic resize, the size-mismatch branch of the training loop,
events_to_stack_polarity, events_to_mask, and the pretrained weight arrays."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import bmc_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import ref_stubs  # noqa: E402  (synthetic recording generator shared with the golden script; reads nothing of the reference)


def rel(a, b):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def collate_case():
    """-> (golden dict, per-recording column dicts); the columns are regenerated from the stored seeds and checked
    against the stored hash, so the test fails loudly (not silently) if numpy's generator ever changed."""
    z = np.load(os.path.join(GOLDEN, "collate.npz"))
    files = [ref_stubs.synth_nfs_file(int(s)) for s in z["file_seeds"]]
    hs = hashlib.sha256()
    for f in files:
        for k in sorted(f):
            if k != "attrs":
                hs.update(np.ascontiguousarray(f[k]).tobytes())
    assert hs.hexdigest() == str(z["columns_sha256"]), "synthetic recording columns differ from the ones the golden was made with"
    return z, files


def test_collate_layout_and_sequence_encoding():
    """a-2: SequenceDataset -> custom_collate -> concat_dict of the reference (dataloader/h5dataloader.py:213-250) on the
    stub recording == oracle raw-column encoder per frame + the [B,L,...] -> sliding-window slicing."""
    z, files = collate_case()
    B, L = z["lr_ranges"].shape[:2]
    seqn = int(z["seqn"])
    assert int(z["n_windows"]) == L - seqn + 1 == 8 and list(z["keys"]) == ["gt_cnt", "inp_cnt"]
    Hl, Wl = (int(v) for v in z["inp_res"])
    Hg, Wg = (int(v) for v in z["gt_res"])
    assert (Hl, Wl, Hg, Wg) == (45, 80, 180, 320)
    inp = np.zeros((B, L, 2, Hl, Wl), np.float32)
    gt = np.zeros((B, L, 2, Hg, Wg), np.float32)
    for b, f in enumerate(files):
        flags = O.augment_flags(int(z["item_seeds"][b]), tuple(str(s) for s in z["augment"]), tuple(float(p) for p in z["augment_prob"]))
        for j in range(L):
            i0, i1 = z["lr_ranges"][b, j]
            inp[b, j] = O.encode_raw_frame_np(f["down8_events/xs"][i0:i1], f["down8_events/ys"][i0:i1],
                                              f["down8_events/ps"][i0:i1], flags, (Hl, Wl))
            g0, g1 = z["gt_ranges"][b, j]
            gt[b, j] = O.encode_raw_frame_np(f["down2_events/xs"][g0:g1], f["down2_events/ys"][g0:g1],
                                             f["down2_events/ps"][g0:g1], flags, (Hg, Wg))
    wins = O.collate_windows(inp, gt, seqn)
    assert len(wins) == int(z["n_windows"])
    for i, w in enumerate(wins):
        for k in ("inp_cnt", "gt_cnt"):
            ref = z["w%d/%s" % (i, k)].astype(np.float32)
            assert w[k].shape == ref.shape
            assert np.array_equal(w[k], ref), (i, k)


@pytest.mark.parametrize("tag", ["ez", "odd", "up", "down", "same"])
def test_bicubic_resize(tag):
    z = np.load(os.path.join(GOLDEN, "bicubic.npz"))
    x = torch.tensor(z[tag + "/x"]).requires_grad_()
    y = O.bicubic_resize(x, z[tag + "/y"].shape[-2:])
    y.backward(torch.tensor(z[tag + "/go"]))
    # ATen evaluates the source coordinate in float32 with a fused multiply-add (one rounding), so does the restatement:
    # what is left is the rounding of the tap weights and of the 16-term sums
    assert rel(y.detach(), z[tag + "/y"]) < 1e-6
    assert rel(x.grad, z[tag + "/gx"]) < 1e-6
    # in float64 the same code is the exact operator: compare against F.interpolate itself
    xd = torch.tensor(z[tag + "/x"], dtype=torch.float64)
    assert rel(O.bicubic_resize(xd, y.shape[-2:]), F.interpolate(xd, size=y.shape[-2:], mode="bicubic", align_corners=False)) < 1e-13


def test_bptt_with_bicubic_resize_branch():
    z = np.load(os.path.join(GOLDEN, "bmcnet_resize.npz"))
    scale, n_c, n_b, B, H, W, nwin, gh, gw = (int(v) for v in z["meta"])
    params, seen = {}, {}
    for k in z.files:
        if k.startswith("sd/"):
            params[k[3:]] = torch.tensor(z[k])
    # rebuild aliasing: identical arrays under alias keys are the same parameter
    canon = {}
    for k, v in list(params.items()):
        key = (v.shape, v.numpy().tobytes())
        if key in canon:
            params[k] = canon[key]
        else:
            canon[key] = v.requires_grad_()
    frames, gts = torch.tensor(z["frames"]), torch.tensor(z["gts"])
    inp = [frames[:, i:i + 2].transpose(1, 2) for i in range(nwin)]
    loss, preds, _ = O.bptt_loss(params, inp, [gts[:, i + 1] for i in range(nwin)], n_c, scale)
    assert abs(loss.item() - float(z["loss"])) < 1e-5 * abs(float(z["loss"]))
    for i in range(nwin):
        assert rel(preds[i].detach(), z["pred%d" % i]) < 1e-6
        assert rel(O.bicubic_resize(preds[i].detach(), (gh, gw)), z["spred%d" % i]) < 2e-6
    loss.backward()
    n = 0
    for k in z.files:
        if k.startswith("grad/"):
            assert rel(params[k[5:]].grad, z[k]) < 2e-5, k
            n += 1
    assert n >= 40


@pytest.mark.parametrize("tag", list("abcdefg"))
def test_events_to_stack_polarity_oracle(tag):
    z = np.load(os.path.join(GOLDEN, "stack_polarity.npz"))
    H, W, bins = (int(v) for v in z[tag + "/meta"])
    st, xa, ya = O.events_to_stack_polarity_np(z[tag + "/xs"], z[tag + "/ys"], z[tag + "/ts"], z[tag + "/ps"], bins, (H, W))
    assert st.shape == z[tag + "/stack"].shape
    assert np.array_equal(st, z[tag + "/stack"])
    assert np.array_equal(xa, z[tag + "/xs_after"]) and np.array_equal(ya, z[tag + "/ys_after"])
    assert np.array_equal(z[tag + "/ps"], z[tag + "/ps_after"])        # the caller's ps is never touched


@pytest.mark.parametrize("tag", list("abcd"))
def test_events_to_mask_oracle(tag):
    z = np.load(os.path.join(GOLDEN, "mask.npz"))
    H, W = (int(v) for v in z[tag + "/meta"])
    mk, xa, ya, pa = O.events_to_mask_np(z[tag + "/xs"], z[tag + "/ys"], z[tag + "/ps"], (H, W))
    assert np.array_equal(mk, z[tag + "/mask"])
    assert np.array_equal(xa, z[tag + "/xs_after"]) and np.array_equal(ya, z[tag + "/ys_after"])
    assert np.array_equal(pa, z[tag + "/ps_after"])


def test_pretrained_weight_arrays_reproduce_reference_outputs():
    """The stored tensors of pretrain/BMCNet_plain_nfs_x4.pth, pushed through the oracle, give the reference's own
    stored predictions (plain_pretrained.npz) -- so the arrays ARE the checkpoint, and the GPU test can use them."""
    zw = np.load(os.path.join(GOLDEN, "plain_pretrained_weights.npz"))
    z = np.load(os.path.join(GOLDEN, "plain_pretrained.npz"))
    uniq = {k[2:]: torch.tensor(zw[k]) for k in zw.files}
    assert len(uniq) == 24 and sum(v.numel() for v in uniq.values()) == 1003296
    params = O.expand_aliases(uniq, [str(k) for k in z["keys"]])
    assert len(params) == 120
    frames = torch.tensor(z["frames"])
    h, pred = torch.zeros(1, 128, 45, 80), torch.zeros(1, 32, 45, 80)
    with torch.no_grad():
        for i in range(2):
            h, pred = O.plain_forward(params, frames[:, i:i + 2].transpose(1, 2), h, pred, i == 0)
            assert rel(pred, z["pred%d" % i]) < 2e-6


# ------------------------------------------------------------------ round 3: the inference loop (SEQN = 3, both metrics)
def test_inference_loop_seqn3_golden():
    """infer_seqn3.npz = the reference's model class driven by the body of infer_BMCNet.py:44-86 (make_golden_r3.py)."""
    z = np.load(os.path.join(GOLDEN, "infer_seqn3.npz"))
    scale, n_c, n_b, B, H, W, seqn, nwin, gh, gw = (int(v) for v in z["meta"])
    params = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd/")}
    res = O.infer_windows(params, torch.tensor(z["frames"]), torch.tensor(z["gts"]), n_c, scale, seqn)
    assert len(res) == nwin
    for i, (pred, esr, base) in enumerate(res):
        ref = torch.tensor(z["pred%d" % i])
        assert float((pred - ref).norm() / ref.norm()) < 2e-6
        assert abs(float(esr) - float(z["esr_mse%d" % i])) < 2e-6 * float(z["esr_mse%d" % i])
        assert abs(float(base) - float(z["bicubic_mse%d" % i])) < 2e-6 * float(z["bicubic_mse%d" % i])


def test_torch_encodings_oracle_matches_reference_golden():
    """events_to_image_torch (bilinear, padding / clipping variants, and interpolation=None) and events_to_voxel_torch of the
    reference (dataloader/encodings.py:16-73, 100-148; golden from tests/golden/make_golden_r3b.py): the numpy restatements
    reproduce the images, the voxel grids AND the mutated inputs bit for bit."""
    import numpy as np
    from oracle import bmc_oracle as O
    g = np.load(os.path.join(GOLDEN, "encodings_torch.npz"))
    H, W = 13, 17
    for pad, clip in ((True, True), (True, False), (False, True)):
        a, b, c = g["xs"].copy(), g["ys"].copy(), g["ps"].copy()
        img = O.events_to_image_torch_np(a, b, c, (H, W), clip_out_of_range=clip, interpolation="bilinear", padding=pad)
        tag = "img_pad%d_clip%d" % (pad, clip)
        assert np.array_equal(img, g[tag]), tag
        assert np.array_equal(a, g[tag + "_xs"]) and np.array_equal(b, g[tag + "_ys"]) and np.array_equal(c, g[tag + "_ps"])
    a, b, c = g["xi"].copy(), g["yi"].copy(), g["ps"].copy()
    assert np.array_equal(O.events_to_image_torch_np(a, b, c, (H, W)), g["imgn"])
    assert np.array_equal(a, g["imgn_xs"]) and np.array_equal(b, g["imgn_ys"]) and np.array_equal(c, g["imgn_ps"])
    a, b = g["xi"].copy(), g["yi"].copy()
    assert np.array_equal(O.events_to_voxel_torch_np(a, b, g["ts"], g["ps"].copy(), 5, (H, W)), g["vox"])
    assert np.array_equal(a, g["vox_xs"]) and np.array_equal(b, g["vox_ys"])
    assert np.array_equal(O.events_to_voxel_torch_np(g["xi"].copy(), g["yi"].copy(), np.zeros_like(g["ts"]), g["ps"].copy(), 5, (H, W)),
                          g["vox_zero_ts"])
    assert np.array_equal(O.events_to_voxel_torch_np(g["xi"][:3].copy(), g["yi"][:3].copy(), g["ts"][:3], g["ps"][:3].copy(), 5, (H, W)),
                          g["vox_three"])
