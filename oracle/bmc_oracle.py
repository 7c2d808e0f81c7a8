"""CPU oracle for the BMCNet bilateral event-SR hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``bmcnet-esr_amd/``)
never imports anything from ``oracle/`` and has no CPU fallback.

It is a functional restatement (plain functions over a ``{state_dict key:
tensor}`` mapping, PyTorch-CPU ops + autograd for the backward) of what the
reference computes on the hot path; nothing here is copied from the reference.
Each function cites the reference lines it follows (paths relative to the
upstream repo root).

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned against outputs of the
reference itself, generated in the build container by
``tests/golden/make_golden.py`` (which imports the reference) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every one of
them.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# event -> count image (dataloader/encodings.py:241-269, 290-305)
# --------------------------------------------------------------------------
def events_to_channels_np(xs, ys, ps, sensor_size=(180, 240)):
    """Two-channel event count image, numpy restatement.

    Follows dataloader/encodings.py:290-305 (events_to_channels) calling
    dataloader/encodings.py:241-269 (events_to_image) twice, including the
    side effects of the first call on the caller's ``xs``/``ys``:

    * first call (positive channel): events outside the sensor get
      x = y = 0 written back into the caller's arrays and their weight zeroed,
      so they contribute nothing;
    * second call (negative channel): the same events now sit at (0, 0), are
      no longer masked, and every formerly-out-of-range *negative* event adds
      p*p = 1 at row H-1 (the vertical flip of y = 0), column 0.

    Returns (img[2,H,W] float32, xs_after, ys_after); inputs are not modified.
    """
    H, W = int(sensor_size[0]), int(sensor_size[1])
    xs = np.array(xs, dtype=np.float32, copy=True)
    ys = np.array(ys, dtype=np.float32, copy=True)
    ps = np.asarray(ps, dtype=np.float32)
    out = np.zeros((2, H, W), dtype=np.float32)
    for ch in range(2):
        wsel = np.where(ps > 0, ps, 0) if ch == 0 else np.where(ps < 0, ps, 0)
        wgt = (ps * wsel).astype(np.float32)
        oob = (xs >= W) | (xs < 0) | (ys >= H) | (ys < 0)
        xs[oob] = 0
        ys[oob] = 0
        wgt = np.where(oob, np.float32(0), wgt)
        xi = xs.astype(np.int64)          # truncation toward zero, like .long()
        yi = H - ys.astype(np.int64) - 1  # vertical flip
        np.add.at(out[ch], (yi, xi), wgt)
    return out, xs, ys


def events_to_voxel_np(xs, ys, ts, ps, num_bins, sensor_size=(180, 240)):
    """Temporal-bilinear voxel grid, numpy restatement of dataloader/encodings.py:272-287 (events_to_voxel calling
    events_to_image :241-269 once per bin).  The first bin's call resets out-of-range coordinates in the caller's
    arrays (and zeroes only ITS temporary weights), so from the second bin on those events are unmasked and land on
    [H-1, 0].  Accumulation in float32, in event order.  Returns (voxel[bins,H,W], xs_after, ys_after)."""
    H, W = int(sensor_size[0]), int(sensor_size[1])
    xs = np.array(xs, dtype=np.float32, copy=True)
    ys = np.array(ys, dtype=np.float32, copy=True)
    ps = np.asarray(ps, dtype=np.float32)
    t = (np.asarray(ts, dtype=np.float32) * np.float32(num_bins - 1)).astype(np.float32)
    out = np.zeros((num_bins, H, W), dtype=np.float32)
    for b in range(num_bins):
        wgt = (ps * np.maximum(np.float32(0), np.float32(1.0) - np.abs(t - np.float32(b)))).astype(np.float32)
        oob = (xs >= W) | (xs < 0) | (ys >= H) | (ys < 0)
        xs[oob] = 0
        ys[oob] = 0
        wgt = np.where(oob, np.float32(0), wgt)
        np.add.at(out[b], (H - ys.astype(np.int64) - 1, xs.astype(np.int64)), wgt)
    return out, xs, ys


def binary_search_f32(t, l, r, x, side="left"):
    """The reference's hand-written search on a sorted float32 array (dataloader/encodings.py:75-97), quirks included:
    it returns the first probe that EQUALS x (either end or the midpoint), not the leftmost / rightmost equal element,
    and with side != 'left' it returns r (one before the insertion point)."""
    while l <= r:
        if t[l] == x:
            return l
        if t[r] == x:
            return r
        mid = l + (r - l) // 2
        if t[mid] == x:
            return mid
        elif t[mid] < x:
            l = mid + 1
        else:
            r = mid - 1
    return l if side == "left" else r


def events_to_stack_no_polarity_np(xs, ys, ts, ps, B, sensor_size=(180, 240)):
    """Event stack without polarity split, numpy restatement of dataloader/encodings.py:202-238: B temporal bins of
    equal width over [ts[0], ts[-1] + 1e-6), each the signed sum of the polarities of its events per pixel
    (events_to_image_torch :16-73 with clip_out_of_range=False, interpolation=None: NO vertical flip).  The bin's event
    range comes from the quirky search above on float32 bounds computed exactly as the reference does
    (ts[0] + delta_t * bi, then + delta_t), so neighbouring bins can overlap by an event.  events_to_image_torch works
    on VIEWS of the caller's arrays: out-of-range events get xs = ys = ps = 0 in place (so an event masked in one bin is
    dead in every later bin).  Returns (stack[B,H,W], xs_after, ys_after, ps_after)."""
    H, W = int(sensor_size[0]), int(sensor_size[1])
    xs = np.array(xs, dtype=np.float32, copy=True)
    ys = np.array(ys, dtype=np.float32, copy=True)
    ps = np.array(ps, dtype=np.float32, copy=True)
    ts = np.asarray(ts, dtype=np.float32)
    out = np.zeros((B, H, W), dtype=np.float32)
    n = len(ts)
    if n <= 3 or np.float32(ts.sum(dtype=np.float32)) == 0:        # :219-220
        return out, xs, ys, ps
    dt = np.float32(np.float32(ts[-1] - ts[0]) + np.float32(1e-6))
    delta_t = np.float32(dt / np.float32(B))
    for bi in range(B):
        tstart = np.float32(ts[0] + np.float32(delta_t * np.float32(bi)))
        tend = np.float32(tstart + delta_t)
        beg = binary_search_f32(ts, 0, n - 1, tstart)
        end = binary_search_f32(ts, 0, n - 1, tend, side="right") + 1
        sl = slice(beg, end)
        x, y, p = xs[sl], ys[sl], ps[sl]                          # views, as in the reference
        oob = (x >= W) | (x < 0) | (y >= H) | (y < 0)
        x[oob] = 0
        y[oob] = 0
        p[oob] = 0
        np.add.at(out[bi], (y.astype(np.int64), x.astype(np.int64)), p)
    return out, xs, ys, ps


def events_to_stack_polarity_np(xs, ys, ts, ps, B, sensor_size=(180, 240)):
    """Event stack with polarity split, numpy restatement of dataloader/encodings.py:151-199: per temporal bin (same
    float32 bounds and quirky search as events_to_stack_no_polarity) two count images, [2,B,H,W] = (positives, negatives),
    each events_to_image_torch(xs[beg:end], ys[beg:end], ps[beg:end] * mask, clip_out_of_range=False): no vertical flip,
    weights p*p.  The coordinate VIEWS are reset in place by the first (positive) call of the first bin that covers an
    event; its weights are a temporary, so the caller's ps stays intact -- every later call sees the event in range
    at (0, 0): an out-of-range NEGATIVE event counts at [0, 0] of the negative image, and an event that a later,
    overlapping bin covers again counts there whatever its sign.  Early return (<= 3 events or all-zero timestamps) is
    [B,H,W] zeros, as in the reference (:165-166).  Returns (stack, xs_after, ys_after)."""
    H, W = int(sensor_size[0]), int(sensor_size[1])
    xs = np.array(xs, dtype=np.float32, copy=True)
    ys = np.array(ys, dtype=np.float32, copy=True)
    ps = np.asarray(ps, dtype=np.float32)
    ts = np.asarray(ts, dtype=np.float32)
    n = len(ts)
    if n <= 3 or np.float32(ts.sum(dtype=np.float32)) == 0:
        return np.zeros((B, H, W), dtype=np.float32), xs, ys
    out = np.zeros((2, B, H, W), dtype=np.float32)
    dt = np.float32(np.float32(ts[-1] - ts[0]) + np.float32(1e-6))
    delta_t = np.float32(dt / np.float32(B))
    for bi in range(B):
        tstart = np.float32(ts[0] + np.float32(delta_t * np.float32(bi)))
        tend = np.float32(tstart + delta_t)
        beg = binary_search_f32(ts, 0, n - 1, tstart)
        end = binary_search_f32(ts, 0, n - 1, tend, side="right") + 1
        sl = slice(beg, end)
        x, y, p = xs[sl], ys[sl], ps[sl]
        for ch in range(2):
            wgt = (p * (np.where(p < 0, 0, p) if ch == 0 else np.where(p > 0, 0, p))).astype(np.float32)
            oob = (x >= W) | (x < 0) | (y >= H) | (y < 0)
            x[oob] = 0
            y[oob] = 0
            wgt[oob] = 0
            np.add.at(out[ch, bi], (y.astype(np.int64), x.astype(np.int64)), wgt)
    return out, xs, ys


def events_to_mask_np(xs, ys, ps, sensor_size=(180, 240)):
    """Binary event mask, numpy restatement of dataloader/encodings.py:308-332: out-of-range events get
    xs = ys = ps = 0 IN PLACE (all three belong to the caller), then mask[(long) y, (long) x] = |p| with
    index_put_(accumulate=False): no vertical flip, and for duplicate pixels the LAST event in order wins (the
    sequential semantics of index_put_; a zeroed out-of-range event can therefore clear [0, 0]).
    Returns (mask[H,W], xs_after, ys_after, ps_after)."""
    H, W = int(sensor_size[0]), int(sensor_size[1])
    xs = np.array(xs, dtype=np.float32, copy=True)
    ys = np.array(ys, dtype=np.float32, copy=True)
    ps = np.array(ps, dtype=np.float32, copy=True)
    oob = (xs >= W) | (xs < 0) | (ys >= H) | (ys < 0)
    xs[oob] = 0
    ys[oob] = 0
    ps[oob] = 0
    mask = np.zeros((H, W), dtype=np.float32)
    xi, yi = xs.astype(np.int64), ys.astype(np.int64)
    for e in range(len(xs)):
        mask[yi[e], xi[e]] = abs(ps[e])
    return mask, xs, ys, ps


def events_to_image_torch_np(xs, ys, ps, sensor_size=(180, 240), clip_out_of_range=True, interpolation=None, padding=True):
    """numpy restatement of events_to_image_torch (dataloader/encodings.py:16-73) with its CPU semantics: float32 arithmetic in
    the reference's operation order, index_put_(accumulate=True) = one pass per corner, events in order within a pass.
    MUTATES xs, ys, ps like the reference (:33-38: out-of-range events are reset to (0, 0) with weight 0).
    xs / ys: float32 (sub-pixel positions) -- or integer arrays for the interpolation=None branch."""
    H, W = sensor_size
    bad = (xs >= W) | (xs < 0) | (ys >= H) | (ys < 0)                       # :33-35
    xs[bad] = 0; ys[bad] = 0; ps[bad] = 0                                    # :36-38
    bil = interpolation == "bilinear"
    ih, iw = (H + 1, W + 1) if (bil and padding) else (H, W)                 # :42-45
    mask = np.ones(xs.shape, np.float32)
    if clip_out_of_range:                                                    # :48-53
        clipx = iw if (interpolation is None and padding is False) else iw - 1
        clipy = ih if (interpolation is None and padding is False) else ih - 1
        mask = np.where(xs >= clipx, np.float32(0), np.float32(1)) * np.where(ys >= clipy, np.float32(0), np.float32(1))
    img = np.zeros((ih, iw), np.float32)
    if bil and not np.issubdtype(xs.dtype, np.integer):                      # :56-64
        xf, yf = xs.astype(np.float32), ys.astype(np.float32)
        pxs, pys = np.floor(xf), np.floor(yf)
        dxs, dys = (xf - pxs).astype(np.float32), (yf - pys).astype(np.float32)
        ix, iy = (pxs * mask).astype(np.int64), (pys * mask).astype(np.int64)
        w = (ps.reshape(-1).astype(np.float32) * mask).astype(np.float32)
        one = np.float32(1.0)
        # interpolate_to_image (:6-13): four index_put_ passes, each adding the events in order
        for (oy, ox, wt) in ((0, 0, w * (one - dxs) * (one - dys)), (0, 1, w * dxs * (one - dys)),
                             (1, 0, w * (one - dxs) * dys), (1, 1, w * dxs * dys)):
            wt = wt.astype(np.float32)
            for k in range(len(wt)):
                img[iy[k] + oy, ix[k] + ox] += wt[k]
    else:                                                                    # :65-70
        ix, iy = xs.astype(np.int64), ys.astype(np.int64)
        pw = ps.astype(np.float32)
        for k in range(len(pw)):
            img[iy[k], ix[k]] += pw[k]
    return img


def events_to_voxel_torch_np(xs, ys, ts, ps, B, sensor_size=(180, 240)):
    """numpy restatement of events_to_voxel_torch (dataloader/encodings.py:100-148), temporal_bilinear=True branch, float32:
    bin b = events_to_image_torch(xs, ys, ps * max(0, 1 - |t_norm - b|), clip_out_of_range=False), t_norm =
    (ts - ts[0]) / (ts[-1] - ts[0] + 1e-6) * (B - 1); all-zero timestamps or <= 3 events give zeros (:121-122).  MUTATES xs, ys
    (the first bin's call resets out-of-range coordinates, :36-37)."""
    H, W = sensor_size
    if float(ts.sum()) == 0 or len(ts) <= 3:
        return np.zeros((B, H, W), np.float32)
    ts = ts.astype(np.float32)
    dt = np.float32(ts[-1] - ts[0]) + np.float32(1e-6)
    t_norm = ((ts - ts[0]) / dt * np.float32(B - 1)).astype(np.float32)
    out = []
    for bi in range(B):
        wts = (ps.astype(np.float32) * np.maximum(np.float32(0), np.float32(1.0) - np.abs(t_norm - np.float32(bi)))).astype(np.float32)
        out.append(events_to_image_torch_np(xs, ys, wts, sensor_size, clip_out_of_range=False))
    return np.stack(out)


def collate_windows(frames_inp, frames_gt, seqn=2):
    """The batch layout the trainer iterates over -- HDF5DataLoaderSequence.custom_collate + concat_dict
    (dataloader/h5dataloader.py:213-250): per time step the items of the batch are stacked on a new dim 0, then every
    run of `seqn` consecutive time steps is stacked on dim 1.  frames_* [B,L,2,H,W] -> list of L-seqn+1 dicts
    {'inp_cnt': [B,seqn,2,H,W], 'gt_cnt': [B,seqn,2,sH,sW]}."""
    L = frames_inp.shape[1]
    return [{"inp_cnt": frames_inp[:, i:i + seqn], "gt_cnt": frames_gt[:, i:i + seqn]} for i in range(L - seqn + 1)]


def encode_raw_frame_np(xs_i16, ys_i16, ps_f64, flags, sensor_size):
    """One dataset item's count image from raw HDF5 columns: get_events (dataloader/h5dataset.py:407-414, the
    int16/float64 columns are concatenated into ONE float64 array), augment_event (:559-578, flips in float64),
    event_formatting (dataloader/base_dataset.py:24-31, float32 cast), then events_to_channels.
    flags: bit0 horizontal, bit1 vertical, bit2 polarity."""
    H, W = int(sensor_size[0]), int(sensor_size[1])
    xs = np.asarray(xs_i16, np.int16).astype(np.float64)
    ys = np.asarray(ys_i16, np.int16).astype(np.float64)
    ps = np.asarray(ps_f64, np.float64).copy()
    if flags & 1:
        xs = W - 1 - xs
    if flags & 2:
        ys = H - 1 - ys
    if flags & 4:
        ps = ps * -1
    img, _, _ = events_to_channels_np(xs.astype(np.float32), ys.astype(np.float32), ps.astype(np.float32), (H, W))
    return img


def augment_flags(seed, mechanisms=("Horizontal", "Vertical", "Polarity"), probs=(0.5, 0.5, 0.5)):
    """Which flips H5Dataset.augment_event applies for a sample seed (dataloader/h5dataset.py:559-578)."""
    import random
    flags = 0
    for i, mech in enumerate(mechanisms):
        bit, s = {"Horizontal": (1, seed), "Vertical": (2, seed + 1), "Polarity": (4, seed + 2)}[mech]
        random.seed(s)
        if random.random() < probs[i]:
            flags |= bit
    return flags


# --------------------------------------------------------------------------
# building blocks (models/submodules.py)
# --------------------------------------------------------------------------
# --------------------------------------------------------------------------
# operand rounding: the contract of the kernels' BMC_MATH_BF16 mode (include/bmc_hip.h), BASELINE configs[3]
# --------------------------------------------------------------------------
# With operand_rounding("bf16") active, EVERY matrix contraction of the path -- each 3x3 / 1x1 convolution and each bmm,
# in its forward, its data gradient and its weight gradient -- rounds BOTH of its operands to bf16 (round to nearest even
# of the float32 value) and accumulates exactly (here: in the tensors' own dtype, float64 in the tests).  Everything else
# (bias / residual adds, ReLU, LayerNorm, softmax, the attention scale, pixel shuffles, resizes, the loss) stays unrounded,
# and so do the stored activations: only what enters a matrix product is rounded, each time it enters one.  That is NOT
# the autograd derivative of a forward with rounded operands (which would contract the unrounded upstream gradient): the
# backward contractions are specified here, as the kernels compute them -- gradients are operands too.
_OPERAND_ROUND = None


class operand_rounding:
    """Context manager: `with operand_rounding("bf16"): ...` (None = the reference's plain float arithmetic)."""

    def __init__(self, mode):
        if mode not in (None, "bf16"):
            raise ValueError("operand_rounding: unknown mode %r" % (mode,))
        self.mode = mode

    def __enter__(self):
        global _OPERAND_ROUND
        self.prev, _OPERAND_ROUND = _OPERAND_ROUND, self.mode
        return self

    def __exit__(self, *exc):
        global _OPERAND_ROUND
        _OPERAND_ROUND = self.prev
        return False


def round_bf16(t: torch.Tensor) -> torch.Tensor:
    """Nearest bf16 (ties to even) of the float32 value of every element, returned in t's dtype."""
    return t.detach().to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _RoundedConv2d(torch.autograd.Function):
    """y = conv2d(r(x), r(w)) + b;  dx = conv2d^T(r(g), r(w));  dw = sum_px r(g) r(x);  db = sum_px g  (r = round_bf16)."""

    @staticmethod
    def forward(ctx, x, w, b):
        xr, wr = round_bf16(x), round_bf16(w)
        ctx.save_for_backward(xr, wr)
        ctx.has_bias = b is not None
        if x.dtype != torch.float64:
            return F.conv2d(xr, wr, b, stride=1, padding=w.shape[-1] // 2)
        # float64 (the precision the GPU tests check against): ATen's double convolution is a slow path on CPU; the same
        # sums as one matrix product over unfolded patches are ~5x faster
        B, _, H, W = xr.shape
        k = wr.shape[-1]
        cols = xr.reshape(B, xr.shape[1], H * W) if k == 1 else F.unfold(xr, k, padding=k // 2)
        y = torch.matmul(wr.reshape(wr.shape[0], -1), cols).reshape(B, wr.shape[0], H, W)
        return y + b.view(1, -1, 1, 1) if b is not None else y

    @staticmethod
    def backward(ctx, g):
        xr, wr = ctx.saved_tensors
        gr = round_bf16(g)
        pad = wr.shape[-1] // 2
        dx = torch.nn.grad.conv2d_input(xr.shape, wr, gr, stride=1, padding=pad) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(xr, wr.shape, gr, stride=1, padding=pad) if ctx.needs_input_grad[1] else None
        db = g.sum(dim=(0, 2, 3)) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class _RoundedBmm(torch.autograd.Function):
    """C = bmm(r(A), r(B));  dA = bmm(r(G), r(B)^T);  dB = bmm(r(A)^T, r(G))."""

    @staticmethod
    def forward(ctx, a, b):
        ar, br = round_bf16(a), round_bf16(b)
        ctx.save_for_backward(ar, br)
        return torch.bmm(ar, br)

    @staticmethod
    def backward(ctx, g):
        ar, br = ctx.saved_tensors
        gr = round_bf16(g)
        da = torch.bmm(gr, br.transpose(1, 2)) if ctx.needs_input_grad[0] else None
        db = torch.bmm(ar.transpose(1, 2), gr) if ctx.needs_input_grad[1] else None
        return da, db


def conv2d(x: torch.Tensor, w: torch.Tensor, b) -> torch.Tensor:
    """Same-padding stride-1 convolution under the active operand rounding."""
    if _OPERAND_ROUND == "bf16":
        return _RoundedConv2d.apply(x, w, b)
    return F.conv2d(x, w, b, stride=1, padding=w.shape[-1] // 2)


def bmm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if _OPERAND_ROUND == "bf16":
        return _RoundedBmm.apply(a, b)
    return torch.bmm(a, b)


def conv(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return conv2d(x, p[name + ".weight"], p[name + ".bias"])


def res_block(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    """x + conv2(relu(conv1(x))) -- models/submodules.py:31-35."""
    return x + conv(p, name + ".conv2", torch.relu(conv(p, name + ".conv1", x)))


def layer_norm_2d(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-6):
    """Per-pixel normalisation over channels, biased variance, eps inside the
    sqrt -- models/submodules.py:127-140,157-166 (forward); autograd supplies
    the backward that submodules.py:141-154 writes by hand."""
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    y = (x - mu) / torch.sqrt(var + eps)
    return y * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def bie(p: Params, name: str, x_1, x_2, x_s):
    """Bilateral information exchange -- models/submodules.py:58-77."""
    b, c, h, w = x_1.shape
    r1 = res_block(p, name + ".conv1", x_1)
    r2 = res_block(p, name + ".conv2", x_2)

    def centre(convf: str, other):
        z = conv(p, name + "." + convf, torch.cat([x_s, other], dim=1))
        z = layer_norm_2d(z, p[name + ".norm_s.weight"], p[name + ".norm_s.bias"])
        return conv(p, name + ".clustering", z).reshape(b, c, h * w)

    c1 = centre("convf1", x_2)
    c2 = centre("convf2", x_1)
    v1 = conv(p, name + ".v1", x_1).reshape(b, c, h * w)
    v2 = conv(p, name + ".v2", x_2).reshape(b, c, h * w)
    s = c ** -0.5
    a1 = torch.softmax(bmm(c1, v1.transpose(1, 2)) * s, dim=-1)      # the scale multiplies the product, unrounded (:69-70)
    a2 = torch.softmax(bmm(c2, v2.transpose(1, 2)) * s, dim=-1)
    o1 = bmm(a1, v1).reshape(b, c, h, w)
    o2 = bmm(a2, v2).reshape(b, c, h, w)
    xs_new = conv(p, name + ".unclustering",
                  torch.cat([c1.reshape(b, c, h, w), c2.reshape(b, c, h, w)], dim=1)) + x_s
    return o1 + r2, o2 + r1, xs_new


def pixel_unshuffle(x: torch.Tensor, r: int) -> torch.Tensor:
    """[B,C,rH,rW] -> [B,C*r*r,H,W], channel = c*r*r + i*r + j --
    models/submodules.py:80-92."""
    b, c, hh, ww = x.shape
    h, w = hh // r, ww // r
    return x.reshape(b, c, h, r, w, r).permute(0, 1, 3, 5, 2, 4).reshape(b, c * r * r, h, w)


def pixel_shuffle(x: torch.Tensor, r: int) -> torch.Tensor:
    """Inverse of pixel_unshuffle (F.pixel_shuffle at models/BMCNet.py:119)."""
    b, c, h, w = x.shape
    co = c // (r * r)
    return x.reshape(b, co, r, r, h, w).permute(0, 1, 4, 2, 5, 3).reshape(b, co, h * r, w * r)


def bilinear_up(x: torch.Tensor, r: int) -> torch.Tensor:
    """x r bilinear, align_corners=False (F.interpolate at models/BMCNet.py:119):
    src = (dst + 0.5)/r - 0.5 clamped at 0, i0 = floor(src), i1 = min(i0+1, n-1)."""
    def axis(n):
        dst = torch.arange(n * r, dtype=torch.float32)
        src = torch.clamp((dst + 0.5) / r - 0.5, min=0.0)
        i0 = torch.floor(src).to(torch.int64)
        i1 = torch.clamp(i0 + 1, max=n - 1)
        lam = (src - i0.to(torch.float32)).to(x.dtype)
        return i0, i1, lam
    _, _, h, w = x.shape
    y0, y1, ly = axis(h)
    x0, x1, lx = axis(w)
    rows = x[:, :, y0, :] * (1 - ly).view(1, 1, -1, 1) + x[:, :, y1, :] * ly.view(1, 1, -1, 1)
    return rows[:, :, :, x0] * (1 - lx).view(1, 1, 1, -1) + rows[:, :, :, x1] * lx.view(1, 1, 1, -1)


def bicubic_resize(x: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(x, size=size, mode='bicubic', align_corners=False) written out -- the size-mismatch branch of the
    training loop (train.py:227-231) and of inference (infer_BMCNet.py:77-78).  ATen semantics: per axis
    scale = in/out, src = fma(scale, dst + 0.5, -0.5) (NOT clamped; in x's dtype, one rounding), i0 = floor(src), t = src - i0, four taps
    i0-1 .. i0+2 with the cubic-convolution weights for A = -0.75, tap indices clamped to [0, in-1].  Differentiable
    (index_select + weighted sums), so autograd provides the transposed operator."""
    A = -0.75

    def axis(n_in, n_out):      # index arithmetic in the tensor's own precision, as ATen does (opmath_t)
        dst = torch.arange(n_out, dtype=x.dtype)
        scale = torch.tensor(float(n_in), dtype=x.dtype) / n_out
        # one rounding, as the fused multiply-add of ATen's float kernels gives: fl(scale * (dst + 0.5) - 0.5)
        src = (scale.double() * (dst.double() + 0.5) - 0.5).to(x.dtype)
        i0 = torch.floor(src)
        t = src - i0
        c1 = lambda v: ((A + 2) * v - (A + 3)) * v * v + 1            # |v| <= 1
        c2 = lambda v: ((A * v - 5 * A) * v + 8 * A) * v - 4 * A      # 1 < |v| < 2
        w = torch.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], 0)                 # [4, n_out]
        idx = torch.stack([torch.clamp(i0.to(torch.int64) - 1 + k, 0, n_in - 1) for k in range(4)], 0)
        return idx, w

    H, W = x.shape[-2:]
    Ho, Wo = int(size[0]), int(size[1])
    iy, wy = axis(H, Ho)
    ix, wx = axis(W, Wo)
    rows = sum(x[..., iy[k], :] * wy[k].view(-1, 1) for k in range(4))               # [..., Ho, W]
    return sum(rows[..., ix[k]] * wx[k] for k in range(4))                           # [..., Ho, Wo]


# --------------------------------------------------------------------------
# BMCNet (models/BMCNet.py)
# --------------------------------------------------------------------------
def parallel_blk(p: Params, name: str, x_1, x_2, x_s, x_1_st, x_2_st, x_1_s_st, x_2_s_st):
    """models/BMCNet.py:19-32."""
    x_1 = res_block(p, name + ".conv1", x_1)
    x_2 = res_block(p, name + ".conv2", x_2)
    x_1_st = res_block(p, name + ".conv1_st", x_1_st)
    x_2_st = res_block(p, name + ".conv2_st", x_2_st)
    x_1, x_1_st, x_1_s_st = bie(p, name + ".lBIE", x_1, x_1_st, x_1_s_st)
    x_2, x_2_st, x_2_s_st = bie(p, name + ".lBIE", x_2, x_2_st, x_2_s_st)
    x_1, x_2, x_s = bie(p, name + ".gBIE", x_1, x_2, x_s)
    return x_1, x_2, x_s, x_1_st, x_2_st, x_1_s_st, x_2_s_st


def _n_blocks(p: Params, prefix: str) -> int:
    idx = {int(k[len(prefix):].split(".")[0]) for k in p if k.startswith(prefix)}
    return max(idx) + 1


def bmcnet_backbone(p: Params, x1p, x1n, x2p, x2n, hp, hn, hs, o, scale: int):
    """models/BMCNet.py:57-84."""
    s2 = scale * scale
    op, on = o[:, :s2], o[:, s2:]
    relu = torch.relu
    xp_st = relu(conv(p, "neuro.conv_fpst", torch.cat([x1p, x2p, hp, op], 1)))
    xn_st = relu(conv(p, "neuro.conv_fnst", torch.cat([x1n, x2n, hn, on], 1)))
    xp_s = relu(conv(p, "neuro.conv_fps", torch.cat([x2p, hp], 1)))
    xn_s = relu(conv(p, "neuro.conv_fns", torch.cat([x2n, hn], 1)))
    both = torch.cat([xp_st, xn_st], 1)
    xs = relu(conv(p, "neuro.conv_fs", torch.cat([both, hs, o], 1)))
    xs_p_st = relu(conv(p, "neuro.conv_fs", torch.cat([both, hp, o], 1)))
    xs_n_st = relu(conv(p, "neuro.conv_fs", torch.cat([both, hn, o], 1)))
    for i in range(_n_blocks(p, "neuro.para_reschunk.")):
        xp_s, xn_s, xs, xp_st, xn_st, xs_p_st, xs_n_st = parallel_blk(
            p, f"neuro.para_reschunk.{i}", xp_s, xn_s, xs, xp_st, xn_st, xs_p_st, xs_n_st)
    x_h = relu(conv(p, "neuro.conv_hs", xs))
    x_h_p = relu(conv(p, "neuro.conv_hp", xs_p_st))
    x_h_n = relu(conv(p, "neuro.conv_hn", xs_n_st))
    x_o = conv(p, "neuro.conv_o", torch.cat([xp_s, xn_s], 1))
    return x_h, x_h_p, x_h_n, x_o


def bmcnet_forward(p: Params, x, x_h, x_h_p, x_h_n, x_o, init: bool, scale: int = 4, repeat: int = 3, operand_round=None):
    """One recurrent window -- models/BMCNet.py:95-121.
    x [B,2,T>=2,H,W]; x_o is [B,2*s*s,H,W] when init else the previous HR
    prediction [B,2,sH,sW].  operand_round="bf16": see operand_rounding above (the backward of the returned tensors
    carries the rounding rule with it: the custom Functions are in the graph)."""
    if operand_round is not None:
        with operand_rounding(operand_round):
            return bmcnet_forward(p, x, x_h, x_h_p, x_h_n, x_o, init, scale, repeat)
    f1, f2 = x[:, :, 0], x[:, :, 1]
    rep = lambda t: t.repeat(1, repeat, 1, 1)
    x1p, x1n = rep(f1[:, 0:1]), rep(f1[:, 1:2])
    x2p, x2n = rep(f2[:, 0:1]), rep(f2[:, 1:2])
    if not init:
        x_o = pixel_unshuffle(x_o, scale)
    # NB models/BMCNet.py:115,118 passes (x_h, x_h_p, x_h_n) positionally into
    # Backbone.forward(xs, hp, hn, hs, o) (models/BMCNet.py:57): the state the
    # caller names x_h is consumed as hp, x_h_p as hn and x_h_n as hs.
    x_h, x_h_p, x_h_n, o = bmcnet_backbone(p, x1p, x1n, x2p, x2n, x_h, x_h_p, x_h_n, x_o, scale)
    pred = pixel_shuffle(o, scale) + bilinear_up(f2[:, :2], scale)
    return x_h, x_h_p, x_h_n, pred


# --------------------------------------------------------------------------
# BMCNet_plain (models/BMCNet_plain.py)
# --------------------------------------------------------------------------
def plain_backbone(p: Params, x1, x2, h, o, scale: int):
    """models/BMCNet_plain.py:19-33."""
    s2 = scale * scale
    relu = torch.relu
    a = relu(conv(p, "neuro.conv_f1", torch.cat([x1, h, o[:, :s2]], 1)))
    b = relu(conv(p, "neuro.conv_f2", torch.cat([x2, h, o[:, s2:]], 1)))
    s = relu(conv(p, "neuro.conv_fs", torch.cat([x1, x2, h, o], 1)))
    for i in range(_n_blocks(p, "neuro.para_reschunk.")):
        a, b, s = bie(p, f"neuro.para_reschunk.{i}", a, b, s)
    x_h = relu(conv(p, "neuro.conv_h", s))
    x_o = conv(p, "neuro.conv_o", torch.cat([a, b], 1))
    return x_h, x_o


def plain_forward(p: Params, x, x_h, x_o, init: bool, scale: int = 4, repeat: int = 3):
    """models/BMCNet_plain.py:44-68."""
    f1, f2 = x[:, :, 0], x[:, :, 1]
    rep = lambda t: t.repeat(1, repeat, 1, 1)
    x1 = torch.cat([rep(f1[:, 0:1]), rep(f2[:, 0:1])], 1)
    x2 = torch.cat([rep(f1[:, 1:2]), rep(f2[:, 1:2])], 1)
    if not init:
        x_o = pixel_unshuffle(x_o, scale)
    x_h, o = plain_backbone(p, x1, x2, x_h, x_o, scale)
    return x_h, pixel_shuffle(o, scale) + bilinear_up(f2[:, :2], scale)


# --------------------------------------------------------------------------
# training step (train.py:202-237) and optimiser (config/train_nfs.yml:28-34)
# --------------------------------------------------------------------------
def bptt_loss(p: Params, inp_windows: Sequence[torch.Tensor], gt_windows: Sequence[torch.Tensor],
              n_c: int, scale: int = 4, plain: bool = False, operand_round=None):
    """Sum over windows of mean-squared error between the SR prediction and
    the HR count image, recurrent state carried without detach --
    train.py:205-234.  inp_windows[i] is [B,2(pol),T,H,W] (already transposed
    as train.py:211 does), gt_windows[i] is [B,2,sH,sW]."""
    if operand_round is not None:
        with operand_rounding(operand_round):
            return bptt_loss(p, inp_windows, gt_windows, n_c, scale, plain)
    B, _, _, H, W = inp_windows[0].shape
    z = lambda c: torch.zeros(B, c, H, W, dtype=inp_windows[0].dtype)
    h, hp, hn, pred = z(n_c), z(n_c), z(n_c), z(2 * scale * scale)
    loss = 0.0
    preds = []
    for i, (x, gt) in enumerate(zip(inp_windows, gt_windows)):
        if plain:
            h, pred = plain_forward(p, x, h, pred, i == 0, scale)
        else:
            h, hp, hn, pred = bmcnet_forward(p, x, h, hp, hn, pred, i == 0, scale)
        preds.append(pred)
        sp = pred                                      # the UNRESIZED prediction is what recurs (train.py:224)
        if pred.shape[-2:] != gt.shape[-2:]:          # train.py:227-231
            sp = bicubic_resize(pred, gt.shape[-2:])
        loss = loss + F.mse_loss(sp, gt)
    return loss, preds, (h, hp, hn)


@torch.no_grad()
def infer_windows(p: Params, frames: torch.Tensor, gts: torch.Tensor, n_c: int, scale: int = 4, seqn: int = 3, gt_size=None):
    """The body of the reference's inference loop -- infer_BMCNet.py:44-86: windows of `seqn` frames (default SEQN = 3,
    :147; only frames 0 and 1 of a window are read by the model, models/BMCNet.py:106-107), zero state before the first
    window (:55-60), (h, hp, hn, prediction) carried (:62-63); per window the metrics esr_mse = MSE(prediction resized to
    the ground truth's size when they differ, gt) (:76-78,84) and the baseline bicubic_mse =
    MSE(bicubic(inp_cnt[:, 1] -> gt_sensor_resolution), gt) (:79,85).  frames [B,L,2,H,W], gts [B,L,2,gh,gw].
    -> list of (prediction, esr_mse, bicubic_mse) per window."""
    B, L, _, H, W = frames.shape
    gt_size = tuple(gts.shape[-2:]) if gt_size is None else tuple(gt_size)
    z = lambda c: torch.zeros(B, c, H, W, dtype=frames.dtype)
    h, hp, hn, pred = z(n_c), z(n_c), z(n_c), z(2 * scale * scale)
    out = []
    for i in range(L - seqn + 1):
        x = frames[:, i:i + seqn].transpose(1, 2)
        gt = gts[:, i + 1]
        h, hp, hn, pred = bmcnet_forward(p, x, h, hp, hn, pred, i == 0, scale)
        esr = pred if tuple(pred.shape[-2:]) == tuple(gt.shape[-2:]) else bicubic_resize(pred, gt.shape[-2:])
        base = bicubic_resize(frames[:, i + 1], gt_size)
        out.append((pred, F.mse_loss(esr, gt), F.mse_loss(base, gt)))
    return out


def adam_amsgrad_step(params, grads, state, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5):
    """One torch.optim.Adam(amsgrad=True, weight_decay=wd) update written out
    (the optimiser train.py:653 builds from config/train_nfs.yml:28-34).
    state: dict with 'step' and per-parameter lists m, v, vmax; in-place."""
    state["step"] += 1
    t = state["step"]
    b1, b2 = betas
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    for i, (w, g) in enumerate(zip(params, grads)):
        g = g + weight_decay * w
        state["m"][i] = b1 * state["m"][i] + (1 - b1) * g
        state["v"][i] = b2 * state["v"][i] + (1 - b2) * g * g
        state["vmax"][i] = torch.maximum(state["vmax"][i], state["v"][i])
        denom = state["vmax"][i].sqrt() / math.sqrt(bc2) + eps
        w -= (lr / bc1) * state["m"][i] / denom
    return params


def unique_params(p: Params):
    """State-dict keys grouped by storage (shared modules register alias keys);
    returns {canonical key: [all keys sharing that tensor]}."""
    seen = {}
    for k, v in p.items():
        seen.setdefault(v.data_ptr(), []).append(k)
    return {ks[0]: ks for ks in seen.values()}


def canonical_key(k: str) -> str:
    """The name named_parameters() reports for the tensor behind state-dict key k: the reference registers one module
    object under several names (models/BMCNet.py:7,9,41,43,46; models/BMCNet_plain.py:9,13; models/submodules.py:45,49)
    and every alias shows up as a key of its own."""
    import re
    k = re.sub(r"para_reschunk\.\d+\.", "para_reschunk.0.", k)
    for a, b in (("neuro.conv_fnst.", "neuro.conv_fpst."), ("neuro.conv_fns.", "neuro.conv_fps."),
                 ("neuro.conv_f2.", "neuro.conv_f1.")):
        if k.startswith(a):
            k = b + k[len(a):]
    parts = k.split(".")
    # module-level aliases: <blk>.conv2 == <blk>.conv1 (a ResidualBlock_noBN, i.e. followed by its own conv1/conv2),
    # <blk>.conv2_st == <blk>.conv1_st, <bie>.convf2 == <bie>.convf1
    for i, q in enumerate(parts):
        if q == "convf2":
            parts[i] = "convf1"
        elif q == "conv2_st":
            parts[i] = "conv1_st"
        elif q == "conv2" and i + 1 < len(parts) and parts[i + 1] in ("conv1", "conv2"):
            parts[i] = "conv1"
    return ".".join(parts)


def expand_aliases(unique: Params, keys) -> Params:
    """{named_parameters() name: tensor} -> full state-dict mapping over `keys` with the reference's aliasing."""
    return {k: unique[canonical_key(k)] for k in keys}
