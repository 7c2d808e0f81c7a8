"""GPU event encodings with the reference's function signature
(reference: dataloader/encodings.py:290-305 events_to_channels, :241-269 events_to_image).

The scatter runs on the MI355X (bmc_events_to_channels, one float atomic per event, bit-exact with the
reference including its out-of-range quirk and its in-place reset of the caller's xs/ys).  CPU tensors
are rejected: there is no CPU fallback."""
import torch

from . import ops


def events_to_channels(xs, ys, ps, sensor_size=(180, 240)):
    """Two-channel event count image [2,H,W] from fp32 event vectors on the GPU.
    As in the reference, out-of-range entries of the caller's xs/ys are reset to 0 in place."""
    assert len(xs) == len(ys) and len(ys) == len(ps)
    if not (xs.is_cuda and ys.is_cuda and ps.is_cuda):
        raise RuntimeError("events_to_channels: tensors must live on the MI355X (no CPU fallback in this build)")
    if not (xs.is_contiguous() and ys.is_contiguous() and ps.is_contiguous()):
        raise RuntimeError("events_to_channels: xs/ys/ps must be contiguous (they are updated in place)")
    off = torch.tensor([0, xs.numel()], dtype=torch.int64, device=xs.device)
    return ops.events_to_channels_batched(xs, ys, ps, off, int(sensor_size[0]), int(sensor_size[1]), mutate=True)[0]


def events_to_voxel(xs, ys, ts, ps, num_bins, sensor_size=(180, 240)):
    """Voxel grid [num_bins,H,W] with temporal bilinear interpolation (reference: dataloader/encodings.py:272-287),
    on GPU tensors; as in the reference the caller's xs/ys lose their out-of-range entries (reset to 0)."""
    assert len(xs) == len(ys) and len(ys) == len(ts) and len(ts) == len(ps)
    if not (xs.is_cuda and ys.is_cuda and ts.is_cuda and ps.is_cuda):
        raise RuntimeError("events_to_voxel: tensors must live on the MI355X (no CPU fallback in this build)")
    off = torch.tensor([0, xs.numel()], dtype=torch.int64, device=xs.device)
    return ops.events_to_voxel_batched(xs, ys, ts, ps, off, int(num_bins), int(sensor_size[0]), int(sensor_size[1]))[0]


def events_to_stack_no_polarity(xs, ys, ts, ps, B, device=None, sensor_size=(180, 240)):
    """Event stack [B,H,W]: B temporal bins, each the signed per-pixel sum of its events' polarities (reference:
    dataloader/encodings.py:202-238, same signature).  On GPU tensors; like the reference it returns zeros for windows
    of <= 3 events or all-zero timestamps, and zeroes the out-of-range events of the caller's xs / ys / ps in place."""
    assert len(xs) == len(ys) and len(ys) == len(ts) and len(ts) == len(ps)
    if not (xs.is_cuda and ys.is_cuda and ts.is_cuda and ps.is_cuda):
        raise RuntimeError("events_to_stack_no_polarity: tensors must live on the MI355X (no CPU fallback in this build)")
    H, W = int(sensor_size[0]), int(sensor_size[1])
    if len(ts) <= 3 or ts.sum() == 0:
        return torch.zeros([B, H, W], device=xs.device)
    return ops.events_to_stack(xs, ys, ts, ps, int(B), H, W)


def events_to_stack_polarity(xs, ys, ts, ps, B, device=None, sensor_size=(180, 240)):
    """Polarity-split event stack [2,B,H,W] (reference: dataloader/encodings.py:151-199, same signature) on GPU tensors;
    like the reference: [B,H,W] zeros for windows of <= 3 events or all-zero timestamps, and the caller's xs / ys lose their
    out-of-range entries (ps is left alone)."""
    assert len(xs) == len(ys) and len(ys) == len(ts) and len(ts) == len(ps)
    if not (xs.is_cuda and ys.is_cuda and ts.is_cuda and ps.is_cuda):
        raise RuntimeError("events_to_stack_polarity: tensors must live on the MI355X (no CPU fallback in this build)")
    H, W = int(sensor_size[0]), int(sensor_size[1])
    if ts.sum() == 0 or len(ts) <= 3:
        return torch.zeros([B, H, W], device=xs.device)
    return ops.events_to_stack_polarity(xs, ys, ts, ps, int(B), H, W)


def events_to_mask(xs, ys, ps, sensor_size=(180, 240)):
    """Binary event mask [H,W] (reference: dataloader/encodings.py:308-332, same signature) on GPU tensors; as in the
    reference the caller's xs / ys / ps lose their out-of-range entries in place."""
    if not (xs.is_cuda and ys.is_cuda and ps.is_cuda):
        raise RuntimeError("events_to_mask: tensors must live on the MI355X (no CPU fallback in this build)")
    if not (xs.is_contiguous() and ys.is_contiguous() and ps.is_contiguous()):
        raise RuntimeError("events_to_mask: xs/ys/ps must be contiguous (they are updated in place)")
    return ops.events_to_mask(xs, ys, ps, int(sensor_size[0]), int(sensor_size[1]))


def events_to_channels_batch(xs, ys, ps, offsets, sensor_size=(180, 240), mutate=True):
    """Many frames in one launch: frame f owns events [offsets[f], offsets[f+1]) -> [nframes,2,H,W]."""
    return ops.events_to_channels_batched(xs, ys, ps, offsets, int(sensor_size[0]), int(sensor_size[1]), mutate=mutate)


def augment_flags(seed, mechanisms=("Horizontal", "Vertical", "Polarity"), probs=(0.5, 0.5, 0.5)):
    """Flip decisions of H5Dataset.augment_event (dataloader/h5dataset.py:559-578) for one sample seed, as the bit
    flags bmc_encode_raw_events takes (bit0 horizontal, bit1 vertical, bit2 polarity).  Same seeding as the reference:
    random.seed(seed), seed+1, seed+2 for H / V / P, one random.random() draw each; the caller's `random` state is
    left as the reference leaves it (re-seeded)."""
    import random
    flags = 0
    for i, mech in enumerate(mechanisms):
        if mech == "Horizontal":
            random.seed(seed)
            if random.random() < probs[i]:
                flags |= 1
        elif mech == "Vertical":
            random.seed(seed + 1)
            if random.random() < probs[i]:
                flags |= 2
        elif mech == "Polarity":
            random.seed(seed + 2)
            if random.random() < probs[i]:
                flags |= 4
    return flags


def raw_events_to_channels_batch(xs_i16, ys_i16, ps_f64, offsets, flips=None, sensor_size=(180, 240)):
    """GPU sequence encoder on raw HDF5 columns (int16 x/y, float64 p) with the flip augmentation folded in:
    replaces get_events -> augment_event -> event_formatting -> events_to_channels of the CPU workers."""
    return ops.encode_raw_events(xs_i16, ys_i16, ps_f64, offsets, flips, int(sensor_size[0]), int(sensor_size[1]))


def events_to_image_torch(xs, ys, ps, device=None, sensor_size=(180, 240), clip_out_of_range=True, interpolation=None, padding=True):
    """events_to_image_torch of the reference (dataloader/encodings.py:16-73) on the GPU: same signature, same side effects
    (out-of-range events are reset in place to (0, 0) with weight 0), same summation order as its CPU index_put_ -> bit-identical.
    xs / ys / ps: contiguous float32 CUDA tensors (sub-pixel positions for interpolation='bilinear')."""
    import ctypes as C
    from . import lib, ops
    if interpolation not in (None, "bilinear"):
        raise ValueError("interpolation must be None or 'bilinear'")
    for t in (xs, ys, ps):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("bmc_hip.encodings.events_to_image_torch needs contiguous float32 CUDA tensors (the reference's long-"
                               "coordinate variant: convert with .float())")
    H, W = sensor_size
    bil = interpolation == "bilinear"
    shape = (H + 1, W + 1) if (bil and padding) else (H, W)
    n = xs.numel()
    out = torch.empty(shape, device=xs.device, dtype=torch.float32)
    ws = torch.empty(lib._ev_torch_ws(n, H, W), device=xs.device, dtype=torch.int32)
    lib.call(lib._ev_img_torch, "bmc_events_to_image_torch", xs.data_ptr(), ys.data_ptr(), ps.data_ptr(), n, H, W,
             1 if clip_out_of_range else 0, 1 if bil else 0, 1 if padding else 0, out.data_ptr(), ws.data_ptr(), ops._stream())
    return out


def events_to_voxel_torch(xs, ys, ts, ps, B, device=None, sensor_size=(180, 240), temporal_bilinear=True):
    """events_to_voxel_torch of the reference (dataloader/encodings.py:100-148), temporal_bilinear=True, on the GPU: [B, H, W];
    zeros for <= 3 events or all-zero timestamps; xs / ys are reset in place where out of range, as the reference does."""
    from . import lib, ops
    if not temporal_bilinear:
        raise NotImplementedError("bmc_hip.encodings.events_to_voxel_torch: only temporal_bilinear=True is implemented")
    for t in (xs, ys, ts, ps):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("bmc_hip.encodings.events_to_voxel_torch needs contiguous float32 CUDA tensors")
    H, W = sensor_size
    n = xs.numel()
    out = torch.empty((B, H, W), device=xs.device, dtype=torch.float32)
    ws = torch.empty(lib._ev_torch_ws(n, H, W), device=xs.device, dtype=torch.int32)
    lib.call(lib._ev_vox_torch, "bmc_events_to_voxel_torch", xs.data_ptr(), ys.data_ptr(), ts.data_ptr(), ps.data_ptr(), n, B, H, W,
             out.data_ptr(), ws.data_ptr(), ops._stream())
    return out
