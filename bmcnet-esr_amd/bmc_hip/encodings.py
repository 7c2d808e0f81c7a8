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
