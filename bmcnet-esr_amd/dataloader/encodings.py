"""GPU event encodings with the reference's function signature
(reference: dataloader/encodings.py:290-305 events_to_channels, :241-269 events_to_image).

The scatter runs on the MI355X (bmc_events_to_channels, one float atomic per event, bit-exact with the
reference including its out-of-range quirk and its in-place reset of the caller's xs/ys).  CPU tensors
are rejected: there is no CPU fallback."""
import torch

from bmc_hip import ops


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


def events_to_channels_batch(xs, ys, ps, offsets, sensor_size=(180, 240), mutate=True):
    """Many frames in one launch: frame f owns events [offsets[f], offsets[f+1]) -> [nframes,2,H,W]."""
    return ops.events_to_channels_batched(xs, ys, ps, offsets, int(sensor_size[0]), int(sensor_size[1]), mutate=mutate)
