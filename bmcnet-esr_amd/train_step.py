"""One BPTT training step with the semantics of the reference's hot loop (train.py:202-237):
zero_grad; for each sliding window: forward with the recurrent state carried WITHOUT detach, MSE against the HR
count image of the window's second frame, losses summed; one backward over all windows; one optimizer step.

Plus the synthetic NFS-shaped event generator used by bench.py / tests."""
import torch
import torch.nn.functional as F

from bmc_hip import ops


def synthetic_events(B, seql, H, W, scale, n_lr, device, seed=3407):
    """Per sample and frame: n_lr LR events and scale^2*n_lr HR events, x~U{0..W-1}, y~U{0..H-1}, p=+-1.
    Returns dict of flat fp32 vectors + int64 frame offsets (frame order: b-major, then time)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for tag, (h, w, n) in {"lr": (H, W, n_lr), "hr": (H * scale, W * scale, n_lr * scale * scale)}.items():
        nf = B * seql
        xs = torch.randint(0, w, (nf * n,), generator=g).to(torch.float32)
        ys = torch.randint(0, h, (nf * n,), generator=g).to(torch.float32)
        ps = (torch.randint(0, 2, (nf * n,), generator=g) * 2 - 1).to(torch.float32)
        off = torch.arange(nf + 1, dtype=torch.int64) * n
        out[tag] = tuple(t.to(device) for t in (xs, ys, ps, off))
    return out


def shard_sequences(inp_cnt, gt_cnt, rank, world):
    """Sequence-batch sharding: rank r of `world` owns sequences [r*B/world, (r+1)*B/world) of the global
    batch -- what DistributedSampler does for the reference's loader (dataloader/h5dataloader.py:191-201); every rank
    then runs the full BPTT on its own sequences and GradAllReducer averages the gradients."""
    B = inp_cnt.shape[0]
    if B % world:
        raise ValueError("global batch %d is not divisible by %d ranks" % (B, world))
    n = B // world
    return inp_cnt[rank * n:(rank + 1) * n], gt_cnt[rank * n:(rank + 1) * n]


def encode_sequence(ev, B, seql, H, W, scale):
    """events -> inp_cnt [B,seql,2,H,W], gt_cnt [B,seql,2,sH,sW] on the GPU (two batched scatter launches);
    what the reference's DataLoader workers do per frame with events_to_channels (dataloader/h5dataset.py:518-526)."""
    xs, ys, ps, off = ev["lr"]
    inp = ops.events_to_channels_batched(xs, ys, ps, off, H, W, mutate=False).view(B, seql, 2, H, W)
    xs, ys, ps, off = ev["hr"]
    gt = ops.events_to_channels_batched(xs, ys, ps, off, H * scale, W * scale, mutate=False).view(
        B, seql, 2, H * scale, W * scale)
    return inp, gt


def bptt_step(model, optimizer, inp_cnt, gt_cnt, n_c, scale, seqn=2, plain=False, loss_fn=F.mse_loss, recompute=False):
    """inp_cnt [B,L,2,H,W], gt_cnt [B,L,2,sH,sW] (device tensors).  Returns (loss, last mse).

    recompute=True keeps only the recurrent state (h, hp, hn, prediction) of every window and re-runs a window's
    forward inside backward (activation checkpointing per window): peak memory drops from
    windows x (one window's activations) to one window's activations, at the price of one extra forward per
    window.  Needed for long sequences / big batches (BASELINE configs[4]: T=16, 8 sequences per GPU); results
    are bit-identical (same kernels, same order)."""
    from torch.utils.checkpoint import checkpoint
    B, L, _, H, W = inp_cnt.shape
    dev = inp_cnt.device
    optimizer.zero_grad()
    loss = 0
    init = True
    z = lambda c: torch.zeros(B, c, H, W, device=dev)
    for i in range(L - seqn + 1):
        x = inp_cnt[:, i:i + seqn].transpose(1, 2)          # [B,2(pol),seqn,H,W]   (train.py:211)
        gt = gt_cnt[:, i + 1]                                # (train.py:213)
        # nn.MSELoss on a prediction of the ground truth's size: head + loss in one kernel pass (models.*.forward_loss)
        fused = loss_fn is F.mse_loss and hasattr(model, "forward_loss") and tuple(gt.shape[-2:]) == (scale * H, scale * W)
        fn = model.forward_loss if fused else model
        extra = (gt,) if fused else ()
        if init:
            state = (z(n_c), z(2 * scale * scale)) if plain else (z(n_c), z(n_c), z(n_c), z(2 * scale * scale))
            out = fn(x, *state, True, *extra)
            init = False
        elif recompute:
            out = checkpoint(fn, x, *state, False, *extra, use_reentrant=False)
        else:
            out = fn(x, *state, False, *extra)
        if fused:
            state, mse = tuple(out[:-1]), out[-1]
        else:
            state = tuple(out)
            # size-mismatch branch of train.py:227-231 (EventZoom: 124x224 prediction vs 124x222 ground truth); the
            # UNRESIZED prediction is what recurs into the next window (train.py:224)
            mse = loss_fn(ops.bicubic_resize(state[-1], gt.shape[-2:]), gt)
        loss = loss + mse
    loss.backward()
    optimizer.step()
    return loss.detach(), mse.detach()
