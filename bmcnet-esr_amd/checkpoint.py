"""Checkpoint / resume for the BPTT trainer.

The reference's live path writes a BARE ``model.state_dict()`` per checkpoint (train.py:555-563) which
``infer_BMCNet.load_model`` consumes (infer_BMCNet.py:106-116); its resume path (``Resumer``,
myutils/utils.py:140-177, train.py:565-603) is dead code expecting a different layout
``{key: {'name':..., 'states':...}}`` and never restores anything.  Here:

* ``save_checkpoint`` writes the same bare state_dict file (drop-in for the reference's inference / load_state_dict),
  plus, next to it, ``<file>.train`` with everything needed to continue training bit-exactly: optimizer and
  scheduler state, iteration, best monitored metric, RNG states -- laid out the way the reference's ``Resumer``
  expects (``{'model'|'optimizer'|'lr_scheduler': {'name', 'states'}, 'trainer': {...}}``) so that code would work too;
* ``resume`` restores all of it.
Writes are atomic (temp file + rename).  With N > 1 ranks only rank 0 writes.
"""
import os

import torch


def _atomic_save(obj, path):
    tmp = path + ".tmp"
    torch.save(obj, tmp)
    os.replace(tmp, path)


def save_checkpoint(path, model, optimizer=None, lr_scheduler=None, iteration=0, monitor_best=None,
                    training_mode="iteration_based_train", rank=0):
    """Bare state_dict at `path` (reference format) + full training state at `path + '.train'`."""
    if rank != 0:
        return
    # alias keys kept; ONE host copy per device tensor, so that torch.save de-duplicates the aliases' storage as it does
    # for the reference's CPU-resident state_dict (a per-key .cpu() would write every shared tensor once per alias)
    host = {}
    sd = {}
    for k, v in model.state_dict().items():
        key = (v.data_ptr(), tuple(v.shape), tuple(v.stride()))
        if key not in host:
            host[key] = v.detach().cpu()
        sd[k] = host[key]
    _atomic_save(sd, path)
    if optimizer is None:
        return
    state = {
        "model": {"name": type(model).__name__, "states": sd},
        "optimizer": {"name": type(optimizer).__name__, "states": optimizer.state_dict()},
        "lr_scheduler": {"name": type(lr_scheduler).__name__ if lr_scheduler is not None else None,
                         "states": lr_scheduler.state_dict() if lr_scheduler is not None else None},
        "trainer": {"training_mode": training_mode, "iteration": int(iteration), "monitor_best": monitor_best},
        "rng": {"torch": torch.get_rng_state(),
                "cuda": torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None},
    }
    _atomic_save(state, path + ".train")


def load_model_state(path, model, strict=True):
    """Load a bare state_dict checkpoint (ours or the reference's, e.g. pretrain/BMCNet_plain_nfs_x4.pth)."""
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict) and "states" in sd["model"]:
        sd = sd["model"]["states"]
    return model.load_state_dict(sd, strict=strict)


def resume(path, model, optimizer=None, lr_scheduler=None, restore_rng=True):
    """Restore model (+ optimizer, scheduler, RNG) from `path` / `path + '.train'`; returns the trainer dict
    ({'iteration', 'monitor_best', 'training_mode'}), iteration = -1 when only the bare file exists."""
    train_path = path if path.endswith(".train") else path + ".train"
    if not os.path.exists(train_path):
        load_model_state(path, model)
        return {"training_mode": None, "iteration": -1, "monitor_best": None}
    st = torch.load(train_path, map_location="cpu", weights_only=False)
    model.load_state_dict(st["model"]["states"], strict=True)
    if optimizer is not None and st["optimizer"]["name"] == type(optimizer).__name__:
        optimizer.load_state_dict(st["optimizer"]["states"])
    if lr_scheduler is not None and st["lr_scheduler"]["states"] is not None \
            and st["lr_scheduler"]["name"] == type(lr_scheduler).__name__:
        lr_scheduler.load_state_dict(st["lr_scheduler"]["states"])
    if restore_rng and st.get("rng"):
        torch.set_rng_state(st["rng"]["torch"])
        if st["rng"]["cuda"] is not None and torch.cuda.is_available():
            try:
                torch.cuda.set_rng_state_all(st["rng"]["cuda"])
            except RuntimeError:
                pass        # different GPU count than at save time
    return st["trainer"]
