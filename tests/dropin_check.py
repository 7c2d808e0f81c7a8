#!/usr/bin/env python3
"""Build-container check of the documented drop-in recipe: with sys.path = [bmcnet-esr_amd,
<reference>] the reference's OWN import block (train.py:16-26, infer_BMCNet.py:11-17) must resolve
  models.*        -> this repo (HIP-backed modules),
  dataloader.*    -> the reference (its CPU workers keep calling its own events_to_channels),
and the reference's SequenceDataset -> HDF5DataLoaderSequence chain (with worker processes) must feed window lists of
the layout the trainer loop consumes.  Run in its own interpreter by tests/test_dropin_imports.py; needs
/root/reference, so it only runs where the reference exists.  Prints one JSON line."""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "bmcnet-esr_amd")
REF = os.environ.get("BMC_REFERENCE", "/root/reference")

sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_stubs  # noqa: E402

ref_stubs.install()
# --- what INTEGRATION.md tells a maintainer to do: our package first, then the reference's repo root
sys.path[:0] = [PKG, REF]

# --- the reference's import block (train.py:16-26), minus what needs absent third-party packages at import time
from config.parser import YAMLParser                                            # noqa: E402,F401  train.py:16
from dataloader.h5dataloader import HDF5DataLoader, HDF5DataLoaderSequence      # noqa: E402,F401  train.py:17
from myutils.utils import *                                                      # noqa: E402,F401,F403  train.py:19
from dataloader.encodings import *                                               # noqa: E402,F401,F403  train.py:24
from models.BMCNet import BMCNet                                                 # noqa: E402  train.py:26
from models.BMCNet_plain import BMCNet_plain                                     # noqa: E402  train_plain.py:25
import dataloader.encodings as enc_mod                                           # noqa: E402
import dataloader.h5dataloader as dl_mod                                         # noqa: E402
import dataloader.h5dataset as ds_mod                                            # noqa: E402
import models.BMCNet as model_mod                                                # noqa: E402
import models.submodules as sub_mod                                              # noqa: E402

import torch                                                                     # noqa: E402
import yaml                                                                      # noqa: E402

res = {"files": {m.__name__: os.path.abspath(m.__file__) for m in (enc_mod, dl_mod, ds_mod, model_mod, sub_mod)}}
for m in (enc_mod, dl_mod, ds_mod):
    assert res["files"][m.__name__].startswith(os.path.abspath(REF) + os.sep), res["files"]
for m in (model_mod, sub_mod):
    assert res["files"][m.__name__].startswith(PKG + os.sep), res["files"]
# the star import of train.py:24 / h5dataset.py:19 still provides the reference's whole encoding surface
for name in ("events_to_channels", "events_to_image", "events_to_mask", "events_to_voxel", "events_to_stack_no_polarity",
             "events_to_stack_polarity", "binary_search_torch_tensor"):
    assert name in globals() and globals()[name].__module__ == "dataloader.encodings", name

# --- the data path of train.py:634 with DataLoader WORKER PROCESSES on the stub recording
cfg = yaml.safe_load(open(os.path.join(REF, "config", "train_nfs.yml")))["train_dataloader"]
with tempfile.TemporaryDirectory() as td:
    paths = [os.path.join(td, "rec%d.h5" % i) for i in range(cfg["batch_size"])]
    for i, p in enumerate(paths):
        ref_stubs.FAKE_FILES[p] = ref_stubs.synth_nfs_file(101 * (i + 1))
    lst = os.path.join(td, "list.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    loader = HDF5DataLoaderSequence(dict(cfg, path_to_datalist_txt=lst, num_workers=2, shuffle=False, pin_memory=False))
    inputs_seq = next(iter(loader))
res["windows"] = len(inputs_seq)
res["inp_cnt"] = list(inputs_seq[0]["inp_cnt"].shape)
res["gt_cnt"] = list(inputs_seq[0]["gt_cnt"].shape)
assert res["windows"] == 8 and res["inp_cnt"] == [2, 2, 2, 45, 80] and res["gt_cnt"] == [2, 2, 2, 180, 320]

# --- the model side of train.py:638-641 / infer_BMCNet.py:106-116: our classes, reference checkpoints load strictly
esr_model = BMCNet(scale=4, n_c=128, n_b=5)
res["bmcnet_keys"] = len(esr_model.state_dict())
plain = BMCNet_plain(4, 128, 5)
ck = os.path.join(REF, "pretrain", "BMCNet_plain_nfs_x4.pth")
res["plain_load"] = str(plain.load_state_dict(torch.load(ck, map_location="cpu"), strict=True))
# there is no CPU fallback behind the reference's loop body (train.py:211-224): on a box without a GPU the call raises
inputs = inputs_seq[0]
input_stack = inputs["inp_cnt"].transpose(1, 2)
z = torch.zeros_like(input_stack[:, 0:1, 0])
try:
    esr_model(input_stack, z.repeat(1, 128, 1, 1), z.repeat(1, 128, 1, 1), z.repeat(1, 128, 1, 1), z.repeat(1, 32, 1, 1), True)
    res["cpu_forward"] = "ran"
except RuntimeError as e:
    res["cpu_forward"] = "raises: " + str(e)[:60]
if not torch.cuda.is_available():
    assert res["cpu_forward"].startswith("raises"), res
print(json.dumps(res))
