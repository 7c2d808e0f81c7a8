"""The drop-in recipe of INTEGRATION.md, exercised in the build container (needs /root/reference: skipped elsewhere):
tests/dropin_check.py runs the reference's own import block with sys.path = [bmcnet-esr_amd, reference] in a fresh
interpreter and checks which file every module resolves to, the DataLoader-worker data path and the strict
checkpoint load."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("BMC_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference repo not present (build container only)")
def test_reference_imports_resolve_and_feed_the_hip_modules():
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([sys.executable, os.path.join(HERE, "dropin_check.py")], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["files"]["dataloader.h5dataloader"].startswith(REF)
    assert res["files"]["dataloader.encodings"].startswith(REF)
    assert "bmcnet-esr_amd" in res["files"]["models.BMCNet"]
    assert res["windows"] == 8 and res["bmcnet_keys"] == 318
    assert res["plain_load"] == "<All keys matched successfully>"


def test_package_does_not_shadow_reference_packages():
    """Nothing under bmcnet-esr_amd/ may be named like a reference package other than `models` (the boundary)."""
    pkg = os.path.join(os.path.dirname(HERE), "bmcnet-esr_amd")
    top = {n.split(".")[0] for n in os.listdir(pkg) if not n.startswith("__")}
    assert not top & {"dataloader", "config", "loss", "logger", "myutils", "generate_dataset", "train", "infer_BMCNet"}, top
