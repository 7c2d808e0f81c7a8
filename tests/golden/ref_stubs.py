"""Stand-ins for the reference's absent third-party imports, for the scripts that IMPORT THE REFERENCE in the build
container (tests/golden/make_golden*.py, tests/test_dropin_imports.py).  Test infrastructure only.

* torchvision, cv2, open3d: empty modules (imported by models/submodules.py:4, dataloader/h5dataset.py:7,
  myutils/vis_events/*.py but untouched on the path exercised here);
* h5py: a dict-backed File whose datasets are numpy arrays (what dataloader/h5dataset.py:34-41,151-156,407-424 reads:
  `.attrs['sensor_resolution']`, `f['down8_events']['ts'][:]`, `f['down8_events/xs'][i0:i1]`);
* matplotlib: `plt.style.use('seaborn-whitegrid')` (dataloader/h5dataset.py:15) names a style matplotlib 3.10 no
  longer ships -> no-op.
"""
import sys
import types

import numpy as np

FAKE_FILES = {}          # path -> {"attrs": {...}, "<group>/<name>": ndarray}


class _Group:
    def __init__(self, store, prefix):
        self._s, self._p = store, prefix

    def __getitem__(self, k):
        return self._s[self._p + "/" + k]


class FakeH5File:
    def __init__(self, path, mode="r"):
        self._s = FAKE_FILES[path]
        self.attrs = self._s["attrs"]

    def __getitem__(self, k):
        if k in self._s:
            return self._s[k]
        if any(key.startswith(k + "/") for key in self._s):
            return _Group(self._s, k)
        raise KeyError(k)

    def close(self):
        pass


def install():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.style.use = lambda *a, **k: None
    for name in ("torchvision", "cv2", "open3d"):
        sys.modules.setdefault(name, types.ModuleType(name))
    h5 = types.ModuleType("h5py")
    h5.File = FakeH5File
    sys.modules["h5py"] = h5


def synth_nfs_file(seed, n_lr=12 * 1024, scale=4, sensor=(360, 640), lr_div=8, jitter=2):
    """Columns of one synthetic NFS-style recording as generate_dataset/tools/event_packagers.py:128-156 stores them
    (xs/ys int16, ts/ps float64): `down8` LR events and the `down2` stream `scale` x finer with scale^2 x the events,
    timestamps sorted; a few coordinates are pushed out of range (jitter) to exercise the encoder's quirk."""
    rng = np.random.default_rng(seed)
    H, W = sensor[0] // lr_div, sensor[1] // lr_div
    out = {"attrs": {"sensor_resolution": np.asarray(sensor)}}
    for prex, (h, w, n) in {"down%d" % lr_div: (H, W, n_lr),
                            "down%d" % (lr_div // scale): (H * scale, W * scale, n_lr * scale * scale)}.items():
        xs = rng.integers(0, w, n).astype(np.int16)
        ys = rng.integers(0, h, n).astype(np.int16)
        bad = rng.integers(0, n, max(n // 500, 1))
        xs[bad[::2]] = w + rng.integers(0, jitter, bad[::2].size)
        ys[bad[1::2]] = -1 - rng.integers(0, jitter, bad[1::2].size)
        out[prex + "_events/xs"] = xs
        out[prex + "_events/ys"] = ys
        out[prex + "_events/ts"] = np.sort(rng.uniform(0.0, 1.0, n))
        out[prex + "_events/ps"] = rng.choice([-1.0, 1.0], n)
    return out
