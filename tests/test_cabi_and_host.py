"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/bmc_hip.h declares, the
drop-in modules keep the reference's state_dict surface, and the host logic (conv channel maps) is consistent."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from bmc_hip import lib
    hdr = open(os.path.join(ROOT, "include", "bmc_hip.h")).read()
    declared = set(re.findall(r"\b(bmc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    for name in declared:
        assert lib.has_symbol(name), name
    assert set(lib.EXPORTS) == declared
    assert lib.bmc_version() >= 100


def test_struct_layout_matches_header():
    """ctypes mirrors of bmc_src_t / bmc_conv_args_t / bmc_pgemm_args_t against a C compile of the header."""
    import ctypes as C
    import subprocess
    import tempfile
    from bmc_hip import lib
    src = '#include <stdio.h>\n#include "bmc_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(bmc_src_t), sizeof(bmc_conv_args_t), sizeof(bmc_pgemm_args_t), sizeof(bmc_chain_fwd_args_t), sizeof(bmc_chain_bwd_args_t));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out] == [C.sizeof(lib.Src), C.sizeof(lib.ConvArgs), C.sizeof(lib.PgemmArgs),
                                      C.sizeof(lib.ChainFwdArgs), C.sizeof(lib.ChainBwdArgs)]


def test_state_dict_surface_matches_reference_counts():
    from models.BMCNet import BMCNet
    from models.BMCNet_plain import BMCNet_plain
    m = BMCNet(4, 128, 5)
    assert len(m.state_dict()) == 318 and len(list(m.parameters())) == 54
    assert sum(p.numel() for p in m.parameters()) == 2731680
    p = BMCNet_plain(4, 128, 5)
    assert len(p.state_dict()) == 120 and len(list(p.parameters())) == 24
    assert sum(q.numel() for q in p.parameters()) == 1003296
    sd = m.state_dict()
    assert sd["neuro.conv_fnst.weight"].data_ptr() == sd["neuro.conv_fpst.weight"].data_ptr()
    assert sd["neuro.para_reschunk.4.lBIE.conv2.conv1.weight"].data_ptr() == sd["neuro.para_reschunk.0.lBIE.conv1.conv1.weight"].data_ptr()
    assert sd["neuro.para_reschunk.0.gBIE.convf2.weight"].data_ptr() == sd["neuro.para_reschunk.0.gBIE.convf1.weight"].data_ptr()


def test_state_dict_keys_equal_golden_reference_keys():
    import numpy as np
    from models.BMCNet import BMCNet
    from models.BMCNet_plain import BMCNet_plain
    g = os.path.join(ROOT, "tests", "golden")
    for tag, cls in (("bmcnet_nc16", BMCNet), ("plain_nc16", BMCNet_plain)):
        z = np.load(os.path.join(g, tag + ".npz"))
        ref_keys = sorted(k[3:] for k in z.files if k.startswith("sd/"))
        scale, n_c, n_b = (int(v) for v in z["meta"][:3])
        assert sorted(cls(scale, n_c, n_b).state_dict().keys()) == ref_keys


def test_pretrained_plain_checkpoint_loads_strictly():
    ck = "/root/reference/pretrain/BMCNet_plain_nfs_x4.pth"
    if not os.path.exists(ck):
        pytest.skip("reference checkpoint not present")
    from models.BMCNet_plain import BMCNet_plain
    m = BMCNet_plain(4, 128, 5)
    assert not any(m.load_state_dict(torch.load(ck, map_location="cpu"), strict=True))


def test_conv_specs_cover_every_reference_input_channel_once():
    from models.BMCNet import BMCNet
    from models.BMCNet_plain import BMCNet_plain
    for m in (BMCNet(4, 32, 1).neuro, BMCNet_plain(4, 32, 1).neuro):
        for name, spec in vars(m).items():
            if not name.startswith("_sp_"):
                continue
            used = sorted(c for c in spec.kmap_host if c >= 0)
            assert spec.kpad % 16 == 0 and spec.kreal == len(used)
            if spec.covers_all:
                assert used == list(range(spec.cin)), name
            else:       # one launch of a convolution evaluated as several launches over column subsets of ONE parameter
                assert name in ("_sp_fs_shared", "_sp_fs_h") and len(set(used)) == len(used) < spec.cin
    b = BMCNet(4, 32, 1).neuro
    assert b._sp_fpst.cin == b.conv_fpst.in_channels and b._sp_fps.cin == b.conv_fps.in_channels
    assert b._sp_o.cin == b.conv_o.in_channels
    # conv_fs: the shared launch and the per-state launch name disjoint columns that together are all of conv_fs.weight's
    both = sorted(c for sp in (b._sp_fs_shared, b._sp_fs_h) for c in sp.kmap_host if c >= 0)
    assert b._sp_fs_shared.cin == b._sp_fs_h.cin == b.conv_fs.in_channels and both == list(range(b.conv_fs.in_channels))
    p = BMCNet_plain(4, 32, 1).neuro
    assert p._sp_f1.cin == p.conv_f1.in_channels and p._sp_fs.cin == p.conv_fs.in_channels


def test_no_cpu_fallback():
    from models.BMCNet_plain import BMCNet_plain
    m = BMCNet_plain(4, 16, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 2, 2, 4, 4), torch.zeros(1, 16, 4, 4), torch.zeros(1, 32, 4, 4), True)
    from bmc_hip.encodings import events_to_channels
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        events_to_channels(torch.zeros(3), torch.zeros(3), torch.ones(3), (4, 4))


def test_unsupported_shapes_fail_loudly():
    from models.BMCNet import BMCNet
    BMCNet(2, 16, 1)           # x2 SR (config/train_nfs.yml: SCALE 2/4/8): 4 sub-pixel channels, padded to the 16-channel granule
    with pytest.raises(NotImplementedError):
        BMCNet(3, 16, 1)       # odd scales: scale^2 is not a multiple of 4
    with pytest.raises(NotImplementedError):
        BMCNet(4, 8, 1)        # n_c = 8 is not a multiple of 16


def test_stacked_weight_cache_follows_parameter_versions():
    """ops.stacked: the stacked per-group weights (v1 / v2) are rebuilt exactly when one of the parameters changed (version
    counter) or is another object (ids can be recycled: weak references), never otherwise."""
    from bmc_hip import ops
    a, b = torch.nn.Parameter(torch.randn(4, 4)), torch.nn.Parameter(torch.randn(4, 4))
    calls = []

    def build():
        calls.append(1)
        return torch.stack([a.detach(), b.detach()])
    s0 = ops.stacked((a, b), build)
    assert ops.stacked((a, b), build) is s0 and len(calls) == 1
    with torch.no_grad():
        a.add_(1.0)                                  # what optimizer.step() does: same object, new version
    s1 = ops.stacked((a, b), build)
    assert s1 is not s0 and len(calls) == 2 and torch.equal(s1[0], a.detach())
    c = torch.nn.Parameter(torch.randn(4, 4))
    s2 = ops.stacked((c, b), lambda: torch.stack([c.detach(), b.detach()]))
    assert s2 is not s1 and torch.equal(s2[0], c.detach())
    assert ops.stacked((a, b), build) is s1 and len(calls) == 2


def test_split_wgrad_plan_of_multi_source_convolutions():
    """ops.split_wgrad (host logic, no launch): dense 128-channel sources without a batch map go to the Winograd weight-gradient
    kernel with their own column offset, everything else into one sub-spec over the SAME weight tensor; single-source and
    all-narrow convolutions are not split; the plan is cached per view set."""
    from bmc_hip import ops
    sp = ops.ConvSpec([list(range(0, 128)), list(range(128, 256)), list(range(256, 272)), list(range(272, 288))])
    plain = [(0, 128, 0, None, 0), (0, 128, 0, None, 4), (0, 16, 0, None, 0), (0, 16, 0, None, 4)]
    big, rest, sub = ops.split_wgrad(sp, plain)
    assert big == [(0, 0), (1, 128)] and rest == [2, 3]
    assert sub.cin == sp.cin and sub.kmap_host == list(range(256, 288)) and not sub.covers_all
    assert ops.split_wgrad(sp, plain) is ops.split_wgrad(sp, plain)
    big, rest, sub = ops.split_wgrad(sp, [(0, 128, 1, 8, 0)] + plain[1:])          # source 0 read through a batch rotation
    assert big == [(1, 128)] and rest == [0, 2, 3] and sub.kmap_host[:128] == list(range(0, 128))
    # padded narrow sources (conv_fpst: 6 real channels in a 16-channel window) keep their -1 entries in the sub-spec
    sp2 = ops.ConvSpec([list(range(0, 6)) + [-1] * 10, list(range(6, 134)), list(range(134, 150))])
    big, rest, sub = ops.split_wgrad(sp2, [(0, 16, 0, None, 0), (0, 128, 0, None, 0), (0, 16, 0, None, 0)])
    assert big == [(1, 6)] and rest == [0, 2] and sub.kmap_host[:8] == [0, 1, 2, 3, 4, 5, -1, -1] and sub.cin == 150
    assert ops.split_wgrad(ops.ConvSpec.dense(128), [(0, 128, 0, None, 0)]) is None
    assert ops.split_wgrad(ops.ConvSpec.dense(16, 32), [(0, 16, 0, None, 0), (0, 32, 0, None, 0)]) is None
