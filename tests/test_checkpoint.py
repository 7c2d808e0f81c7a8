"""Checkpoint / resume host logic (CPU): bare state_dict compatibility + bit-exact optimizer/scheduler resume."""
import os

import pytest
import torch


def _toy():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=1e-5, amsgrad=True)
    sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.95)
    return m, opt, sch


def _train(m, opt, sch, steps, seed):
    g = torch.Generator().manual_seed(seed)
    for i in range(steps):
        x = torch.randn(4, 6, generator=g)
        opt.zero_grad()
        m(x).pow(2).mean().backward()
        opt.step()
        if i % 2 == 1:
            sch.step()


def test_resume_continues_bit_exactly(tmp_path):
    from checkpoint import resume, save_checkpoint
    m, opt, sch = _toy()
    _train(m, opt, sch, 4, seed=1)
    p = str(tmp_path / "checkpoint-iteration4.pth")
    save_checkpoint(p, m, opt, sch, iteration=4, monitor_best=0.5)
    _train(m, opt, sch, 3, seed=2)
    ref = [q.detach().clone() for q in m.parameters()]
    m2, opt2, sch2 = _toy()
    tr = resume(p, m2, opt2, sch2)
    assert tr["iteration"] == 4 and tr["monitor_best"] == 0.5 and tr["training_mode"] == "iteration_based_train"
    _train(m2, opt2, sch2, 3, seed=2)
    for a, b in zip(ref, m2.parameters()):
        assert torch.equal(a, b)
    assert sch2.get_last_lr() == sch.get_last_lr()


def test_bare_file_is_reference_format_and_keeps_alias_keys(tmp_path):
    from checkpoint import load_model_state, resume, save_checkpoint
    from models.BMCNet_plain import BMCNet_plain
    m = BMCNet_plain(4, 16, 2)
    p = str(tmp_path / "model_best_until_iteration7.pth")
    save_checkpoint(p, m)                                     # model only -> just the bare file
    assert not os.path.exists(p + ".train")
    sd = torch.load(p, map_location="cpu")
    assert list(sd.keys()) == list(m.state_dict().keys()) and len(sd) > len(list(m.parameters()))     # alias keys present
    m2 = BMCNet_plain(4, 16, 2)
    assert not any(load_model_state(p, m2))
    assert resume(p, m2)["iteration"] == -1
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def test_reference_pretrained_checkpoint_roundtrip(tmp_path):
    ck = "/root/reference/pretrain/BMCNet_plain_nfs_x4.pth"
    if not os.path.exists(ck):
        pytest.skip("reference checkpoint not present")
    from checkpoint import load_model_state, save_checkpoint
    from models.BMCNet_plain import BMCNet_plain
    m = BMCNet_plain(4, 128, 5)
    load_model_state(ck, m)
    p = str(tmp_path / "re.pth")
    save_checkpoint(p, m)
    a, b = torch.load(ck, map_location="cpu"), torch.load(p, map_location="cpu")
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
