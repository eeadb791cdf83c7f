"""Row N4: the sync-free Logger writes the records the reference's graphs/logger.py writes (fixture: oracle/make_golden.py
logger_fixture(), produced by the reference logger itself on a seeded 25-step stream)."""
import json
import os

import pytest
import torch

from conftest import GOLDEN


def _close(a, b, path=""):
    if isinstance(b, dict):
        assert isinstance(a, dict) and list(a.keys()) == list(b.keys()), (path, list(a.keys()), list(b.keys()))
        for k in b:
            _close(a[k], b[k], f"{path}.{k}")
    elif isinstance(b, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _close(x, y, f"{path}[{i}]")
    elif isinstance(b, float):
        assert a == pytest.approx(b, rel=2e-5, abs=2e-6), (path, a, b)
    else:
        assert a == b, (path, a, b)


def _run(dev, tmp_path):
    from graphs.logger import Logger
    fx = torch.load(os.path.join(GOLDEN, "logger.pt"), weights_only=False)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.Unet_experts = torch.nn.ModuleList([torch.nn.Conv2d(3, 5, 3) for _ in range(2)])
            self.VIT_experts = torch.nn.ModuleList([torch.nn.Linear(7, 4) for _ in range(3)])
            self.Unet_router = torch.nn.Linear(6, 4)
            self.vit_router = torch.nn.Linear(6, 4)
            self.cross_attn = torch.nn.Linear(5, 5)
    net = Net()
    net.load_state_dict(fx["net_state"])
    net = net.to(dev)
    lg = Logger(log_dir=str(tmp_path), run_name="fx", log_interval=10)
    d = lambda t: t.to(dev)
    for st, gs in zip(fx["steps"], fx["grads"]):
        for p, g in zip(net.parameters(), gs):
            p.grad = d(g.clone())
        lg.log_training_step(step=st["step"], loss_dict={k: d(v) for k, v in st["loss"].items()}, zeta=st["zeta"],
                             log_var=torch.tensor(st["log_var"], device=dev), lr=st["lr"], sigma=d(st["sigma"]), p_mean=-1.2, p_std=1.6)
        lg.log_router_statistics(step=st["step"], unet_probs=d(st["unet_probs"]), vit_probs=d(st["vit_probs"]), sigma=d(st["sigma"]),
                                 p_mean=-1.2, p_std=1.6)
        lg.log_scaling_gating(scaling_factors=d(st["scaling"]), gate_weights=d(st["gate"]), sigma=d(st["sigma"]))
        lg.log_gradients(step=st["step"], model=net)
        lg.log_weight_statistics(step=st["step"], model=net)
    for k, ref in fx["files"].items():
        got = [json.loads(l) for l in open(getattr(lg, k))]
        _close(got, ref, k)


def test_logger_matches_reference_records(tmp_path):
    _run("cpu", tmp_path)


@pytest.mark.gpu
def test_logger_matches_reference_records_on_device(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _run("cuda", tmp_path)
