"""pytest plumbing: the `gpu` marker and import paths.

* repo root                      -> `oracle.*` (checker; tests only)
* heterogeneous-moe-..._amd/     -> `models.*` (drop-in mirror) and `hdmoe_hip.*`
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_components():
    import torch
    return torch.load(os.path.join(GOLDEN, "components.pt"), weights_only=False)


@pytest.fixture(scope="session", params=[1, 2])
def golden_full(request):
    import torch
    return torch.load(os.path.join(GOLDEN, f"full_config{request.param}.pt"), weights_only=False)


@pytest.fixture(scope="session")
def golden_sampler():
    import torch
    return torch.load(os.path.join(GOLDEN, "sampler.pt"), weights_only=False)


@pytest.fixture(scope="session", params=[1, 2, 3, 4])
def golden_wide(request):
    """BASELINE configs 1-4 at their real widths: outputs of the reference under oracle/recipe.py's weights/inputs."""
    import torch
    return torch.load(os.path.join(GOLDEN, f"wide_config{request.param}.pt"), weights_only=False)


def wide_setup(g):
    """(model class, constructor kwargs, recipe state_dict, recipe inputs) of a wide fixture; modules built on CPU."""
    from Utils import configs
    from models import model_config1, model_config2
    from oracle.recipe import fill_state, make_inputs
    bc = configs.BASELINE_CONFIGS[g["cfg_id"]]
    kw = configs.model_kwargs(**bc["over"])
    cls = (model_config1 if bc["module"] == 1 else model_config2).preconditioned_HDMOEM
    model = cls(**kw)
    state = fill_state(model.state_dict(), g["seed"])
    inp = make_inputs(g["B"], kw["IN_in_channels"], kw["IN_img_resolution"], kw["num_experts"], 77, kw["text_emb_dim"], g["seed"])
    return bc["module"], model, kw, state, inp
