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
