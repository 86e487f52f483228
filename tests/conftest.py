import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=["fp32mfma", "bf16x6"])
def both_gemm_paths(request, monkeypatch):
    """Runs a sampler test once per arithmetic path of the denoiser's qkv / MLP Linears (include/after_hip.h:
    after_denoiser_set_gemm_path): the fp32 MFMA kernel at every size, and the bf16-split kernel (gemm_x6.hip)
    at every size -- the default dispatches between the two by the number of token rows, so together they cover
    it.  Read by after_denoiser_create; modules opt in with pytest.mark.usefixtures("both_gemm_paths")."""
    monkeypatch.setenv("AFTER_GEMM_X6", "0" if request.param == "fp32mfma" else "2")
    return request.param
