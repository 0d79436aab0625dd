import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Everything native is built in-tree once per session (no-op when up to date)."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    from nanosnp_amd import _lib
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def pileup_weights():
    from tests.helpers import load_pileup_weights
    return load_pileup_weights()


@pytest.fixture(params=["device", "host"])
def tok_mode(request, monkeypatch):
    """the text pipelines with the mpileup text cut into columns on the device (nsnp_mpileup_tokenise, the default) and on the host cores
    (nsnp_mpileup_parse_into): both must give the same bytes"""
    monkeypatch.setenv("NSNP_TOKENISE", request.param)
    return request.param
