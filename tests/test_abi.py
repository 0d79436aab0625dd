"""The C-ABI library loads without a GPU and exports every symbol the headers declare."""
import ctypes
import os
import re

from tests.helpers import ROOT


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nsnp_[a-z0-9_]+)\s*\(", src)))


def test_hip_library_exports_every_declared_symbol():
    names = _declared("nanosnp.h")
    assert len(names) >= 18
    lib = ctypes.CDLL(os.path.join(ROOT, "nanosnp_amd", "libnanosnp_hip.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_host_library_exports_every_declared_symbol():
    names = _declared("nsnp_host.h")
    lib = ctypes.CDLL(os.path.join(ROOT, "nanosnp_amd", "libnanosnp_host.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    from nanosnp_amd import _lib
    assert set(_declared("nanosnp.h")) == set(_lib.EXPORTS)
    lib = _lib.load()
    assert lib.nsnp_version() >= 100
    assert lib.nsnp_strerror(-5).decode() == "device is not gfx950"


def test_no_cpu_fallback():
    """Without a GPU the product path refuses to run instead of routing through the oracle."""
    import pytest
    import torch
    from nanosnp_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.NanoSNPError):
        _lib.Context(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nanosnp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".c", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f
