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


def test_gather_argument_rules_without_a_device():
    """nsnp_gather_check = the device-free part of nsnp_gather_results' validation, identical on every rank: a bad plan
    must fail everywhere before anything is posted (a root that returned early would leave its peers in an unmatched send)"""
    import numpy as np
    from nanosnp_amd import _lib
    lib = _lib.load()
    off = np.array([0, 40, 40, 100], np.int64)
    chk = lambda rank, world, nbytes, o, root: lib.nsnp_gather_check(rank, world, nbytes, o.ctypes.data if o is not None else None, root)
    assert chk(0, 3, 40, off, 0) == 0 and chk(1, 3, 0, off, 0) == 0 and chk(2, 3, 60, off, 2) == 0
    assert chk(0, 3, 41, off, 0) == -1                      # local_bytes disagrees with this rank's slot (byte_off mismatch)
    assert chk(2, 3, 40, off, 0) == -1
    assert chk(0, 3, 40, off, 3) == -1 and chk(0, 3, 40, off, -1) == -1          # root out of range
    assert chk(3, 3, 0, off, 0) == -1 and chk(-1, 3, 0, off, 0) == -1            # rank out of range
    assert chk(0, 0, 0, off, 0) == -1                                            # empty world
    assert chk(0, 3, 40, None, 0) == -1                                          # the table is required on every rank
    assert chk(0, 3, -1, off, 0) == -1
    assert chk(0, 2, 40, np.array([0, 40, 30], np.int64), 0) == -1               # offsets must not decrease
    assert chk(0, 2, 39, np.array([1, 40, 50], np.int64), 0) == -1               # and start at 0
    # without a communicator (and without a context) the entry itself refuses
    assert lib.nsnp_gather_results(None, None, 0, None, off.ctypes.data, 0, None) == -1
    assert lib.nsnp_comm_attach(None, None, 0, 1) == -1
