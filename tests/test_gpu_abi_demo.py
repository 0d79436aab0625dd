"""The C ABI driven from a plain C program (examples/abi_demo.c: no Python, no PyTorch in that process) gives the same calls as
the Python binding on the same synthetic input."""
import os
import subprocess

import numpy as np
import pytest

from nanosnp_amd import host

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_through_the_abi(tmp_path, pileup_weights):
    import torch
    from nanosnp_amd import _lib
    exe = tmp_path / "abi_demo"
    libdir = os.path.join(ROOT, "nanosnp_amd")
    cc = subprocess.run(["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "examples", "abi_demo.c"), "-o", str(exe), "-L" + libdir, "-lnanosnp_hip", "-lnanosnp_host",
                         "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    wfile = tmp_path / "weights.f32"
    np.concatenate([np.ascontiguousarray(w, np.float32).ravel() for w in pileup_weights]).tofile(wfile)
    n = 777
    run = subprocess.run([str(exe), str(wfile), str(n)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    rows = [l.split("\t") for l in run.stdout.splitlines()]
    assert len(rows) == n
    # the same input through the Python binding
    cols = host.synth_columns(4242, n * 33, coverage=30.0, max_depth=144, het_rate=0.02, window=33)
    ctx = _lib.Context(0)
    ctx.pileup_load_weights(pileup_weights)
    counts, depth, flags = ctx.pileup_encode_columns(torch.from_numpy(cols.bases).cuda(), torch.from_numpy(cols.col_off).cuda(),
                                                     torch.from_numpy(cols.ref).cuda())
    centers = torch.arange(n, dtype=torch.int64, device="cuda") * 33 + 16
    gt, zy = ctx.pileup_forward_windows(counts, centers)
    ga, za, gm, zm, _ = ctx.pileup_postprocess(gt, zy)
    torch.cuda.synchronize()
    assert [int(r[1]) for r in rows] == ga.cpu().tolist() and [int(r[2]) for r in rows] == za.cpu().tolist()
    assert np.allclose([float(r[3]) for r in rows], gm.cpu().numpy(), atol=2e-6)
    assert np.allclose([float(r[4]) for r in rows], zm.cpu().numpy(), atol=2e-6)
    assert len(set(int(r[1]) for r in rows)) > 3            # the demo input is not degenerate
    ctx.close()
