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


def test_rccl_gather_entry_world_one(gpu_ctx):
    """nsnp_comm_* / nsnp_gather_results (the optional C-ABI gather over RCCL): a one-rank communicator on the GPU box - the unique
    id, ncclCommInitRank through the run-time resolved library, the root's own block copied into place, empty blocks allowed.
    (Two ranks need two GPUs: RCCL refuses a second rank on the same device; the PyTorch ranks of bench.py gather through
    torch.distributed, see include/nanosnp.h.)"""
    import torch
    from nanosnp_amd import _lib
    c = _lib.Context(0)
    uid = _lib.Context.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    c.comm_init(uid, 0, 1)
    x = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
    out = c.gather_bytes(x, [x.numel() * 4])
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.float32), x)
    empty = torch.empty(0, dtype=torch.float32, device="cuda")
    out0 = c.gather_bytes(empty, [0])
    assert out0.numel() == 0
    with pytest.raises(_lib.NanoSNPError):
        c.comm_init(uid, 0, 1)                        # one communicator per context
    with pytest.raises(_lib.NanoSNPError):
        gpu_ctx.gather_bytes(x, [x.numel() * 4]) if hasattr(gpu_ctx, "_comm") else _lib.check(_lib.load().nsnp_gather_results(
            gpu_ctx.handle, None, 0, None, None, 0, None), gpu_ctx.handle, "nsnp_gather_results")     # no communicator bound
    c.close()
