#!/usr/bin/env python3
"""Memory-pattern ceilings for k_hap_features at L = 33, D = 90, int32 planes (tools/probes/feat_skeleton.hip): loads + stores with
almost no arithmetic in the kernel's mapping (one 132-byte row per wave load) and with four rows per 16-byte wave load, against the
product kernel on the same planes."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
so = os.path.join(ROOT, "tools", "probes", "libfeat_skeleton.so")
src = os.path.join(ROOT, "tools", "probes", "feat_skeleton.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, src], check=True)
lib = C.CDLL(so)
from nanosnp_amd import _lib
N, D, L = 16384, 90, 33
g = torch.Generator(device="cuda").manual_seed(1)
planes = [torch.randint(-2, 5, (N, D, L), dtype=torch.int32, device="cuda", generator=g) for _ in range(4)]
out = torch.empty((N, 105, L), dtype=torch.float32, device="cuda")
ref_row = torch.zeros((N, L), dtype=torch.int32, device="cuda")
ctx = _lib.Context(0)
s = torch.cuda.current_stream().cuda_stream
bytes_ = N * (4 * D * L * 4 + 105 * L * 4)
def t(fn, reps=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
p = [C.c_void_p(x.data_ptr()) for x in planes]
ms = t(lambda: ctx.hap_features(planes[0], planes[1], planes[2], planes[3], ref_row))
print(f"product kernel                           {ms:.4f} ms  {bytes_ / ms / 1e9:.2f} TB/s  ({bytes_ / ms / 8e9:.3f} of 8 TB/s)")
for which, name in ((0, "rows of 33 lanes (product mapping), row stores"), (1, "4 rows per 16-B load, row stores"), (2, "4 rows per 16-B load, flat 8-B stores")):
    for U in ((2, 4, 6) if which == 0 else (1, 2, 3)):
        ms = t(lambda: lib.feat_skeleton(which, U, *p, C.c_int64(N), D, C.c_void_p(out.data_ptr()), C.c_void_p(s)))
        print(f"skeleton: {name:48s} U={U}  {ms:.4f} ms  {bytes_ / ms / 1e9:.2f} TB/s")
ms = t(lambda: lib.feat_skeleton(3, 0, *p, C.c_int64(N), D, C.c_void_p(out.data_ptr()), C.c_void_p(s)))
print(f"stores only (row stores of 33 lanes)     {ms:.4f} ms  {N * 105 * L * 4 / ms / 1e9:.2f} TB/s of writes")
