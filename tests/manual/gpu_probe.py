#!/usr/bin/env python3
"""Development probe run through gpurun: parity of every HIP entry point against the goldens /
oracle plus rough timings.  Not part of the test suite (tests/ holds the real checks)."""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nanosnp_amd import _lib, host
from oracle import oracle
from tests.helpers import golden, load_pileup_weights

dev = torch.device("cuda:0")
ctx = _lib.Context(0)
print("device", torch.cuda.get_device_name(0))

# ---- forward vs golden
w = load_pileup_weights()
ctx.pileup_load_weights(w)
z = np.load(golden("pileup_fwd.npz"))
x = torch.from_numpy(z["x"].astype(np.int32)).to(dev)
gt, zy = ctx.pileup_forward(x)
torch.cuda.synchronize()
gt, zy = gt.cpu().numpy(), zy.cpu().numpy()
print("pileup fwd vs golden: max|d| gt %.3g zy %.3g argmax agree %.4f" % (
    np.abs(gt - z["gt"]).max(), np.abs(zy - z["zy"]).max(), (gt.argmax(1) == z["gt"].argmax(1)).mean()))
for n in (1, 15, 17, 127, 129):
    g2, z2 = ctx.pileup_forward(x[:n].contiguous())
    print("  N=%d max|d| %.3g" % (n, np.abs(g2.cpu().numpy() - z["gt"][:n]).max()))

# ---- encode vs golden .pd
for tag in ("g1", "adv"):
    text = gzip.open(golden(f"encode_{tag}.mpileup.gz")).read()
    fa = gzip.open(golden(f"encode_{tag}.fa.gz")).read()
    pd = gzip.open(golden(f"encode_{tag}.pd.gz")).read()
    seq = np.frombuffer(b"".join(fa.split(b"\n")[1:]), np.uint8)
    pos, col_off, bases = host.mpileup_parse(text)
    ref = seq[pos - 1]
    oc, od, of = oracle.encode_columns(bases, col_off, ref)
    c, d, f = ctx.pileup_encode_columns(torch.from_numpy(bases).to(dev), torch.from_numpy(col_off).to(dev),
                                        torch.from_numpy(np.ascontiguousarray(ref)).to(dev))
    torch.cuda.synchronize()
    print(tag, "encode counts", np.array_equal(c.cpu().numpy(), oc), "depth", np.array_equal(d.cpu().numpy(), od),
          "flags", np.array_equal(f.cpu().numpy(), of))
    if not np.array_equal(c.cpu().numpy(), oc):
        bad = np.nonzero((c.cpu().numpy() != oc).any(1))[0]
        print("  bad columns", bad[:10], len(bad))
        b0 = bad[0]; print(bases[col_off[b0]:col_off[b0+1]].tobytes()); print(c.cpu().numpy()[b0]); print(oc[b0])
    centers, n = ctx.pileup_select_sites(torch.from_numpy(pos).to(dev), f)
    xs = ctx.pileup_gather_windows(c, centers)
    gx, names, gpos, gref = host.pd_parse(pd)
    print("  sites", n, "x equal golden", np.array_equal(xs.cpu().numpy(), gx), "pos", np.array_equal(pos[centers.cpu().numpy()], gpos))
    g1, z1 = ctx.pileup_forward(xs)
    g2, z2 = ctx.pileup_forward_windows(c, centers)
    print("  forward_windows == forward", torch.equal(g1, g2), torch.equal(z1, z2))

# ---- hap features vs golden
z = np.load(golden("hap_features.npz"))
for tag in "ph":
    arrs = [torch.from_numpy(z[f"{tag}_{k}"].astype(np.int32)).to(dev) for k in ("seq", "bq", "mq", "hap", "ref")]
    out = ctx.hap_features(*arrs)
    ref = np.concatenate([z[f"{tag}_feat"], z[f"{tag}_ref"].astype(np.float64)[:, None, :]], axis=1).astype(np.float32)
    print("hap features", tag, "bit-equal to golden(float32 cast):", np.array_equal(out.cpu().numpy(), ref))

# ---- timing
def bench(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / iters

for N in (4096, 32768, 131072):
    xx = torch.randint(-20, 40, (N, 33, 18), dtype=torch.int32, device=dev)
    ctx.reserve(min(N, 65536))
    gt = torch.empty((N, 21), device=dev); zy = torch.empty((N, 3), device=dev)
    t = bench(lambda: ctx.pileup_forward(xx, gt, zy))
    print("forward N=%d: %.3f ms  %.2f M sites/s" % (N, t * 1e3, N / t / 1e6))

M = 33 * 65536
cols = host.synth_columns(20260001, M, coverage=30, window=33)
b = torch.from_numpy(cols.bases).to(dev); co = torch.from_numpy(cols.col_off).to(dev); rf = torch.from_numpy(cols.ref).to(dev)
t = bench(lambda: ctx.pileup_encode_columns(b, co, rf))
print("encode M=%d cols (%.1f MB): %.3f ms  %.2f G cols/s  %.1f GB/s alg" % (M, cols.bases.size / 1e6, t * 1e3, M / t / 1e9,
      (cols.bases.size + M * (1 + 72)) / t / 1e9))
planes = host.synth_hap_planes(5, 4096, 30, 90, 33)
pt = [torch.from_numpy(a).to(dev) for a in planes]
t = bench(lambda: ctx.hap_features(*pt))
print("hap features N=4096 D=90 L=33: %.3f ms  %.2f M sites/s  %.1f GB/s" % (t * 1e3, 4096 / t / 1e6, 4096 * (4 * 90 * 33 * 4 + 105 * 33 * 4) / t / 1e9))

# ---- hap forward timing
from tests.helpers import seeded_hap_weights
ws = seeded_hap_weights(12, H=256)
ctx.hap_load_weights(ws)
for N, prec in ((512, 0), (4096, 0), (4096, 1), (16384, 1)):
    ctx.set_option("hap_precision", prec)
    xp = torch.randn((N, 105, 33), device=dev) * 100
    xh = torch.randn((N, 105, 11), device=dev) * 100
    t = bench(lambda: ctx.hap_forward(xp, xh), iters=3)
    print("hap forward precision %d" % prec, end=" "); print("N=%d: %.2f ms  %.1f k sites/s  %.1f TFLOP/s (353.7 MFLOP/site alg)" % (N, t * 1e3, N / t / 1e3, 353.7e6 * N / t / 1e12))
