#!/usr/bin/env python3
"""What would cheaper correction terms cost in accuracy?  numpy model of the PileupModel forward (real weights, golden inputs)
with every product evaluated as f16x3 (what the kernels do), with the two correction terms in fp8 e4m3 (scaled by 2^11), with
the h_lo term dropped, and in plain fp16.  DESIGN.md section 9."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanosnp_amd.fixtures import load_pileup_weights
from tests.helpers import golden
w = load_pileup_weights()
z = np.load(golden("pileup_fwd.npz"))
x = z["x"].astype(np.float32)[:256]
def f16(a): return a.astype(np.float16).astype(np.float32)
def split(a):
    hi = f16(a); lo = f16(a - hi); return hi, lo
def fp8_e4m3(a):
    # round to 1-4-3 (max 448, min subnormal 2^-9), round-to-nearest-even via scaling trick
    a = np.asarray(a, np.float32); s = np.sign(a); m = np.abs(a)
    m = np.minimum(m, 448.0)
    e = np.floor(np.log2(np.maximum(m, 2.0**-20)))
    e = np.maximum(e, -6.0)                      # subnormals share the exponent of 2^-6
    q = 2.0 ** (e - 3)                           # 3 mantissa bits
    return (s * np.round(m / q) * q).astype(np.float32)
SITE_MODES = {}
def matvec(W, v, mode, site="other"):
    mode = SITE_MODES.get(site, mode)
    # W [R,K], v [N,K] -> [N,R]
    Wh, Wl = split(W); vh, vl = split(v)
    main = vh @ Wh.T
    if mode == "fp32": return v @ W.T
    if mode == "f16x3": return main + vh @ Wl.T + vl @ Wh.T
    if mode == "f16x1": return main
    if mode == "fp8corr":
        S = 2.0 ** 11
        return main + (fp8_e4m3(vh) @ fp8_e4m3(Wl * S).T + fp8_e4m3(vl * S) @ fp8_e4m3(Wh).T) / S
    if mode == "f16hi_wlo":  # keep W_lo term in fp16, drop h_lo
        return main + vh @ Wl.T
    raise ValueError(mode)
def sig(a): return 1 / (1 + np.exp(-a))
def lstm_dir(xs, Wih, Whh, bih, bhh, rev, mode, steps=None, site="l0"):
    N, T, _ = xs.shape; H = Whh.shape[1]
    h = np.zeros((N, H), np.float32); c = np.zeros((N, H), np.float32); out = np.zeros((N, T, H), np.float32)
    order = range(T - 1, -1, -1) if rev else range(T)
    for k, t in enumerate(order):
        if steps is not None and k >= steps: break
        g = matvec(Wih, xs[:, t], mode, site + "_ih") + matvec(Whh, h, mode, site + "_hh") + bih + bhh
        i, f, gg, o = sig(g[:, :H]), sig(g[:, H:2*H]), np.tanh(g[:, 2*H:3*H]), sig(g[:, 3*H:])
        c = f * c + i * gg; h = o * np.tanh(c); out[:, t] = h
    return out
def forward(mode):
    h0 = np.concatenate([lstm_dir(x, *w[0:4], False, mode), lstm_dir(x, *w[4:8], True, mode)], 2)
    h1 = np.concatenate([lstm_dir(h0, *w[8:12], False, mode, 17, "l1"), lstm_dir(h0, *w[12:16], True, mode, 17, "l1")], 2)[:, 16]
    p = matvec(w[16], h1, mode) + w[17]
    d = np.tanh(matvec(w[18], p, mode) + w[19])
    lg = matvec(w[20], d, mode) + w[21]; lz = matvec(w[22], d, mode) + w[23]
    sm = lambda a: np.exp(a - a.max(1, keepdims=True)) / np.exp(a - a.max(1, keepdims=True)).sum(1, keepdims=True)
    return sm(lg), sm(lz)
ref = forward("fp32")
print("numpy fp32 vs golden:", np.abs(ref[0] - z["gt"][:256]).max())
for mode in ("f16x3", "fp8corr", "f16hi_wlo", "f16x1"):
    g = forward(mode)
    print(f"{mode:10s} max |dp| vs fp32: gt {np.abs(g[0]-ref[0]).max():.2e}  zy {np.abs(g[1]-ref[1]).max():.2e}   argmax flips {int((g[0].argmax(1)!=ref[0].argmax(1)).sum())}")

for name, sm in (("layer-1 input projection without h0_lo", {"l1_ih": "f16hi_wlo"}), ("layer-1 input projection with fp8 corrections", {"l1_ih": "fp8corr"}),
                 ("layer-1 input projection in plain fp16", {"l1_ih": "f16x1"})):
    SITE_MODES.clear(); SITE_MODES.update(sm)
    g = forward("f16x3")
    print(f"f16x3 except {name}: gt {np.abs(g[0]-ref[0]).max():.2e}  zy {np.abs(g[1]-ref[1]).max():.2e}")
