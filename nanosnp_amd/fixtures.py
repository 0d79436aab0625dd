"""Weights and synthetic model inputs used by bench.py, __graft_entry__.smoke(), tools/ and the tests.

* ``load_pileup_weights`` -- the values of the shipped ``PileupModel/models/ont_pileup.chkpt`` as a fixture
  (``nanosnp_amd/data/ont_pileup_weights.npz``, shipped with the package; written by tests/golden/make_golden.py from the
  checkpoint; data, not code).  Nothing here reads the test tree.
* ``seeded_hap_weights`` / ``seeded_cat_weights`` -- stand-ins for the HaplotypeModel checkpoints that are absent from
  the reference tree (``.MISSING_LARGE_BLOBS``): numpy PCG64 streams (stable across platforms) in state-dict order, the
  exact arrays tests/golden/make_golden.py loaded into the reference's modules when it recorded their outputs.
* ``synth_cat_groups`` -- legacy CatModel group tensors shaped like ``dataset.PredictDataset`` builds them.
"""
from __future__ import annotations

import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

PILEUP_WEIGHT_KEYS = (
    [f"encoder.lstm.{n}_l{l}{d}" for l in (0, 1) for d in ("", "_reverse")
     for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    + ["encoder.output_proj.weight", "encoder.output_proj.bias",
       "forward_layer.dense.weight", "forward_layer.dense.bias",
       "forward_layer.genotype_layer.weight", "forward_layer.genotype_layer.bias",
       "forward_layer.zygosity_layer.weight", "forward_layer.zygosity_layer.bias"]
)



def load_pileup_weights(path=None):
    """The 24 tensors LSTMNetwork.predict uses, in state-dict order."""
    z = np.load(path or os.path.join(DATA, "ont_pileup_weights.npz"))
    return [np.ascontiguousarray(z[k], dtype=np.float32) for k in PILEUP_WEIGHT_KEYS]


def hap_weight_names(n_layers=3):
    names = []
    for enc in ("pileup_encoder", "haplotype_encoder"):
        for l in range(n_layers):
            for d in ("", "_reverse"):
                for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    names.append(f"{enc}.lstm.{n}_l{l}{d}")
        names += [f"{enc}.output_proj.weight", f"{enc}.output_proj.bias"]
    names += ["forward_layer.dense.weight", "forward_layer.dense.bias",
              "forward_layer.genotype_layer.weight", "forward_layer.genotype_layer.bias",
              "forward_layer.zygosity_layer.weight", "forward_layer.zygosity_layer.bias"]
    return names


def hap_weight_shapes(F=105, H=256, n_layers=3, n_gt=10, n_zy=3):
    shapes = {}
    for enc in ("pileup_encoder", "haplotype_encoder"):
        for l in range(n_layers):
            I = F if l == 0 else 2 * H
            for d in ("", "_reverse"):
                shapes[f"{enc}.lstm.weight_ih_l{l}{d}"] = (4 * H, I)
                shapes[f"{enc}.lstm.weight_hh_l{l}{d}"] = (4 * H, H)
                shapes[f"{enc}.lstm.bias_ih_l{l}{d}"] = (4 * H,)
                shapes[f"{enc}.lstm.bias_hh_l{l}{d}"] = (4 * H,)
        shapes[f"{enc}.output_proj.weight"] = (H, 2 * H)
        shapes[f"{enc}.output_proj.bias"] = (H,)
    shapes["forward_layer.dense.weight"] = (H, 2 * H)
    shapes["forward_layer.dense.bias"] = (H,)
    shapes["forward_layer.genotype_layer.weight"] = (n_gt, H)
    shapes["forward_layer.genotype_layer.bias"] = (n_gt,)
    shapes["forward_layer.zygosity_layer.weight"] = (n_zy, H)
    shapes["forward_layer.zygosity_layer.bias"] = (n_zy,)
    return shapes


def seeded_hap_weights(seed, F=105, H=256, n_layers=3, n_gt=10, n_zy=3, ih_scale=0.002, head_scale=8.0):
    """U(-1/sqrt(H), 1/sqrt(H)) like torch's default LSTM/Linear init, from numpy PCG64.
    The input-layer weights are scaled down so that count-valued features (up to ~5000)
    do not saturate every gate."""
    rng = np.random.default_rng(seed)
    shapes = hap_weight_shapes(F, H, n_layers, n_gt, n_zy)
    k = 1.0 / np.sqrt(H)
    out = []
    for name in hap_weight_names(n_layers):
        w = rng.uniform(-k, k, size=shapes[name]).astype(np.float32)
        if name.endswith("weight_ih_l0") or name.endswith("weight_ih_l0_reverse"):
            w *= np.float32(ih_scale)
        if "genotype_layer.weight" in name or "zygosity_layer.weight" in name:
            w *= np.float32(head_scale)   # spread the logits so that parity errors are visible
        out.append(w)
    return out


# the seeded HaplotypeModel weights tests/golden/make_golden.py `twostage` loaded into the reference's module when it wrote two_stage.npz
TWO_STAGE_HAP_WEIGHTS = dict(seed=13, H=256, ih_scale=0.03, head_scale=120.0)


# ---- legacy CatModel (HaplotypeModel/model.py:201-360, crnn.py:84-190) ---------------------------
CAT_CHANNELS = (10, 32, 64, 128, 128, 256, 256)


def cat_weight_names():
    """Floating-point tensors of CatModel.state_dict() in order (num_batches_tracked skipped)."""
    names = []
    for i in range(6):
        b = f"haplotype_base.cnn.conv{i}.base.conv{i}_base_"
        names += [b + "conv1.weight", b + "conv1.bias"]
        names += [b + "bn1." + s for s in ("weight", "bias", "running_mean", "running_var")]
        names += [b + "conv2.weight", b + "conv2.bias"]
        names += [b + "bn2." + s for s in ("weight", "bias", "running_mean", "running_var")]
        s = f"haplotype_base.cnn.conv{i}.shortcut.conv{i}_shortcut_conv1."
        names += [s + "weight", s + "bias"]
    for r in range(2):
        for sfx in ("", "_reverse"):
            names += [f"haplotype_base.rnn.{r}.rnn.{k}_l0{sfx}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        names += [f"haplotype_base.rnn.{r}.embedding.weight", f"haplotype_base.rnn.{r}.embedding.bias"]
    for l in range(3):
        for sfx in ("", "_reverse"):
            names += [f"haplotype_percentage.rnn.{k}_l{l}{sfx}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    names += ["haplotype_percentage.out_layer.weight", "haplotype_percentage.out_layer.bias",
              "out_layer.weight", "out_layer.bias"]
    return names


def cat_weight_shapes(H=256):
    sh = {}
    for i in range(6):
        ci, co = CAT_CHANNELS[i], CAT_CHANNELS[i + 1]
        b = f"haplotype_base.cnn.conv{i}.base.conv{i}_base_"
        sh[b + "conv1.weight"] = (co, ci, 3, 3); sh[b + "conv1.bias"] = (co,)
        sh[b + "conv2.weight"] = (co, co, 3, 3); sh[b + "conv2.bias"] = (co,)
        for bn in ("bn1.", "bn2."):
            for s in ("weight", "bias", "running_mean", "running_var"):
                sh[b + bn + s] = (co,)
        s = f"haplotype_base.cnn.conv{i}.shortcut.conv{i}_shortcut_conv1."
        sh[s + "weight"] = (co, ci, 1, 1); sh[s + "bias"] = (co,)
    for r in range(2):
        for sfx in ("", "_reverse"):
            p = f"haplotype_base.rnn.{r}.rnn."
            sh[p + "weight_ih_l0" + sfx] = (4 * H, H); sh[p + "weight_hh_l0" + sfx] = (4 * H, H)
            sh[p + "bias_ih_l0" + sfx] = (4 * H,); sh[p + "bias_hh_l0" + sfx] = (4 * H,)
        sh[f"haplotype_base.rnn.{r}.embedding.weight"] = (H, 2 * H); sh[f"haplotype_base.rnn.{r}.embedding.bias"] = (H,)
    for l in range(3):
        for sfx in ("", "_reverse"):
            p = "haplotype_percentage.rnn."
            sh[p + f"weight_ih_l{l}{sfx}"] = (4 * H, 20 if l == 0 else 2 * H); sh[p + f"weight_hh_l{l}{sfx}"] = (4 * H, H)
            sh[p + f"bias_ih_l{l}{sfx}"] = (4 * H,); sh[p + f"bias_hh_l{l}{sfx}"] = (4 * H,)
    sh["haplotype_percentage.out_layer.weight"] = (H, 2 * H); sh["haplotype_percentage.out_layer.bias"] = (H,)
    sh["out_layer.weight"] = (10, 2 * H); sh["out_layer.bias"] = (10,)
    return sh


def seeded_cat_weights(seed):
    """torch-default-like init (U(-1/sqrt(fan_in), ..)) with non-trivial BatchNorm statistics."""
    rng = np.random.default_rng(seed)
    shapes = cat_weight_shapes()
    out = []
    for name in cat_weight_names():
        s = shapes[name]
        if name.endswith("running_var"):
            w = rng.uniform(0.5, 1.5, size=s)
        elif name.endswith("running_mean"):
            w = rng.normal(0.0, 0.2, size=s)
        elif ".bn" in name and name.endswith("weight"):
            w = rng.normal(1.0, 0.1, size=s)
        elif ".bn" in name and name.endswith("bias"):
            w = rng.normal(0.0, 0.1, size=s)
        else:
            fan_in = int(np.prod(s[1:])) if len(s) > 1 else 256
            if ".rnn." in name and "embedding" not in name:
                fan_in = 256
            k = 1.0 / np.sqrt(fan_in)
            w = rng.uniform(-k, k, size=s)
            # scaled so that the ten probabilities differ visibly from site to site (the default init gives
            # a nearly constant, bias-dominated output that would hide input-dependent parity errors)
            if ".rnn." in name and "weight" in name and "embedding" not in name:
                w *= 4.0
            if name.startswith("haplotype_percentage.rnn.weight_ih_l0"):
                w *= 10.0
            if "embedding.weight" in name or name == "haplotype_percentage.out_layer.weight":
                w *= 6.0
            if name == "out_layer.weight":
                w *= 10.0
            if name.endswith("bias") and ("embedding" in name or "out_layer" in name):
                w *= 0.0
        out.append(w.astype(np.float32))
    return out


def synth_cat_groups(seed, N, L=11):
    """g0 (surrounding columns) and g1 (adjacent heterozygous sites) as dataset.PredictDataset builds them
    (HaplotypeModel/dataset.py:862-915): [N,40,L,5] = per tag 20 rows of (base, baseq, mapq, mask, phase);
    base: -2 padding, -1 deletion, 0 not covered, 1..4 ACGT."""
    rng = np.random.default_rng(seed)
    gs = []
    for _ in range(2):
        g = np.zeros((N, 40, L, 5), np.float32)
        for n in range(N):
            for tag in range(2):
                depth = int(rng.integers(0 if n % 7 == 3 else 2, 21))
                rows = slice(tag * 20, tag * 20 + 20)
                base = np.full((20, L), -2, np.int64)
                cons = rng.integers(1, 5, size=L)
                if depth:
                    b = np.tile(cons, (depth, 1))
                    noise = rng.random((depth, L))
                    b = np.where(noise < 0.08, rng.integers(1, 5, size=(depth, L)), b)
                    b = np.where((noise >= 0.08) & (noise < 0.12), -1, b)
                    b = np.where((noise >= 0.12) & (noise < 0.16), 0, b)
                    base[:depth] = b
                bq = np.where(base > 0, rng.integers(1, 61, size=(20, L)), 0)
                mq = np.where(base != -2, np.tile(rng.integers(1, 61, size=(20, 1)), (1, L)), 0)
                g[n, rows, :, 0] = base
                g[n, rows, :, 1] = bq
                g[n, rows, :, 2] = mq
                g[n, rows, :, 3] = (base != -2)
                g[n, rows, :, 4] = tag + 1
        gs.append(g)
    return gs[0], gs[1]
