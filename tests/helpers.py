"""Shared test helpers: fixture paths, seeded weight generators, tolerance constants.

The seeded weight generators stand in for the HaplotypeModel checkpoints that are absent from
the reference tree (.MISSING_LARGE_BLOBS): tools/make_golden.py loads these exact arrays
into the reference's model_dev.LSTMNetwork and records its outputs; the tests regenerate the
same arrays (numpy PCG64 streams are stable across platforms) instead of committing 33 MB.
"""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# north_star: float outputs within 1e-4 abs of the reference PyTorch CPU path
PROB_ATOL = 1e-4

PILEUP_WEIGHT_KEYS = (
    [f"encoder.lstm.{n}_l{l}{d}" for l in (0, 1) for d in ("", "_reverse")
     for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    + ["encoder.output_proj.weight", "encoder.output_proj.bias",
       "forward_layer.dense.weight", "forward_layer.dense.bias",
       "forward_layer.genotype_layer.weight", "forward_layer.genotype_layer.bias",
       "forward_layer.zygosity_layer.weight", "forward_layer.zygosity_layer.bias"]
)


def golden(name):
    return os.path.join(GOLDEN, name)


def load_pileup_weights():
    """The 24 tensors LSTMNetwork.predict uses, in state-dict order."""
    z = np.load(golden("ont_pileup_weights.npz"))
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


def seeded_hap_weights(seed, F=105, H=256, n_layers=3, n_gt=10, n_zy=3):
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
            w *= np.float32(0.002)
        if "genotype_layer.weight" in name or "zygosity_layer.weight" in name:
            w *= np.float32(8.0)   # spread the logits so that parity errors are visible
        out.append(w)
    return out
