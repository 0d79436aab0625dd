"""Host-side mirror of the reference's PileupModel network for inference.

``LSTMNetwork`` keeps the reference's constructor / ``predict`` surface
(PileupModel/model.py:85-119, used by PileupModel/predict.py:37-65,208-214) but owns no torch
modules: weights are handed once to the HIP library, ``predict`` is one C-ABI call.
"""
from __future__ import annotations

import numpy as np

from . import _lib

# state-dict keys in the order nsnp_pileup_load_weights expects (ont_pileup.chkpt)
ENCODER_KEYS = [f"lstm.{n}_l{l}{d}" for l in (0, 1) for d in ("", "_reverse")
                for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")] + \
               ["output_proj.weight", "output_proj.bias"]
FORWARD_KEYS = ["dense.weight", "dense.bias", "genotype_layer.weight", "genotype_layer.bias",
                "zygosity_layer.weight", "zygosity_layer.bias"]

# PileupModel/config/ont_pileup.yaml:6-20 -- the only architecture the kernels are built for
EXPECTED_MODEL_CONFIG = {"feature_dim": 18, "gt_num_class": 21, "zy_num_class": 3,
                         "enc": {"type": "lstm", "hidden_size": 64, "output_size": 128, "n_layers": 2,
                                 "bidirectional": True},
                         "joint": {"inner_size": 256}}


def _check_config(model_cfg):
    if model_cfg is None:
        return
    def get(d, k):
        return d[k] if isinstance(d, dict) else getattr(d, k)
    for k, v in EXPECTED_MODEL_CONFIG.items():
        got = get(model_cfg, k)
        if isinstance(v, dict):
            for kk, vv in v.items():
                if get(got, kk) != vv:
                    raise _lib.NanoSNPError(f"unsupported model config: {k}.{kk}={get(got, kk)!r}, kernels are built for {vv!r}")
        elif got != v:
            raise _lib.NanoSNPError(f"unsupported model config: {k}={got!r}, kernels are built for {v!r}")


class _SubModule:
    """Stands in for ``model.encoder`` / ``model.forward_layer`` so that the reference's
    ``pred_model.encoder.load_state_dict(checkpoint['encoder'])`` lines keep working
    (PileupModel/predict.py:213-214)."""

    def __init__(self, owner, keys, prefix):
        self._owner, self._keys, self._prefix = owner, keys, prefix
        self.state = None

    def load_state_dict(self, sd):
        missing = [k for k in self._keys if k not in sd]
        if missing:
            raise KeyError(f"{self._prefix}: missing keys {missing}")
        self.state = [np.ascontiguousarray(
            sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], dtype=np.float32)
            for k in self._keys]
        self._owner._maybe_upload()


class LSTMNetwork:
    """``LSTMNetwork(config.model)``; ``.encoder.load_state_dict``; ``.forward_layer.load_state_dict``;
    ``.predict(inputs) -> (gt_prob[N,21], zy_prob[N,3])`` -- the reference interface, HIP inside."""

    def __init__(self, config=None, device=0, chunk_sites=None, ctx=None):
        _check_config(config)
        self.ctx = ctx if ctx is not None else _lib.Context(device, chunk_sites)
        self.encoder = _SubModule(self, ENCODER_KEYS, "encoder")
        self.forward_layer = _SubModule(self, FORWARD_KEYS, "forward_layer")
        self._loaded = False

    # torch.nn.Module look-alikes used by the reference's predict.py
    def to(self, device):
        return self

    def eval(self):
        return self

    def _maybe_upload(self):
        if self.encoder.state is not None and self.forward_layer.state is not None:
            self.ctx.pileup_load_weights(self.encoder.state + self.forward_layer.state)
            self._loaded = True

    def load_weight_list(self, tensors):
        """24 arrays in state-dict order (nanosnp_amd/data/ont_pileup_weights.npz order)."""
        self.ctx.pileup_load_weights(tensors)
        self._loaded = True
        return self

    @classmethod
    def from_checkpoint(cls, path, device=0, **kw):
        """``torch.save`` dict {encoder, forward_layer, ...} as PileupModel/utils.py:67-77 writes it."""
        import torch
        ck = torch.load(path, map_location="cpu", weights_only=False)
        m = cls(None, device, **kw)
        m.encoder.load_state_dict(ck["encoder"])
        m.forward_layer.load_state_dict(ck["forward_layer"])
        return m

    @classmethod
    def from_npz(cls, path, device=0, **kw):
        z = np.load(path)
        m = cls(None, device, **kw)
        m.encoder.load_state_dict({k: z["encoder." + k] for k in ENCODER_KEYS})
        m.forward_layer.load_state_dict({k: z["forward_layer." + k] for k in FORWARD_KEYS})
        return m

    def predict(self, inputs, stream=None):
        """inputs: cuda tensor [N,33,18]; int32 (preferred, the position_matrix as stored) or the
        float tensor the reference builds at predict.py:49 (integral values, converted back)."""
        import torch
        if not self._loaded:
            raise _lib.NanoSNPError("weights not loaded")
        if inputs.dim() != 3 or tuple(inputs.shape[1:]) != (33, 18):
            raise ValueError(f"expected [N,33,18], got {tuple(inputs.shape)}")
        if not inputs.is_cuda:
            raise _lib.NanoSNPError("inputs must live on the GPU (no CPU path)")
        x = inputs if inputs.dtype == torch.int32 else inputs.to(torch.int32)
        return self.ctx.pileup_forward(x.contiguous(), stream=stream)
