"""Host-side mirror of the reference's HaplotypeModel network for inference (boundary B3).

``LSTMNetwork`` keeps the surface ``HaplotypeModel/predict_dev.py:27-48,69-71`` uses -- ``LSTMNetwork(config)``,
``.to(device)``, ``.load_state_dict(torch.load(path))``, ``.eval()``, ``.predict(pileup_x, haplotype_x)``
(``HaplotypeModel/model_dev.py:108-143``) -- but owns no torch modules: the 58 weight tensors are handed once to the
HIP library (``nsnp_hap_load_weights``), ``predict`` is one C-ABI call (``nsnp_hap_forward``).  There is no CPU path.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def state_dict_keys(n_layers=3):
    """keys of model_dev.LSTMNetwork.state_dict() in the order nsnp_hap_load_weights expects
    (pileup_encoder 26, haplotype_encoder 26, forward_layer 6; the loss modules hold no tensors)"""
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


# HaplotypeModel/config/ont_haplotype.yaml:7-16 -- the architecture family the kernels are built for
_DEFAULT = {"pileup_dim": 105, "haplotype_dim": 105, "pileup_length": 33, "haplotype_length": 11,
            "hidden_size": 256, "lstm_layers": 3, "gt_num_class": 10, "zy_num_class": 3}


def _get(cfg, key):
    if cfg is None:
        return _DEFAULT[key]
    m = cfg["model"] if isinstance(cfg, dict) else getattr(cfg, "model")
    return m[key] if isinstance(m, dict) else getattr(m, key)


class LSTMNetwork:
    """``LSTMNetwork(config)``; ``.load_state_dict(sd)``; ``.predict(pileup_x[N,105,33], haplotype_x[N,105,11]) ->
    (gt_prob[N,10], zy_prob[N,3])`` -- the reference interface, HIP inside."""

    def __init__(self, config=None, device=0, ctx=None):
        self.dims = {k: int(_get(config, k)) for k in _DEFAULT}
        d = self.dims
        if d["pileup_dim"] != d["haplotype_dim"]:
            raise _lib.NanoSNPError("pileup_dim and haplotype_dim must agree (one feature reduction feeds both encoders)")
        if (d["pileup_length"], d["haplotype_length"]) != (33, 11):
            raise _lib.NanoSNPError(f"unsupported window lengths {d['pileup_length']}/{d['haplotype_length']}: kernels are built for 33/11")
        if d["hidden_size"] % 64 or d["lstm_layers"] != 3:
            raise _lib.NanoSNPError("hidden_size must be a multiple of 64 and lstm_layers 3 (include/nanosnp.h: nsnp_hap_load_weights)")
        self.ctx = ctx if ctx is not None else _lib.Context(device)
        self._loaded = False

    # torch.nn.Module look-alikes used by predict_dev.py
    def to(self, device):
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd, strict=True):
        keys = state_dict_keys(self.dims["lstm_layers"])
        missing = [k for k in keys if k not in sd]
        if missing:
            raise KeyError(f"missing keys {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        arrs = [np.ascontiguousarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], dtype=np.float32)
                for k in keys]
        return self.load_weight_list(arrs)

    def load_weight_list(self, tensors):
        """58 arrays in state-dict order"""
        d = self.dims
        self.ctx.hap_load_weights(tensors, n_features=d["pileup_dim"], hidden=d["hidden_size"], n_layers=d["lstm_layers"],
                                  n_gt=d["gt_num_class"], n_zy=d["zy_num_class"])
        self._loaded = True
        return self

    def predict(self, pileup_x, haplotype_x, stream=None):
        """pileup_x [N,105,33], haplotype_x [N,105,11] cuda tensors (any float dtype: predict_dev.py:35-36 casts to float32)"""
        import torch
        if not self._loaded:
            raise _lib.NanoSNPError("weights not loaded")
        d = self.dims
        if pileup_x.dim() != 3 or tuple(pileup_x.shape[1:]) != (d["pileup_dim"], d["pileup_length"]) or \
                tuple(haplotype_x.shape) != (pileup_x.shape[0], d["haplotype_dim"], d["haplotype_length"]):
            raise ValueError(f"expected [N,{d['pileup_dim']},33] and [N,{d['haplotype_dim']},11], got "
                             f"{tuple(pileup_x.shape)} and {tuple(haplotype_x.shape)}")
        if not (pileup_x.is_cuda and haplotype_x.is_cuda):
            raise _lib.NanoSNPError("inputs must live on the GPU (no CPU path)")
        xp = pileup_x.to(torch.float32).contiguous()
        xh = haplotype_x.to(torch.float32).contiguous()
        return self.ctx.hap_forward(xp, xh, stream=stream)
