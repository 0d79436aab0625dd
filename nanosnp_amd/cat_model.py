"""Host-side mirror of the legacy ``CatModel`` (HaplotypeModel/model.py:201-360, crnn.py:84-190) as HaplotypeModel/predict.py uses
it: ``CatModel(nc0, nc1, nc2, nclass, nh)``, ``.load_state_dict(sd)``, ``.eval()``, ``.to(device)``, ``.predict(g0, g1, g2, g3) ->
softmax probabilities [N, nclass]`` - HIP inside (include/nanosnp.h: nsnp_cat_load_weights, nsnp_cat_forward)."""
from __future__ import annotations

import numpy as np

from . import _lib
from .fixtures import cat_weight_names


class CatModel:
    def __init__(self, nc0=5, nc1=5, nc2=2, nclass=10, nh=256, device=0, ctx=None):
        if (int(nc0), int(nc1), int(nclass), int(nh)) != (5, 5, 10, 256):
            raise _lib.NanoSNPError("CatModel is built for nc0 = nc1 = 5, nclass = 10, nh = 256 (predict.py:78-86, config_prev/cat45.yaml)")
        self.ctx = ctx if ctx is not None else _lib.Context(device)
        self._loaded = False

    # torch.nn.Module look-alikes
    def to(self, device):
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd, strict=True):
        """sd: CatModel.state_dict() - the 132 floating-point tensors are taken in the module's order (num_batches_tracked and the loss
        criterion's buffers are not used by predict)"""
        keys = cat_weight_names()
        missing = [k for k in keys if k not in sd]
        if missing:
            raise KeyError(f"missing keys {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        arrs = [np.ascontiguousarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], dtype=np.float32) for k in keys]
        return self.load_weight_list(arrs)

    def load_weight_list(self, tensors):
        self.ctx.cat_load_weights(tensors)
        self._loaded = True
        return self

    def predict(self, g0, g1, g2=None, g3=None, stream=None):
        """g0, g1: [N, 40, 11, 5] cuda tensors (dataset.py:862-915; any float / integer dtype); g2, g3 (edge matrices) are accepted
        and unused, as in model.py:332-358.  -> softmax probabilities [N, 10]."""
        if not self._loaded:
            raise _lib.NanoSNPError("weights not loaded")
        if not (g0.is_cuda and g1.is_cuda):
            raise _lib.NanoSNPError("inputs must live on the GPU (no CPU path)")
        return self.ctx.cat_forward(g0, g1, stream=stream)
