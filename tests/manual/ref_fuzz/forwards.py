#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): the reference's own modules with RANDOM weights - PileupModel/model.py
LSTMNetwork.predict (torch CPU) and, in a second process, HaplotypeModel/model_dev.py LSTMNetwork.predict - against the fp32 oracle on
the same weights and inputs: the oracle is pinned by the three shipped checkpoints and seeded HaplotypeModel weights; this checks that
nothing in it leans on how those weights look (scales 0.1 .. 3 times torch's default initialisation, large biases, a dominating channel).
    python tests/manual/ref_fuzz/forwards.py pileup|haplotype|cat [N_SEEDS]"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch, yaml
for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object}), ("tables", {"Filters": lambda **k: None})):
    m = types.ModuleType(name); [setattr(m, k, v) for k, v in attrs.items()]; sys.modules[name] = m
from oracle import oracle
which = sys.argv[1] if len(sys.argv) > 1 else "pileup"
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
REF = "/root/reference"
torch.set_num_threads(8)
bad = 0
if which == "pileup":
    sys.path.insert(0, os.path.join(REF, "PileupModel"))
    from model import LSTMNetwork
    from utils import AttrDict
    cfg = AttrDict(yaml.load(open(os.path.join(REF, "PileupModel/config/ont_pileup.yaml")), Loader=yaml.FullLoader))
    for seed in range(seeds):
        torch.manual_seed(seed); rng = np.random.default_rng(seed)
        m = LSTMNetwork(cfg.model); m.eval()
        scale = [0.1, 0.5, 1.0, 2.0, 3.0, 1.0][seed % 6]
        with torch.no_grad():
            for n_, p in list(m.encoder.named_parameters()) + list(m.forward_layer.named_parameters()):
                p.mul_(scale)
                if seed % 3 == 1 and n_.endswith("bias_ih_l0"): p.add_(2.0)
                if seed % 3 == 2 and n_ == "lstm.weight_ih_l0": p[:, 3].mul_(30.0)
        ws = [v.detach().numpy().astype(np.float32) for v in list(m.encoder.state_dict().values())[:18] + list(m.forward_layer.state_dict().values())[:6]]
        for kind in range(3):
            x = [rng.integers(0, 50, (128, 33, 18)) - 10, rng.integers(-144, 145, (128, 33, 18)), rng.poisson(2, (128, 33, 18)) * (rng.random((128, 33, 18)) < 0.3)][kind].astype(np.int32)
            with torch.no_grad():
                gt, zy = m.predict(torch.from_numpy(x).type(torch.FloatTensor))
            og, oz = oracle.pileup_forward(ws, x, nthreads=8)
            f64 = oracle.pileup_forward_f64(ws, x)
            d = max(np.abs(og - gt.numpy()).max(), np.abs(oz - zy.numpy()).max())
            dr = max(np.abs(f64[0] - gt.numpy()).max(), np.abs(f64[1] - zy.numpy()).max())       # the reference's own distance from float64
            ok = d <= max(1e-5, 3 * dr)
            bad += not ok
            print(f"pileup seed {seed} scale {scale} input {kind}: |oracle - reference| {d:.1e} (reference vs float64 {dr:.1e}) {'ok' if ok else 'LOOK'}", flush=True)
elif which == "cat":
    # the legacy CatModel (HaplotypeModel/model.py:332-358) with torch's OWN random initialisation (BatchNorm statistics randomised: a fresh
    # module has mean 0 / variance 1) against oracle.cat_forward on the module's state dict
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    from model import CatModel
    from nanosnp_amd.fixtures import cat_weight_names, synth_cat_groups
    for seed in range(seeds):
        torch.manual_seed(500 + seed)
        m = CatModel(nc0=5, nc1=5, nc2=2, nclass=10, nh=256); m.eval()
        sd = m.state_dict()
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for k in sd:
                if k.endswith("running_var"): sd[k].copy_(torch.rand(sd[k].shape, generator=g) + 0.5)
                elif k.endswith("running_mean"): sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.2)
        ws = [sd[k].numpy().astype(np.float32) for k in cat_weight_names()]
        g0, g1 = synth_cat_groups(900 + seed, 48)
        with torch.no_grad():
            gt = m.predict(torch.from_numpy(g0), torch.from_numpy(g1), None, None).numpy()
        og = oracle.cat_forward(ws, g0, g1, nthreads=8)
        d = float(np.abs(og - gt).max())
        ok = d <= 2e-5
        bad += not ok
        print(f"cat seed {seed}: |oracle - reference| {d:.1e} {'ok' if ok else 'LOOK'}", flush=True)
else:
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    from model_dev import LSTMNetwork
    from utils import AttrDict
    from nanosnp_amd import host
    from nanosnp_amd.fixtures import hap_weight_names
    cfg = AttrDict({"model": {"pileup_dim": 105, "haplotype_dim": 105, "pileup_length": 33, "haplotype_length": 11, "hidden_size": 256,
                              "lstm_layers": 3, "gt_num_class": 10, "zy_num_class": 3, "dropout": 0.1}})
    for seed in range(seeds):
        torch.manual_seed(100 + seed); rng = np.random.default_rng(seed)
        m = LSTMNetwork(cfg); m.eval()
        scale = [1.0, 0.3, 2.0][seed % 3]
        sd = m.state_dict()
        with torch.no_grad():
            for k in sd:
                if sd[k].dtype.is_floating_point:
                    sd[k].mul_(scale)
                    if k.endswith("weight_ih_l0") or k.endswith("weight_ih_l0_reverse"): sd[k].mul_(0.01)      # count-valued features
        ws = [sd[k].numpy().astype(np.float32) for k in hap_weight_names()]
        n = 24
        feats = []
        for L in (33, 11):
            seq, bq, mq, hap, ref_row = host.synth_hap_planes(900 + seed + L, n, 30, 90, L)
            feats.append(oracle.hap_features_batch(seq, bq, mq, hap, ref_row, nthreads=8))
        xp, xh = feats
        with torch.no_grad():
            gt, zy = m.predict(torch.from_numpy(xp), torch.from_numpy(xh))
        og, oz = oracle.hap_forward(ws, xp, xh, nthreads=8)
        d = max(np.abs(og - gt.numpy()).max(), np.abs(oz - zy.numpy()).max())
        ok = d <= 2e-5
        bad += not ok
        print(f"haplotype seed {seed} scale {scale}: |oracle - reference| {d:.1e} {'ok' if ok else 'LOOK'}", flush=True)
print("bad", bad)
