#!/usr/bin/env python3
"""CPU timing of the REFERENCE ITSELF for the hot path (BASELINE.json configs[0]).

Development container only (needs /root/reference; nothing here travels to the GPU box):

  * PileupModel/model.py LSTMNetwork.predict (the call of PileupModel/predict.py:51) imported with stub
    modules, shipped ont_pileup.chkpt, CPU torch, on 1,000 synthetic 30x windows (generator G2 of
    SURVEY.md 8(d), encoded by the oracle): batch 64 (configs[0]) and batch 1000 (predict.py:205),
    torch threads = all cores and 1; median of 5 timed repeats after one warm-up.
  * the reference's compiled DNA_CreateCanSnpTensor (oracle/_ref, -num_threads 1) on a G1 contig of
    200,000 columns: columns/s and candidate sites/s.

  * (own interpreter: the two packages' module names collide) HaplotypeModel/model_dev.py LSTMNetwork.predict
    (predict_dev.py:39) with seeded weights on 256 G3 sites, batch 64 and 256; dataset_dev.get_frequency_feature
    per call at D = 90 / 180, L = 33; legacy model.CatModel.predict on 256 sites (BASELINE configs[2] / [4]).

Writes profiles/r03_reference_cpu.json (read by bench.py for the note beside `cpu_baseline`).

    python tests/manual/time_reference_cpu.py
"""
from __future__ import annotations

import json
import os
import statistics
import subprocess
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("NANOSNP_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


def _stub_modules():
    for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object}),
                        ("tables", {"Filters": lambda **k: None}), ("pysam", {})):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m


def time_forward():
    import torch
    import yaml
    from nanosnp_amd import host
    from oracle import oracle
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "PileupModel"))
    from model import LSTMNetwork          # noqa: E402  (reference module)
    from utils import AttrDict             # noqa: E402
    cfg = AttrDict(yaml.load(open(os.path.join(REF, "PileupModel/config/ont_pileup.yaml")), Loader=yaml.FullLoader))
    m = LSTMNetwork(cfg.model)
    ck = torch.load(os.path.join(REF, "PileupModel/models/ont_pileup.chkpt"), map_location="cpu", weights_only=False)
    m.encoder.load_state_dict(ck["encoder"]); m.forward_layer.load_state_dict(ck["forward_layer"]); m.eval()
    n = 1000
    cols = host.synth_columns(20260000, n * 33, coverage=30, window=33)
    counts, _, _ = oracle.encode_columns(cols.bases, cols.col_off, cols.ref)
    x = torch.from_numpy(counts.reshape(n, 33, 18))
    out = []
    ncpu = os.cpu_count() or 1
    for threads in (ncpu, 1):
        torch.set_num_threads(threads)
        for batch in (64, 1000):
            def run():
                t0 = time.perf_counter()
                for b0 in range(0, n, batch):
                    ft = x[b0:b0 + batch].type(torch.FloatTensor)            # predict.py:49
                    gt, zy = m.predict(ft)                                    # predict.py:51
                    gt.detach().cpu().numpy(); zy.detach().cpu().numpy()     # predict.py:52-53
                return time.perf_counter() - t0
            run()
            ts = [run() for _ in range(5)]
            med = statistics.median(ts)
            out.append({"what": "LSTMNetwork.predict (reference module, CPU torch %s)" % torch.__version__,
                        "windows": n, "batch": batch, "threads": threads, "median_s": med, "sites_per_s": n / med})
            print(out[-1])
    return out


def time_encode():
    from nanosnp_amd import host
    refdir = os.path.join(ROOT, "oracle", "_ref")
    exe = os.path.join(refdir, "DNA_CreateCanSnpTensor")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    from oracle import oracle
    n_cols = 200_000
    cols = host.synth_columns(20260001, n_cols, coverage=30, window=0)            # G1: the columns carry their reference bases
    _, depth, _ = oracle.encode_columns(cols.bases, cols.col_off, cols.ref)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "ref.fa")
        with open(fa, "wb") as f:
            f.write(b">chrB\n")
            s = cols.ref.tobytes()
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + b"\n")
        open(fa + ".fai", "w").write(f"chrB\t{n_cols}\t6\t60\t61\n")
        pile = os.path.join(d, "pile"); os.makedirs(pile)
        lines = []
        off = cols.col_off
        bases = cols.bases.tobytes()
        for i in range(n_cols):
            b = bases[off[i]:off[i + 1]]
            dp = max(int(depth[i]), 1)
            lines.append(b"chrB\t%d\tN\t%d\t%s\t%s\n" % (i + 1, dp, b if b else b"*", b"I" * dp))
        open(os.path.join(pile, "chrB.mpileup"), "wb").write(b"".join(lines))
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            subprocess.run([exe, "-reference", fa, "-chr_pileup_dir", pile, "-output_dir", os.path.join(d, f"t{rep}"),
                            "-min_af", "0.12", "-snp_min_af", "0.12", "-indel_min_af", "0.12", "-min_coverage", "6",
                            "-flanking_base", "16", "-num_threads", "1", "chrB"], check=True, capture_output=True)
            ts.append(time.perf_counter() - t0)
        med = statistics.median(ts[1:])
        n_sites = sum(1 for _ in open(os.path.join(d, "t1", "chrB.tensor")))
    r = {"what": "DNA_CreateCanSnpTensor (reference C++, g++ -O3, -num_threads 1), file in -> .tensor text out",
         "columns": n_cols, "median_s": med, "columns_per_s": n_cols / med, "candidate_sites": n_sites, "sites_per_s": n_sites / med}
    print(r)
    return r


def time_haplotype():
    """runs in its own interpreter (python time_reference_cpu.py --haplotype): prints one JSON object"""
    import torch
    from nanosnp_amd import host
    from nanosnp_amd.fixtures import (cat_weight_names, hap_weight_names, seeded_cat_weights, seeded_hap_weights,
                                      synth_cat_groups)
    from oracle import oracle
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    import dataset_dev                      # noqa: E402  (reference modules)
    from model import CatModel              # noqa: E402
    from model_dev import LSTMNetwork       # noqa: E402
    from utils import AttrDict              # noqa: E402
    ncpu = os.cpu_count() or 1
    out = {"forward": [], "features": [], "cat": []}
    cfg = AttrDict({"model": {"pileup_dim": 105, "haplotype_dim": 105, "pileup_length": 33, "haplotype_length": 11,
                              "hidden_size": 256, "lstm_layers": 3, "gt_num_class": 10, "zy_num_class": 3, "dropout": 0.1}})
    m = LSTMNetwork(cfg)
    m.load_state_dict({k: torch.from_numpy(w) for k, w in zip(hap_weight_names(), seeded_hap_weights(12))}, strict=False)
    m.eval()
    n = 256
    pp = host.synth_hap_planes(20260400, n, 30, 90, 33); ph = host.synth_hap_planes(20260500, n, 30, 90, 11)
    xp = torch.from_numpy(oracle.hap_features_batch(*pp)); xh = torch.from_numpy(oracle.hap_features_batch(*ph))
    for threads in (ncpu, 1):
        torch.set_num_threads(threads)
        for batch in (64, 256):
            def run():
                t0 = time.perf_counter()
                for b0 in range(0, n, batch):
                    with torch.no_grad():
                        gt, zy = m.predict(xp[b0:b0 + batch].type(torch.FloatTensor), xh[b0:b0 + batch].type(torch.FloatTensor))   # predict_dev.py:35-39
                    gt.detach().cpu().numpy()
                return time.perf_counter() - t0
            run()
            med = statistics.median([run() for _ in range(3)])
            out["forward"].append({"what": "model_dev.LSTMNetwork.predict (reference module, seeded weights, CPU torch %s)" % torch.__version__,
                                   "sites": n, "batch": batch, "threads": threads, "median_s": med, "sites_per_s": n / med})
            print(out["forward"][-1], file=sys.stderr)
    for D in (90, 180):
        pl = host.synth_hap_planes(7, 64, D / 3, D, 33)
        def runf():
            t0 = time.perf_counter()
            for i in range(64):
                dataset_dev.get_frequency_feature(pl[0][i], pl[1][i], pl[2][i], pl[3][i])      # dataset_dev.py:342
            return (time.perf_counter() - t0) / 64
        runf()
        med = statistics.median([runf() for _ in range(3)])
        out["features"].append({"what": "dataset_dev.get_frequency_feature (numpy, one call; two calls per site)", "D": D, "L": 33,
                                "ms_per_call": med * 1e3, "sites_per_s_one_worker": 1.0 / (med * (1 + 11 / 33))})
        print(out["features"][-1], file=sys.stderr)
    cm = CatModel(nc0=5, nc1=5, nc2=2, nclass=10, nh=256)
    cm.load_state_dict({k: torch.from_numpy(w) for k, w in zip(cat_weight_names(), seeded_cat_weights(21))}, strict=False)
    cm.eval()
    g0, g1 = synth_cat_groups(321, n)
    g0, g1 = torch.from_numpy(g0), torch.from_numpy(g1)
    for threads in (ncpu, 1):
        torch.set_num_threads(threads)
        def runc():
            t0 = time.perf_counter()
            for b0 in range(0, n, 64):
                with torch.no_grad():
                    cm.predict(g0[b0:b0 + 64], g1[b0:b0 + 64], None, None).numpy()
            return time.perf_counter() - t0
        runc()
        med = statistics.median([runc() for _ in range(3)])
        out["cat"].append({"what": "model.CatModel.predict (reference module, seeded weights)", "sites": n, "batch": 64, "threads": threads,
                           "median_s": med, "sites_per_s": n / med})
        print(out["cat"][-1], file=sys.stderr)
    print(json.dumps(out))


def main():
    if not os.path.isdir(REF):
        sys.exit(f"{REF} is not mounted: the reference can only be timed in the development container")
    if "--haplotype" in sys.argv:
        return time_haplotype()
    cpu = ""
    for l in open("/proc/cpuinfo"):
        if l.startswith("model name"):
            cpu = l.split(":", 1)[1].strip(); break
    res = {"host": {"cpu": cpu, "logical_cpus": os.cpu_count()}, "forward": time_forward()}
    try:
        res["encode"] = time_encode()
    except Exception as e:                                        # the encode timing is secondary
        res["encode"] = {"error": repr(e)}
    hp = subprocess.run([sys.executable, os.path.abspath(__file__), "--haplotype"], capture_output=True, text=True)
    if hp.returncode == 0:
        res["haplotype"] = json.loads(hp.stdout.strip().splitlines()[-1])
    else:
        res["haplotype"] = {"error": hp.stderr[-2000:]}
    json.dump(res, open(os.path.join(ROOT, "profiles", "r03_reference_cpu.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
