#!/usr/bin/env python3
"""CPU timing of the REFERENCE ITSELF for the hot path (BASELINE.json configs[0]).

Development container only (needs /root/reference; nothing here travels to the GPU box):

  * PileupModel/model.py LSTMNetwork.predict (the call of PileupModel/predict.py:51) imported with stub
    modules, shipped ont_pileup.chkpt, CPU torch, on 1,000 synthetic 30x windows (generator G2 of
    SURVEY.md 8(d), encoded by the oracle): batch 64 (configs[0]) and batch 1000 (predict.py:205),
    torch threads = all cores and 1; median of 5 timed repeats after one warm-up.
  * the reference's compiled DNA_CreateCanSnpTensor (oracle/_ref, -num_threads 1) on a G1 contig of
    200,000 columns: columns/s and candidate sites/s.

Writes profiles/r02_reference_cpu.json (read by bench.py for the note beside `cpu_baseline`).

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


def main():
    if not os.path.isdir(REF):
        sys.exit(f"{REF} is not mounted: the reference can only be timed in the development container")
    cpu = ""
    for l in open("/proc/cpuinfo"):
        if l.startswith("model name"):
            cpu = l.split(":", 1)[1].strip(); break
    res = {"host": {"cpu": cpu, "logical_cpus": os.cpu_count()}, "forward": time_forward()}
    try:
        res["encode"] = time_encode()
    except Exception as e:                                        # the encode timing is secondary
        res["encode"] = {"error": repr(e)}
    json.dump(res, open(os.path.join(ROOT, "profiles", "r02_reference_cpu.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
