#!/usr/bin/env python3
"""Stress of the streamed stage-5 pipeline (hap_pipeline.predict_haplotype_bins): random runs of site files - 0 to 400 sites each, depths
from 1 to 120 that differ between files AND between the two plane sets, int8 and int32 storage mixed, values beyond int8 in the middle
of a run (the narrowing restarts), contigs the reference lacks, windows hanging over both contig ends, pass sizes from 1 site to the
whole run - the csv against the oracle chain (host reference rows -> oracle features -> oracle forward -> argmax -> the row formatter),
and against itself for every pass size.  Test infrastructure (loads oracle/)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib, host, sitefile
from nanosnp_amd.fixtures import seeded_hap_weights
from nanosnp_amd.hap_pipeline import DeviceReference, predict_haplotype_bins
from oracle import oracle

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ws = seeded_hap_weights(12, H=256)
    ctx = _lib.Context(0); ctx.hap_load_weights(ws)
    bad = 0
    for r in range(rounds):
        rng = np.random.default_rng(8800 + r + int(os.environ.get("NSNP_STRESS_SEED", "0")))
        refs = {c: rng.choice(list(b"ACGTacgtN"), int(rng.integers(300, 5000)), p=[.23, .23, .23, .23, .02, .02, .01, .01, .02]).astype(np.uint8) for c in ("ctgA", "ctgB")}
        tmp = tempfile.mkdtemp()
        files, rows_want = [], []
        for fi in range(int(rng.integers(1, 5))):
            n = int(rng.choice([0, 1, 7, 130, 400]))
            Dp, Dh = int(rng.choice([1, 30, 90, 120])), int(rng.choice([1, 25, 90]))
            contig = str(rng.choice(["ctgA", "ctgB", "ctgMissing"], p=[.5, .4, .1]))
            L = len(refs.get(contig, np.zeros(1000)))
            pp = host.synth_hap_planes(100 * r + fi, n, 30, Dp, 33); ph = host.synth_hap_planes(100 * r + fi + 50, n, 30, Dh, 11)
            if rng.random() < 0.3 and n:
                pp[2][rng.integers(0, n), 0, 0] = 255                      # one value beyond int8 somewhere in the run
            posn = np.sort(rng.choice(np.arange(-20, L + 40), n, replace=False)) if n else np.zeros(0, int)
            cands = [f"{contig}:{p}" for p in posn]
            hpos = [[f"{contig}:{p + 37 * (k - 5)}" for k in range(11)] for p in posn]
            planes = dict(zip(sitefile.HAP_PLANES, (ph[0], ph[3], ph[1], ph[2], pp[0], pp[3], pp[1], pp[2])))
            path = os.path.join(tmp, f"f{fi}.bin")
            sitefile.write_haplotype_bin(path, cands, hpos, planes, plane_dtype=str(rng.choice(["int8", "int32"])))
            files.append(path)
            if n:
                c2, h2, pl = sitefile.read_haplotype_bin(path, mmap=False)          # (sorted by position as write_to_bins does)
                rp = host.haplotype_ref_rows(refs, c2, 33); rh = host.haplotype_ref_rows(refs, c2, 11, position_lists=h2)
                g = lambda k: np.asarray(pl[k], np.int32)
                xp = oracle.hap_features_batch(g("pileup_sequences"), g("pileup_baseq"), g("pileup_mapq"), g("pileup_hap"), rp, nthreads=8)
                xh = oracle.hap_features_batch(g("haplotype_sequences"), g("haplotype_baseq"), g("haplotype_mapq"), g("haplotype_hap"), rh, nthreads=8)
                ogt, _ = oracle.hap_forward(ws, xp, xh, nthreads=8)
                rows_want.append((c2, ogt))
        ref = DeviceReference(refs, 0)
        outs = {}
        for ps in (1, 50, 128, 16384):
            if ps == 1 and sum(len(c) for c, _ in rows_want) > 200:
                continue
            o = os.path.join(tmp, f"o{ps}.csv")
            predict_haplotype_bins(ctx, files, ref, o, pass_sites=ps)
            outs[ps] = open(o, "rb").read()
        first = next(iter(outs.values()))
        same = all(v == first for v in outs.values())
        got = [l.split("\t") for l in first.decode().splitlines()]
        k = 0; worst = 0.0; flips = 0; ok_rows = True
        labels = ["AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT"]
        for c2, ogt in rows_want:
            for i, cp in enumerate(c2):
                ctg, pos = cp.split(":")
                row = got[k]; k += 1
                ok_rows = ok_rows and row[0] == ctg and row[1] == pos
                near = np.sort(ogt[i])[-1] - np.sort(ogt[i])[-2] < 1e-4
                if row[2] != labels[int(ogt[i].argmax())]:
                    flips += not near
                else:
                    q_or, _ = host.calculate_score(np.float32(ogt[i].max()), host.SCORE_FLOAT64)
                    worst = max(worst, abs(float(row[3]) - q_or))
        ok = same and ok_rows and k == len(got) and flips == 0 and worst <= 0.02
        bad += not ok
        print(f"round {r}: {len(files)} files, {k} sites, pass sizes {sorted(outs)}: identical across pass sizes {same}, rows in order {ok_rows}, "
              f"genotype flips {flips}, worst |QUAL - oracle QUAL| {worst:.3f} -> {'ok' if ok else 'FAIL'}", flush=True)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
