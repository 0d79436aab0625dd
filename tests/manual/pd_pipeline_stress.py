#!/usr/bin/env python3
"""Stress of the streamed window-file pipeline (pipeline.predict_pileup_bins): random runs of .pd.bin files - 0 to 3000 windows each,
int16 and int32 counts on disk mixed, a count beyond int16 in the middle of a run, several contigs per file in any order, batch sizes
1000 / 64 / 7, pass sizes from 1 window to the whole run - the VCF against the plain chain (nsnp_pileup_forward on the whole array ->
argmax / max / coverage slice with numpy -> nsnp_vcf_format_batches per file) and across pass sizes.  Test infrastructure."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import host, sitefile
from nanosnp_amd.fixtures import load_pileup_weights
from nanosnp_amd.pileup_model import LSTMNetwork
from nanosnp_amd.pipeline import predict_pileup_bins

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    m = LSTMNetwork().load_weight_list(load_pileup_weights())
    ctx = m.ctx
    fai = "ctgA\t900000\t6\t60\t61\nctgB\t900000\t6\t60\t61\nctgC\t900000\t6\t60\t61\n"
    header = host.vcf_header(fai).encode()
    bad = 0
    for r in range(rounds):
        rng = np.random.default_rng(9900 + r + int(os.environ.get("NSNP_STRESS_SEED", "0")))
        tmp = tempfile.mkdtemp()
        bs = int(rng.choice([1000, 64, 7]))
        files, want = [], header
        for fi in range(int(rng.integers(1, 5))):
            n = int(rng.choice([0, 1, 9, 700, 3000]))
            cols = host.synth_columns(31000 + 10 * r + fi, max(n, 1) * 33, coverage=float(rng.choice([8, 30, 60])), window=33)
            counts, _, _ = ctx.pileup_encode_columns(torch.from_numpy(cols.bases).cuda(), torch.from_numpy(cols.col_off).cuda(), torch.from_numpy(cols.ref).cuda())
            x = ctx.pileup_gather_windows(counts, torch.arange(max(n, 1), dtype=torch.int64, device="cuda") * 33 + 16).cpu().numpy()[:n]
            if n and rng.random() < 0.3:
                x[rng.integers(0, n), rng.integers(0, 33), rng.integers(0, 18)] = 40000
            names = [str(c) for c in rng.choice(["ctgA", "ctgB", "ctgC"], n)]
            pos = rng.integers(1, 800000, n).astype(np.int64)
            refb = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
            position = [f"{c}:{int(p)}:{'N' * 16}{chr(int(b))}{'N' * 16}" for c, p, b in zip(names, pos, refb)]
            path = os.path.join(tmp, f"f{fi}.pd.bin")
            sitefile.write_pileup_bin(path, x, position, matrix_dtype=str(rng.choice(["int16", "int32"])))
            files.append(path)
            if n:
                gt, zy = ctx.pileup_forward(torch.from_numpy(np.ascontiguousarray(x)).cuda())
                gt, zy = gt.cpu().numpy(), zy.cpu().numpy()
                uniq = list(dict.fromkeys(names)); tbl = host.ContigTable(uniq)
                cov = x[:, 16, [0, 1, 2, 3, 9, 10, 11, 12]].astype(np.float32)
                text, _ = host.vcf_format_batches(tbl, np.array([uniq.index(c) for c in names], np.int32), pos, refb, gt.argmax(1).astype(np.uint8),
                                                  zy.argmax(1).astype(np.uint8), gt.max(1), zy.max(1), cov, batch_size=bs)
                want += text
        outs = {}
        for ps, narrow in ((1, True), (50, True), (777, False), (65536, True)):
            if ps == 1 and len(want) > 40000:
                continue
            o = os.path.join(tmp, f"o{ps}.vcf")
            predict_pileup_bins(m, files, fai, o, batch_size=bs, pass_sites=ps, narrow=narrow)
            outs[ps] = open(o, "rb").read()
        ok = all(v == want for v in outs.values())
        bad += not ok
        print(f"round {r}: {len(files)} files, batch size {bs}, {want.count(10) - header.count(10)} rows, pass sizes {sorted(outs)}: {'identical to the plain chain' if ok else 'DIFFER'}", flush=True)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
