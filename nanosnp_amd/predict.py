"""The two predict loops of the reference on the MI355X path.

    predict_pileup     PileupModel/predict.py:37-195     position_matrix -> pileup.vcf
    predict_haplotype  HaplotypeModel/predict_dev.py:27-48  read planes -> haplotype.csv

The loops keep the reference's batch structure (its VCF rows depend on the batch boundary, see
nanosnp_amd/csrc/nsnp_vcf.c) but every per-site Python statement is gone: forward, argmax/max and
the coverage slice run on the device, the text rows are produced by the native writer.
"""
from __future__ import annotations

import numpy as np

from . import host

COV_CHANNELS = [0, 1, 2, 3, 9, 10, 11, 12]        # predict.py:63


def predict_pileup(model, x, contig_names, positions, reference_bases, fai_text, output_file,
                   batch_size=1000, score_mode=host.SCORE_FLOAT64, device_batch=65536):
    """model: nanosnp_amd.pileup_model.LSTMNetwork; x: int32 [N,33,18] (numpy or cuda tensor);
    contig_names/positions/reference_bases: what PredictDataset yields (dataset.py:141-146).
    Returns the number of VCF rows written."""
    import torch
    ctx = model.ctx
    table = host.ContigTable(list(contig_names))
    xt = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32))
    xt = xt.to(torch.device("cuda", ctx.device), torch.int32).contiguous()          # the context's device, not torch's current one
    pos = np.asarray(positions, np.int64)
    refb = np.asarray(reference_bases, np.uint8)
    n = xt.shape[0]
    cov_idx = torch.tensor(COV_CHANNELS, device=xt.device)
    # device part in passes of `device_batch` sites; the text rows are then produced for the reference's batches of
    # `batch_size` sites (a row can depend on its batch: nsnp_vcf.c) by one native call, OpenMP over the batches
    outs = []
    for b0 in range(0, n, device_batch):
        xb = xt[b0:b0 + device_batch]
        gt, zy = model.predict(xb)
        ga, za, gm, zm, _ = ctx.pileup_postprocess(gt, zy)
        cov = xb[:, 16, :].index_select(1, cov_idx).to(torch.float32)            # predict.py:63 on a FloatTensor
        outs.append((ga, za, gm, zm, cov))
    if outs:
        ga, za, gm, zm, cov = [torch.cat([o[i] for o in outs]).cpu().numpy() for i in range(5)]
    else:
        ga = za = np.empty(0, np.uint8); gm = zm = np.empty(0, np.float32); cov = np.empty((0, 8), np.float32)
    text, rows = host.vcf_format_batches(table, table.ids, pos, refb, ga, za, gm, zm, cov, batch_size=batch_size,
                                         score_mode=score_mode)
    with open(output_file, "wb") as f:
        f.write(host.vcf_header(fai_text).encode())
        f.write(text)
    return rows


def predict_haplotype(ctx, planes_pileup, planes_haplotype, candidate_positions, output_file,
                      batch_size=1000, score_mode=host.SCORE_FLOAT64, device_batch=16384):
    """ctx: a Context with hap weights loaded; planes_*: (seq, baseq, mapq, hap, ref_row) int32
    arrays [N,D,33] / [N,D,11]; candidate_positions: "ctg:pos" strings (dataset_dev.py:331-333).
    batch_size is the reference's DataLoader batch (predict_dev.py:27-33): a haplotype.csv row does not depend on it (unlike a
    pileup.vcf row), so the device works in passes of `device_batch` sites - the forward of 1,000 sites runs at 40 % of the rate of
    16,384 - and the rows are written by one native call; a probability the reference's loop would fail on (score_mode 0, p = 1)
    fails here as it does there, whatever the batch."""
    import torch
    del batch_size
    n = len(candidate_positions)
    ctgs, poss = zip(*[p.split(":") for p in candidate_positions]) if n else ((), ())
    table = host.ContigTable(list(ctgs))
    pos = np.array([int(p) for p in poss], np.int64)
    # read planes handed over as int8 arrays (every value fits: base codes, HP, qualities <= 93, padding -2) cross PCIe
    # and HBM at a quarter of the bytes, same features bit for bit (nsnp_hap_features_i8); reference rows stay int32
    def dtypes(planes):
        narrow = all(a.dtype == np.int8 for a in planes[:4])
        return [np.dtype(np.int8 if narrow else np.int32)] * 4 + [np.dtype(np.int32)]
    tp, th = dtypes(planes_pileup), dtypes(planes_haplotype)
    dev = torch.device("cuda", ctx.device)
    ga_all, gm_all = [], []
    for b0 in range(0, n, int(device_batch)):
        sl = slice(b0, b0 + int(device_batch))
        dp = [torch.from_numpy(np.ascontiguousarray(a[sl], dtype=t)).to(dev) for a, t in zip(planes_pileup, tp)]
        dh = [torch.from_numpy(np.ascontiguousarray(a[sl], dtype=t)).to(dev) for a, t in zip(planes_haplotype, th)]
        xp = ctx.hap_features(*dp)
        xh = ctx.hap_features(*dh)
        gt, _ = ctx.hap_forward(xp, xh)
        gm, ga = gt.max(dim=1)
        ga_all.append(ga.to(torch.uint8)); gm_all.append(gm)
    with open(output_file, "wb") as f:
        if n:
            f.write(host.hap_csv_format(table, table.ids, pos, torch.cat(ga_all).cpu().numpy(), torch.cat(gm_all).cpu().numpy(), score_mode))
    return n
