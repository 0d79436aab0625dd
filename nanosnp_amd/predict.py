"""The two predict loops of the reference on the MI355X path.

    predict_pileup     PileupModel/predict.py:37-195     position_matrix -> pileup.vcf
    predict_haplotype  HaplotypeModel/predict_dev.py:27-48  read planes -> haplotype.csv
    predict            PileupModel/predict.py:37           the reference's call shape over window files (streamed)
    predict_dev        HaplotypeModel/predict_dev.py:27    the reference's call shape over haplotype bins (streamed)

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
                      batch_size=1000, score_mode=host.SCORE_FLOAT64, device_batch=16384, narrow=True, stats=None):
    """ctx: a Context with hap weights loaded; planes_*: (seq, baseq, mapq, hap, ref_row) integer arrays [N,D,33] / [N,D,11] in host
    memory (numpy or memmap; int32 as the reference's bins hold them, or int8); candidate_positions: "ctg:pos" strings
    (dataset_dev.py:331-333).  The sites are STREAMED (nanosnp_amd.hap_pipeline.stream_haplotype): passes of `device_batch` sites are
    staged into pinned buffers on all host cores (int32 planes narrowed to int8 on the way when every value fits: a quarter of the
    bytes over PCIe, the same features bit for bit), copied on a copy stream and reduced + forwarded on the compute stream, three
    passes in flight; nothing is pageable, nothing blocks per pass.
    batch_size is the reference's DataLoader batch (predict_dev.py:27-33): a haplotype.csv row does not depend on it (unlike a
    pileup.vcf row), so the device works in passes of `device_batch` sites - the forward of 1,000 sites runs at 40 % of the rate of
    16,384 - and the rows are written by one native call; a probability the reference's loop would fail on (score_mode 0, p = 1)
    fails here as it does there, whatever the batch."""
    from .hap_pipeline import HapArraySource, stream_haplotype
    del batch_size
    n = len(candidate_positions)
    src = HapArraySource(planes_pileup, planes_haplotype, list(candidate_positions))
    calls = stream_haplotype(ctx, src, None, pass_sites=int(device_batch), narrow=narrow, stats=stats)
    with open(output_file, "wb") as f:
        if n:
            f.write(host.hap_csv_format(calls.table, calls.contig_id, calls.pos, calls.gt_arg, calls.gt_max, score_mode))
    return n


# ---- the reference's own call shapes ------------------------------------------------------------------------------------------------
def predict(model, testing_paths, reference_index_file, batch_size, output_file, device=None, **kw):
    """PileupModel/predict.py:37 `predict(model, testing_paths, reference_index_file, batch_size, output_file, device)`, argument for
    argument, over this repository's window files: model = nanosnp_amd.pileup_model.LSTMNetwork (it carries its device: `device` is
    accepted and ignored unless it names another GPU than the model's), testing_paths a list of `.pd.bin` files or a directory,
    reference_index_file the path of the .fai.  Streams through pipeline.predict_pileup_bins; returns the rows written."""
    from .pipeline import predict_pileup_bins
    _same_device(model.ctx, device)
    return predict_pileup_bins(model, testing_paths, reference_index_file, output_file, batch_size=batch_size, **kw)


def predict_dev(model, test_data, reference_path, batch_size, pileup_length, haplotype_length, output_file, device=None, **kw):
    """HaplotypeModel/predict_dev.py:27 `predict(model, test_data, reference_path, batch_size, pileup_length, haplotype_length,
    output_file, device)`, argument for argument, over this repository's haplotype site files: model =
    nanosnp_amd.haplotype_model.LSTMNetwork, test_data the directory of bins, reference_path the FASTA.  batch_size only sized the
    reference's DataLoader batches (its rows do not depend on it): accepted, unused.  Streams through
    hap_pipeline.predict_haplotype_bins; returns the rows written."""
    from .hap_pipeline import predict_haplotype_bins
    if (int(pileup_length), int(haplotype_length)) != (33, 11):
        raise ValueError("the HaplotypeModel of this path takes windows of 33 and groups of 11 columns (config/ont_haplotype.yaml)")
    _same_device(model.ctx, device)
    return predict_haplotype_bins(model.ctx, test_data, reference_path, output_file, **kw)


def _same_device(ctx, device):
    if device is None:
        return
    import torch
    d = torch.device(device)
    if d.type != "cuda" or (d.index is not None and d.index != ctx.device):
        raise ValueError(f"the model lives on cuda:{ctx.device}; predict(..., device={device!r}) names another device (there is no CPU path)")
