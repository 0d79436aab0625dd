"""Stage s1 + s2 of the reference pipeline in one pass on the device:

    <chr>.mpileup text + FASTA  ->  column encode -> candidate windows -> PileupModel -> pileup.vcf

replacing DNA_CreateCanSnpTensor -> DNA_CreatePredictData -> make_bin_predict_data.py ->
PileupModel/predict.py (dna_sv_tensor/src/scripts/make_predict_data.sh:184-234,
scripts/s2_pileup_model_predict.sh:11-16) and the four text/HDF5 files between them.
File reading and VCF writing are host work (native readers / writer in libnanosnp_host.so);
everything between lives in HBM.
"""
from __future__ import annotations

import numpy as np

from . import host
from .predict import COV_CHANNELS


def call_contig(model, mpileup_text: bytes, contig: str, chr_seq: np.ndarray, min_af=0.12, min_coverage=6,
                batch_size=1000, score_mode=host.SCORE_FLOAT64):
    """One contig: returns (vcf_rows: bytes, n_sites, n_rows).  model: pileup_model.LSTMNetwork."""
    import torch
    ctx = model.ctx
    pos, col_off, bases = host.mpileup_parse(mpileup_text)
    if pos.size == 0:
        return b"", 0, 0
    if pos.max() > chr_seq.size or pos.min() < 1:
        raise ValueError(f"{contig}: position outside the reference sequence")
    ref = np.ascontiguousarray(chr_seq[pos - 1])
    dev = "cuda"
    d_bases = torch.from_numpy(bases if bases.size else np.zeros(1, np.uint8)).to(dev)
    d_off = torch.from_numpy(col_off).to(dev)
    d_ref = torch.from_numpy(ref).to(dev)
    d_pos = torch.from_numpy(pos).to(dev)
    counts, depth, flags = ctx.pileup_encode_columns(d_bases, d_off, d_ref, min_af, min_coverage)
    centers, n_sites = ctx.pileup_select_sites(d_pos, flags)
    if n_sites == 0:
        return b"", 0, 0
    gt, zy = ctx.pileup_forward_windows(counts, centers)
    ga, za, gm, zm, _ = ctx.pileup_postprocess(gt, zy)
    cov = counts[centers][:, COV_CHANNELS].to(torch.float32)              # predict.py:63
    c_host = centers.cpu().numpy()
    table = host.ContigTable([contig])
    ids = np.zeros(n_sites, np.int32)
    site_pos = pos[c_host]
    site_ref = ref[c_host] & 0xDF                                          # make_predict_data/main.cpp:91 upper-cases
    ga, za, gm, zm, cov = (t.cpu().numpy() for t in (ga, za, gm, zm, cov))
    # the VCF rows depend on the batch boundary: one native call formats every batch (OpenMP over the batches)
    text, rows = host.vcf_format_batches(table, ids, site_pos, site_ref, ga, za, gm, zm, cov, batch_size=batch_size,
                                         score_mode=score_mode)
    return text, n_sites, rows


def call_variants(model, contigs, fasta_path, fai_text, output_file, **kw):
    """contigs: iterable of (name, path to <name>.mpileup).  Writes pileup.vcf; returns total rows."""
    total = 0
    with open(output_file, "wb") as f:
        f.write(host.vcf_header(fai_text).encode())
        for name, path in contigs:
            seq = host.fasta_load_contig(fasta_path, name)
            with open(path, "rb") as g:
                text = g.read()
            rows_text, _, rows = call_contig(model, text, name, seq, **kw)
            f.write(rows_text)
            total += rows
    return total
