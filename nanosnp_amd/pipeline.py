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
    """One contig: returns (vcf_rows: bytes, n_sites, n_rows).  model: pileup_model.LSTMNetwork.

    Under an initialised torch.distributed process group (one process per GPU, torchrun) the contig's columns are
    statically sharded over the ranks (nanosnp_amd.dist.shard_columns: each rank encodes its range plus a 16-column halo
    each side, re-computed not exchanged), every rank runs the forward on the sites centred in its own range, and the
    per-site calls are gathered to rank 0 in rank = position order, where the rows are formatted exactly as a single
    process would format them (the reference's batches of `batch_size` sites run over the whole site list).  Ranks other
    than 0 return (b"", n_sites_total, 0)."""
    import torch
    import torch.distributed as tdist
    from .dist import gather_varlen, shard_columns
    ctx = model.ctx
    pos, col_off, bases = host.mpileup_parse(mpileup_text)
    if pos.size == 0:
        return b"", 0, 0
    if pos.max() > chr_seq.size or pos.min() < 1:
        raise ValueError(f"{contig}: position outside the reference sequence")
    ref = np.ascontiguousarray(chr_seq[pos - 1])
    sharded = tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
    rank, world = (tdist.get_rank(), tdist.get_world_size()) if sharded else (0, 1)
    M = int(pos.size)
    c_lo, c_hi, own_lo, own_hi = shard_columns(M, rank, world, halo=16)
    dev = torch.device("cuda", ctx.device)          # the context's device, not torch's current one (one GPU per rank under torchrun)
    b0, b1 = int(col_off[c_lo]), int(col_off[c_hi])
    d_bases = torch.from_numpy(bases[b0:b1] if b1 > b0 else np.zeros(1, np.uint8)).to(dev)
    d_off = torch.from_numpy(col_off[c_lo:c_hi + 1] - b0).to(dev)
    d_ref = torch.from_numpy(ref[c_lo:c_hi]).to(dev)
    d_pos = torch.from_numpy(pos[c_lo:c_hi]).to(dev)
    n_loc = 0
    if c_hi > c_lo:
        counts, depth, flags = ctx.pileup_encode_columns(d_bases, d_off, d_ref, min_af, min_coverage)
        centers, n_loc = ctx.pileup_select_sites(d_pos, flags)
        if n_loc:
            owned = (centers >= own_lo - c_lo) & (centers < own_hi - c_lo)      # halo columns belong to the neighbours
            centers = centers[owned].contiguous()
            n_loc = int(centers.shape[0])
    if n_loc:
        gt, zy = ctx.pileup_forward_windows(counts, centers)
        ga, za, gm, zm, _ = ctx.pileup_postprocess(gt, zy)
        cov = counts[centers][:, COV_CHANNELS].to(torch.float32)              # predict.py:63
        # compact call rows: global column index, argmax / max of both heads, the eight coverage channels (all exact in float64)
        rows = torch.cat([(centers + c_lo).to(torch.float64)[:, None], ga.to(torch.float64)[:, None], za.to(torch.float64)[:, None],
                          gm.to(torch.float64)[:, None], zm.to(torch.float64)[:, None], cov.to(torch.float64)], dim=1)
    else:
        rows = torch.zeros((0, 13), dtype=torch.float64, device=dev)
    if sharded:
        backend_dev = dev if tdist.get_backend() == "nccl" else "cpu"
        rows = gather_varlen(rows.to(backend_dev))
        if rank != 0:
            n_tot = torch.zeros(1, dtype=torch.int64, device=backend_dev)
            tdist.broadcast(n_tot, src=0)
            return b"", int(n_tot.item()), 0
        tdist.broadcast(torch.tensor([rows.shape[0]], dtype=torch.int64, device=backend_dev), src=0)
    n_sites = int(rows.shape[0])
    if n_sites == 0:
        return b"", 0, 0
    r = rows.cpu().numpy()
    c_host = r[:, 0].astype(np.int64)
    table = host.ContigTable([contig])
    ids = np.zeros(n_sites, np.int32)
    site_pos = pos[c_host]
    site_ref = ref[c_host] & 0xDF                                          # make_predict_data/main.cpp:91 upper-cases
    # the VCF rows depend on the batch boundary: one native call formats every batch (OpenMP over the batches)
    text, n_rows = host.vcf_format_batches(table, ids, site_pos, site_ref, r[:, 1].astype(np.uint8), r[:, 2].astype(np.uint8),
                                           r[:, 3].astype(np.float32), r[:, 4].astype(np.float32), r[:, 5:13].astype(np.float32),
                                           batch_size=batch_size, score_mode=score_mode)
    return text, n_sites, n_rows


def call_variants(model, contigs, fasta_path, fai_text, output_file, **kw):
    """contigs: iterable of (name, path to <name>.mpileup).  Writes pileup.vcf (rank 0 only under torch.distributed: see
    call_contig); returns total rows."""
    import torch.distributed as tdist
    root = not (tdist.is_available() and tdist.is_initialized()) or tdist.get_rank() == 0
    total = 0
    f = open(output_file, "wb") if root else None
    try:
        if root:
            f.write(host.vcf_header(fai_text).encode())
        for name, path in contigs:
            seq = host.fasta_load_contig(fasta_path, name)
            with open(path, "rb") as g:
                text = g.read()
            rows_text, _, rows = call_contig(model, text, name, seq, **kw)
            if root:
                f.write(rows_text)
            total += rows
    finally:
        if f:
            f.close()
    return total
