"""Stage-4 read-matrix builder: haplotagged reads -> per-group read x position matrices -> the padded planes the haplotype
feature kernel consumes.

    read_matrices   create_pileup_haplotype.single_group_pileup_haplotype_feature :22-134   the two pileup passes over the
                    alignment file: coverage filter of the groups, then one row per read name over the union of the groups'
                    positions (base code / HP tag / base quality / mapping quality; 0 = the read does not cover the column)
    group_planes    :137-207 + write_to_bins.py:15-61   per group the 11 support columns and the 33-wide window, reads kept when
                    their centre base is non-zero, ordered by the HP tag at the centre, padded with -2 / cut at D rows -- on the
                    device through nsnp_hap_arrange_reads (the planes never visit the host again before the feature kernel)

The alignment file is whatever the caller's BAM library hands over: any object with pysam's ``pileup(contig, start, end,
min_base_quality=0, min_mapping_quality=0)`` iteration (columns with ``.pos``, ``.n``, ``.pileups``; pileup reads with
``.alignment.{query_name, has_tag, get_tag, query_sequence, query_qualities, mapping_quality}``, ``.is_del``, ``.is_refskip``,
``.query_position``) -- a ``pysam.AlignmentFile`` works unchanged (htslib is not part of this repository's image; the tests
drive a stand-in with synthetic reads and compare with the reference function run on the same stand-in).
"""
from __future__ import annotations

import numpy as np

BASE_TO_INT = {"A": 1, "C": 2, "G": 3, "T": 4}        # create_pileup_haplotype.py:7


class ReadMatrices:
    """positions: sorted 1-based columns; seq / hap / baseq / mapq: int32 [n_reads, n_positions] in first-seen read order"""

    def __init__(self, contig, positions, names, seq, hap, baseq, mapq, groups):
        self.contig, self.positions, self.names = contig, positions, names
        self.seq, self.hap, self.baseq, self.mapq, self.groups = seq, hap, baseq, mapq, groups
        self._col = {p: i for i, p in enumerate(positions)}

    def columns(self, wanted):
        return np.array([self._col[int(p)] for p in wanted], np.int64)


def read_matrices(samfile, groups, max_coverage, pileup_flanking_size=16):
    """groups: [[(contig, position)] * (2 * adjacent_size + 1)] (or objects with .ctgname / .position as the reference's SNPItem),
    all on one contig.  Returns ReadMatrices, or None when no group survives the coverage filter or a read carries a base
    outside ACGT at a wanted column (the reference's bare `except` then returns nothing for the whole call, :209-214)."""
    def cp(item):
        return (item.ctgname, int(item.position)) if hasattr(item, "position") else (item[0], int(item[1]))
    groups = [[cp(it) for it in g] for g in groups]
    if not groups:
        return None
    ctg = groups[0][0][0]
    assert all(g[0][0] == ctg for g in groups)
    positions = sorted({p for g in groups for _, p in g})
    pset = set(positions)
    failed = set()
    for col in samfile.pileup(ctg, positions[0], positions[-1], min_base_quality=0, min_mapping_quality=0):       # :39-47
        p = col.pos + 1
        if p < positions[0]:
            continue
        if p > positions[-1]:
            break
        if p in pset and col.n > max_coverage:
            failed.add(p)
    if failed:
        groups = [g for g in groups if not any(p in failed for _, p in g)]                                      # :50-60
    if not groups:
        return None
    ext = set()
    for g in groups:                                                                                             # :73-83
        mid = len(g) // 2
        for k, (_, p) in enumerate(g):
            if k == mid:
                ext.update(range(p - pileup_flanking_size, p + pileup_flanking_size + 1))
            else:
                ext.add(p)
    ext = sorted(ext)
    col_of = {p: i for i, p in enumerate(ext)}
    row_of, names = {}, []
    cells = []                                            # (row, col, seq, hap, baseq, mapq)
    for col in samfile.pileup(ctg, ext[0], ext[-1], min_base_quality=0, min_mapping_quality=0):                 # :89-134
        p = col.pos + 1
        if p < ext[0]:
            continue
        if p > ext[-1]:
            break
        i = col_of.get(p)
        if i is None:
            continue
        # The reference asserts here (:98 the coverage of EVERY wanted column - the first pass looked at the group positions only, a
        # window column can still be deeper; :104 the HP tag is 1, 2 or 3) inside its bare try / except (:209-214), which prints and
        # returns its still empty lists: nothing for this chunk of groups.  Same outcome here.
        if col.n > max_coverage:
            return None
        for pr in col.pileups:
            aln = pr.alignment
            tag = aln.get_tag("HP") if aln.has_tag("HP") else 3
            if tag not in (1, 2, 3):
                return None
            r = row_of.get(aln.query_name)
            if r is None:
                r = row_of[aln.query_name] = len(names)
                names.append(aln.query_name)
            if not pr.is_del and not pr.is_refskip:
                code = BASE_TO_INT.get(str.upper(aln.query_sequence[pr.query_position]))
                if code is None:
                    return None
                cells.append((r, i, code, tag, int(aln.query_qualities[pr.query_position]), int(aln.mapping_quality)))
            elif pr.is_del:
                cells.append((r, i, -1, tag, 0, int(aln.mapping_quality)))
    R, P = len(names), len(ext)
    seq = np.zeros((R, P), np.int32); hap = np.zeros((R, P), np.int32)
    bq = np.zeros((R, P), np.int32); mq = np.zeros((R, P), np.int32)
    if cells:
        c = np.array(cells, np.int64)
        seq[c[:, 0], c[:, 1]] = c[:, 2]; hap[c[:, 0], c[:, 1]] = c[:, 3]
        bq[c[:, 0], c[:, 1]] = c[:, 4]; mq[c[:, 0], c[:, 1]] = c[:, 5]
    return ReadMatrices(ctg, ext, names, seq, hap, bq, mq, groups)


def group_slices(rm, pileup_flanking_size=16):
    """per group the column sets of :141 (its positions) and :178 (the window around its centre) and the position strings"""
    out = []
    for g in rm.groups:
        gpos = [p for _, p in g]
        centre = gpos[len(gpos) // 2]
        wpos = list(range(centre - pileup_flanking_size, centre + pileup_flanking_size + 1))
        out.append(dict(candidate=f"{rm.contig}:{centre}", haplotype_positions=[f"{rm.contig}:{p}" for p in gpos],
                        hap_cols=rm.columns(gpos), pile_cols=rm.columns(wpos)))
    return out


def group_planes(ctx, rm, max_haplotype_depth, max_pileup_depth, pileup_flanking_size=16):
    """-> (candidate_positions, haplotype_positions, (seq, baseq, mapq, hap) int32 cuda [N, D_h, 11], the same [N, D_p, 33],
    depths_h, depths_p): reads filtered on the centre base, HP-sorted, padded with -2 and cut at D, exactly what
    write_to_bins.py stores and nanosnp_amd.predict.predict_haplotype consumes (plus reference rows)."""
    import torch
    sl = group_slices(rm, pileup_flanking_size)
    dev = torch.device("cuda", ctx.device)
    full = [torch.from_numpy(a).to(dev) for a in (rm.seq, rm.baseq, rm.mapq, rm.hap)]
    outs = []
    for key, D in (("hap_cols", max_haplotype_depth), ("pile_cols", max_pileup_depth)):
        idx = torch.from_numpy(np.stack([s[key] for s in sl])).to(dev)                   # [N, L]
        mats = [f[:, idx].permute(1, 0, 2).contiguous() for f in full]                    # [N, R, L] views of the one matrix
        outs.append(ctx.hap_arrange_reads(*mats, int(D)))
    (hs, hb, hm, hh, dh), (ps, pb, pm, ph, dp) = outs
    return ([s["candidate"] for s in sl], [s["haplotype_positions"] for s in sl], (hs, hb, hm, hh), (ps, pb, pm, ph), dh, dp)
