"""Final call-set merge (stage s6) and stage-4 group selection, host side.

    merge_calls      scripts/merge.py:15-145   pileup.vcf + haplotype.csv -> final VCF
    select_groups    HaplotypeModel/select_hetesnp_homosnp.py:75-177  low-quality candidates + their
                     5 high-quality heterozygous neighbours each side

Text in, text out, as in the reference (these stages are not accelerable math); written as single
passes over arrays instead of per-row dictionaries.
"""
from __future__ import annotations

import bisect
import math

MAJOR_CONTIGS = ["chr" + str(a) for a in list(range(1, 23)) + ["X", "Y"]] + [str(a) for a in list(range(1, 23)) + ["X", "Y"]]


def merge_calls(pileup_vcf_text: str, haplotype_csv_text: str, quality_threshold=15.0) -> str:
    """scripts/merge.py:15-145.  Pileup calls with QUAL <= threshold are replaced by the haplotype call of
    the same site when that one has QUAL >= 13 (INFO 'H'); the others keep the pileup call (INFO 'P') when
    it is a variant with QUAL >= 13 (or above the threshold)."""
    cat = {}
    for row in haplotype_csv_text.splitlines():
        if not row.strip():
            continue
        ctg, pos, gt, qual = row.strip().split("\t")
        cat[(ctg, pos)] = (gt, qual)
    out = []
    insert_hp = True
    for line in pileup_vcf_text.splitlines(keepends=True):
        if line.startswith("#"):
            out.append(line)
            if insert_hp:
                out.append('##INFO=<ID=P,Number=0,Type=Flag,Description="Result from pileup model">\n')
                out.append('##INFO=<ID=H,Number=0,Type=Flag,Description="Result from haplotype model">\n')
                insert_hp = False
            continue
        fields = line.strip().split("\t")
        ref, quality, filt, ctg, chr_offset = fields[3], float(fields[5]), fields[6], fields[0], int(fields[1])
        depth, af = fields[-1].split(":")[-2:]
        depth, af = int(depth), float(af)

        def keep_pileup():
            f2 = line.strip().split("\t")
            f2[7] = "P"
            out.append("\t".join(f2) + "\n")

        if quality <= quality_threshold:
            hit = cat.get((ctg, str(chr_offset)))
            if hit is None:                                           # KeyError branch, merge.py:122-133
                if filt != "RefCall" and quality >= 13:
                    keep_pileup()
                continue
            gt, qual = hit[0], float(hit[1])
            if qual < 13:                                             # merge.py:68-80
                if filt != "RefCall" and quality >= 13:
                    keep_pileup()
                continue
            if ref in gt:
                if gt[0] == gt[1]:
                    continue                                          # haplotype model says hom-ref
                new_gt, new_zy = gt.replace(ref, ""), "0/1"
            else:
                if gt[0] == gt[1]:
                    new_gt, new_zy = gt[0], "1/1"
                else:
                    new_gt, new_zy = ",".join(sorted(gt)), "1/2"
            quality = qual
            for sym in ("D", "I"):                                    # merge.py:100-111 ('D' test first, elif 'I')
                if sym in new_gt:
                    if new_zy in ("0/1", "1/1"):
                        new_gt = None
                    else:
                        new_gt, new_zy = gt.replace(sym, ""), "0/1"
                    break
            if new_gt is None:
                continue
            out.append("{0}\t{1}\t.\t{2}\t{3}\t{4}\t{5}\t{6}\t{7}\t{8}\n".format(
                ctg, chr_offset, ref, new_gt, str(quality), "PASS", "H", "GT:GQ:DP:AF",
                new_zy + ":%s:%d:%f" % (str(int(quality)), depth, af)))
        elif filt != "RefCall":
            keep_pileup()
    return "".join(out)


def parse_vcf_for_groups(vcf_text: str, quality_threshold):
    """select_hetesnp_homosnp.py:82-104: {contig: {pos: (genotype, quality)}} without the confident
    homozygous calls"""
    contigs = {}
    for row in vcf_text.splitlines():
        if not row or row[0] == "#":
            continue
        c = row.strip().split()
        genotype = c[9].split(":")[0].replace("|", "/")
        quality = float(c[5])
        if genotype in ("0/0", "1/1") and quality >= quality_threshold:
            continue
        contigs.setdefault(c[0], {})[int(c[1])] = (genotype, quality)
    return contigs


def select_groups(vcf_text: str, quality_threshold=19.0, adjacent_size=5, support_quality=14.0, nthreads=10,
                  reference_bug=True):
    """{contig: [[(pos, genotype, quality)] * (2*adjacent_size+1)]}: every site with quality < threshold
    together with its adjacent_size nearest heterozygous ('0/1') sites of quality >= support_quality on
    each side (select_hetesnp_homosnp.py:180-230; thresholds of scripts/s4...sh:57-65).

    reference_bug=True reproduces find_adjacent_sites storing its result after the contig loop
    (:226, one indentation level too shallow): of every chunk of ceil(n_contigs / nthreads) contigs only
    the LAST contig's groups survive.  With False every contig is kept."""
    contigs = parse_vcf_for_groups(vcf_text, quality_threshold)
    order = MAJOR_CONTIGS + list(contigs.keys())
    names = sorted(contigs.keys(), key=lambda x: order.index(x))
    step = math.ceil(len(names) / nthreads) if names else 1
    chunks = [names[i:i + step] for i in range(0, len(names), step)]
    groups = {}
    for chunk in chunks:
        for ci, contig in enumerate(chunk):
            d = contigs[contig]
            all_pos = sorted(d)
            het_idx = [i for i, p in enumerate(all_pos) if d[p][1] >= support_quality and d[p][0] == "0/1"]
            ctg_groups = []
            for i, p in enumerate(all_pos):
                if not d[p][1] < quality_threshold:
                    continue
                k = bisect.bisect_left(het_idx, i)                    # supports strictly left of i: het_idx[:k]
                left = het_idx[max(0, k - adjacent_size):k]
                k2 = bisect.bisect_right(het_idx, i)
                right = het_idx[k2:k2 + adjacent_size]
                if len(left) != adjacent_size or len(right) != adjacent_size:
                    continue
                ctg_groups.append([(all_pos[j], d[all_pos[j]][0], d[all_pos[j]][1]) for j in left + [i] + right])
            if not reference_bug or ci == len(chunk) - 1:
                groups[contig] = ctg_groups
    return groups
