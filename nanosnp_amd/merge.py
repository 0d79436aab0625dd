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

_CHROMOSOMES = [str(n) for n in range(1, 23)] + ["X", "Y"]
MAJOR_CONTIGS = ["chr" + c for c in _CHROMOSOMES] + _CHROMOSOMES            # UCSC names first, then Ensembl names (select_hetesnp_homosnp.py:14)

_INFO_HEADER = ('##INFO=<ID=P,Number=0,Type=Flag,Description="Result from pileup model">\n'
                '##INFO=<ID=H,Number=0,Type=Flag,Description="Result from haplotype model">\n')
_MIN_CALL_QUAL = 13.0        # scripts/merge.py:68,122: below it neither model's call is written


def haplotype_alleles(ref: str, pair: str):
    """The two genotype letters of a haplotype call (ACGT, D, I) against the site's reference base -> (ALT text, GT text), or None
    when the merged file holds no row for the site.  Decision table of scripts/merge.py:82-111:

        ref among the letters, both equal      -> None (homozygous reference)
        ref among the letters                  -> the other letter, 0/1
        both equal                             -> that letter, 1/1
        two different non-reference letters    -> both, sorted, 1/2
        then an indel letter in ALT ('D' is looked at before 'I'): one ALT allele -> None; two -> the other letter, 0/1"""
    first, second = pair[0], pair[1]
    if ref in pair:
        if first == second:
            return None
        alt, zygosity = pair.replace(ref, ""), "0/1"
    elif first == second:
        alt, zygosity = first, "1/1"
    else:
        alt, zygosity = ",".join(sorted(pair)), "1/2"
    for letter in "DI":
        if letter in alt:
            return (pair.replace(letter, ""), "0/1") if zygosity == "1/2" else None
    return alt, zygosity


def merge_calls(pileup_vcf_text: str, haplotype_csv_text: str, quality_threshold=15.0) -> str:
    """scripts/merge.py:15-145 as one pass with a decision per row.  A pileup row whose QUAL is at most the threshold is REPLACED by
    the haplotype model's call of the same site when that call has QUAL >= 13 (INFO 'H', or no row at all: haplotype_alleles);
    every other row stays as the pileup model wrote it (INFO 'P') provided it is a variant (FILTER != RefCall) whose QUAL is above
    the threshold or at least 13.  Byte-identical to the reference's output (tests/test_next_rows.py)."""
    calls = {}
    for record in haplotype_csv_text.splitlines():
        record = record.strip()
        if record:
            contig, position, letters, score = record.split("\t")
            calls[contig, position] = (letters, score)
    merged, header_pending = [], True
    for raw in pileup_vcf_text.splitlines(keepends=True):
        if raw.startswith("#"):
            merged.append(raw)
            if header_pending:
                merged.append(_INFO_HEADER)
                header_pending = False
            continue
        col = raw.strip().split("\t")
        position, pileup_qual = int(col[1]), float(col[5])
        sample = col[-1].split(":")
        depth, allele_frequency = int(sample[-2]), float(sample[-1])
        variant = col[6] != "RefCall"
        hap = calls.get((col[0], str(position))) if pileup_qual <= quality_threshold else None
        hap_qual = float(hap[1]) if hap is not None else None
        if hap is not None and hap_qual >= _MIN_CALL_QUAL:
            decided = haplotype_alleles(col[3], hap[0])
            if decided is not None:
                alt, zygosity = decided
                merged.append(f"{col[0]}\t{position}\t.\t{col[3]}\t{alt}\t{hap_qual}\tPASS\tH\tGT:GQ:DP:AF\t"
                              f"{zygosity}:{int(hap_qual)}:{depth:d}:{allele_frequency:f}\n")
        elif variant and (pileup_qual > quality_threshold or pileup_qual >= _MIN_CALL_QUAL):
            col[7] = "P"
            merged.append("\t".join(col) + "\n")
    return "".join(merged)


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
