"""Host-side readers and generators (libnanosnp_host.so)."""
import numpy as np
import pytest

from nanosnp_amd import host


def test_synth_is_deterministic_and_thread_independent(monkeypatch):
    a = host.synth_columns(42, 5000, coverage=30)
    b = host.synth_columns(42, 5000, coverage=30)
    assert np.array_equal(a.bases, b.bases) and np.array_equal(a.col_off, b.col_off) and np.array_equal(a.ref, b.ref)
    c = host.synth_columns(43, 5000, coverage=30)
    assert not np.array_equal(a.ref, c.ref)
    # ~Poisson(30) symbols per column, grammar characters only
    per_col = np.diff(a.col_off)
    assert 30 < per_col.mean() < 40
    assert set(np.unique(a.bases)) <= set(b"ACGTacgt*#+-123^I$")


def test_synth_windows_layout():
    w = host.synth_columns(7, 33 * 10, window=33)
    assert w.pos[32] + 1 != w.pos[33]            # a gap separates consecutive windows
    assert np.all(np.diff(w.pos[:33]) == 1)


def test_mpileup_roundtrip():
    cols = host.synth_columns(3, 300)
    text = cols.mpileup_text("chr1")
    pos, col_off, bases = host.mpileup_parse(text)
    assert np.array_equal(pos, cols.pos) and np.array_equal(col_off, cols.col_off) and np.array_equal(bases, cols.bases)
    # \r\n line ends, consecutive tabs and a missing final newline (cpp_aux.cpp:43-59, line_reader.cpp:95-127)
    t2 = text.replace(b"\n", b"\r\n").replace(b"\tN\t", b"\t\tN\t")[:-2]
    pos2, col_off2, bases2 = host.mpileup_parse(t2)
    assert np.array_equal(pos2, cols.pos) and np.array_equal(bases2, cols.bases)
    with pytest.raises(host.HostError):
        host.mpileup_parse(b"chr1\t5\tN\n")


def test_empty_inputs():
    pos, col_off, bases = host.mpileup_parse(b"")
    assert pos.size == 0 and col_off.tolist() == [0]
    x, names, p, r = host.pd_parse(b"")
    assert x.shape == (0, 33, 18) and names == []


def test_fasta_with_and_without_fai(tmp_path):
    rng = np.random.default_rng(0)
    s1 = rng.choice(list(b"ACGTN"), 1234).astype(np.uint8)
    fa = tmp_path / "a.fa"
    host.write_fasta(str(fa), "ctgA", s1, line=50)
    assert np.array_equal(host.fasta_load_contig(str(fa), "ctgA"), s1)
    # second contig, no index
    with open(fa, "ab") as f:
        f.write(b">ctgB some description\nACGTAC\nGG\n")
    (tmp_path / "a.fa.fai").unlink()
    assert bytes(host.fasta_load_contig(str(fa), "ctgB")) == b"ACGTACGG"
    assert np.array_equal(host.fasta_load_contig(str(fa), "ctgA"), s1)
    with pytest.raises(host.HostError):
        host.fasta_load_contig(str(fa), "nope")


def test_pd_parse_fields():
    row = " ".join(str(i - 300) for i in range(594)) + " "
    text = (row + "\tchr7:12345:" + "A" * 16 + "G" + "C" * 16 + "\t30-XT 5\n").encode()
    x, names, pos, refb = host.pd_parse(text * 3)
    assert x.shape == (3, 33, 18) and x[1].ravel().tolist() == [i - 300 for i in range(594)]
    assert names == ["chr7"] * 3 and pos.tolist() == [12345] * 3 and refb.tolist() == [ord("G")] * 3


def test_hap_planes_shape_and_padding():
    seq, bq, mq, hap, ref = host.synth_hap_planes(1, 8, coverage=30, depth=90, length=33)
    assert seq.shape == (8, 90, 33) and ref.shape == (8, 33)
    assert set(np.unique(seq)) <= {-2, -1, 0, 1, 2, 3, 4}
    pad = (seq == -2)
    assert np.array_equal(pad, hap == -2) and np.array_equal(pad, bq == -2)
    # rows are ordered by HP at the centre column (create_pileup_haplotype.py:158-165)
    for n in range(8):
        hp = hap[n, :, 16]; hp = hp[hp > 0]
        assert np.all(np.diff(hp) >= 0)


def test_haplotype_ref_rows_follow_the_reference_quirks():
    """dataset_dev.py:106-120: {'A':1,'C':2,'G':3,'T':4,'N':0}, anything else / out of range -> 0,
    negative 0-based indices wrap around (Python indexing).  (Corner cases on a hand-made sequence; the pin by the reference's
    own PileupFeature / HaplotypeFeature classes is tests/test_two_stage_host.py.)"""
    seq = b"ACGTNacgtRACGTACGTACGTACGTACGTACGTACGTAC"
    refs = {"c1": np.frombuffer(seq, np.uint8)}
    base2int = {"A": 1, "C": 2, "G": 3, "T": 4, "N": 0}

    def ref_impl(ctg, pos, length):        # the reference loop, restated with Python strings
        row = []
        for j in range(pos - length // 2, pos + length // 2 + 1):
            try:
                row.append(base2int[{"c1": seq.decode()}[ctg][j - 1]])
            except Exception:
                row.append(0)
        return row
    cands = ["c1:20", "c1:3", "c1:38", "nope:5"]
    got = host.haplotype_ref_rows(refs, cands, 33)
    for i, c in enumerate(cands):
        ctg, pos = c.split(":")
        assert got[i].tolist() == ref_impl(ctg, int(pos), 33), c
    groups = [[f"c1:{p}" for p in range(2, 24, 2)]]
    g = host.haplotype_ref_rows(refs, ["c1:12"], 11, position_lists=groups)
    assert g.shape == (1, 11) and g[0].tolist() == [base2int.get(seq.decode()[p - 1], 0) for p in range(2, 24, 2)]


def test_line_cuts_and_halo_ranges_of_the_streamed_pipeline():
    """nanosnp_amd.pipeline.line_cuts / halo_range: parts of whole lines that tile the text, halos of up to 16 whole lines that stop
    at the ends of the text; the same on bytes and on a memory-mapped file"""
    import mmap
    import tempfile
    from nanosnp_amd.pipeline import halo_range, line_cuts
    cols = host.synth_columns(3, 500, coverage=12)
    text = cols.mpileup_text_native("chr9").tobytes()
    assert text == cols.mpileup_text("chr9")
    starts = [0] + [i + 1 for i, ch in enumerate(text) if ch == 10]
    with tempfile.TemporaryFile() as f:
        f.write(text); f.flush()
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        for buf in (text, mm):
            for parts in (1, 2, 3, 7, 64, 501):
                cuts = line_cuts(buf, parts)
                assert cuts[0] == 0 and cuts[-1] == len(text) and all(a <= b for a, b in zip(cuts, cuts[1:])) and all(c in starts for c in cuts)
                for a, b in zip(cuts, cuts[1:]):
                    if b <= a:
                        continue
                    a2, b2, n_lo, n_hi = halo_range(buf, a, b)
                    ia, ib = starts.index(a), starts.index(b)
                    assert n_lo == min(16, ia) and n_hi == min(16, len(starts) - 1 - ib)
                    assert a2 == starts[ia - n_lo] and b2 == starts[ib + n_hi]
                    pos, off, bases = host.mpileup_parse_range(buf, a2, b2)
                    assert np.array_equal(pos, cols.pos[ia - n_lo:ib + n_hi])
                    assert np.array_equal(bases, cols.bases[cols.col_off[ia - n_lo]:cols.col_off[ib + n_hi]])
        mm.close()
    # caller-supplied (pinned-style) buffers, and the too-small case
    out = (np.empty(600, np.int64), np.empty(601, np.int64), np.empty(len(text), np.uint8))
    pos, off, bases = host.mpileup_parse_range(text, 0, len(text), out=out)
    assert np.array_equal(pos, cols.pos) and np.array_equal(off, cols.col_off) and np.array_equal(bases, cols.bases) and pos.base is out[0]
    with pytest.raises(host.HostError):
        host.mpileup_parse_range(text, 0, len(text), out=(np.empty(10, np.int64), np.empty(11, np.int64), np.empty(len(text), np.uint8)))
