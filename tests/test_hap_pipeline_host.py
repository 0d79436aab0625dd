"""Host pieces of the streamed stage-5 pipeline (nanosnp_amd/hap_pipeline.py, nsnp_stage.c): staging reads, position-field parsing and
the reference-row gather - the latter on CPU tensors against host.haplotype_ref_rows, the per-site restatement of
HaplotypeModel/dataset_dev.py:106-120,150-162."""
import os

import numpy as np
import pytest

from nanosnp_amd import host, sitefile
from nanosnp_amd.hap_pipeline import DeviceReference, HapArraySource, HapBinSource, _as_fields


def test_stage_values_from_files_and_arrays(tmp_path):
    rng = np.random.default_rng(1)
    a = rng.integers(-2, 94, 3_000_001).astype(np.int32)
    p = tmp_path / "a.bin"
    with open(p, "wb") as f:
        f.write(b"x" * 64); f.write(a.tobytes())
    fd = os.open(p, os.O_RDONLY)
    try:
        for off, n in ((0, a.size), (5, 1_234_567), (a.size - 3, 3), (17, 0)):
            d32 = np.full(n + 2, 77, np.int32); d8 = np.full(n + 2, 77, np.int8)
            assert host.stage_values(d32, n, fd=fd, src_off=64 + 4 * off) == 0 and np.array_equal(d32[:n], a[off:off + n]) and (d32[n:] == 77).all()
            assert host.stage_values(d8, n, fd=fd, src_off=64 + 4 * off) == 0 and np.array_equal(d8[:n], a[off:off + n]) and (d8[n:] == 77).all()
            assert host.stage_values(d8, n, src=a, src_off=4 * off) == 0 and np.array_equal(d8[:n], a[off:off + n])
            assert host.stage_values(d32, n, src=a, src_off=4 * off) == 0 and np.array_equal(d32[:n], a[off:off + n])
        a8 = a.astype(np.int8)
        d8 = np.empty(1000, np.int8)
        assert host.stage_values(d8, 1000, src=a8, src_off=11, src_dtype=np.int8) == 0 and np.array_equal(d8, a8[11:1011])
        # values that do not fit int8 are counted (the caller then sends int32), on the file path and on the memory path
        b = a.copy(); b[[5, 1_500_000, 2_999_999]] = (255, -129, 128)
        with open(p, "r+b") as f:
            f.seek(64); f.write(b.tobytes())
        d8 = np.empty(b.size, np.int8)
        assert host.stage_values(d8, b.size, fd=fd, src_off=64) == 3
        assert host.stage_values(d8, b.size, src=b) == 3
        assert host.stage_values(d8, 1_400_000, src=b) == 1
        with pytest.raises(host.HostError):
            host.stage_values(np.empty(10, np.int32), 10, fd=fd, src_off=64 + 4 * a.size)       # past the end of the file
        with pytest.raises(host.HostError):
            host.stage_values(np.empty(4, np.int32), 4, src=a8, src_dtype=np.int8)             # int8 -> int32 is not offered
        with pytest.raises(host.HostError):
            host.stage_values(np.empty(3, np.int8), 4, src=a)
    finally:
        os.close(fd)


def test_position_fields():
    tbl = host.ContigTable(["chr1", "chr10", "ctgA"])
    f = _as_fields(["chr1:12", "chr10:0", "ctgA:+7", "chrX:99", "chr1: 345 ", "chr1:-3", "chr1:000123456789012345"])
    pos, ctg = host.parse_ctg_pos(f, tbl)
    assert pos.tolist() == [12, 0, 7, 99, 345, -3, 123456789012345] and ctg.tolist() == [0, 1, 2, -1, 0, 0, 0]
    f2 = _as_fields([["chr1:1", "chr10:2"], ["ctgA:3", "zz:4"]])
    pos, ctg = host.parse_ctg_pos(f2, tbl)
    assert pos.shape == (2, 2) and pos.tolist() == [[1, 2], [3, 4]] and ctg.tolist() == [[0, 1], [2, -1]]
    wide = np.zeros((3, 300), np.uint8); wide[:, :7] = np.frombuffer(b"chr1:42", np.uint8)       # StringAtom(300) fields
    assert host.parse_ctg_pos(wide, tbl)[0].tolist() == [42, 42, 42]
    f3 = _as_fields(["chr1:1_000", "chr1: +1_2_3 ", "chr1:-0_0"])                          # int() takes single underscores between digits (PEP 515)
    assert host.parse_ctg_pos(f3, tbl)[0].tolist() == [int(s.split(":")[1]) for s in ("chr1:1_000", "chr1: +1_2_3 ", "chr1:-0_0")] == [1000, 123, 0]
    for bad in ("chr1", "chr1:", "chr1:1:2", "chr1:12a", "chr1:1.5", ":", "chr1:--1", "chr1:_1", "chr1:1_", "chr1:1__0", "chr1:+_1", "chr1:1 0"):
        with pytest.raises(host.HostError):
            host.parse_ctg_pos(_as_fields(["chr1:5", bad]), tbl)
    assert host.parse_ctg_pos(np.zeros((0, 9), np.uint8), tbl)[0].shape == (0,)
    big = _as_fields([f"chr10:{i}" for i in range(100_000)])
    pos, ctg = host.parse_ctg_pos(big, tbl)
    assert np.array_equal(pos, np.arange(100_000)) and (ctg == 1).all()


def test_pd_position_fields_and_int16_staging(tmp_path):
    """the two host pieces of pipeline.predict_pileup_bins: the `position` strings of a .pd.bin read as PileupModel/dataset.py:127-132
    reads them (strip, split(':') into exactly three parts, int(pos), ord(seq[16])), and int32 counts narrowed to int16 while staged"""
    tbl = host.ContigTable(["chr1", "chr10"])
    seq = "ACGTACGTACGTACGT" + "G" + "TTTTTTTTTTTTTTTT"
    fields = [f"chr1:12:{seq}", f" chr10: 7 :{seq[:16]}c{seq[17:]} ", f"chrUn:+99:{'N' * 33}", f"chr1:-5:{seq[:17]}", f"chr1:000123:{seq}extra"]
    rows = np.zeros((len(fields), sitefile.POSITION_WIDTH), np.uint8)
    for i, f in enumerate(fields):
        rows[i, :len(f)] = np.frombuffer(f.encode(), np.uint8)
    pos, ctg, refb = host.parse_ctg_pos_ref(rows, tbl)
    want = [(f.strip().split(":")) for f in fields]
    assert pos.tolist() == [int(w[1]) for w in want] == [12, 7, 99, -5, 123]
    assert refb.tolist() == [ord(w[2][16]) for w in want] and ctg.tolist() == [0, 1, -1, 0, 0]
    for bad in ("chr1:12", f"chr1:12:{seq}:x", f"chr1:1x:{seq}", f"chr1::{seq}", "chr1:12:" + seq[:16], f"chr1:1.0:{seq}", ""):
        r = rows[:2].copy(); r[1] = 0; r[1, :len(bad)] = np.frombuffer(bad.encode(), np.uint8)
        with pytest.raises(host.HostError):
            host.parse_ctg_pos_ref(r, tbl)
    assert [a.shape for a in host.parse_ctg_pos_ref(np.zeros((0, 83), np.uint8), tbl)] == [(0,)] * 3
    many = np.zeros((50_000, 83), np.uint8)
    txt = [f"chr10:{i * 3}:{seq}" for i in range(50_000)]
    for i, f in enumerate(txt):
        many[i, :len(f)] = np.frombuffer(f.encode(), np.uint8)
    pos, ctg, refb = host.parse_ctg_pos_ref(many, tbl)
    assert np.array_equal(pos, np.arange(50_000) * 3) and (ctg == 1).all() and (refb == ord("G")).all()
    # int32 -> int16: exact when every value fits, counted when not; from a file and from memory
    rng = np.random.default_rng(2)
    a = rng.integers(-300, 301, 2_000_003).astype(np.int32)
    a[[0, 7, 1_999_999]] = (32767, -32768, 1234)
    p = tmp_path / "c.bin"
    with open(p, "wb") as f:
        f.write(b"y" * 128); f.write(a.tobytes())
    fd = os.open(p, os.O_RDONLY)
    try:
        for off, n in ((0, a.size), (3, 1_000_001), (a.size - 1, 1), (9, 0)):
            d = np.full(n + 2, 77, np.int16)
            assert host.stage_values(d, n, fd=fd, src_off=128 + 4 * off) == 0 and np.array_equal(d[:n], a[off:off + n]) and (d[n:] == 77).all()
            assert host.stage_values(d, n, src=a, src_off=4 * off) == 0 and np.array_equal(d[:n], a[off:off + n])
        b = a.copy(); b[[1, 1_000_000, 2_000_002]] = (32768, -32769, 1 << 20)
        d = np.empty(b.size, np.int16)
        assert host.stage_values(d, b.size, src=b) == 3 and host.stage_values(d, 1_000_000, src=b) == 1
        a16 = a.astype(np.int16)
        d = np.empty(500, np.int16)
        assert host.stage_values(d, 500, src=a16, src_off=6, src_dtype=np.int16) == 0 and np.array_equal(d, a16[3:503])
        with pytest.raises(host.HostError):
            host.stage_values(np.empty(4, np.int32), 4, src=a16, src_dtype=np.int16)             # widening is not offered
    finally:
        os.close(fd)
    # the coverage slice of predict.py:63 from a staged pass (int16 or int32 values), as float32
    ch = [0, 1, 2, 3, 9, 10, 11, 12]
    for dt in (np.int16, np.int32):
        x = rng.integers(-40, 4000, (20_001, 33, 18)).astype(dt)
        assert np.array_equal(host.window_channels(x.reshape(-1), 20_001, 16, ch), x[:, 16, ch].astype(np.float32))
        out = np.full((5, 8), -1, np.float32)
        host.window_channels(x.reshape(-1), 3, 0, ch, out=out)
        assert np.array_equal(out[:3], x[:3, 0, ch]) and (out[3:] == -1).all()
    assert host.window_channels(np.zeros(0, np.int16), 0, 16, ch).shape == (0, 8)
    for bad in (lambda: host.window_channels(np.zeros(594, np.int16), 1, 16, [18]), lambda: host.window_channels(np.zeros(594, np.int16), 1, 33, ch),
                lambda: host.window_channels(np.zeros(593, np.int16), 1, 16, ch), lambda: host.window_channels(np.zeros(594, np.int8), 1, 16, ch)):
        with pytest.raises(host.HostError):
            bad()


def test_reference_rows_follow_the_reference_quirks_on_any_device():
    """DeviceReference.rows (torch gather; here on CPU tensors) against host.haplotype_ref_rows: lower-case / N / IUPAC bases, positions
    past the end, NEGATIVE 0-based indices that wrap like Python indexing, unknown contigs, an empty contig"""
    import torch
    rng = np.random.default_rng(5)
    refs = {"c1": rng.choice(list(b"ACGTacgtNRY"), 500).astype(np.uint8), "c2": rng.choice(list(b"ACGT"), 40).astype(np.uint8), "c3": np.zeros(0, np.uint8)}
    dr = DeviceReference(refs, torch.device("cpu"))
    cands, hlists = [], []
    for i in range(400):
        ctg = ["c1", "c2", "c3", "zz"][int(rng.integers(0, 4))]
        p = int(rng.choice([1, 2, 5, 16, 17, 30, 39, 40, 41, 56, 57, 480, 499, 500, 501, 520, 0, -3, -30, -600]))
        cands.append(f"{ctg}:{p}")
        hlists.append([f"{['c1', 'c2', 'zz'][int(rng.integers(0, 3))]}:{int(rng.integers(-60, 560))}" for _ in range(11)])
    want_p = host.haplotype_ref_rows(refs, cands, 33)
    want_h = host.haplotype_ref_rows(refs, cands, 11, position_lists=hlists)
    cp, cc = host.parse_ctg_pos(_as_fields(cands), dr.table)
    hp, hc = host.parse_ctg_pos(_as_fields(hlists), dr.table)
    off = torch.arange(-16, 17)[None, :]
    got_p = dr.rows(torch.from_numpy(cc)[:, None].expand(-1, 33), torch.from_numpy(cp)[:, None] - 1 + off)
    got_h = dr.rows(torch.from_numpy(hc), torch.from_numpy(hp) - 1)
    assert np.array_equal(got_p.numpy(), want_p) and np.array_equal(got_h.numpy(), want_h)
    assert want_p.max() == 4 and (want_p == 0).any()
    dr.add_names(["zz"])                                   # known by name now, still without a sequence: same rows, ids stay
    cp2, cc2 = host.parse_ctg_pos(_as_fields(cands), dr.table)
    assert np.array_equal(cc2[cc >= 0], cc[cc >= 0]) and (cc2 >= 0).all()
    assert np.array_equal(dr.rows(torch.from_numpy(cc2)[:, None].expand(-1, 33), torch.from_numpy(cp2)[:, None] - 1 + off).numpy(), want_p)


def test_sources_hand_out_the_same_bytes(tmp_path):
    n, D = 37, 12
    pp = host.synth_hap_planes(3, n, 30, D, 33); ph = host.synth_hap_planes(4, n, 30, D, 11)
    cands = [f"ctgA:{100 + 3 * i}" for i in range(n)]
    hpos = [[f"ctgA:{100 + 3 * i + k}" for k in range(11)] for i in range(n)]
    names = dict(zip(sitefile.HAP_PLANES, (ph[0], ph[3], ph[1], ph[2], pp[0], pp[3], pp[1], pp[2])))
    for dt in ("int8", "int32"):
        p = tmp_path / f"s_{dt}.bin"
        sitefile.write_haplotype_bin(p, cands, hpos, names, plane_dtype=dt)
        src = HapBinSource(p)
        arr = HapArraySource(pp[:4], ph[:4], cands, hpos)
        assert (src.n, src.Dp, src.Dh, src.elem) == (n, D, D, 1 if dt == "int8" else 4) and arr.elem == 4
        for name, want in (("pileup_baseq", pp[1]), ("haplotype_hap", ph[3])):
            for lo, hi in ((0, n), (5, 9), (n - 1, n)):
                cnt = want[lo:hi].size
                for s_ in (src, arr):
                    d8 = np.empty(cnt, np.int8)
                    assert s_.stage_plane(name, lo, hi, d8) == 0 and np.array_equal(d8, want[lo:hi].reshape(-1))
                d32 = np.empty(cnt, np.int32)
                if dt == "int32":
                    assert src.stage_plane(name, lo, hi, d32) == 0 and np.array_equal(d32, want[lo:hi].reshape(-1))
        cf, hf = src.position_fields(3, 8)
        tbl = host.ContigTable(["ctgA"])
        assert host.parse_ctg_pos(cf.reshape(5, -1), tbl)[0].tolist() == [100 + 3 * i for i in range(3, 8)]
        assert host.parse_ctg_pos(hf, tbl)[0][1].tolist() == [100 + 3 * 4 + k for k in range(11)]
        src.close()
    (tmp_path / "nope.bin").write_bytes(b"NSNPBIN1" + bytes(8))            # a valid container without the arrays
    with pytest.raises(sitefile.SiteFileError):
        HapBinSource(tmp_path / "nope.bin")


def test_load_reference_file_reads_a_fasta_as_get_truth_does(tmp_path):
    """get_truth.py:88-104 (checked against the reference's function in the development container): the name ends at the first blank,
    every '>' of it is dropped, sequence lines are stripped and joined with their case, a header without sequence leaves no entry, the
    later of two equal names wins, text in front of the first header lands under ''"""
    cases = {">chr1 desc\nACGT\nacgtN\n>chr2\tx\nAAAA\n": {"chr1": b"ACGTacgtN", "chr2\tx": b"AAAA"},
             "ACGT\n>c1\nAA\n>c1\nCC\n": {"": b"ACGT", "c1": b"CC"},
             ">c1\n\n>c2\nGG \n  TT\n": {"c2": b"GGTT"},
             ">c>1 a\nAC\r\nGT\r\n": {"c1": b"ACGT"},
             "": {}, ">only\n": {}, ">a\nAC\n>b\n>c\nGG": {"a": b"AC", "c": b"GG"}, "\n\n>x\nA\n\nC\n": {"x": b"AC"}}
    for k, (text, want) in enumerate(cases.items()):
        p = tmp_path / f"r{k}.fa"
        p.write_bytes(text.encode())
        assert host.load_reference_file(p) == want, text
