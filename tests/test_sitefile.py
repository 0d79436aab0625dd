"""Flat binary site containers (nanosnp_amd/sitefile.py) against the arrays the reference's HDF5 bins hold."""
import gzip
import os

import numpy as np
import pytest

from nanosnp_amd import host, sitefile
from tests.helpers import golden


def test_array_container_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    arrays = {"a": rng.integers(-5, 5, (7, 33, 18)).astype(np.int32), "b": rng.random((3, 2, 2, 2)).astype(np.float32),
              "bytes": np.frombuffer(b"hello world", np.uint8), "empty": np.empty((0, 33, 18), np.int32),
              "i64": np.arange(5, dtype=np.int64), "scalar_like": np.array([3], np.uint8)}
    p = tmp_path / "x.bin"
    sitefile.write_arrays(p, arrays)
    for mm in (True, False):
        back = sitefile.read_arrays(p, mmap=mm)
        assert list(back) == list(arrays)
        for k, v in arrays.items():
            assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v)
            if mm and v.size:
                assert back[k].offset % 64 == 0            # device copies start on a 64-byte boundary
    with pytest.raises(sitefile.SiteFileError):
        sitefile.write_arrays(tmp_path / "y.bin", {"five_d": np.zeros((1, 1, 1, 1, 1))})
    (tmp_path / "junk.bin").write_bytes(b"not a site file")
    with pytest.raises(sitefile.SiteFileError):
        sitefile.read_arrays(tmp_path / "junk.bin")


@pytest.mark.parametrize("name", ["encode_g1", "encode_adv", "encode_end"])
def test_pd_to_bin_holds_what_the_reference_bin_holds(tmp_path, name):
    """make_bin_predict_data.py:48-77 (text -> arrays) followed by PileupModel/dataset.py:118-139 (arrays -> fields),
    restated below, against the native .pd parser of the device pipeline"""
    pd = gzip.open(golden(f"{name}.pd.gz")).read()
    p = tmp_path / f"{name}.pd.bin"
    n = sitefile.pd_to_bin(pd, p)
    names, pos, refb, x = sitefile.read_pileup_bin(p)
    x2, names2, pos2, refb2 = host.pd_parse(pd)
    assert n == len(names) == x2.shape[0] and n > 50
    assert np.array_equal(x, x2) and names == names2 and np.array_equal(pos, pos2) and np.array_equal(refb, refb2)
    # the reference's own transformation of each line
    lines = [l for l in pd.decode().split("\n") if l.strip()]
    alts = sitefile.read_alt_info(p)
    for i in (0, 1, n // 2, n - 1):
        tensor_s, position_s, alt_s = lines[i].split("\t")
        want = np.array([int(v) for v in tensor_s.split()], dtype="int32").reshape(33, 18)
        assert np.array_equal(x[i], want)
        ctg, q, seq = position_s.strip().split(":")
        assert (names[i], pos[i], refb[i]) == (ctg, int(q), ord(seq[16]))
        assert alts[i] == alt_s.strip()


def test_pileup_bin_counts_are_int16_on_disk_when_they_fit(tmp_path):
    """write_pileup_bin's default keeps the counts as int16 (half the file and half the bytes the streamed predict loop stages) unless a
    value does not fit, or the caller asks for the reference's Int32Atom; read_pileup_bin hands out int32 either way"""
    rng = np.random.default_rng(5)
    x = rng.integers(-144, 145, (37, 33, 18)).astype(np.int32)
    position = [f"chr1:{i + 1}:{'ACGT' * 8}A" for i in range(37)]
    sizes = {}
    for md, want in (("int16", np.int16), ("int32", np.int32)):
        p = tmp_path / f"{md}.bin"
        sitefile.write_pileup_bin(p, x, position, matrix_dtype=md)
        assert sitefile.array_index(p)["position_matrix"][0] == want
        got = sitefile.read_pileup_bin(p)[3]
        assert got.dtype == np.int32 and np.array_equal(got, x)
        sizes[md] = os.path.getsize(p)
    assert abs(sizes["int32"] - sizes["int16"] - x.size * 2) < 64                        # arrays start on 64-byte boundaries
    for edge, want in ((32767, np.int16), (-32768, np.int16), (32768, np.int32), (-32769, np.int32)):
        y = x.copy(); y[5, 16, 3] = edge
        sitefile.write_pileup_bin(tmp_path / "e.bin", y, position)
        assert sitefile.array_index(tmp_path / "e.bin")["position_matrix"][0] == want
        assert np.array_equal(sitefile.read_pileup_bin(tmp_path / "e.bin")[3], y)
    sitefile.write_pileup_bin(tmp_path / "s.bin", x.astype(np.int16), position)            # int16 in, int16 out
    assert np.array_equal(sitefile.read_pileup_bin(tmp_path / "s.bin")[3], x)
    with pytest.raises(sitefile.SiteFileError):
        sitefile.write_pileup_bin(tmp_path / "f.bin", x.astype(np.float32), position)
    with pytest.raises(sitefile.SiteFileError):
        sitefile.write_pileup_bin(tmp_path / "f.bin", x, position, matrix_dtype="int8")


def test_pileup_bin_rejects_malformed_input(tmp_path):
    with pytest.raises(sitefile.SiteFileError):
        sitefile.pd_to_bin(b"1 2 3\tchr1:5:" + b"A" * 33 + b"\t10-A 3 \n", tmp_path / "a.bin")          # not 594 values
    with pytest.raises(sitefile.SiteFileError):
        sitefile.write_pileup_bin(tmp_path / "b.bin", np.zeros((2, 33, 18), np.int32), ["chr1:1:" + "A" * 33])
    sitefile.write_pileup_bin(tmp_path / "c.bin", np.zeros((1, 33, 18), np.int32), ["chr1:12:ACG"])       # ref33 too short
    with pytest.raises(sitefile.SiteFileError):
        sitefile.read_pileup_bin(tmp_path / "c.bin")
    assert sitefile.pd_to_bin(b"", tmp_path / "e.bin") == 0
    names, pos, refb, x = sitefile.read_pileup_bin(tmp_path / "e.bin")
    assert names == [] and x.shape == (0, 33, 18)


def test_haplotype_bin_sorts_by_position_and_cuts_depth(tmp_path):
    """write_to_bins.py:5-8 (argsort by integer position), :39-42,54-61 (depth clamp), padding value -2 kept"""
    rng = np.random.default_rng(3)
    n, dh, dp = 9, 12, 20
    posn = rng.permutation(np.arange(1000, 1000 + 50 * n, 50))
    cand = [f"ctgA:{p}" for p in posn]
    hpos = [[f"ctgA:{p + k}" for k in range(11)] for p in posn]
    planes = {}
    for name in sitefile.HAP_PLANES:
        d, l = (dh, 11) if name.startswith("haplotype") else (dp, 33)
        a = rng.integers(-1, 5, (n, d, l)).astype(np.int32)
        a[:, d - 3:] = -2
        planes[name] = a
    p = tmp_path / "ctgA_1000_1400.bin"
    order = np.argsort(posn, kind="stable")
    big = {k: v.copy() for k, v in planes.items()}
    big["pileup_mapq"][2, 0, 0] = 255                                       # one value beyond int8: the whole file stays int32
    for kw, src, want_dtype in (({}, planes, np.int8), ({"plane_dtype": "int32"}, planes, np.int32), ({}, big, np.int32)):
        sitefile.write_haplotype_bin(p, cand, hpos, src, max_haplotype_depth=10, max_pileup_depth=64, **kw)
        c2, h2, pl2 = sitefile.read_haplotype_bin(p)
        assert c2 == [cand[i] for i in order] and h2 == [hpos[i] for i in order]
        for name in sitefile.HAP_PLANES:
            want = src[name][order]
            if name.startswith("haplotype"):
                want = want[:, :10]
            assert pl2[name].dtype == want_dtype and np.array_equal(pl2[name], want)
        idx = sitefile.array_index(p)
        assert idx["pileup_mapq"][:2] == (np.dtype(want_dtype), (n, dp, 33)) and idx["pileup_mapq"][2] % 64 == 0
    (tmp_path / "nope.bin").write_bytes(b"NSNPBIN1" + bytes(8))            # a valid container without the arrays
    with pytest.raises(sitefile.SiteFileError):
        sitefile.read_haplotype_bin(tmp_path / "nope.bin")


@pytest.mark.gpu
def test_bin_file_feeds_the_predict_loop(tmp_path, pileup_weights):
    """a memory-mapped site file gives the same pileup.vcf as the arrays it was written from"""
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.predict import predict_pileup
    z = np.load(golden("pileup_vcf.npz"))
    names = list(z["names"])
    position = [f"{c}:{int(p)}:{'N' * 16}{chr(int(r))}{'N' * 16}" for c, p, r in zip(names, z["pos"], z["refb"])]
    p = tmp_path / "sites.pd.bin"
    sitefile.write_pileup_bin(p, z["x"], position)
    n2, p2, r2, x2 = sitefile.read_pileup_bin(p)
    m = LSTMNetwork().load_weight_list(pileup_weights)
    fai = bytes(z["fai"]).decode()
    a = predict_pileup(m, z["x"].astype(np.int32), names, z["pos"], z["refb"], fai, str(tmp_path / "a.vcf"), batch_size=1000)
    b = predict_pileup(m, np.asarray(x2), n2, p2, r2, fai, str(tmp_path / "b.vcf"), batch_size=1000)
    assert a == b and (tmp_path / "a.vcf").read_bytes() == (tmp_path / "b.vcf").read_bytes()
