"""The streamed stage-5 path (nanosnp_amd/hap_pipeline.py): haplotype site file -> pinned staging -> copy stream -> haplotype features
+ HaplotypeModel forward -> haplotype.csv, the counterpart of HaplotypeModel/predict_dev.py:27-48 + dataset_dev.py:92-172,337-349.
The csv must not depend on the pass size, the on-disk dtype, the narrowing or the number of ranks, and must equal the rows made from
the per-site host restatement of the reference rows + the oracle chain."""
import os
import types
import re

import numpy as np
import pytest

from nanosnp_amd import host, sitefile
from tests.helpers import PROB_ATOL, seeded_hap_weights

pytestmark = pytest.mark.gpu
LABELS = ["AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT"]


def _make_bin(tmp_path, n, seed, D=90, dtype="int8", name="ctgA_1_9.bin", mapq255=False, contig="ctgA", ref_len=30_000):
    rng = np.random.default_rng(seed)
    pp = host.synth_hap_planes(seed, n, 30, D, 33); ph = host.synth_hap_planes(seed + 1, n, 30, D, 11)
    if mapq255:
        pp[2][pp[2] == 60] = 255                          # "mapping quality not available" of the SAM specification: beyond int8
    seq = rng.choice(list(b"ACGTacgtN"), ref_len, p=[.23, .23, .23, .23, .02, .02, .01, .01, .02]).astype(np.uint8)
    posn = np.sort(rng.choice(np.arange(1, ref_len + 40), n, replace=False))          # some windows hang over both ends
    posn[:min(2, n)] = (3, 9)[:min(2, n)]
    cands = [f"{contig}:{p}" for p in posn]
    hpos = [[f"{contig}:{max(1, p + 37 * (k - 5))}" for k in range(11)] for p in posn]
    planes = dict(zip(sitefile.HAP_PLANES, (ph[0], ph[3], ph[1], ph[2], pp[0], pp[3], pp[1], pp[2])))
    path = tmp_path / name
    sitefile.write_haplotype_bin(path, cands, hpos, planes, plane_dtype=dtype)
    return path, {contig: seq}, cands, hpos, pp, ph


@pytest.fixture(scope="module")
def hctx():
    from nanosnp_amd import _lib
    c = _lib.Context(0)
    c.hap_load_weights(seeded_hap_weights(12, H=256))
    yield c
    c.close()


def test_streamed_stage5_equals_the_per_site_restatement_and_the_oracle(tmp_path, hctx):
    from nanosnp_amd.hap_pipeline import DeviceReference, HapBinSource, predict_haplotype_bins, stream_haplotype
    from nanosnp_amd.predict import predict_haplotype
    from oracle import oracle
    n = 300
    path, refs, cands, hpos, pp, ph = _make_bin(tmp_path, n, 41)
    ref = DeviceReference(refs, 0)
    out = tmp_path / "haplotype.csv"
    st = {}
    assert predict_haplotype_bins(hctx, [path], ref, str(out), stats=st) == n
    assert st["passes"] == 1 and st["passes_int8"] == 1 and st["sites"] == n
    rows = out.read_text().splitlines()
    # the reference's call shape: predict(model, test_data, reference_path, ...) - a FASTA path (predict_dev.py:28 load_reference_file)
    fa = tmp_path / "ref.fa"
    fa.write_bytes(b"".join(b">" + k.encode() + b" some description\n" + b"\n".join(bytes(v[i:i + 70]) for i in range(0, len(v), 70)) + b"\n" for k, v in refs.items()))
    out_fa = tmp_path / "from_fasta.csv"
    assert predict_haplotype_bins(hctx, [path], str(fa), str(out_fa)) == n and out_fa.read_bytes() == out.read_bytes()
    # ... and predict_dev.py:27 argument for argument: predict(model, test_data, reference_path, batch_size, pileup_length, haplotype_length, output_file, device)
    from nanosnp_amd import predict as nsnp_predict
    bins = tmp_path / "bins_dir"; bins.mkdir()
    (bins / path.name).write_bytes(path.read_bytes())
    model = types.SimpleNamespace(ctx=hctx)
    out_pd = tmp_path / "predict_dev.csv"
    assert nsnp_predict.predict_dev(model, str(bins), str(fa), 1000, 33, 11, str(out_pd), "cuda:0") == n and out_pd.read_bytes() == out.read_bytes()
    with pytest.raises(ValueError):
        nsnp_predict.predict_dev(model, str(bins), str(fa), 1000, 33, 21, str(out_pd))
    # (a) the array entry with reference rows made by the per-site host restatement of dataset_dev.py:106-120,150-162
    rp = host.haplotype_ref_rows(refs, cands, 33)
    rh = host.haplotype_ref_rows(refs, cands, 11, position_lists=hpos)
    assert (rp == 0).any() and rp.max() == 4
    out2 = tmp_path / "arrays.csv"
    predict_haplotype(hctx, list(pp[:4]) + [rp], list(ph[:4]) + [rh], cands, str(out2))
    assert out2.read_bytes() == out.read_bytes()
    # (b) probabilities against the oracle chain on the same planes and rows
    src = HapBinSource(path)
    calls = stream_haplotype(hctx, src, ref, keep_probabilities=True, pass_sites=128)
    src.close()
    xp = oracle.hap_features_batch(*pp[:4], rp); xh = oracle.hap_features_batch(*ph[:4], rh)
    ogt, _ = oracle.hap_forward(seeded_hap_weights(12, H=256), xp, xh, nthreads=8)
    assert np.abs(calls.probabilities - ogt).max() < PROB_ATOL
    assert np.array_equal(calls.gt_arg, calls.probabilities.argmax(1)) and np.array_equal(calls.gt_max, calls.probabilities.max(1))
    for j, r in enumerate(rows):
        ctg, pos, gt, q = r.split("\t")
        assert f"{ctg}:{pos}" == cands[j] and gt == LABELS[int(calls.gt_arg[j])] and re.fullmatch(r"\d+\.\d+", q)
        want_q, ok = host.calculate_score(calls.gt_max[j])
        assert ok and float(q) == want_q


def test_streamed_stage5_is_independent_of_pass_size_dtype_and_narrowing(tmp_path, hctx):
    from nanosnp_amd.hap_pipeline import DeviceReference, predict_haplotype_bins
    n = 700
    p8, refs, cands, hpos, pp, ph = _make_bin(tmp_path, n, 43, dtype="int8", name="a8.bin")
    p32, *_ = _make_bin(tmp_path, n, 43, dtype="int32", name="a32.bin")
    assert sitefile.array_index(p8)["pileup_mapq"][0] == np.int8 and sitefile.array_index(p32)["pileup_mapq"][0] == np.int32
    ref = DeviceReference(refs, 0)
    want = None
    for path, kw in ((p8, {}), (p8, dict(pass_sites=256)), (p8, dict(pass_sites=100)), (p8, dict(pass_sites=7)),
                     (p32, {}), (p32, dict(pass_sites=300)), (p32, dict(narrow=False)), (p32, dict(narrow=False, pass_sites=129)),
                     (p8, dict(pass_sites=n)), (p8, dict(pass_sites=n - 1)), (p8, dict(pass_sites=n + 1))):
        out = tmp_path / "o.csv"
        st = {}
        assert predict_haplotype_bins(hctx, [path], ref, str(out), stats=st, **kw) == n
        got = out.read_bytes()
        want = want if want is not None else got
        assert got == want, (path, kw)
        ps = kw.get("pass_sites", 16384)
        want_passes = 1 if n <= ps else 1 + -(-(n - max(1, ps // 4)) // ps)           # (the first pass of a run is a quarter pass)
        assert st["passes"] == want_passes and st["passes_int8"] == (0 if kw.get("narrow") is False else st["passes"])
        per_site = (33 + 11) * 90 * (4 if kw.get("narrow") is False else 1) * 4
        assert st["bytes_staged"] == n * per_site
    assert want.count(b"\n") == n


def test_streamed_stage5_edge_inputs(tmp_path, hctx):
    from nanosnp_amd.hap_pipeline import DeviceReference, HapBinSource, predict_haplotype_bins, stream_haplotype
    p, refs, cands, hpos, pp, ph = _make_bin(tmp_path, 90, 47, dtype="int32", name="e.bin", mapq255=True)
    ref = DeviceReference(refs, 0)
    out = tmp_path / "e.csv"
    # a mapping quality of 255 does not fit int8: the narrowing call notices and the planes travel as int32; same rows as narrow=False
    st = {}
    assert predict_haplotype_bins(hctx, [p], ref, str(out), pass_sites=32, stats=st) == 90
    assert st.get("narrow_restarts") == 1 and st["passes_int8"] == 0
    out2 = tmp_path / "e2.csv"
    predict_haplotype_bins(hctx, [p], ref, str(out2), narrow=False)
    assert out.read_bytes() == out2.read_bytes()
    assert sitefile.array_index(_make_bin(tmp_path, 5, 47, dtype="int8", name="e8.bin", mapq255=True)[0])["pileup_mapq"][0] == np.int32
    # no sites, one site, a directory of bins in os.listdir order
    d = tmp_path / "bins"; d.mkdir()
    p0, refs0, *_ = _make_bin(d, 0, 1, name="z_0_0.bin")
    p1, refs1, c1, *_ = _make_bin(d, 1, 2, name="y_1_1.bin", contig="ctgB")
    p2, refs2, c2, *_ = _make_bin(d, 40, 3, name="x_1_9.bin", contig="ctgC")
    ref_all = DeviceReference({**refs1, **refs2}, 0)
    assert predict_haplotype_bins(hctx, str(d), ref_all, str(out)) == 41
    got = out.read_text().splitlines()
    order = [f for f in os.listdir(d)]
    want_first = {"y_1_1.bin": "ctgB", "x_1_9.bin": "ctgC"}[[f for f in order if f != "z_0_0.bin"][0]]
    assert got[0].startswith(want_first + "\t") and len(got) == 41
    src = HapBinSource(p0)
    calls = stream_haplotype(hctx, src, ref_all)
    assert calls.pos.size == 0 and calls.probabilities is None
    # a candidate on a contig the reference does not hold: reference rows 0 (the reference's bare except), the name in the csv
    px, _, cx, *_ = _make_bin(tmp_path, 6, 5, name="u.bin", contig="unplaced_7")
    assert predict_haplotype_bins(hctx, [px], ref_all, str(out)) == 6
    assert all(l.startswith("unplaced_7\t") for l in out.read_text().splitlines())
    srcx = HapBinSource(px)
    cz = stream_haplotype(hctx, srcx, ref_all, keep_probabilities=True)
    zeros = DeviceReference({"unplaced_7": b""}, 0)
    cz2 = stream_haplotype(hctx, srcx, zeros, keep_probabilities=True)
    assert np.array_equal(cz.probabilities, cz2.probabilities)
    # a position field that is not "ctg:pos": the reference raises (dataset_dev.py:109-110); so does this
    planes = sitefile.read_arrays(px, mmap=False)
    bad = [f"unplaced_7:{i}" for i in range(5)] + ["unplaced_7"]
    pb = tmp_path / "bad.bin"
    sitefile.write_haplotype_bin(pb, [f"unplaced_7:{i}" for i in range(6)], [[f"unplaced_7:{i}"] * 10 + [bad[i]] for i in range(6)],
                                 {k: planes[k] for k in sitefile.HAP_PLANES})
    with pytest.raises(host.HostError):
        predict_haplotype_bins(hctx, [pb], ref_all, str(out))
    assert predict_haplotype_bins(hctx, [px], ref_all, str(out)) == 6                    # the context is usable afterwards
    with pytest.raises(ValueError):
        stream_haplotype(hctx, srcx, None)                                              # a bin carries no reference rows


def test_a_directory_of_bins_is_one_pipeline(tmp_path, hctx):
    """several files through predict_haplotype_bins = the concatenation of their single-file csvs (files of different depth, an empty
    one in the middle, one that needs the int32 restart at the end: rows already written are not repeated)"""
    from nanosnp_amd.hap_pipeline import DeviceReference, predict_haplotype_bins
    specs = [("a.bin", 150, 61, dict(D=90)), ("b.bin", 0, 62, dict(D=90)), ("c.bin", 77, 63, dict(D=40)), ("d.bin", 1, 64, dict(D=90)),
             ("e.bin", 90, 65, dict(D=90, dtype="int32", mapq255=True))]
    paths, refs = [], {}
    for name, n, seed, kw in specs:
        p, r, *_ = _make_bin(tmp_path, n, seed, name=name, contig="ctg" + name[0].upper(), **kw)
        paths.append(p); refs.update(r)
    ref = DeviceReference(refs, 0)
    singles = b""
    for p in paths:
        o = tmp_path / "one.csv"
        predict_haplotype_bins(hctx, [p], ref, str(o), pass_sites=64)
        singles += o.read_bytes()
    for ps in (64, 16384, 50):
        o = tmp_path / "all.csv"
        st = {}
        assert predict_haplotype_bins(hctx, paths, ref, str(o), pass_sites=ps, stats=st) == 318
        assert o.read_bytes() == singles, ps
        assert st.get("narrow_restarts") == 1 and st["sites"] >= 318


def _rank_worker(rank, world, port, tmp, q):
    import torch.distributed as dist
    from nanosnp_amd import _lib
    from nanosnp_amd.hap_pipeline import predict_haplotype_bins
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = _lib.Context(0)
        c.hap_load_weights(seeded_hap_weights(12, H=256))
        z = np.load(os.path.join(tmp, "ref.npz"))
        out = os.path.join(tmp, f"sharded_{rank}.csv")
        rows = predict_haplotype_bins(c, [os.path.join(tmp, "a.bin"), os.path.join(tmp, "b.bin")], {k: z[k] for k in z.files}, out, pass_sites=64)
        q.put((rank, rows, os.path.exists(out)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_site_sharded_stage5_writes_the_single_process_csv(tmp_path, hctx, world):
    """every rank streams its shard_range of every bin (all on cuda:0 of the one-GPU box, the gather over gloo), rank 0 writes: the csv
    of the single-process run; the second bin is smaller than the number of ranks x 2 (an almost empty shard)"""
    import socket
    import torch.multiprocessing as mp
    from nanosnp_amd.hap_pipeline import predict_haplotype_bins
    pa, refa, *_ = _make_bin(tmp_path, 333, 51, name="a.bin", contig="ctgA")
    pb, refb, *_ = _make_bin(tmp_path, 4, 52, name="b.bin", contig="ctgB")
    np.savez(tmp_path / "ref.npz", **refa, **refb)
    want = tmp_path / "single.csv"
    assert predict_haplotype_bins(hctx, [pa, pb], {**refa, **refb}, str(want), pass_sites=64) == 337
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == (0, 337, True) and all(r[1] == 0 and not r[2] for r in res[1:])
    assert (tmp_path / "sharded_0.csv").read_bytes() == want.read_bytes()


def test_the_reference_written_stage5_rows_through_the_file_path(tmp_path):
    """tests/golden/two_stage.npz (planes, group positions and the haplotype.csv the reference's predict_dev.py wrote) through
    write_haplotype_bin -> predict_haplotype_bins with the reference rows gathered on the device"""
    from tools.hap_e2e_bench import two_stage_fixture_check
    r = two_stage_fixture_check(0, str(tmp_path))
    assert r["ok"] and r["rows"] == 19, r
