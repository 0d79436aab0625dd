"""End-to-end predict loops on the GPU vs what the reference's predict() wrote."""
import os
import re

import numpy as np
import pytest

from nanosnp_amd import host
from tests.helpers import golden

pytestmark = pytest.mark.gpu


def test_pileup_vcf_end_to_end(tmp_path, pileup_weights):
    """position_matrix -> pileup.vcf.  Probabilities differ from CPU torch by ~1e-7, which can move a
    QUAL by one unit in its second decimal; everything else must be identical."""
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.predict import predict_pileup
    z = np.load(golden("pileup_vcf_modes.npz"))          # the wrappers default to NumPy 1.x scalar promotion (score_mode 1)
    m = LSTMNetwork().load_weight_list(pileup_weights)
    out = tmp_path / "pileup.vcf"
    rows = predict_pileup(m, z["x"].astype(np.int32), list(z["names"]), z["pos"], z["refb"],
                          bytes(z["fai"]).decode(), str(out), batch_size=1000)
    got = out.read_bytes().decode().splitlines()
    want = bytes(z["vcf_np1_bs1000"]).decode().splitlines()
    assert len(got) == len(want) and rows == sum(1 for l in want if not l.startswith("#"))
    # every row that differs is checked on its own (no budget of tolerated rows): only QUAL / GQ may differ, the GPU's probabilities
    # of that site are within 1e-6 of the reference's (golden gt / zy), and the reference's QUAL is the QUAL of a probability at most
    # 1e-6 away from the GPU's - the two sit on either side of a rounding boundary of the two-decimal score
    import torch
    from tests.helpers import qual_reachable
    site = {(str(n), int(p)): j for j, (n, p) in enumerate(zip(z["names"], z["pos"]))}
    gt_gpu, zy_gpu = m.predict(torch.from_numpy(z["x"].astype(np.int32)).cuda())
    gt_gpu, zy_gpu = gt_gpu.cpu().numpy(), zy_gpu.cpu().numpy()
    assert np.abs(gt_gpu - z["gt"]).max() < 1e-6 and np.abs(zy_gpu - z["zy"]).max() < 1e-6
    n_qual_diff = 0
    for g, w in zip(got, want):
        if g == w:
            continue
        gf, wf = g.split("\t"), w.split("\t")
        assert gf[:5] == wf[:5] and gf[6:9] == wf[6:9], (g, w)
        gs, ws = gf[9].split(":"), wf[9].split(":")
        assert gs[0] == ws[0] and gs[2:] == ws[2:] and int(gs[1]) == int(float(gf[5])) and int(ws[1]) == int(float(wf[5])), (g, w)
        j = site[(gf[0], int(gf[1]))]
        assert qual_reachable(float(wf[5]), zy_gpu[j].max(), gt_gpu[j].max(), refcall=gf[6] == "RefCall"), (g, w, zy_gpu[j].max(), gt_gpu[j].max())
        n_qual_diff += 1
    print("rows whose QUAL differs by a rounding boundary:", n_qual_diff, "of", len(want))


def test_haplotype_csv_end_to_end(tmp_path, gpu_ctx):
    from nanosnp_amd.predict import predict_haplotype
    from oracle import oracle
    from tests.helpers import seeded_hap_weights
    ws = seeded_hap_weights(12, H=256)
    gpu_ctx.hap_load_weights(ws)
    n = 70
    pp = host.synth_hap_planes(31, n, 30, 90, 33)
    ph = host.synth_hap_planes(32, n, 30, 90, 11)
    cands = [f"chr{1 + i % 2}:{1000 + 7 * i}" for i in range(n)]
    out = tmp_path / "haplotype.csv"
    predict_haplotype(gpu_ctx, pp, ph, cands, str(out), batch_size=32)
    rows = out.read_text().splitlines()
    assert len(rows) == n
    ogt, _ = oracle.hap_forward(ws, oracle.hap_features_batch(*pp), oracle.hap_features_batch(*ph), nthreads=8)
    labels = ["AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT"]
    for j, r in enumerate(rows):
        ctg, pos, gt, q = r.split("\t")
        assert f"{ctg}:{pos}" == cands[j] and re.fullmatch(r"\d+\.\d+", q)
        top2 = np.sort(ogt[j])[-2:]
        if top2[1] - top2[0] > 1e-3:                      # unambiguous argmax
            assert gt == labels[int(ogt[j].argmax())]
        want_q, ok = host.calculate_score(ogt[j].max())
        assert ok and abs(float(q) - want_q) <= 0.0101
    # the same planes handed over as int8: identical csv
    out8 = tmp_path / "haplotype8.csv"
    predict_haplotype(gpu_ctx, [a.astype(np.int8) for a in pp[:4]] + [pp[4]], [a.astype(np.int8) for a in ph[:4]] + [ph[4]],
                      cands, str(out8), batch_size=32)
    assert out8.read_bytes() == out.read_bytes()


def test_mpileup_to_vcf_pipeline(tmp_path, pileup_weights, tok_mode):
    """s1+s2 in one pass: mpileup text + FASTA -> VCF, against the rows the reference's predict() wrote
    for the same contig's sites (golden pileup_vcf.npz holds the chrS sites first, batch 1000)."""
    import gzip
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_variants
    z = np.load(golden("pileup_vcf.npz"))
    m = LSTMNetwork().load_weight_list(pileup_weights)
    fa = tmp_path / "ref.fa"
    fa.write_bytes(gzip.open(golden("encode_g1.fa.gz")).read())
    mp = tmp_path / "chrS.mpileup"
    mp.write_bytes(gzip.open(golden("encode_g1.mpileup.gz")).read())
    out = tmp_path / "pileup.vcf"
    rows = call_variants(m, [("chrS", str(mp))], str(fa), "chrS\t6100\t6\t60\t61\n", str(out))
    got = [l for l in out.read_text().splitlines() if not l.startswith("#")]
    ref_rows = [l for l in bytes(z["vcf_bs1000"]).decode().splitlines() if l.startswith("chrS\t")]
    assert rows == len(got) == len(ref_rows)
    # The golden batch of 1000 also held chrT sites, and a fallback row takes its ALT from the other sites of its batch
    # (predict.py:102-109), so the reference's text for a chrS-only batch is rebuilt from the reference's OWN probabilities of the chrS
    # sites by the row formatter (byte-identical to predict() on 72 runs: tests/test_vcf.py).  Against that text every row must be
    # equal, except a QUAL / GQ that a probability at most 1e-6 away from the GPU's explains - no budget of tolerated rows.
    import torch
    from nanosnp_amd.predict import COV_CHANNELS
    from tests.helpers import qual_reachable
    sel = np.flatnonzero(np.asarray(z["names"]) == "chrS")
    x = z["x"][sel].astype(np.int32)
    gt_ref, zy_ref = z["gt"][sel], z["zy"][sel]
    table = host.ContigTable(["chrS"])
    want_text, want_rows = host.vcf_format_batches(table, np.zeros(sel.size, np.int32), z["pos"][sel], z["refb"][sel],
                                                   gt_ref.argmax(1).astype(np.uint8), zy_ref.argmax(1).astype(np.uint8), gt_ref.max(1), zy_ref.max(1),
                                                   x[:, 16, COV_CHANNELS].astype(np.float32), batch_size=1000, score_mode=host.SCORE_FLOAT64)
    want = want_text.decode().splitlines()
    assert want_rows == len(want) == len(got)
    # (the rebuilt text equals the reference's own rows wherever no other batch member is involved)
    assert sum(a != b for a, b in zip(want, ref_rows)) <= 3 and all(a.split("\t")[:4] == b.split("\t")[:4] for a, b in zip(want, ref_rows))
    gt_gpu, zy_gpu = m.predict(torch.from_numpy(x).cuda())
    gt_gpu, zy_gpu = gt_gpu.cpu().numpy(), zy_gpu.cpu().numpy()
    assert np.abs(gt_gpu - gt_ref).max() < 1e-6 and np.abs(zy_gpu - zy_ref).max() < 1e-6
    site = {int(p): j for j, p in enumerate(z["pos"][sel])}
    for g, w in zip(got, want):
        if g == w:
            continue
        gf, wf = g.split("\t"), w.split("\t")
        assert gf[:5] == wf[:5] and gf[6:9] == wf[6:9], (g, w)
        gs, ws = gf[9].split(":"), wf[9].split(":")
        assert gs[0] == ws[0] and gs[2:] == ws[2:], (g, w)
        j = site[int(gf[1])]
        assert qual_reachable(float(wf[5]), zy_gpu[j].max(), gt_gpu[j].max(), refcall=gf[6] == "RefCall", score_mode=host.SCORE_FLOAT64), (g, w)


def test_streamed_pipeline_is_independent_of_the_chunk_size(tmp_path, pileup_weights, tok_mode):
    """pipeline.call_contig works the text off in chunks of whole lines (parse of chunk k + 1 on the host beside the device work of
    chunk k, 16 lines of halo re-parsed): the VCF is byte-identical whatever the chunk size - one chunk, a few, or chunks shorter
    than a window - on a contig with position gaps and lower-case / N reference bases (encode_g1) and on a 40 k-column G1 contig"""
    import gzip
    from nanosnp_amd import host
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_contig
    m = LSTMNetwork().load_weight_list(pileup_weights)
    text = gzip.open(golden("encode_g1.mpileup.gz")).read()
    fa = tmp_path / "ref.fa"
    fa.write_bytes(gzip.open(golden("encode_g1.fa.gz")).read())
    seq = host.fasta_load_contig(str(fa), "chrS")
    ref_rows, n_sites, n_rows = call_contig(m, text, "chrS", seq, chunk_bytes=1 << 30)
    assert n_sites > 50 and n_rows > 0
    for cb in (100_000, 20_000, 3_000, 700):
        st = {}
        rows, ns, nr = call_contig(m, text, "chrS", seq, chunk_bytes=cb, stats=st)
        assert (ns, nr) == (n_sites, n_rows) and rows == ref_rows, cb
        assert st["chunks"] >= len(text) // cb and st["columns"] == text.count(b"\n") and st["sites"] == n_sites
    cols = host.synth_columns(20261111, 40_000, coverage=30, het_rate=0.03)
    big = cols.mpileup_text_native("chrB")
    seqb = cols.ref.copy()
    want = call_contig(m, big, "chrB", seqb, chunk_bytes=1 << 30)
    got = call_contig(m, big, "chrB", seqb, chunk_bytes=256 << 10)
    assert got == want and want[1] > 500
    # the opt-in form that formats the complete batches of finished chunks on a writer thread (measured slower, kept off): same bytes
    for cb in (256 << 10, 90_000, 1 << 30):
        for bs in (1000, 64, 7):
            a = call_contig(m, big, "chrB", seqb, chunk_bytes=cb, batch_size=bs, rows_beside=True)
            b = call_contig(m, big, "chrB", seqb, chunk_bytes=1 << 30, batch_size=bs)
            assert bytes(a[0]) == bytes(b[0]) and a[1:] == b[1:], (cb, bs)
    assert call_contig(m, b"", "chrB", seqb, rows_beside=True) == (b"", 0, 0)


def test_rows_unpack_writes_device_and_pinned_outputs(gpu_ctx):
    """nsnp_pileup_rows_unpack: the float64 call rows of a streamed text run -> the formatter's typed arrays, written by the kernel into
    device tensors or straight into pinned host memory (no copy); 0 rows, 1 row, a ragged count; wrong shapes / pageable outputs refused"""
    import torch
    from nanosnp_amd import _lib
    rng = np.random.default_rng(8)
    for n in (0, 1, 70001):
        rows = np.concatenate([rng.integers(1, 2 ** 40, (n, 1)).astype(np.float64), rng.integers(0, 21, (n, 1)).astype(np.float64),
                               rng.integers(0, 3, (n, 1)).astype(np.float64), rng.random((n, 2)).astype(np.float32).astype(np.float64),
                               rng.integers(-3000, 3000, (n, 8)).astype(np.float64)], axis=1)
        r = torch.from_numpy(rows).cuda()
        for pinned in (False, True):
            kw = dict(pin_memory=True) if pinned else dict(device="cuda")
            outs = (torch.zeros(n + 3, dtype=torch.int64, **kw), torch.zeros(n + 3, dtype=torch.uint8, **kw), torch.zeros(n + 3, dtype=torch.uint8, **kw),
                    torch.zeros(n + 3, dtype=torch.float32, **kw), torch.zeros(n + 3, dtype=torch.float32, **kw), torch.zeros((n + 3, 8), dtype=torch.float32, **kw))
            gpu_ctx.pileup_rows_unpack(r, outs)
            torch.cuda.synchronize()
            got = [t.cpu().numpy() for t in outs]
            assert np.array_equal(got[0][:n], rows[:, 0].astype(np.int64)) and np.array_equal(got[1][:n], rows[:, 1].astype(np.uint8))
            assert np.array_equal(got[2][:n], rows[:, 2].astype(np.uint8)) and np.array_equal(got[3][:n], rows[:, 3].astype(np.float32))
            assert np.array_equal(got[4][:n], rows[:, 4].astype(np.float32)) and np.array_equal(got[5][:n], rows[:, 5:].astype(np.float32))
            assert all(not g[n:].any() for g in got)                               # nothing written behind the last row
    r = torch.zeros((4, 13), dtype=torch.float64, device="cuda")
    good = (torch.zeros(4, dtype=torch.int64, device="cuda"), torch.zeros(4, dtype=torch.uint8, device="cuda"), torch.zeros(4, dtype=torch.uint8, device="cuda"),
            torch.zeros(4, dtype=torch.float32, device="cuda"), torch.zeros(4, dtype=torch.float32, device="cuda"), torch.zeros((4, 8), dtype=torch.float32, device="cuda"))
    with pytest.raises(_lib.NanoSNPError):
        gpu_ctx.pileup_rows_unpack(r[:, :12].contiguous(), good)
    with pytest.raises(_lib.NanoSNPError):
        gpu_ctx.pileup_rows_unpack(r, good[:5] + (torch.zeros((4, 8), dtype=torch.float32),))          # pageable host memory
    with pytest.raises(_lib.NanoSNPError):
        gpu_ctx.pileup_rows_unpack(r, (good[0][:3],) + good[1:])


def test_a_run_over_several_contigs_writes_each_contigs_rows_in_order(tmp_path, pileup_weights, tok_mode):
    """pipeline.call_variants / call_contigs: the rows of contig c are formatted and written on a writer thread (own stream) while contig
    c + 1 streams - the file is the header + the rows call_contig gives for every contig on its own, in order; an empty contig and a
    contig without a site among them; per batch size"""
    from nanosnp_amd import host
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_contig, call_contigs, call_variants
    m = LSTMNetwork().load_weight_list(pileup_weights)
    contigs, fasta, fai = [], b"", ""
    for i, (n, cov) in enumerate(((4000, 30), (0, 30), (2500, 60), (40, 30), (6000, 12))):
        name = f"ctg{i}"
        cols = host.synth_columns(20261300 + i, max(n, 1), coverage=cov, het_rate=0.05)
        text = bytes(cols.mpileup_text_native(name)) if n else b""
        seq = cols.ref.copy()
        (tmp_path / f"{name}.mpileup").write_bytes(text)
        fasta += b">" + name.encode() + b"\n" + b"\n".join(bytes(seq[a:a + 60]) for a in range(0, seq.size, 60)) + b"\n"
        fai += f"{name}\t{seq.size}\t0\t60\t61\n"
        contigs.append((name, text, seq))
    (tmp_path / "ref.fa").write_bytes(fasta)
    for bs in (1000, 64):
        want, want_rows, want_sites = host.vcf_header(fai).encode(), 0, 0
        for name, text, seq in contigs:
            t, ns, nr = call_contig(m, text, name, seq, batch_size=bs, chunk_bytes=100_000)
            want += bytes(t); want_rows += nr; want_sites += ns
        assert want_rows > 300
        out = tmp_path / f"run{bs}.vcf"
        rows = call_variants(m, [(name, str(tmp_path / f"{name}.mpileup")) for name, _, _ in contigs], str(tmp_path / "ref.fa"), fai, str(out),
                             batch_size=bs, chunk_bytes=100_000)
        assert rows == want_rows and out.read_bytes() == want
        st = {}
        with open(tmp_path / "again.vcf", "wb") as f:
            assert call_contigs(m, contigs, f, batch_size=bs, chunk_bytes=1 << 30, stats=st) == (want_sites, want_rows)
        assert host.vcf_header(fai).encode() + (tmp_path / "again.vcf").read_bytes() == want and st["vcf_rows"] == want_rows
    with open(tmp_path / "none.vcf", "wb") as f:
        assert call_contigs(m, [], f) == (0, 0)


def test_streamed_pipeline_edge_inputs(pileup_weights, tok_mode):
    """call_contig on the inputs a real run meets at its edges: no text, one line, fewer columns than a window, a last line without
    its newline, CRLF line ends, a chunk size below one line (every line its own chunk), and the same contig again on the same model
    after a larger one (the pinned / device buffer sets kept on the model are re-used): always the rows of the one-chunk run"""
    from nanosnp_amd import host
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_contig
    m = LSTMNetwork().load_weight_list(pileup_weights)
    cols = host.synth_columns(20261212, 3000, coverage=30, het_rate=0.05)
    text = bytes(cols.mpileup_text_native("chrE"))
    seq = cols.ref.copy()
    lines = text.split(b"\n")[:-1]
    want = call_contig(m, text, "chrE", seq, chunk_bytes=1 << 30)
    assert want[1] > 30 and want[2] > 0
    want_rows = bytes(want[0])
    assert call_contig(m, b"", "chrE", seq) == (b"", 0, 0)
    for k in (1, 16, 32):                                           # fewer columns than a 33-column window: nothing to call
        r = call_contig(m, b"\n".join(lines[:k]) + b"\n", "chrE", seq)
        assert r[1] == 0 and bytes(r[0]) == b""
    r = call_contig(m, text[:-1], "chrE", seq, chunk_bytes=50_000)  # the last line without its newline
    assert bytes(r[0]) == want_rows and r[1:] == want[1:]
    r = call_contig(m, b"\r\n".join(lines) + b"\r\n", "chrE", seq, chunk_bytes=40_000)
    assert bytes(r[0]) == want_rows and r[1:] == want[1:]
    r = call_contig(m, text, "chrE", seq, chunk_bytes=64)           # below one line: one line per chunk, ~3000 chunks
    assert bytes(r[0]) == want_rows and r[1:] == want[1:]
    big_cols = host.synth_columns(20261213, 30_000, coverage=30, het_rate=0.03)
    big = call_contig(m, big_cols.mpileup_text_native("chrF"), "chrF", big_cols.ref.copy(), chunk_bytes=200_000)
    assert big[1] > 300
    r = call_contig(m, text, "chrE", seq, chunk_bytes=70_000)       # smaller again: the larger buffer sets are kept and re-used
    assert bytes(r[0]) == want_rows and r[1:] == want[1:]
    with pytest.raises(ValueError):
        call_contig(m, text, "chrE", seq[:100])                     # positions beyond the reference sequence


def test_streamed_pipeline_refuses_blank_lines_and_grows_its_column_buffers(pileup_weights, tok_mode):
    """(a) the pipeline's halo bookkeeping takes one line = one column: an empty line (which the tolerant parser steps over) near a
    chunk cut would shift the chunk's own range - the text is refused (the reference aborts on such a line); (b) the pinned / device
    column buffers are budgeted at 24 text bytes per line and grow when a chunk holds more, shorter lines: a shallow contig of 19-byte
    lines gives the rows of its one-chunk run at every chunk size"""
    from nanosnp_amd import host
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_contig
    m = LSTMNetwork().load_weight_list(pileup_weights)
    cols = host.synth_columns(20261313, 3000, coverage=30, het_rate=0.05)
    text = bytes(cols.mpileup_text_native("chrE"))
    seq = cols.ref.copy()
    lines = text.split(b"\n")[:-1]
    want = call_contig(m, text, "chrE", seq, chunk_bytes=1 << 30)
    cut = len(b"\n".join(lines[:1500])) + 1
    for at in (1490, 1500, 1510, 5, 2999):
        bad = b"\n".join(lines[:at]) + b"\n\n" + b"\n".join(lines[at:]) + b"\n"
        for cb in (1 << 30, cut):
            with pytest.raises(host.HostError, match="empty line"):
                call_contig(m, bad, "chrE", seq, chunk_bytes=cb)
    r = call_contig(m, text, "chrE", seq, chunk_bytes=cut)          # the model is usable afterwards
    assert bytes(r[0]) == bytes(want[0]) and r[1:] == want[1:]
    # (b) 60,000 lines of 19 bytes, no quality field (only columns 0, 1 and 4 are read: main.cpp:162-172)
    n = 60_000
    shallow = b"".join(b"c\t%d\tN\t6\t%s\n" % (i + 1, b"AAACCC" if i % 37 == 0 else (b"AAAAAg" if i % 11 == 0 else b"AaAaAa")) for i in range(n))
    assert len(shallow) < 24 * n
    seq2 = np.full(n, ord("A"), np.uint8)
    m2 = LSTMNetwork().load_weight_list(pileup_weights)             # fresh buffer sets
    st = {}
    one = call_contig(m2, shallow, "c", seq2, chunk_bytes=1 << 30, stats=st)
    assert one[1] > 1000 and st["columns"] == n
    for cb in (300_000, 50_000):
        r = call_contig(m2, shallow, "c", seq2, chunk_bytes=cb)
        assert bytes(r[0]) == bytes(one[0]) and r[1:] == one[1:], cb


def test_streamed_pd_bin_files_write_the_array_path_vcf(tmp_path, pileup_weights):
    """pipeline.predict_pileup_bins (PileupModel/predict.py:37-195 over .pd.bin files, streamed: pread -> pinned -> copy stream -> forward
    -> calls -> rows) against predict.predict_pileup on the same arrays (itself held to the reference's VCF above): byte-identical for
    every pass size, with the counts narrowed to int16 or sent as int32, for several files in one run (the reference's batches restart
    with every file), an empty file among them, a count beyond int16, and a directory argument (its *.bin files in os.listdir order)"""
    import os
    from nanosnp_amd import sitefile
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import predict_pileup_bins
    from nanosnp_amd.predict import predict_pileup
    z = np.load(golden("pileup_vcf_modes.npz"))
    x = z["x"].astype(np.int32)
    names, pos, refb = list(z["names"]), z["pos"], z["refb"]
    fai = bytes(z["fai"]).decode()
    position = [f"{c}:{int(p)}:{'N' * 16}{chr(int(r))}{'N' * 16}" for c, p, r in zip(names, pos, refb)]
    n = x.shape[0]
    m = LSTMNetwork().load_weight_list(pileup_weights)
    cuts = [0, n // 3, n // 3, n - 5, n]                               # three files + an empty one
    d = tmp_path / "bins"; d.mkdir()
    files, files32, want = [], [], b""
    (tmp_path / "bins32").mkdir()
    for i in range(4):
        a, b = cuts[i], cuts[i + 1]
        p = d / f"part{i}.pd.bin"
        sitefile.write_pileup_bin(p, x[a:b], position[a:b])                                  # int16 counts on disk (the default)
        sitefile.write_pileup_bin(tmp_path / "bins32" / p.name, x[a:b], position[a:b], matrix_dtype="int32")   # the reference's Int32Atom layout
        files.append(str(p)); files32.append(str(tmp_path / "bins32" / p.name))
        o = tmp_path / "arr.vcf"
        predict_pileup(m, x[a:b], names[a:b], pos[a:b], refb[a:b], fai, str(o), batch_size=100)
        body = o.read_bytes()
        header = host.vcf_header(fai).encode()
        assert body.startswith(header)
        want += body[len(header):]
    want = header + want
    for kw in (dict(), dict(pass_sites=64), dict(pass_sites=7), dict(narrow=False), dict(narrow=False, pass_sites=50), dict(pass_sites=n)):
        o = tmp_path / "stream.vcf"
        st = {}
        rows = predict_pileup_bins(m, files32, fai, str(o), batch_size=100, stats=st, **kw)
        assert o.read_bytes() == want, kw
        assert rows == want.count(b"\n") - header.count(b"\n") and st["sites"] == n
        assert st["passes_int16"] == (0 if kw.get("narrow") is False else st["passes"])
        st = {}
        assert predict_pileup_bins(m, files, fai, str(o), batch_size=100, stats=st, **kw) == rows and o.read_bytes() == want, kw
        assert st["passes_int16"] == st["passes"]                                            # int16 files travel as they are
        mixed = [files[0], files32[1], files32[2], files[3]]
        assert predict_pileup_bins(m, mixed, fai, str(o), batch_size=100, **kw) == rows and o.read_bytes() == want, kw
    # the reference's own call shape (PileupModel/predict.py:37): predict(model, testing_paths, reference_index_file, batch_size, output_file, device)
    from nanosnp_amd import predict as nsnp_predict
    fai_path = tmp_path / "ref.fa.fai"; fai_path.write_text(fai)
    o = tmp_path / "shape.vcf"
    assert nsnp_predict.predict(m, files, str(fai_path), 100, str(o), "cuda") == rows and o.read_bytes() == want
    with pytest.raises(ValueError):
        nsnp_predict.predict(m, files, str(fai_path), 100, str(o), "cpu")
    # a directory: the reference takes os.listdir order and only names ending in .bin (predict.py:215)
    (d / "notes.txt").write_text("not a bin")
    order = [os.path.join(str(d), f) for f in os.listdir(d) if f.endswith(".bin")]
    o1, o2 = tmp_path / "dir.vcf", tmp_path / "list.vcf"
    predict_pileup_bins(m, str(d), fai, str(o1), batch_size=100)
    predict_pileup_bins(m, order, fai, str(o2), batch_size=100)
    assert o1.read_bytes() == o2.read_bytes()
    # a count beyond int16: that pass and the later ones travel as int32, same rows as narrow=False
    xb = x[:200].copy(); xb[150, 3, 2] = 40000
    pb = tmp_path / "big.pd.bin"
    sitefile.write_pileup_bin(pb, xb, position[:200])                                        # does not fit int16: stored as int32
    assert sitefile.array_index(pb)["position_matrix"][0] == np.int32 and sitefile.array_index(files[0])["position_matrix"][0] == np.int16
    st = {}
    predict_pileup_bins(m, [str(pb)], fai, str(o1), batch_size=100, pass_sites=64, stats=st)
    predict_pileup_bins(m, [str(pb)], fai, str(o2), batch_size=100, narrow=False)
    assert o1.read_bytes() == o2.read_bytes() and 0 < st["passes_int16"] < st["passes"]
    # a malformed position field: the reference raises (dataset.py:127)
    sitefile.write_pileup_bin(pb, x[:3], ["chr1:5:" + "A" * 33, "chr1:6", "chr1:7:" + "A" * 33])
    with pytest.raises(host.HostError):
        predict_pileup_bins(m, [str(pb)], fai, str(o1))
    assert predict_pileup_bins(m, files, fai, str(o1), batch_size=100) == want.count(b"\n") - header.count(b"\n")      # the model is usable afterwards


def _pd_rank_worker(rank, world, port, tmp, q):
    import torch.distributed as dist
    from nanosnp_amd.fixtures import load_pileup_weights
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import predict_pileup_bins
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = LSTMNetwork().load_weight_list(load_pileup_weights())
        out = os.path.join(tmp, f"sharded_{rank}.vcf")
        files = [os.path.join(tmp, f"part{i}.pd.bin") for i in range(4)]
        rows = predict_pileup_bins(m, files, open(os.path.join(tmp, "fai.txt")).read(), out, batch_size=100, pass_sites=64)
        alone = os.path.join(tmp, f"alone_{rank}.vcf")                      # distributed=False inside the group: the whole job by this process
        rows_alone = predict_pileup_bins(m, files, open(os.path.join(tmp, "fai.txt")).read(), alone, batch_size=100, distributed=False)
        q.put((rank, rows, os.path.exists(out), rows_alone))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_site_sharded_window_files_write_the_single_process_vcf(tmp_path, pileup_weights, world):
    """pipeline.predict_pileup_bins under a process group: every rank takes its shard_range of every file's windows (all on cuda:0 of
    the one-GPU box, the gather over gloo) cut at batch boundaries and formats its own rows - they depend on the batch a site falls into -,
    rank 0 writes the gathered text: the single-process VCF; an empty file and a file smaller than the number of ranks among them; contigs met in different
    orders by different ranks"""
    import socket
    import torch.multiprocessing as mp
    from nanosnp_amd import sitefile
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import predict_pileup_bins
    z = np.load(golden("pileup_vcf_modes.npz"))
    x = z["x"].astype(np.int32)
    names, pos, refb = list(z["names"]), z["pos"], z["refb"]
    fai = bytes(z["fai"]).decode()
    position = [f"{c}:{int(p)}:{'N' * 16}{chr(int(r))}{'N' * 16}" for c, p, r in zip(names, pos, refb)]
    n = x.shape[0]
    cuts = [0, n // 3, n // 3, n - 2, n]                                   # a third, an empty file, most of the rest, two sites
    files = []
    for i in range(4):
        a, b = cuts[i], cuts[i + 1]
        sitefile.write_pileup_bin(tmp_path / f"part{i}.pd.bin", x[a:b], position[a:b], matrix_dtype="int32" if i == 2 else "int16")
        files.append(str(tmp_path / f"part{i}.pd.bin"))
    (tmp_path / "fai.txt").write_text(fai)
    m = LSTMNetwork().load_weight_list(pileup_weights)
    want = tmp_path / "single.vcf"
    n_rows = predict_pileup_bins(m, files, fai, str(want), batch_size=100, pass_sites=64)
    assert n_rows > 100
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pd_rank_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][:3] == (0, n_rows, True) and all(r[1] == 0 and not r[2] for r in res[1:]) and all(r[3] == n_rows for r in res)
    assert (tmp_path / "sharded_0.vcf").read_bytes() == want.read_bytes()
    for r in range(world):
        assert (tmp_path / f"alone_{r}.vcf").read_bytes() == want.read_bytes()
