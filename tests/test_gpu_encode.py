"""Pileup encode / site selection / window gather on the GPU: bit-exact vs the reference goldens
and the oracle."""
import gzip

import numpy as np
import pytest

from nanosnp_amd import host
from tests.helpers import golden

pytestmark = pytest.mark.gpu


def _enc(ctx, bases, col_off, ref, **kw):
    import torch
    b = torch.from_numpy(np.ascontiguousarray(bases if bases.size else np.zeros(1, np.uint8))).cuda()
    c, d, f = ctx.pileup_encode_columns(b, torch.from_numpy(np.ascontiguousarray(col_off)).cuda(),
                                        torch.from_numpy(np.ascontiguousarray(ref)).cuda(), **kw)
    torch.cuda.synchronize()
    return c, d, f


@pytest.mark.parametrize("reader", ["device", "host"])
@pytest.mark.parametrize("tag", ["g1", "adv", "end", "cut", "pos", "rdr"])
def test_reference_fixture_end_to_end(gpu_ctx, tag, reader):
    """mpileup text -> tokenise -> encode -> select -> gather == the tensors the reference programs wrote from the same text; the text
    cut into columns by the device tokeniser (nsnp_mpileup_tokenise) or by the host one (nsnp_mpileup_parse)"""
    import torch
    text = gzip.open(golden(f"encode_{tag}.mpileup.gz")).read()
    fa = gzip.open(golden(f"encode_{tag}.fa.gz")).read()
    pd = gzip.open(golden(f"encode_{tag}.pd.gz")).read()
    seq = np.frombuffer(b"".join(fa.split(b"\n")[1:]), np.uint8)
    pos, col_off, bases = host.mpileup_parse(text)
    if reader == "device":
        dpos, doff, dbases, dref = gpu_ctx.mpileup_tokenise(torch.from_numpy(np.frombuffer(text, np.uint8).copy()).cuda(), torch.from_numpy(seq.copy()).cuda())
        assert np.array_equal(dpos.cpu().numpy(), pos) and np.array_equal(doff.cpu().numpy(), col_off) and np.array_equal(dbases.cpu().numpy(), bases)
        c, d, f = gpu_ctx.pileup_encode_columns(dbases, doff, dref)
    else:
        ref = seq[pos - 1]
        c, d, f = _enc(gpu_ctx, bases, col_off, ref)
    centers, n = gpu_ctx.pileup_select_sites(torch.from_numpy(pos).cuda(), f)
    x = gpu_ctx.pileup_gather_windows(c, centers)
    gx, names, gpos, gref = host.pd_parse(pd)
    assert n == gx.shape[0]
    assert np.array_equal(x.cpu().numpy(), gx)
    assert np.array_equal(pos[centers.cpu().numpy()], gpos)
    pd_depth = np.array([int(l.split(b"\t")[2].split(b"-")[0]) for l in pd.splitlines()])
    assert np.array_equal(d.cpu().numpy()[centers.cpu().numpy()], pd_depth)


@pytest.mark.parametrize("seed,cov", [(1, 30), (2, 60), (3, 5)])
def test_random_columns_vs_oracle(gpu_ctx, seed, cov):
    from oracle import oracle
    cols = host.synth_columns(seed, 200_000, coverage=cov)
    ref = cols.ref.copy()
    rng = np.random.default_rng(seed)
    ref[rng.random(ref.size) < 0.05] |= 0x20
    ref[rng.random(ref.size) < 0.01] = ord("N")
    c, d, f = _enc(gpu_ctx, cols.bases, cols.col_off, ref)
    oc, od, of = oracle.encode_columns(cols.bases, cols.col_off, ref)
    assert np.array_equal(c.cpu().numpy(), oc)
    assert np.array_equal(d.cpu().numpy(), od)
    assert np.array_equal(f.cpu().numpy(), of)


def test_thresholds_are_parameters(gpu_ctx):
    from oracle import oracle
    cols = host.synth_columns(9, 20_000, coverage=12)
    for af, mc in ((0.12, 6), (0.2, 10), (0.0, 0), (1.0, 1)):
        c, d, f = _enc(gpu_ctx, cols.bases, cols.col_off, cols.ref, min_af=af, min_coverage=mc)
        oc, od, of = oracle.encode_columns(cols.bases, cols.col_off, cols.ref, af, mc)
        assert np.array_equal(f.cpu().numpy(), of), (af, mc)
    # the reference program's two thresholds apart (-snp_min_af / -indel_min_af): nsnp_pileup_encode_columns2; deep columns (beyond the
    # threshold table: the exact 128-bit test) among them
    deep = host.synth_columns(10, 3000, coverage=400, max_depth=1200)
    for cset in (cols, deep):
        for snp, ind in ((0.12, 0.3), (0.3, 0.05), (0.0, 1.0), (1.0, 0.0), (0.12, 0.12)):
            c, d, f = _enc(gpu_ctx, cset.bases, cset.col_off, cset.ref, min_af=snp, indel_min_af=ind, min_coverage=6)
            oc, od, of = oracle.encode_columns(cset.bases, cset.col_off, cset.ref, snp, 6, indel_min_af=ind)
            assert np.array_equal(f.cpu().numpy(), of) and np.array_equal(c.cpu().numpy(), oc), (snp, ind)
    _, _, f1 = _enc(gpu_ctx, cols.bases, cols.col_off, cols.ref, min_af=0.12, indel_min_af=0.5)
    _, _, f2 = _enc(gpu_ctx, cols.bases, cols.col_off, cols.ref, min_af=0.12)
    assert not np.array_equal(f1.cpu().numpy(), f2.cpu().numpy())                      # (the second threshold does something on this data)


def test_handwritten_edge_columns(gpu_ctx):
    """empty column, '*' only, >60-base indel skipped, '^' swallowing '+', digit-free '+', truncated
    indel at the end of the string, 12 distinct alleles (table overflow path), same allele 9x"""
    from oracle import oracle
    cols = [b"", b"*", b"A+61" + b"C" * 61 + b"A", b"^+A^-a$", b"A+CA", b"AAA+5AC", b"a-2",
            b"".join(b"A+%d%s" % (k + 1, b"ACGTACGTACGT"[:k + 1]) for k in range(12)) + b"A+2AC" * 3,
            b"g-3acg" * 9 + b"G-3ACG" * 2, b"NNNnnn", b"#*#*", b"T" * 1000 + b"+2GG" * 500]
    bases = np.frombuffer(b"".join(cols), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int64)
    ref = np.frombuffer(b"ACGTNacgtnAC", np.uint8)[:len(cols)]
    c, d, f = _enc(gpu_ctx, bases, off, ref)
    oc, od, of = oracle.encode_columns(bases, off, ref)
    assert np.array_equal(c.cpu().numpy(), oc), (c.cpu().numpy(), oc)
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(f.cpu().numpy(), of)
    # an allele the END OF THE COLUMN cuts short is an allele of its own (tensor_maker.cpp:101 reads the declared length past the string:
    # the key holds the NUL) - values as the reference's binaries give them (tests/golden/encode_cut.*): I1 / D1 / i1 / d1 = largest count of ONE allele
    cut = [(b"AAAAAAAAAAAA+2AC+2AC+3AC", {"I": 3, "I1": 2}), (b"AAAAAAAAAAAAaaaaaa-2ac-2ac-3ac", {"d": 3, "d1": 2}),
           (b"AAAAAAAAAAAA+2AC+2AC+2AC+3AC", {"I": 4, "I1": 3}), (b"AAAAAAAAAAAA+1A+2A", {"I": 2, "I1": 1}), (b"AAAAAAAAAAAA+3", {"i": 1, "i1": 1}),
           (b"AAAAAAAAAAAA+2ac+2ac+3ac", {"i": 3, "i1": 2}), (b"AAAA+2AC+2AC+2AC", {"I": 3, "I1": 3})]
    ch = {"I": 4, "I1": 5, "D": 6, "D1": 7, "i": 13, "i1": 14, "d": 15, "d1": 16}
    bases = np.frombuffer(b"".join(c for c, _ in cut), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(c) for c, _ in cut])]).astype(np.int64)
    ref = np.frombuffer(b"A" * len(cut), np.uint8)
    c, d, f = _enc(gpu_ctx, bases, off, ref)
    oc, od, of = oracle.encode_columns(bases, off, ref)
    assert np.array_equal(c.cpu().numpy(), oc) and np.array_equal(f.cpu().numpy(), of)
    for k, (_, want) in enumerate(cut):
        got = {name: int(oc[k, i]) for name, i in ch.items()}
        assert all(got[n] == v for n, v in want.items()) and sum(got.values()) == sum(want.values()), (k, got, want)


def test_opener_dense_columns(gpu_ctx):
    """columns made of construct openers: more flagged openers in one column than the wave's entry list holds (exact path), a wave
    whose openers need several segments of the list, openers shadowed by the bytes an earlier one consumes, four-digit lengths,
    alleles longer than the four bytes the multiplicity compare takes at once, columns at both ends of the 253-byte fast path"""
    from oracle import oracle
    rng = np.random.default_rng(5)
    cols = [b"^" * 200, b"+" * 150, b"-" * 253, b"^+" * 100, b"A+1^C-1+G", b"A+1000" + b"C" * 30, b"T-0A+0", b"^^^A",
            b"A+6ACGTAC" * 10 + b"A+6ACGTAG" * 7 + b"A+6ACGTAC", b"c-7acgtacg" * 5 + b"c-7acgtacc" * 6,
            b"A" * 253, b"A" * 254, b"C" * 250 + b"+1G", b"G" * 249 + b"-2AC", b"^" * 127 + b"A" * 126]
    for _ in range(120):                                          # a whole wave of indel-only columns: ~ 64 x 40 openers
        n = int(rng.integers(20, 60))
        cols.append(b"".join(bytes(rng.choice(list(b"ACGTacgt"), 1)) + (b"+" if rng.random() < .5 else b"-") + b"2" +
                             bytes(rng.choice(list(b"ACGT"), 2)) for _ in range(n)))
    bases = np.frombuffer(b"".join(cols), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int64)
    ref = rng.choice(list(b"ACGTN"), len(cols)).astype(np.uint8)
    c, d, f = _enc(gpu_ctx, bases, off, ref)
    oc, od, of = oracle.encode_columns(bases, off, ref)
    bad = np.nonzero((c.cpu().numpy() != oc).any(1))[0]
    assert bad.size == 0, (bad[:5], c.cpu().numpy()[bad[:2]], oc[bad[:2]])
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(f.cpu().numpy(), of)


def test_entry_list_capacity_boundaries(gpu_ctx):
    """the opener list of a wave's segment holds 208 entries: columns and whole waves (64 consecutive columns) with exactly 208 and
    209 openers, a wave whose first column fills the list alone, read starts (dropped from the compacted list) between counted indels"""
    from oracle import oracle
    four = b"A+1CA-1GA+1CA-2GT"                                    # four openers, all counted
    waves = [
        [four] * 52 + [b"ACGTacgt"] * 12,                           # 208: one segment
        [four] * 52 + [b"AC^]GT"] + [b"ACGT"] * 11,                  # 209: two segments
        [b"+" * 208] + [four] * 63,                                 # the first column fills a segment alone
        [b"+" * 209] + [b"A"] * 63,                                 # one column beyond the list: exact path
        [b"^+" * 60 + b"A+2CC" * 20] + [b"^!A+3ACG^~c-3acg" * 3] * 63,   # read starts whose quality byte is an opener; mixed lists
        [b"A+2AC" * 3 + b"^IA+2AC" * 2 + b"a-2ac^Ia-2ac"] * 64,       # equal alleles around read starts: multiplicities 5 and 2
    ]
    cols = [c for w in waves for c in w]
    bases = np.frombuffer(b"".join(cols), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int64)
    ref = np.frombuffer((b"ACGTN" * len(cols))[:len(cols)], np.uint8).copy()
    c, d, f = _enc(gpu_ctx, bases, off, ref)
    oc, od, of = oracle.encode_columns(bases, off, ref)
    bad = np.nonzero((c.cpu().numpy() != oc).any(1))[0]
    assert bad.size == 0, (bad[:5], c.cpu().numpy()[bad[:2]], oc[bad[:2]])
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(f.cpu().numpy(), of)


def test_single_class_columns_at_the_8bit_counter_limit(gpu_ctx):
    """columns of ONE symbol class starting on a 4-byte boundary: 256 equal bytes span exactly 64 words (the fast path's
    word limit) but wrap an 8-bit class counter into its neighbour - they must take the exact path"""
    from oracle import oracle
    cols = [b"A" * 256, b"a" * 256, b"A" * 255 + b"C", b"*" * 256, b"#" * 252, b"T" * 254 + b"^]", b"g" * 253, b"G" * 256]
    cols = [c + b"C" * ((-len(c)) % 4) if i < 4 else c for i, c in enumerate(cols)]          # the first four stay word-aligned
    bases = np.frombuffer(b"".join(cols), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int64)
    ref = np.frombuffer(b"ACGTNacg", np.uint8).copy()
    c, d, f = _enc(gpu_ctx, bases, off, ref)
    oc, od, of = oracle.encode_columns(bases, off, ref)
    assert np.array_equal(c.cpu().numpy(), oc), (c.cpu().numpy(), oc)
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(f.cpu().numpy(), of)


def test_empty_call(gpu_ctx):
    import torch
    e8 = torch.zeros(1, dtype=torch.uint8, device="cuda")
    c, d, f = gpu_ctx.pileup_encode_columns(e8, torch.zeros(1, dtype=torch.int64, device="cuda"), e8[:0])
    assert c.shape == (0, 18)
    centers, n = gpu_ctx.pileup_select_sites(torch.zeros(0, dtype=torch.int64, device="cuda"), e8[:0])
    assert n == 0


@pytest.mark.parametrize("m", [1, 32, 33, 34, 1000, 100_003])
def test_select_sites_vs_oracle(gpu_ctx, m):
    """position gaps, candidates at run edges, runs shorter than a window"""
    import torch
    from oracle import oracle
    rng = np.random.default_rng(m)
    step = np.where(rng.random(m) < 0.02, rng.integers(2, 5, m), 1)
    pos = np.cumsum(step).astype(np.int64)
    flags = np.where(rng.random(m) < 0.3, 8, 0).astype(np.uint8) | rng.integers(0, 8, m).astype(np.uint8)
    want = oracle.select_sites(pos, flags)
    got, n = gpu_ctx.pileup_select_sites(torch.from_numpy(pos).cuda(), torch.from_numpy(flags).cuda())
    assert n == len(want)
    assert np.array_equal(got.cpu().numpy(), want)
    if len(want) > 3:       # capacity smaller than the result: count still exact, prefix written
        got2, n2 = gpu_ctx.pileup_select_sites(torch.from_numpy(pos).cuda(), torch.from_numpy(flags).cuda(), cap=3)
        assert n2 == len(want) and np.array_equal(got2.cpu().numpy(), want[:3])
    # positions that repeat or step back: every step inside the window counts, a gap of two and a repeat must not cancel (main.cpp:174-178)
    u = rng.random(m)
    step2 = np.where(u < 0.01, 0, np.where(u < 0.02, -rng.integers(1, 20, m), np.where(u < 0.04, 2, 1)))
    pos2 = (np.cumsum(step2) + 1000).astype(np.int64)
    flags2 = np.where(rng.random(m) < 0.5, 8, 0).astype(np.uint8)
    want2 = oracle.select_sites(pos2, flags2)
    got3, n3 = gpu_ctx.pileup_select_sites(torch.from_numpy(pos2).cuda(), torch.from_numpy(flags2).cuda())
    assert n3 == len(want2) and np.array_equal(got3.cpu().numpy(), want2)


def test_full_size_invariants_1m_columns(gpu_ctx):
    """Size-independent checks at benchmark scale (33 x 32768 columns): the depth identity
    depth = -(ref_upper + ref_lower) + '*' + '#', non-negative non-reference channels, I1 <= I, and
    a random sample of columns against the oracle."""
    from oracle import oracle
    n = 32768
    cols = host.synth_columns(20260000, n * 33, coverage=30, window=33)
    c, d, f = _enc(gpu_ctx, cols.bases, cols.col_off, cols.ref)
    c, d = c.cpu().numpy(), d.cpu().numpy()
    ridx = np.searchsorted(np.frombuffer(b"ACGT", np.uint8), cols.ref)
    rows = np.arange(c.shape[0])
    assert np.array_equal(d, -(c[rows, ridx] + c[rows, 9 + ridx]) + c[:, 8] + c[:, 17])
    assert (c[:, 5] <= c[:, 4]).all() and (c[:, 7] <= c[:, 6]).all() and (c[:, 14] <= c[:, 13]).all()
    sample = np.random.default_rng(0).choice(c.shape[0], 20000, replace=False)
    sb = np.concatenate([cols.bases[cols.col_off[i]:cols.col_off[i + 1]] for i in sample])
    so = np.concatenate([[0], np.cumsum(cols.col_off[sample + 1] - cols.col_off[sample])]).astype(np.int64)
    oc, od, _ = oracle.encode_columns(sb, so, cols.ref[sample])
    assert np.array_equal(c[sample], oc) and np.array_equal(d[sample], od)


def test_deep_and_indel_heavy_columns_take_the_sub_batch_and_rescan_paths(gpu_ctx):
    """waves whose 64 columns exceed the LDS stage (split into sub-batches), single columns longer than
    the stage (global path), columns beyond the fast path's 253 bytes (exact path out of LDS) and waves with more openers than one
    segment of the entry list holds"""
    from oracle import oracle
    rng = np.random.default_rng(21)
    cols = []
    for c in range(700):
        kind = c % 7
        depth = [30, 144, 400, 30, 2000, 60, 30][kind]
        parts = []
        for r in range(depth):
            sym = "ACGTacgt*#"[int(rng.integers(0, 10))]
            s = sym
            if kind in (1, 3, 4) and rng.random() < (0.6 if kind != 4 else 0.1):
                L = int(rng.integers(1, 6))
                seq = "".join(rng.choice(list("ACGT"), L))
                if rng.random() < 0.5:
                    seq = ["A", "AC", "ACG"][int(rng.integers(0, 3))]
                s += ("+" if rng.random() < 0.5 else "-") + str(len(seq)) + (seq.lower() if sym.islower() else seq)
            parts.append(s)
        cols.append("".join(parts).encode())
    bases = np.frombuffer(b"".join(cols), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int64)
    ref = rng.choice(list(b"ACGTN"), len(cols)).astype(np.uint8)
    c, d, f = _enc(gpu_ctx, bases, off, ref)
    oc, od, of = oracle.encode_columns(bases, off, ref)
    bad = np.nonzero((c.cpu().numpy() != oc).any(1))[0]
    assert bad.size == 0, (bad[:5], c.cpu().numpy()[bad[:2]], oc[bad[:2]])
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(f.cpu().numpy(), of)


def test_af_test_in_integers_equals_the_float64_division(gpu_ctx):
    """the kernel decides (double)count / depth >= min_af with one 128-bit integer comparison (pileup_encode.hip: AfThreshold);
    the oracle divides in float64 as tensor_maker.cpp:195-228 does.  Flags must agree for every threshold class: ordinary values,
    quotients that hit the threshold exactly (k / depth for many depths), powers of two, zero / negative / tiny / huge / NaN."""
    from oracle import oracle
    cols = host.synth_columns(11, 60_000, coverage=30)
    rng = np.random.default_rng(3)
    # small depths too, so that simple fractions (1/3, 2/7, ...) occur as quotients
    cols2 = host.synth_columns(12, 60_000, coverage=6)
    thresholds = [0.12, 0.3, 0.5, 0.25, 1.0, 0.0, -1.0, 1e-30, 5e-324, 2.5, 1e12, float("inf"), float("nan"),
                  1 / 3, 2 / 7, 3 / 8, 0.1, 0.2, np.nextafter(0.25, 0), np.nextafter(0.25, 1), np.nextafter(1 / 3, 0), np.nextafter(1 / 3, 1),
                  float(rng.random()), float(rng.random())]
    for cs in (cols, cols2):
        for a in thresholds:
            c, d, f = _enc(gpu_ctx, cs.bases, cs.col_off, cs.ref, min_af=a, min_coverage=2)
            oc, od, of = oracle.encode_columns(cs.bases, cs.col_off, cs.ref, a, 2)
            assert np.array_equal(f.cpu().numpy(), of), a
            assert np.array_equal(c.cpu().numpy(), oc) and np.array_equal(d.cpu().numpy(), od)


def test_select_sites_range_and_call_rows_equal_the_elementwise_forms(gpu_ctx):
    """the two fused entries of the streamed text pipeline: nsnp_pileup_select_sites_range = select_sites + where the chunk's own sites lie
    in the ascending list (into pinned memory, no host round trip), nsnp_pileup_call_rows = the [n,13] float64 rows in one launch"""
    import torch
    from nanosnp_amd.predict import COV_CHANNELS
    cols = host.synth_columns(20260123, 120_000, coverage=30, het_rate=0.05)
    dev = torch.device("cuda")
    pos_np = np.arange(1, 120_001, dtype=np.int64)
    pos_np[50_000:] += 7                                               # a gap: windows across it are not emitted
    pos = torch.from_numpy(pos_np).to(dev)
    c, d, f = _enc(gpu_ctx, cols.bases, cols.col_off, cols.ref)
    centers, n = gpu_ctx.pileup_select_sites(pos, f)
    assert n > 1000
    for lo, hi in ((0, 120_000), (16, 119_984), (40_000, 80_000), (60_000, 60_000), (119_999, 120_000)):
        meta = torch.full((4,), -1, dtype=torch.int64, pin_memory=True)
        center = gpu_ctx.pileup_select_sites_range(pos, f, lo, hi, meta)
        torch.cuda.synchronize()
        m = meta.tolist()
        assert m[0] == m[3] == n and torch.equal(center[:n], centers)
        cn = centers.cpu().numpy()
        assert m[1] == int((cn < lo).sum()) and m[2] == int((cn < hi).sum())
    meta_d = torch.zeros(4, dtype=torch.int64, device=dev)            # meta on the device works as well
    gpu_ctx.pileup_select_sites_range(pos, f, 100, 5000, meta_d)
    cn = centers.cpu().numpy()
    assert meta_d.tolist() == [n, int((cn < 100).sum()), int((cn < 5000).sum()), n]
    # rows
    k = centers.shape[0]
    g = torch.Generator(device="cpu").manual_seed(3)
    ga = torch.randint(0, 21, (k,), generator=g, dtype=torch.uint8).to(dev); za = torch.randint(0, 3, (k,), generator=g, dtype=torch.uint8).to(dev)
    gm = torch.rand(k, generator=g).to(dev); zm = torch.rand(k, generator=g).to(dev)
    rows = gpu_ctx.pileup_call_rows(c, centers, pos, ga, za, gm, zm)
    cov = c.index_select(0, centers).index_select(1, torch.tensor(COV_CHANNELS, device=dev)).to(torch.float64)
    f64 = lambda t: t.to(torch.float64)[:, None]
    want = torch.cat([f64(pos.index_select(0, centers)), f64(ga), f64(za), f64(gm), f64(zm), cov], dim=1)
    assert rows.dtype == torch.float64 and torch.equal(rows, want)
    assert gpu_ctx.pileup_call_rows(c, centers[:0], pos, ga[:0], za[:0], gm[:0], zm[:0]).shape == (0, 13)
