"""nsnp_mpileup_tokenise (mpileup_tokenise.hip) against the oracle's byte-at-a-time restatement of the reference's reader
(oracle.mpileup_tokenise: line_reader.cpp:95-127, cpp_aux.cpp:43-59, make_candidate_snp_tensor/main.cpp:162-172) and against the host
tokeniser of libnanosnp_host.so: positions, column offsets, column-5 bytes and reference bytes bit for bit, through the C ABI."""
import numpy as np
import pytest

from nanosnp_amd import host
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1], ids=["three_launches", "one_launch"])
def ctx(request):
    """both forms of the tokeniser: three launches (default: no workgroup waits for another) and the opt-in chained scan in one launch"""
    from nanosnp_amd import _lib
    c = _lib.Context(0)
    c.set_option("tok_fused", request.param)
    return c


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _check(ctx, text, seq=None, shift=0):
    """device result == oracle result; shift: the text starts `shift` bytes into its device buffer (an unaligned pointer)"""
    import torch
    t = np.frombuffer(text, np.uint8)
    buf = torch.zeros(t.size + shift + 64, dtype=torch.uint8, device="cuda")
    buf[shift:shift + t.size] = _dev(t)
    buf[shift + t.size:] = ord("\n")                   # (bytes behind the text must not be looked at: they would add lines)
    if shift:
        buf[:shift] = ord("\n")
    d_seq = _dev(seq) if seq is not None else None
    pos, off, bases, ref = ctx.mpileup_tokenise(buf[shift:shift + t.size], d_seq)
    opos, ooff, obases = oracle.mpileup_tokenise(t)
    assert np.array_equal(pos.cpu().numpy(), opos)
    assert np.array_equal(off.cpu().numpy(), ooff)
    assert np.array_equal(bases.cpu().numpy(), obases)
    if seq is not None:
        assert np.array_equal(ref.cpu().numpy(), seq[opos - 1])
    return opos.size


def test_synthetic_contig_equals_oracle_and_host_parser(ctx):
    cols = host.synth_columns(20260000, 60000, coverage=30)
    text = cols.mpileup_text_native("chrS")
    seq = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(1).integers(0, 4, 70000)]
    n = _check(ctx, bytes(text), seq)
    assert n == 60000
    hpos, hoff, hbases = host.mpileup_parse(bytes(text))
    pos, off, bases, _ = ctx.mpileup_tokenise(_dev(np.frombuffer(bytes(text), np.uint8)))
    assert np.array_equal(pos.cpu().numpy(), hpos) and np.array_equal(off.cpu().numpy(), hoff) and np.array_equal(bases.cpu().numpy(), hbases)
    assert np.array_equal(hbases, cols.bases) and np.array_equal(hoff, cols.col_off)


@pytest.mark.parametrize("shift", [0, 1, 7, 15, 16, 33])
def test_any_alignment_and_text_end(ctx, shift):
    lines = [b"c\t%d\tN\t3\tA+1Ga^]t$\tIII" % (100 + i) for i in range(700)]
    for tail in (b"\n", b"", b"\r\n", b"\r"):
        _check(ctx, b"\n".join(lines) + tail, shift=shift)


def test_reader_corner_cases(ctx):
    """runs of tabs collapse, a leading / trailing tab, CRLF, a missing quality column, more than six fields, atoll-style positions
    (white space, sign, trailing junk, no digits at all), a '\\r' inside a line, another contig name"""
    rng = np.random.default_rng(5)
    lines = []
    for i in range(5000):
        b = bytes(rng.choice(np.frombuffer(b"ACGTacgt*#.,+-^$0123456789N", np.uint8), int(rng.integers(1, 90))))
        f = [b"chr1", b"%d" % (i + 1), b"N", b"%d" % len(b), b, b"I" * int(rng.integers(1, 40))]
        u = rng.random()
        if u < 0.05: f[1] = b"+000" + f[1] + b"xyz"
        elif u < 0.08: f[1] = b"  " + f[1]
        elif u < 0.10: f[1] = b"-" + f[1]
        elif u < 0.11: f[1] = b"x" + f[1]
        elif u < 0.12: f[1] = b"\r" + f[1]
        elif u < 0.13: f[1] = b" "
        elif u < 0.14: f[1] = b"12345678901234567890123"              # wraps like the host's multiply-and-add
        l = b"\t".join(f)
        u = rng.random()
        if u < 0.05: l = l.replace(b"\t", b"\t\t\t", int(rng.integers(1, 6)))
        elif u < 0.08: l = b"\t" + l + b"\t"
        elif u < 0.12: l = b"\t".join(f[:5])
        elif u < 0.16: l = l + b"\r"
        elif u < 0.18: l = l + b"\textra\tfields"
        elif u < 0.20: l = l.replace(b"I", b"\r", 1) if b"I" in l else l
        elif u < 0.22: f[4] = f[4] + b"\r" + f[4]; l = b"\t".join(f)     # a '\r' inside column 5 stays in it
        lines.append(l)
    text = b"\n".join(lines) + b"\n"
    assert _check(ctx, text) == 5000
    assert _check(ctx, text[:-1], shift=3) == 5000


def test_lines_longer_than_a_tile_and_tiny_texts(ctx):
    rng = np.random.default_rng(9)
    big = lambda n: bytes(rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), n))
    lines = [b"c\t1\tN\t9\t" + big(30000) + b"\t" + b"I" * 30000,          # column 5 and the qualities both span tiles
             b"c\t2\tN\t1\tA\tI",
             b"c" * 20000 + b"\t3\tN\t1\t" + big(8192) + b"\tI",               # a contig name longer than a tile
             b"c\t" + b"0" * 9000 + b"4\tN\t1\tG\tI",                         # a position token longer than a tile
             b"c\t5\tN\t1\t" + big(8191), b"c\t6\tN\t1\t" + big(8193) + b"\tI"]
    _check(ctx, b"\n".join(lines) + b"\n")
    _check(ctx, b"\n".join(lines))
    for text in (b"a\tb\tc\td\te", b"a\t7\tc\td\te\n", b"a\t7\tc\td\te\tf\n" * 3):
        _check(ctx, text)
    import torch
    pos, off, bases, ref = ctx.mpileup_tokenise(torch.zeros(0, dtype=torch.uint8, device="cuda"))
    assert pos.numel() == 0 and off.tolist() == [0] and bases.numel() == 0


def test_text_the_reference_cannot_read_is_refused(ctx):
    ok = b"c\t1\tN\t1\tA\tI\n"
    for bad, what in ((ok + b"c\t2\tN\t1\n" + ok, "fewer than five"), (ok + b"\n" + ok, "empty line"), (b"\n" + ok, "empty line"),
                      (ok + b"\r\n" + ok, "empty line"), (ok + ok + b"\r", "empty line"), (ok + b"c\t\t\t2\t\tN\t1\n", "fewer than five")):
        with pytest.raises(ValueError, match=what):
            ctx.mpileup_tokenise(_dev(np.frombuffer(bad, np.uint8)))
        with pytest.raises(ValueError):
            oracle.mpileup_tokenise(bad)
    seq = np.frombuffer(b"ACGTACGT", np.uint8)
    for p in (0, 9, -3):
        with pytest.raises(ValueError, match="outside the reference"):
            ctx.mpileup_tokenise(_dev(np.frombuffer(ok + b"c\t%d\tN\t1\tA\tI\n" % p, np.uint8)), _dev(seq))
    pos, off, bases, ref = ctx.mpileup_tokenise(_dev(np.frombuffer(ok + b"c\t8\tN\t1\tA\tI\n", np.uint8)), _dev(seq))
    assert ref.tolist() == [ord("A"), ord("T")]


def test_capacity_is_respected(ctx):
    """too few columns / bytes: status ERANGE, meta says what is needed, nothing is written beyond the capacities"""
    import torch
    text = b"".join(b"c\t%d\tN\t4\tACGT\tIIII\n" % i for i in range(1, 3001))
    d = _dev(np.frombuffer(text, np.uint8))
    for cap, cap_b in ((100, 100000), (5000, 999), (2999, 11999), (3000, 12000)):
        pos = torch.full((cap + 8,), -7, dtype=torch.int64, device="cuda")
        off = torch.full((cap + 1 + 8,), -7, dtype=torch.int64, device="cuda")
        bases = torch.full((cap_b + 64,), 255, dtype=torch.uint8, device="cuda")
        meta = torch.zeros(4, dtype=torch.int64, pin_memory=True)
        ctx.mpileup_tokenise_into(d, None, pos[:cap], off[:cap + 1], bases[:cap_b], None, meta)
        torch.cuda.synchronize()
        m, nb, status, _ = meta.tolist()
        assert (m, nb) == (3000, 12000) and bool(status & ctx.TOK_ERANGE) == (cap < 3000 or cap_b < 12000)
        assert (pos[cap:] == -7).all() and (off[cap + 1:] == -7).all() and (bases[cap_b:] == 255).all()
        if not status:
            assert off[:cap + 1].tolist() == list(range(0, 12001, 4)) and bytes(bases[:cap_b].cpu().numpy()) == b"ACGT" * 3000


def test_many_tiles_and_repeated_calls(ctx):
    """a text of ~3,000 tiles (the chained scan's look-back windows span several rounds of 64 descriptors), tokenised repeatedly into the
    same buffers (descriptors and status words are re-armed by every call): always the oracle's columns"""
    import torch
    cols = host.synth_columns(20260001, 280_000, coverage=30)
    text = np.frombuffer(bytes(cols.mpileup_text_native("chrS")), np.uint8)
    opos, ooff, obases = oracle.mpileup_tokenise(text)
    d = _dev(text)
    for _ in range(4):
        pos, off, bases, _ = ctx.mpileup_tokenise(d)
        assert np.array_equal(pos.cpu().numpy(), opos) and np.array_equal(off.cpu().numpy(), ooff) and np.array_equal(bases.cpu().numpy(), obases)


def test_random_bytes_equal_the_oracle_reader(ctx):
    """lines of arbitrary bytes (no tab / newline inside a field, '\\r' anywhere), runs of tabs, CRLF, atoll-style positions: the generator of
    tests/test_host.py's host-tokeniser fuzz, here through the device tokeniser at several alignments"""
    from tests.test_host import _random_mpileup_text
    rng = np.random.default_rng(20260607)
    for rnd in range(6):
        text = _random_mpileup_text(rng, 4000)
        _check(ctx, text, shift=int(rng.integers(0, 40)))
