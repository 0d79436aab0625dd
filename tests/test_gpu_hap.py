"""Haplotype feature reduction (and forward) on the GPU vs goldens / oracle."""
import numpy as np
import pytest

from nanosnp_amd import host
from tests.helpers import golden

pytestmark = pytest.mark.gpu


def _feat(ctx, planes):
    import torch
    out = ctx.hap_features(*[torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda() for a in planes])
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("tag", ["p", "h"])
def test_golden_features_bit_identical(gpu_ctx, tag):
    """float64 math + fp32 cast == dataset_dev.get_frequency_feature + predict_dev.py:35 cast"""
    z = np.load(golden("hap_features.npz"))
    planes = [z[f"{tag}_{k}"] for k in ("seq", "bq", "mq", "hap", "ref")]
    got = _feat(gpu_ctx, planes)
    want = np.concatenate([z[f"{tag}_feat"], z[f"{tag}_ref"].astype(np.float64)[:, None, :]], 1).astype(np.float32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("cov,D,L", [(30, 90, 33), (30, 90, 11), (60, 180, 33), (60, 180, 11), (2, 1, 33), (100, 300, 33)])
def test_random_planes_vs_oracle(gpu_ctx, cov, D, L):
    from oracle import oracle
    planes = host.synth_hap_planes(D * 1000 + L, 200, coverage=cov, depth=D, length=L)
    got = _feat(gpu_ctx, planes)
    want = oracle.hap_features_batch(*planes, nthreads=4)
    assert np.array_equal(got, want)


def test_int8_planes_give_the_same_features(gpu_ctx):
    """nsnp_hap_features_i8: the planes as int8 (all values fit), a quarter of the bytes, bit-identical features"""
    import torch
    planes = host.synth_hap_planes(4242, 300, coverage=60, depth=180, length=33)
    want = _feat(gpu_ctx, planes)
    narrow = [torch.from_numpy(a.astype(np.int8)).cuda() for a in planes[:4]] + [torch.from_numpy(planes[4]).cuda()]
    got = gpu_ctx.hap_features(*narrow)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    from nanosnp_amd._lib import NanoSNPError
    with pytest.raises(NanoSNPError):
        gpu_ctx.hap_features(narrow[0], narrow[1].to(torch.int32), narrow[2], narrow[3], narrow[4])


@pytest.mark.parametrize("D", [1, 2, 7, 90, 91, 300, 1000])
@pytest.mark.parametrize("narrow", [False, True])
def test_adversarial_planes_at_the_pileup_window(gpu_ctx, D, narrow):
    """L = 33 on planes that are NOT sorted by haplotype tag, on rows that belong to several read sets or to none, with the tag only in
    column 32, with odd and tiny depths, with depths beyond the flush period of the packed sums, and (int32) with qualities the packed
    fields cannot hold: bit-identical to the oracle.  (Written for the round-5 kernel experiments - two rows per wave step, LDS-staged
    persistent workgroups: docs/rounds/r05.md - which it caught nothing wrong with and which were dropped for being slower.)"""
    import torch
    from oracle import oracle
    rng = np.random.default_rng(100 + D)
    N, L = 24, 33
    seq = rng.integers(-2, 5, (N, D, L)).astype(np.int32)
    hap = rng.integers(0, 4, (N, D, 1)).astype(np.int32) * (rng.random((N, D, L)) < 0.9)      # one tag per row, zeros sprinkled in
    hap[0] = rng.integers(-2, 4, (D, L))                                                     # site 0: every row a mix of tags
    hap[1] = 0; hap[1, :, 32] = rng.integers(0, 4, D)                                        # site 1: the tag only in column 32
    hap[2] = np.sort(rng.integers(1, 4, D))[:, None]                                         # site 2: sorted, as the reference's bins are
    hap[3] = 0                                                                               # site 3: no row in any set
    bq = rng.integers(0, 94, (N, D, L)).astype(np.int32); mq = rng.integers(0, 61, (N, D, L)).astype(np.int32)
    seq[4, D // 2:] = -2; hap[4, D // 2:] = -2; bq[4, D // 2:] = -2; mq[4, D // 2:] = -2     # site 4: padding rows behind the reads
    if not narrow:
        bq[5] = rng.integers(0, 2**30, (D, L)); mq[5, ::3] = rng.integers(2040, 2056, mq[5, ::3].shape)   # beyond the 16-bit fields
        bq[6, :, 32] = 2**31 - 1
    ref = rng.integers(0, 5, (N, L)).astype(np.int32)
    dt = np.int8 if narrow else np.int32
    dev = [torch.from_numpy(a.astype(dt)).cuda() for a in (seq, bq, mq, hap)] + [torch.from_numpy(ref).cuda()]
    got = gpu_ctx.hap_features(*dev)
    torch.cuda.synchronize()
    want = oracle.hap_features_batch(seq, bq, mq, hap, ref, nthreads=4)
    assert np.array_equal(got.cpu().numpy(), want)


def test_mixed_hp_rows_and_large_values(gpu_ctx):
    """a row with several HP values belongs to several read sets (np.any semantics); int32-range
    qualities need 64-bit sums"""
    from oracle import oracle
    rng = np.random.default_rng(1)
    N, D, L = 16, 40, 33
    seq = rng.integers(-2, 5, (N, D, L)).astype(np.int32)
    hap = rng.integers(-2, 4, (N, D, L)).astype(np.int32)
    bq = rng.integers(0, 2**30, (N, D, L)).astype(np.int32)
    mq = rng.integers(-5, 2**30, (N, D, L)).astype(np.int32)
    ref = rng.integers(0, 5, (N, L)).astype(np.int32)
    got = _feat(gpu_ctx, (seq, bq, mq, hap, ref))
    want = oracle.hap_features_batch(seq, bq, mq, hap, ref)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("D,L", [(40, 11), (700, 11), (130, 33), (64, 64), (50, 7), (33, 32)])
def test_packed_sums_boundaries_and_the_exact_path(gpu_ctx, D, L):
    """the running sums are packed (8-bit counts, 16-bit quality sums, flushed every 28 rows of a lane); qualities outside [0, 2048)
    bypass them: values around the boundary, negatives and int32 extremes mixed with ordinary ones, several rows per wave
    (L <= 32), depths that need several flushes"""
    from oracle import oracle
    rng = np.random.default_rng(D * 100 + L)
    N = 24
    seq = rng.integers(-2, 5, (N, D, L)).astype(np.int32)
    hap = rng.integers(-2, 4, (N, D, L)).astype(np.int32)
    pool = np.array([0, 1, 40, 93, 2046, 2047, 2048, 2049, 65535, 65536, -1, -2048, 2**31 - 1, -2**31], np.int64)
    pick = lambda: np.where(rng.random((N, D, L)) < 0.9, rng.integers(0, 94, (N, D, L)), pool[rng.integers(0, pool.size, (N, D, L))]).astype(np.int32)
    bq, mq = pick(), pick()
    seq[0] = 1; hap[0] = 1; bq[0] = 2047; mq[0] = 2047              # one column class at the largest packed value, every row in HP1
    seq[1] = 2; bq[1] = 2048; mq[1] = -1                             # everything through the exact path
    ref = rng.integers(0, 5, (N, L)).astype(np.int32)
    got = _feat(gpu_ctx, (seq, bq, mq, hap, ref))
    want = oracle.hap_features_batch(seq, bq, mq, hap, ref)
    assert np.array_equal(got, want)


# ---- HaplotypeModel forward -------------------------------------------------------------------------
@pytest.fixture(scope="module", params=[2, 1, 0], ids=["bf16x3", "f16x3", "fp32"])
def hap_model(request, gpu_ctx):
    """the forward tests run in all three modes: exact fp32 (library default), bf16x3 (full fp32 operand width on the bf16 pipe) and the opt-in f16x3"""
    from tests.helpers import seeded_hap_weights
    ws = seeded_hap_weights(12, H=256)
    gpu_ctx.hap_load_weights(ws)
    gpu_ctx.set_option("hap_precision", request.param)
    yield gpu_ctx, ws
    gpu_ctx.set_option("hap_precision", 0)


def _hfwd(ctx, xp, xh):
    import torch
    gt, zy = ctx.hap_forward(torch.from_numpy(np.ascontiguousarray(xp)).cuda(), torch.from_numpy(np.ascontiguousarray(xh)).cuda())
    torch.cuda.synchronize()
    return gt.cpu().numpy(), zy.cpu().numpy()


def test_hap_forward_golden_of_the_reference_module(hap_model):
    """model_dev.LSTMNetwork.predict with the seeded weights (trained weights are absent upstream)"""
    from tests.helpers import PROB_ATOL
    ctx, _ = hap_model
    z = np.load(golden("hap_fwd_h256.npz"))
    assert int(z["seed"]) == 12
    gt, zy = _hfwd(ctx, z["xp"], z["xh"])
    assert np.abs(gt - z["gt"]).max() < PROB_ATOL and np.abs(zy - z["zy"]).max() < PROB_ATOL


@pytest.mark.parametrize("prec", [0, 1, 2], ids=["fp32", "f16x3", "bf16x3"])
def test_hap_forward_golden_with_site_dependent_outputs(prec):
    """hap_fwd_h256x.npz (48 sites, three genotype classes, p_max 0.37 .. 0.90; reference module with scaled seeded weights)"""
    from nanosnp_amd import _lib
    from tests.helpers import PROB_ATOL, seeded_hap_weights
    z = np.load(golden("hap_fwd_h256x.npz"))
    c = _lib.Context(0)
    c.hap_load_weights(seeded_hap_weights(int(z["seed"]), H=256, ih_scale=0.03, head_scale=120.0))
    c.set_option("hap_precision", prec)
    gt, zy = _hfwd(c, z["xp"], z["xh"])
    assert np.abs(gt - z["gt"]).max() < PROB_ATOL and np.abs(zy - z["zy"]).max() < PROB_ATOL
    assert np.array_equal(gt.argmax(1), z["gt"].argmax(1))
    c.close()


@pytest.mark.parametrize("prec,narrow", [(0, False), (0, True), (2, False), (1, False)], ids=["fp32-int32planes", "fp32-int8planes", "bf16x3", "f16x3"])
def test_features_and_forward_large_golden_incl_edge_sites(prec, narrow):
    """hap_fwd_large.npz: 256 sites through the reference's own get_frequency_feature + ref row + LSTMNetwork.predict, incl. all-padding
    planes, depth-1 sites, saturated features and deletion-only sites; here read planes -> nsnp_hap_features -> nsnp_hap_forward"""
    import torch
    from nanosnp_amd import _lib
    from tests.helpers import PROB_ATOL, seeded_hap_weights
    z = np.load(golden("hap_fwd_large.npz"))
    c = _lib.Context(0)
    c.hap_load_weights(seeded_hap_weights(int(z["seed"]), H=256, ih_scale=0.03, head_scale=120.0))
    c.set_option("hap_precision", prec)
    xs = []
    for t in ("p", "h"):
        pl = [torch.from_numpy(z[f"{t}_{k}"].astype(np.int8 if narrow else np.int32)).cuda() for k in ("seq", "bq", "mq", "hap")]
        xs.append(c.hap_features(*pl, torch.from_numpy(z[f"{t}_ref"].astype(np.int32)).cuda()))
    gt, zy = c.hap_forward(xs[0], xs[1])
    torch.cuda.synchronize()
    gt, zy = gt.cpu().numpy(), zy.cpu().numpy()
    dev = np.maximum(np.abs(gt - z["gt"]).max(1), np.abs(zy - z["zy"]).max(1))
    print("max |dp| %.3g at site %d" % (dev.max(), dev.argmax()))
    assert np.isfinite(gt).all()
    if prec != 1:
        # the default arithmetic AND the bf16x3 mode (24 significand bits per operand) hold the contract on every site, the 32 edge
        # sites with their saturated feature sums included - no relaxed bound
        assert dev.max() < PROB_ATOL
    else:
        # f16x3 carries 21-22 significand bits per operand and its DOCUMENTED bound (include/nanosnp.h) is 1e-4 on inputs whose
        # features stay below 2048 and 2e-4 beyond: on this fixture (heads scaled x120 so that errors show, feature sums up to 8,370
        # on the saturated sites) the worst site measures 1.5e-4; the G3 sites stay inside 1e-4
        assert dev[32:].max() < PROB_ATOL and dev.max() < 2e-4
    top2 = np.sort(z["gt"], 1)[:, -2:]
    assert np.all((gt.argmax(1) == z["gt"].argmax(1)) | (top2[:, 1] - top2[:, 0] < 1e-3))
    c.close()


def test_full_stage5_pool_properties():
    """the whole 150,000-site stage-5 pool of the bench (generator G3, int8 read planes) through features + forward in both arithmetic
    modes: probabilities finite, rows sum to 1, fp32 and f16x3 agree within the port's tolerance and on the argmax wherever the top
    two classes are further apart than that tolerance"""
    import torch
    from nanosnp_amd import _lib
    from tests.helpers import seeded_hap_weights
    n, chunk = 150_000, 16384
    c = _lib.Context(0)
    c.hap_load_weights(seeded_hap_weights(12, H=256))
    out = {0: [], 1: [], 2: []}
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        xs = []
        for L, sd in ((33, 20260400), (11, 20260500)):
            pl = host.synth_hap_planes(sd + c0, m, 30, 90, L)
            xs.append(c.hap_features(*[torch.from_numpy(a.astype(np.int8)).cuda() for a in pl[:4]], torch.from_numpy(pl[4]).cuda()))
        for prec in (0, 1, 2):
            c.set_option("hap_precision", prec)
            gt, zy = c.hap_forward(xs[0], xs[1])
            out[prec].append(torch.cat([gt, zy], 1))
    p32, p16, pb3 = torch.cat(out[0]), torch.cat(out[1]), torch.cat(out[2])
    assert p32.shape == (n, 13) and bool(torch.isfinite(p32).all()) and bool(torch.isfinite(p16).all()) and bool(torch.isfinite(pb3).all())
    d3 = float((p32 - pb3).abs().max())
    print("stage-5 pool: max |p_fp32 - p_bf16x3| =", d3)
    assert d3 < 5e-6                                    # fp32 summation-order noise (the two fp32-width paths sum in different orders)
    for p in (p32, p16, pb3):
        assert float((p[:, :10].sum(1) - 1).abs().max()) < 1e-5 and float((p[:, 10:].sum(1) - 1).abs().max()) < 1e-5
        assert float(p.min()) >= 0.0
    d = float((p32 - p16).abs().max())
    print("stage-5 pool: max |p_fp32 - p_f16x3| =", d)
    assert d < 1e-4
    top2 = p32[:, :10].topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2e-4
    assert bool((p32[:, :10].argmax(1) == p16[:, :10].argmax(1))[clear].all()) and int(clear.sum()) > n // 2
    c.close()


@pytest.mark.parametrize("n", [1, 127, 128, 129, 300])
def test_hap_forward_vs_oracle_ragged(hap_model, n):
    from oracle import oracle
    from tests.helpers import PROB_ATOL
    ctx, ws = hap_model
    pp = host.synth_hap_planes(900 + n, n, 30, 90, 33)
    ph = host.synth_hap_planes(950 + n, n, 30, 90, 11)
    xp = oracle.hap_features_batch(*pp, nthreads=4)
    xh = oracle.hap_features_batch(*ph, nthreads=4)
    gt, zy = _hfwd(ctx, xp, xh)
    ogt, ozy = oracle.hap_forward(ws, xp, xh, nthreads=8)
    assert np.abs(gt - ogt).max() < PROB_ATOL and np.abs(zy - ozy).max() < PROB_ATOL
    assert np.allclose(gt.sum(1), 1, atol=1e-5)


def test_hap_forward_sensitive_inputs(hap_model):
    """random O(1) features make the network input-sensitive (synthetic count features saturate the
    seeded-weight gates), so layer/step/direction mix-ups show up as O(0.1) errors"""
    from oracle import oracle
    from tests.helpers import PROB_ATOL
    ctx, ws = hap_model
    rng = np.random.default_rng(4)
    n = 96
    xp = (rng.standard_normal((n, 105, 33)) * 300).astype(np.float32)
    xh = (rng.standard_normal((n, 105, 11)) * 300).astype(np.float32)
    gt, zy = _hfwd(ctx, xp, xh)
    ogt, ozy = oracle.hap_forward(ws, xp, xh, nthreads=8)
    assert ogt.std(0).max() > 1e-3          # outputs really differ between sites
    assert np.abs(gt - ogt).max() < PROB_ATOL and np.abs(zy - ozy).max() < PROB_ATOL


def test_hap_features_into_forward_pipeline(hap_model):
    """predict_dev.py:34-39 end to end on the device: planes -> features -> forward"""
    import torch
    from oracle import oracle
    from tests.helpers import PROB_ATOL
    ctx, ws = hap_model
    n = 64
    pp = host.synth_hap_planes(1, n, 30, 90, 33)
    ph = host.synth_hap_planes(2, n, 30, 90, 11)
    fp = ctx.hap_features(*[torch.from_numpy(a).cuda() for a in pp])
    fh = ctx.hap_features(*[torch.from_numpy(a).cuda() for a in ph])
    gt, zy = ctx.hap_forward(fp, fh)
    ogt, ozy = oracle.hap_forward(ws, oracle.hap_features_batch(*pp), oracle.hap_features_batch(*ph), nthreads=8)
    assert np.abs(gt.cpu().numpy() - ogt).max() < PROB_ATOL


def test_hap_unsupported_shape_is_an_error(gpu_ctx):
    from nanosnp_amd import _lib
    from tests.helpers import seeded_hap_weights
    with pytest.raises(_lib.NanoSNPError):
        gpu_ctx.hap_load_weights(seeded_hap_weights(11, H=32), hidden=32)


def test_hap_arrange_reads_vs_oracle(gpu_ctx):
    import torch
    from oracle import oracle
    rng = np.random.default_rng(3)
    N, R, L = 50, 150, 33
    for D_out in (90, 40, 200):
        seq = rng.integers(-1, 5, (N, R, L)).astype(np.int32)
        seq[rng.random((N, R)) < 0.25, L // 2] = 0
        hap = np.where(seq != 0, rng.integers(1, 4, (N, R, 1)), 0).astype(np.int32)
        bq = rng.integers(0, 60, (N, R, L)).astype(np.int32); mq = rng.integers(0, 61, (N, R, L)).astype(np.int32)
        n_reads = rng.integers(0, R + 1, N).astype(np.int32)
        outs = gpu_ctx.hap_arrange_reads(*[torch.from_numpy(a).cuda() for a in (seq, bq, mq, hap)], D_out,
                                         n_reads=torch.from_numpy(n_reads).cuda())
        torch.cuda.synchronize()
        for n in range(N):
            want = oracle.hap_arrange(seq[n], bq[n], mq[n], hap[n], D_out, rows=n_reads[n])
            for k in range(4):
                assert np.array_equal(outs[k][n].cpu().numpy(), want[k]), (n, k)
            assert int(outs[4][n]) == want[4]
    # an HP tag is whatever integer the BAM holds: every int32 value sorts, INT_MAX included (once the kernel's mark of a dropped row),
    # for the two rank paths (R <= 64: four waves share a row's comparisons; R > 64)
    for R2 in (40, 64, 65):
        seq = rng.integers(-1, 5, (7, R2, 11)).astype(np.int32)
        hap = rng.choice([1, 2, 2 ** 31 - 1, -2 ** 31, 0, -5], (7, R2, 11)).astype(np.int32)
        bq = rng.integers(0, 60, (7, R2, 11)).astype(np.int32); mq = rng.integers(0, 61, (7, R2, 11)).astype(np.int32)
        outs = gpu_ctx.hap_arrange_reads(*[torch.from_numpy(a).cuda() for a in (seq, bq, mq, hap)], 50)
        for n in range(7):
            want = oracle.hap_arrange(seq[n], bq[n], mq[n], hap[n], 50)
            assert all(np.array_equal(outs[k][n].cpu().numpy(), want[k]) for k in range(4)) and int(outs[4][n]) == want[4], (R2, n)


def test_hap_forward_f16x3_mode(gpu_ctx):
    """every fp32 product as three fp16 MFMAs (hi.hi + lo.hi + hi.lo), fp32 accumulate: same goldens and
    tolerance as the exact-fp32 mode"""
    import torch
    from nanosnp_amd import _lib
    from oracle import oracle
    from tests.helpers import PROB_ATOL, seeded_hap_weights
    ws = seeded_hap_weights(12, H=256)
    c = _lib.Context(0)
    c.hap_load_weights(ws)
    z = np.load(golden("hap_fwd_h256.npz"))
    c.set_option("hap_precision", 0)
    g32, z32 = _hfwd(c, z["xp"], z["xh"])
    c.set_option("hap_precision", 1)
    g16, z16 = _hfwd(c, z["xp"], z["xh"])
    assert np.abs(g16 - z["gt"]).max() < PROB_ATOL and np.abs(z16 - z["zy"]).max() < PROB_ATOL
    assert np.abs(g16 - g32).max() < 2e-5
    rng = np.random.default_rng(4)
    for n in (1, 129, 300):
        xp = (rng.standard_normal((n, 105, 33)) * 300).astype(np.float32)
        xh = (rng.standard_normal((n, 105, 11)) * 300).astype(np.float32)
        gt, zy = _hfwd(c, xp, xh)
        ogt, ozy = oracle.hap_forward(ws, xp, xh, nthreads=8)
        assert np.abs(gt - ogt).max() < PROB_ATOL and np.abs(zy - ozy).max() < PROB_ATOL, n
    # count-valued features up to several thousand (beyond fp16's exact integers): the lo half carries the rest
    pp = host.synth_hap_planes(77, 64, 60, 180, 33); ph = host.synth_hap_planes(78, 64, 60, 180, 11)
    xp = oracle.hap_features_batch(*pp); xh = oracle.hap_features_batch(*ph)
    gt, zy = _hfwd(c, xp, xh)
    ogt, ozy = oracle.hap_forward(ws, xp, xh, nthreads=8)
    assert np.abs(gt - ogt).max() < PROB_ATOL
    # features beyond the fp16 range saturate (both halves) instead of turning into infinities / NaNs
    xb = xp[:4].copy(); xb[0, 3, 5] = 1.0e5; xb[1, 40, 16] = -3.0e6; xb[2, :, 0] = 7.0e4
    gt, zy = _hfwd(c, xb, xh[:4])
    assert np.isfinite(gt).all() and np.isfinite(zy).all() and np.allclose(gt.sum(1), 1.0, atol=1e-5)
    c.close()


def test_hap_forward_bf16x3_mode(gpu_ctx):
    """hap_precision 2: weights as three bf16 planes, fp32 activations split on their way into LDS, six bf16 MFMAs per product, fp32
    accumulation: the full fp32 operand width, so (unlike f16x3) count-valued features of several thousand and features far beyond the
    fp16 range keep the 1e-4 contract against the oracle with no saturation"""
    from nanosnp_amd import _lib
    from oracle import oracle
    from tests.helpers import PROB_ATOL, seeded_hap_weights
    ws = seeded_hap_weights(12, H=256)
    c = _lib.Context(0)
    c.hap_load_weights(ws)
    z = np.load(golden("hap_fwd_h256.npz"))
    g32, z32 = _hfwd(c, z["xp"], z["xh"])
    c.set_option("hap_precision", 2)
    g3, z3 = _hfwd(c, z["xp"], z["xh"])
    assert np.abs(g3 - z["gt"]).max() < PROB_ATOL and np.abs(z3 - z["zy"]).max() < PROB_ATOL
    assert np.abs(g3 - g32).max() < 2e-6 and np.abs(z3 - z32).max() < 2e-6
    rng = np.random.default_rng(4)
    for n in (1, 129, 300):
        xp = (rng.standard_normal((n, 105, 33)) * 300).astype(np.float32)
        xh = (rng.standard_normal((n, 105, 11)) * 300).astype(np.float32)
        gt, zy = _hfwd(c, xp, xh)
        ogt, ozy = oracle.hap_forward(ws, xp, xh, nthreads=8)
        assert np.abs(gt - ogt).max() < PROB_ATOL and np.abs(zy - ozy).max() < PROB_ATOL, n
    pp = host.synth_hap_planes(77, 64, 60, 180, 33); ph = host.synth_hap_planes(78, 64, 60, 180, 11)
    xp = oracle.hap_features_batch(*pp); xh = oracle.hap_features_batch(*ph)
    xb = xp.copy(); xb[0, 3, 5] = 1.0e5; xb[1, 40, 16] = -3.0e6; xb[2, :, 0] = 7.0e4          # beyond the fp16 range: carried exactly
    gt, zy = _hfwd(c, xb, xh)
    ogt, ozy = oracle.hap_forward(ws, xb, xh, nthreads=8)
    assert np.isfinite(gt).all() and np.abs(gt - ogt).max() < PROB_ATOL and np.abs(zy - ozy).max() < PROB_ATOL
    assert np.array_equal(gt, _hfwd(c, xb, xh)[0])                                             # run-to-run deterministic
    c.close()


def test_hap_arrange_reads_vs_the_reference_function(gpu_ctx):
    """nsnp_hap_arrange_reads on the read matrices of tests/golden/hap_arrange.npz against what the reference's
    single_group_pileup_haplotype_feature returned for them (create_pileup_haplotype.py:140-207)"""
    import torch
    from tests.test_oracle_golden import _check_arranged_against_reference

    def arrange(seq, bq, mq, hap, D):
        outs = gpu_ctx.hap_arrange_reads(*[torch.from_numpy(a[None]).cuda() for a in (seq, bq, mq, hap)], D)
        torch.cuda.synchronize()
        return tuple(o[0].cpu().numpy() for o in outs[:4]) + (int(outs[4][0].item()),)
    _check_arranged_against_reference(arrange, np.load(golden("hap_arrange.npz")))


def test_reference_style_haplotype_model_interface():
    """nanosnp_amd.haplotype_model.LSTMNetwork mirrors model_dev.LSTMNetwork as predict_dev.py:35-39,69-71 uses it"""
    import torch
    from nanosnp_amd import _lib
    from nanosnp_amd.haplotype_model import LSTMNetwork, state_dict_keys
    from tests.helpers import PROB_ATOL, hap_weight_names, seeded_hap_weights
    assert state_dict_keys() == hap_weight_names()
    cfg = {"model": {"pileup_dim": 105, "haplotype_dim": 105, "pileup_length": 33, "haplotype_length": 11, "hidden_size": 256,
                     "lstm_layers": 3, "gt_num_class": 10, "zy_num_class": 3, "dropout": 0.1}}
    m = LSTMNetwork(cfg).to("cuda")
    m.load_state_dict({k: torch.from_numpy(w) for k, w in zip(hap_weight_names(), seeded_hap_weights(12, H=256))})
    m.eval()
    z = np.load(golden("hap_fwd_h256.npz"))
    x_pileup = torch.from_numpy(z["xp"]).type(torch.FloatTensor).to("cuda")            # predict_dev.py:35-36
    x_haplotype = torch.from_numpy(z["xh"]).type(torch.FloatTensor).to("cuda")
    gt, zy = m.predict(x_pileup, x_haplotype)
    assert np.abs(gt.cpu().numpy() - z["gt"]).max() < PROB_ATOL and np.abs(zy.cpu().numpy() - z["zy"]).max() < PROB_ATOL
    with pytest.raises(Exception):
        LSTMNetwork(cfg).predict(x_pileup, x_haplotype)                               # weights not loaded
    with pytest.raises(_lib.NanoSNPError):
        LSTMNetwork({"model": dict(cfg["model"], hidden_size=100)})
    with pytest.raises(KeyError):
        LSTMNetwork(cfg).load_state_dict({})


def test_features_and_forward_are_capturable_in_a_hip_graph(hap_model):
    """nsnp_hap_load_weights reserves the forward's workspace, so nsnp_hap_features + nsnp_hap_forward neither allocate nor
    synchronise: the pair can be captured once and replayed (round 2 allocated 0.8 GB inside the first forward)"""
    import ctypes
    import torch
    from nanosnp_amd import _lib
    ctx, _ = hap_model
    n = 300
    pp = host.synth_hap_planes(7001, n, 30, 90, 33); ph = host.synth_hap_planes(7002, n, 30, 90, 11)
    dp = [torch.from_numpy(a).cuda() for a in pp]; dh = [torch.from_numpy(a).cuda() for a in ph]
    xp = ctx.hap_features(*dp); xh = ctx.hap_features(*dh)
    gt, zy = ctx.hap_forward(xp, xh)                             # eager reference
    want = (gt.clone(), zy.clone())
    torch.cuda.synchronize()
    P, lib = ctypes.c_void_p, _lib.load()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            sp = P(torch.cuda.current_stream().cuda_stream)
            rc = 0
            for pl, out, L in ((dp, xp, 33), (dh, xh, 11)):
                rc = rc or lib.nsnp_hap_features(ctx.handle, *[P(t.data_ptr()) for t in pl], n, 90, L, P(out.data_ptr()), sp)
            rc = rc or lib.nsnp_hap_forward(ctx.handle, P(xp.data_ptr()), P(xh.data_ptr()), n, P(gt.data_ptr()), P(zy.data_ptr()), sp)
            assert rc == 0
    for _ in range(2):
        gt.zero_(); zy.zero_(); xp.zero_(); xh.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(gt, want[0]) and torch.equal(zy, want[1])


def test_pass_size_option_multi_pass_ragged():
    """"hap_pass_sites": N larger than the internal pass and not a multiple of it or of the 128-site tile - the passes see the same
    arithmetic, so the result does not depend on the pass size (bit for bit); bad values are refused; N = 0 is a no-op"""
    import torch
    from nanosnp_amd import _lib
    from nanosnp_amd._lib import NanoSNPError
    from tests.helpers import seeded_hap_weights
    c = _lib.Context(0)
    c.hap_load_weights(seeded_hap_weights(12, H=256))
    n = 700
    rng = np.random.default_rng(3)
    xp = torch.from_numpy((rng.standard_normal((n, 105, 33)) * 20).astype(np.float32)).cuda()
    xh = torch.from_numpy((rng.standard_normal((n, 105, 11)) * 20).astype(np.float32)).cuda()
    ref = c.hap_forward(xp, xh)
    for ps in (128, 256, 640, 1024):
        c.set_option("hap_pass_sites", ps)
        got = c.hap_forward(xp, xh)
        torch.cuda.synchronize()
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), ps
    for bad in (0, 100, 129, 1 << 20):
        with pytest.raises(NanoSNPError):
            c.set_option("hap_pass_sites", bad)
    e = c.hap_forward(xp[:0], xh[:0])
    assert e[0].shape == (0, 10) and e[1].shape == (0, 3)
    c.close()
