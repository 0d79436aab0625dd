"""PileupModel forward on the GPU (through the C ABI) vs the reference goldens and the oracle.
Tolerance: 1e-4 absolute on probabilities (BASELINE.json north_star); measured ~5e-7."""
import numpy as np
import pytest

from tests.helpers import PROB_ATOL, golden
from nanosnp_amd.fixtures import DATA

WEIGHTS_NPZ = __import__("os").path.join(DATA, "ont_pileup_weights.npz")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[2, 1, 0], ids=["bf16x3", "f16x3", "fp32"])
def model(request, gpu_ctx, pileup_weights):
    """the tests of this module that take `model` run in all three arithmetic modes: exact fp32 MFMA (library default), bf16x3
    (three bf16 terms per operand = the full fp32 significand, six bf16 MFMAs per product) and the opt-in f16x3"""
    gpu_ctx.pileup_load_weights(pileup_weights)
    gpu_ctx.set_option("pileup_precision", request.param)
    gpu_ctx.test_precision = request.param
    yield gpu_ctx
    gpu_ctx.set_option("pileup_precision", 0)


def _fwd(ctx, x_np):
    import torch
    x = torch.from_numpy(np.ascontiguousarray(x_np, dtype=np.int32)).cuda()
    gt, zy = ctx.pileup_forward(x)
    torch.cuda.synchronize()
    return gt.cpu().numpy(), zy.cpu().numpy()


# the three checkpoints PileupModel/models/ ships (one architecture: config/ont_pileup.yaml = config/hg001_mix_without_balance.yaml:6-20):
# (fixture with the reference's outputs, fixture with the weights) - ont_pileup keeps its weights in a file of their own
CHECKPOINTS = {"ont_pileup": ("pileup_fwd.npz", None),
               "hg001_e13": ("pileup_fwd_hg001_e13.npz", "pileup_fwd_hg001_e13.npz"),
               "hg001_e186": ("pileup_fwd_hg001_e186.npz", "pileup_fwd_hg001_e186.npz")}


@pytest.mark.parametrize("ckpt", list(CHECKPOINTS))
def test_golden_outputs_of_the_reference_model(model, pileup_weights, ckpt):
    """LSTMNetwork.predict of the reference with every shipped checkpoint on the same 256 inputs, in all three arithmetics"""
    from tests.helpers import load_pileup_weights
    out_file, w_file = CHECKPOINTS[ckpt]
    z = np.load(golden(out_file))
    x = np.load(golden("pileup_fwd.npz"))["x"]
    model.pileup_load_weights(load_pileup_weights(golden(w_file) if w_file else None))
    try:
        gt, zy = _fwd(model, x)
    finally:
        model.pileup_load_weights(pileup_weights)
    assert np.abs(gt - z["gt"]).max() < PROB_ATOL
    assert np.abs(zy - z["zy"]).max() < PROB_ATOL
    assert np.array_equal(gt.argmax(1), z["gt"].argmax(1))
    assert np.array_equal(zy.argmax(1), z["zy"].argmax(1))
    assert np.allclose(gt.sum(1), 1, atol=1e-5) and np.allclose(zy.sum(1), 1, atol=1e-5)


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 127, 128, 129, 1000])
def test_ragged_batch_sizes_vs_oracle(model, pileup_weights, n):
    from oracle import oracle
    rng = np.random.default_rng(n)
    x = (rng.integers(0, 40, (n, 33, 18)) * (rng.random((n, 33, 18)) < 0.4)).astype(np.int32)
    x[:, :, 0] = -rng.integers(0, 60, (n, 33))
    gt, zy = _fwd(model, x)
    assert gt.shape == (n, 21) and zy.shape == (n, 3)
    if n:
        ogt, ozy = oracle.pileup_forward(pileup_weights, x, nthreads=4)
        assert np.abs(gt - ogt).max() < PROB_ATOL and np.abs(zy - ozy).max() < PROB_ATOL


def test_extreme_counts(model, pileup_weights):
    """depth-144 saturation, all-zero windows, large negatives: gates saturate, nothing overflows"""
    from oracle import oracle
    x = np.zeros((6, 33, 18), np.int32)
    x[1] = 144; x[2] = -144; x[3, :, ::2] = 10000; x[4, 16] = -100000; x[5, ::3] = 2**20
    gt, zy = _fwd(model, x)
    assert np.isfinite(gt).all() and np.isfinite(zy).all()
    ogt, ozy = oracle.pileup_forward(pileup_weights, x)
    assert np.abs(gt - ogt).max() < PROB_ATOL and np.abs(zy - ozy).max() < PROB_ATOL


def test_sites_are_independent_and_chunking_is_invisible(model, pileup_weights):
    """Size-independent properties used at full batch sizes: a site's result does not depend on its
    batch neighbours or on the internal chunk size; repeated runs are bit-identical."""
    import torch
    from nanosnp_amd import _lib
    rng = np.random.default_rng(0)
    n = 4096 + 77
    x = torch.from_numpy((rng.integers(0, 50, (n, 33, 18)) - 10).astype(np.int32)).cuda()
    gt, zy = model.pileup_forward(x)
    gt2, zy2 = model.pileup_forward(x)
    assert torch.equal(gt, gt2) and torch.equal(zy, zy2)
    perm = torch.randperm(n, device="cuda")
    gtp, zyp = model.pileup_forward(x[perm].contiguous())
    assert torch.equal(gtp, gt[perm]) and torch.equal(zyp, zy[perm])
    small = _lib.Context(0, chunk_sites=1000)        # forces 5 internal chunks
    small.pileup_load_weights(pileup_weights)
    small.set_option("pileup_precision", model.test_precision)
    gts, zys = small.pileup_forward(x)
    assert torch.equal(gts, gt) and torch.equal(zys, zy)
    sub, _ = model.pileup_forward(x[100:133].contiguous())
    assert torch.equal(sub, gt[100:133])
    small.close()


def test_full_batch_4096_checksum_vs_oracle_sample(model, pileup_weights):
    """BASELINE config 2 batch size: every 16th site of a 4096 batch against the oracle."""
    import torch
    from nanosnp_amd import host
    from oracle import oracle
    n = 4096
    cols = host.synth_columns(20260001, n * 33, coverage=30, window=33)
    counts, _, _ = oracle.encode_columns(cols.bases, cols.col_off, cols.ref)
    x = counts.reshape(n, 33, 18)
    gt, zy = _fwd(model, x)
    idx = np.arange(0, n, 16)
    ogt, ozy = oracle.pileup_forward(pileup_weights, x[idx], nthreads=8)
    assert np.abs(gt[idx] - ogt).max() < PROB_ATOL and np.abs(zy[idx] - ozy).max() < PROB_ATOL
    assert np.allclose(gt.sum(1), 1, atol=1e-5)


def test_forward_windows_reads_the_count_matrix_in_place(model):
    import torch
    rng = np.random.default_rng(3)
    m = 5000
    counts = torch.from_numpy(rng.integers(-30, 40, (m, 18)).astype(np.int32)).cuda()
    centers = torch.from_numpy(np.sort(rng.choice(np.arange(16, m - 16), 300, replace=False)).astype(np.int64)).cuda()
    x = model.pileup_gather_windows(counts, centers)
    ref = torch.stack([counts[c - 16:c + 17] for c in centers.tolist()])
    assert torch.equal(x, ref)
    g1, z1 = model.pileup_forward(x)
    g2, z2 = model.pileup_forward_windows(counts, centers)
    assert torch.equal(g1, g2) and torch.equal(z1, z2)


def test_postprocess_matches_predict_py(model):
    """argmax / max / depth of PileupModel/predict.py:54-65"""
    import torch
    z = np.load(golden("pileup_fwd.npz"))
    x = torch.from_numpy(z["x"].astype(np.int32)).cuda()
    gt, zy = model.pileup_forward(x)
    ga, za, gm, zm, depth = model.pileup_postprocess(gt, zy, x)
    gtn, zyn = gt.cpu().numpy(), zy.cpu().numpy()
    assert np.array_equal(ga.cpu().numpy(), np.argmax(gtn, 1)) and np.array_equal(za.cpu().numpy(), np.argmax(zyn, 1))
    assert np.array_equal(gm.cpu().numpy(), np.max(gtn, 1)) and np.array_equal(zm.cpu().numpy(), np.max(zyn, 1))
    cov = z["x"].astype(np.int64)[:, 16][:, [0, 1, 2, 3, 9, 10, 11, 12]]
    want = np.array([-1 * c[np.where(c < 0)].sum() for c in cov])
    assert np.array_equal(depth.cpu().numpy(), want)


def test_reference_style_interface(pileup_weights):
    """nanosnp_amd.pileup_model.LSTMNetwork mirrors PileupModel/model.py + predict.py:208-214"""
    import torch
    from nanosnp_amd.pileup_model import LSTMNetwork
    m = LSTMNetwork.from_npz(WEIGHTS_NPZ).to("cuda").eval()
    z = np.load(golden("pileup_fwd.npz"))
    feature_tensor = torch.from_numpy(z["x"].astype(np.int32)).type(torch.FloatTensor).to("cuda")   # predict.py:49
    gt, zy = m.predict(feature_tensor)
    assert np.abs(gt.cpu().numpy() - z["gt"]).max() < PROB_ATOL
    with pytest.raises(Exception):
        LSTMNetwork({"feature_dim": 18, "gt_num_class": 21, "zy_num_class": 3,
                     "enc": {"type": "lstm", "hidden_size": 128, "output_size": 128, "n_layers": 2, "bidirectional": True},
                     "joint": {"inner_size": 256}})
    with pytest.raises(Exception):
        LSTMNetwork().predict(feature_tensor)      # weights not loaded


def test_workgroup_shape_does_not_change_results(model, pileup_weights):
    """the launcher picks 1/2/4/8 waves per recurrence workgroup from the batch size; all give
    bit-identical probabilities"""
    import torch
    from nanosnp_amd import _lib
    rng = np.random.default_rng(11)
    x = torch.from_numpy((rng.integers(0, 50, (777, 33, 18)) - 10).astype(np.int32)).cuda()
    ref_gt, ref_zy = model.pileup_forward(x)
    c = _lib.Context(0)
    c.pileup_load_weights(pileup_weights)
    c.set_option("pileup_precision", model.test_precision)
    for w in (1, 2, 4, 8):
        c.set_option("recurrence_waves", w)
        gt, zy = c.pileup_forward(x)
        assert torch.equal(gt, ref_gt) and torch.equal(zy, ref_zy), w
    for bad in (3, 6):          # 6 had a build for one kernel only: the others would have skipped 5/6 of the sites
        with pytest.raises(_lib.NanoSNPError):
            c.set_option("recurrence_waves", bad)
    c.close()


def test_fp32_register_stationary_kernels_equal_the_lds_image_kernels_bit_for_bit(pileup_weights):
    """exact fp32: the register-stationary layer-0 / fused layer-1 kernels (default) issue, per accumulator, the same
    k-ordered MFMA chain and the same cell expressions as the LDS-image kernels K1 / K2 + K3: identical bits for every
    workgroup shape, ragged sizes, either kernel in front of the other, and the windows entry point"""
    import torch
    from nanosnp_amd import _lib
    from oracle import oracle
    c = _lib.Context(0)
    c.pileup_load_weights(pileup_weights)
    rng = np.random.default_rng(41)
    xn = (rng.integers(0, 60, (1003, 33, 18)) - 15).astype(np.int32)
    xn[5, 3] = 5000; xn[700, 16, 2] = -70000; xn[300, 10, 4] = 2**24 + 3          # the fp32 mode carries the whole int32 -> float cast
    x = torch.from_numpy(xn).cuda()
    c.set_option("l0_register_stationary", 0); c.set_option("l1_register_stationary", 0); c.set_option("head_split", 0)
    old = c.pileup_forward(x)
    c.set_option("head_split", 1)                     # heads with the output tiles split over 8 waves: same chains
    got = c.pileup_forward(x)
    assert torch.equal(got[0], old[0]) and torch.equal(got[1], old[1])
    for st in (0, 1):                                 # waves 4-7 of the eight-wave layer-1 kernel running a group's next input part early
        c.set_option("l1_stagger", st); c.set_option("l1_register_stationary", 2)
        got = c.pileup_forward(x)
        assert torch.equal(got[0], old[0]) and torch.equal(got[1], old[1]), st
    c.set_option("l1_stagger", 0)
    c.set_option("l0_input_weights_in_lds", 1); c.set_option("l0_register_stationary", 1); c.set_option("l0_site_groups", 1)
    got = c.pileup_forward(x)                         # layer 0 with its input-part fragments in LDS (four workgroups per SIMD set)
    assert torch.equal(got[0], old[0]) and torch.equal(got[1], old[1])
    c.set_option("l0_input_weights_in_lds", 0)
    for l0, l1 in ((1, 1), (1, 2), (1, 0), (0, 1), (0, 2)):      # layer 1: 1 = four waves x four tiles (default), 2 = eight waves x two tiles
        c.set_option("l0_register_stationary", l0); c.set_option("l1_register_stationary", l1)
        for g0 in ((0, 1, 2, 4) if l0 else (0,)):
            for g1 in ({0: (0,), 1: (0, 1, 2), 2: (0, 2, 4)}[l1]):
                c.set_option("l0_site_groups", g0); c.set_option("l1_site_groups", g1)
                got = c.pileup_forward(x)
                assert torch.equal(got[0], old[0]) and torch.equal(got[1], old[1]), (l0, l1, g0, g1)
    c.set_option("l0_register_stationary", 1); c.set_option("l1_register_stationary", 1)
    c.set_option("l0_site_groups", 0); c.set_option("l1_site_groups", 0)
    for n in (1, 15, 16, 17, 31, 33, 63, 64, 65, 200):
        gn, zn = c.pileup_forward(x[:n].contiguous())
        assert torch.equal(gn, old[0][:n]) and torch.equal(zn, old[1][:n]), n
    og, oz = oracle.pileup_forward(pileup_weights, xn[:320], nthreads=8)
    assert np.abs(old[0][:320].cpu().numpy() - og).max() < PROB_ATOL and np.abs(old[1][:320].cpu().numpy() - oz).max() < PROB_ATOL
    c.close()


def test_f16x3_precision_mode(pileup_weights):
    """every product as 3 fp16 MFMAs with fp32 accumulation (pileup_forward_f16x3.hip): same goldens,
    same 1e-4 tolerance; measured ~1e-6"""
    import torch
    from nanosnp_amd import _lib
    c = _lib.Context(0)
    c.pileup_load_weights(pileup_weights)
    z = np.load(golden("pileup_fwd.npz"))
    x = torch.from_numpy(z["x"].astype(np.int32)).cuda()
    g32, z32 = c.pileup_forward(x)
    c.set_option("pileup_precision", 1)
    g16, z16 = c.pileup_forward(x)
    torch.cuda.synchronize()
    d_gold = max(np.abs(g16.cpu().numpy() - z["gt"]).max(), np.abs(z16.cpu().numpy() - z["zy"]).max())
    d_32 = max((g16 - g32).abs().max().item(), (z16 - z32).abs().max().item())
    print("f16x3 vs golden", d_gold, "vs fp32 path", d_32)
    assert d_gold < PROB_ATOL and d_32 < 2e-5
    assert np.array_equal(g16.cpu().numpy().argmax(1), z["gt"].argmax(1))
    # deep counts (> 2048: the fp16 hi part is no longer exact, the lo part carries the rest) and extremes
    from oracle import oracle
    xe = np.zeros((8, 33, 18), np.int32)
    xe[1] = 144; xe[2] = -144; xe[3, :, ::2] = 10000; xe[4, 16] = -60000; xe[5, ::3] = 4099; xe[6] = 2049; xe[7, :, 1] = 33001
    ge, ze = c.pileup_forward(torch.from_numpy(xe).cuda())
    oge, oze = oracle.pileup_forward(pileup_weights, xe)
    assert np.isfinite(ge.cpu().numpy()).all()
    assert np.abs(ge.cpu().numpy() - oge).max() < PROB_ATOL and np.abs(ze.cpu().numpy() - oze).max() < PROB_ATOL
    # ragged sizes, wave shapes, windows entry point
    rng = np.random.default_rng(5)
    xr = torch.from_numpy((rng.integers(0, 50, (1000, 33, 18)) - 10).astype(np.int32)).cuda()
    ref = c.pileup_forward(xr)
    for w in (1, 2, 4, 8):
        # (the exact-fp32 path is bit-identical across workgroup shapes; here the compiler's code for the
        #  8/4-wave variants differs from the 2/1-wave ones by one ulp in a few rows)
        c.set_option("recurrence_waves", w)
        got = c.pileup_forward(xr)
        assert (got[0] - ref[0]).abs().max().item() < 5e-7 and (got[1] - ref[1]).abs().max().item() < 5e-7, w
        again = c.pileup_forward(xr)
        assert torch.equal(got[0], again[0]) and torch.equal(got[1], again[1])      # run-to-run deterministic
    c.set_option("recurrence_waves", 0)
    for n in (1, 17, 129):
        gn, zn = c.pileup_forward(xr[:n].contiguous())
        assert torch.equal(gn, ref[0][:n])
    c.close()


def test_fused_layer1_kernel_equals_the_two_kernel_path(pileup_weights):
    """f16x3: projection fused into the layer-1 recurrence (no Xp1 round trip) computes the same sums in
    the same order as projection kernel + recurrence kernel"""
    import torch
    from nanosnp_amd import _lib
    c = _lib.Context(0)
    c.pileup_load_weights(pileup_weights)
    c.set_option("pileup_precision", 1)
    c.set_option("l1_register_stationary", 0)          # the LDS-image / ring kernel (K23); K23r has its own test below
    rng = np.random.default_rng(9)
    for n in (1, 191, 192, 193, 1000, 5000):
        x = torch.from_numpy((rng.integers(0, 50, (n, 33, 18)) - 10).astype(np.int32)).cuda()
        c.set_option("fused_l1", 0)
        g0, z0 = c.pileup_forward(x)
        c.set_option("fused_l1", 1)
        g1, z1 = c.pileup_forward(x)
        torch.cuda.synchronize()
        assert (g0 - g1).abs().max().item() < 5e-7 and (z0 - z1).abs().max().item() < 5e-7, n
    z = np.load(golden("pileup_fwd.npz"))
    gt, zy = c.pileup_forward(torch.from_numpy(z["x"].astype(np.int32)).cuda())
    assert np.abs(gt.cpu().numpy() - z["gt"]).max() < PROB_ATOL and np.abs(zy.cpu().numpy() - z["zy"]).max() < PROB_ATOL
    c.close()


def test_register_stationary_layer0_kernel(pileup_weights):
    """f16x3 layer 0: the register-stationary kernel (weights in VGPRs, h exchanged through LDS; default) against the
    LDS-image kernel and the oracle, for every workgroup shape, ragged sizes and counts beyond the exact fp16 range"""
    import torch
    from nanosnp_amd import _lib
    from oracle import oracle
    c = _lib.Context(0)
    c.pileup_load_weights(pileup_weights)
    c.set_option("pileup_precision", 1)
    rng = np.random.default_rng(21)
    xn = (rng.integers(0, 50, (1003, 33, 18)) - 10).astype(np.int32)
    xn[5, 3] = 5000; xn[700, 16, 2] = -70000; xn[1002] = 2049          # steps where the lo part of x is needed
    xn[300, 10, 4] = 200000; xn[301, 20] = -131008                      # beyond what two fp16 halves carry: saturates
    x = torch.from_numpy(xn).cuda()
    c.set_option("l0_register_stationary", 0)
    old = c.pileup_forward(x)
    c.set_option("l0_register_stationary", 1)
    ref = None
    for g in (0, 1, 2, 4):
        c.set_option("l0_site_groups", g)
        got = c.pileup_forward(x)
        if ref is None:
            ref = got
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), g      # a site never depends on its group
    with pytest.raises(_lib.NanoSNPError):
        c.set_option("l0_site_groups", 3)
    # gate rows are pre-scaled by log2(e) before the fp16 split, so the two kernels differ by rounding only
    assert (ref[0] - old[0]).abs().max().item() < 5e-6 and (ref[1] - old[1]).abs().max().item() < 5e-6
    og, oz = oracle.pileup_forward(pileup_weights, xn[:302], nthreads=8)
    assert np.abs(ref[0][:302].cpu().numpy() - og).max() < PROB_ATOL and np.abs(ref[1][:302].cpu().numpy() - oz).max() < PROB_ATOL
    og, oz = oracle.pileup_forward(pileup_weights, xn[-3:])
    assert np.abs(ref[0][-3:].cpu().numpy() - og).max() < PROB_ATOL
    assert torch.isfinite(ref[0]).all() and torch.isfinite(old[0]).all()
    og, oz = oracle.pileup_forward(pileup_weights, xn[698:702])
    assert np.abs(ref[0][698:702].cpu().numpy() - og).max() < PROB_ATOL and np.abs(old[0][698:702].cpu().numpy() - og).max() < PROB_ATOL
    for n in (1, 15, 63, 65, 300):
        gn, zn = c.pileup_forward(x[:n].contiguous())
        assert torch.equal(gn, ref[0][:n]) and torch.equal(zn, ref[1][:n]), n
    c.close()


def test_register_stationary_layer1_kernel(pileup_weights):
    """f16x3 layer 1: the register-stationary kernel (default) against the LDS-image / ring kernel, the unfused two-kernel
    path and the oracle; ragged sizes; either layer-0 kernel in front of it"""
    import torch
    from nanosnp_amd import _lib
    from oracle import oracle
    c = _lib.Context(0)
    c.pileup_load_weights(pileup_weights)
    c.set_option("pileup_precision", 1)
    rng = np.random.default_rng(31)
    xn = (rng.integers(0, 60, (777, 33, 18)) - 15).astype(np.int32)
    x = torch.from_numpy(xn).cuda()
    ref = c.pileup_forward(x)                                   # defaults: both register-stationary kernels
    c.set_option("l1_register_stationary", 0)
    ring = c.pileup_forward(x)
    c.set_option("fused_l1", 0)
    two = c.pileup_forward(x)
    c.set_option("fused_l1", 1); c.set_option("l1_register_stationary", 1); c.set_option("l0_register_stationary", 0)
    mixed = c.pileup_forward(x)
    c.set_option("l0_register_stationary", 1)
    for other in (ring, two, mixed):
        assert (ref[0] - other[0]).abs().max().item() < 5e-6 and (ref[1] - other[1]).abs().max().item() < 5e-6
    og, oz = oracle.pileup_forward(pileup_weights, xn[:256], nthreads=8)
    assert np.abs(ref[0][:256].cpu().numpy() - og).max() < PROB_ATOL and np.abs(ref[1][:256].cpu().numpy() - oz).max() < PROB_ATOL
    again = c.pileup_forward(x)
    assert torch.equal(again[0], ref[0]) and torch.equal(again[1], ref[1])
    for g in (2, 4, 0):                                         # eight-wave kernel with 32 / 64 sites per workgroup / the default kernel
        c.set_option("l1_site_groups", g)
        got = c.pileup_forward(x)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), g
    c.set_option("l1_register_stationary", 2)                 # the eight-wave x two-tile kernel against the default four x four
    got = c.pileup_forward(x)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    c.set_option("l1_register_stationary", 1)
    with pytest.raises(_lib.NanoSNPError):
        c.set_option("l1_site_groups", 3)
    for n in (1, 31, 32, 33, 63, 64, 65, 200):
        gn, zn = c.pileup_forward(x[:n].contiguous())
        assert torch.equal(gn, ref[0][:n]) and torch.equal(zn, ref[1][:n]), n
    c.close()


def test_forward_is_capturable_in_a_hip_graph(pileup_weights):
    """after one warm-up call (workspace, LDS attributes) encode + forward + postprocess make no allocation and no
    synchronisation: the sequence can be captured once and replayed (include/nanosnp.h: nsnp_ctx_reserve)"""
    import torch
    from nanosnp_amd import _lib, host
    c = _lib.Context(0, chunk_sites=4096)
    c.pileup_load_weights(pileup_weights)
    cols = host.synth_columns(5, 33 * 1024, coverage=30, window=33)
    b = torch.from_numpy(cols.bases).cuda(); off = torch.from_numpy(cols.col_off).cuda(); rf = torch.from_numpy(cols.ref).cuda()
    centers = (torch.arange(1024, dtype=torch.int64, device="cuda") * 33 + 16)
    counts, depth, flags = c.pileup_encode_columns(b, off, rf)
    gt = torch.empty((1024, 21), device="cuda"); zy = torch.empty((1024, 3), device="cuda")
    c.pileup_forward_windows(counts, centers, gt, zy)            # warm-up
    want = (gt.clone(), zy.clone())
    torch.cuda.synchronize()
    P = __import__("ctypes").c_void_p
    lib = _lib.load()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            sp = P(torch.cuda.current_stream().cuda_stream)
            rc = lib.nsnp_pileup_encode_columns(c.handle, P(b.data_ptr()), P(off.data_ptr()), P(rf.data_ptr()), 33 * 1024,
                                                __import__("ctypes").c_double(0.12), 6, P(counts.data_ptr()), P(depth.data_ptr()),
                                                P(flags.data_ptr()), sp)
            rc = rc or lib.nsnp_pileup_forward_windows(c.handle, P(counts.data_ptr()), P(centers.data_ptr()), 1024, P(gt.data_ptr()), P(zy.data_ptr()), sp)
            assert rc == 0
    for _ in range(3):
        gt.zero_(); zy.zero_(); counts.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(gt, want[0]) and torch.equal(zy, want[1])
    c.close()


@pytest.mark.parametrize("opts", [{}, {"pileup_precision": 1}, {"pileup_precision": 2}, {"head_split": 0}], ids=["fp32", "f16x3", "bf16x3", "fp32-one-wave-heads"])
def test_fused_forward_and_argmax_equals_the_two_calls(pileup_weights, opts):
    """nsnp_pileup_forward_windows_calls = nsnp_pileup_forward_windows + nsnp_pileup_postprocess (predict.py:51-57), bit for bit:
    the fp32 heads kernel writes argmax / max itself, the other paths run the two kernels back to back; ragged N, ties included"""
    import torch
    from nanosnp_amd import _lib, host
    c = _lib.Context(0, chunk_sites=2048)
    c.pileup_load_weights(pileup_weights)
    for k, v in opts.items():
        c.set_option(k, v)
    n = 4099
    cols = host.synth_columns(77, 33 * n, coverage=30, window=33)
    counts, _, _ = c.pileup_encode_columns(torch.from_numpy(cols.bases).cuda(), torch.from_numpy(cols.col_off).cuda(), torch.from_numpy(cols.ref).cuda())
    counts[33 * 5:33 * 6] = 0                                       # an all-zero window: many equal probabilities
    centers = torch.arange(n, dtype=torch.int64, device="cuda") * 33 + 16
    gt, zy = c.pileup_forward_windows(counts, centers)
    ga, za, gm, zm, _ = c.pileup_postprocess(gt, zy)
    gt2, zy2, ga2, za2, gm2, zm2 = c.pileup_forward_windows_calls(counts, centers)
    torch.cuda.synchronize()
    assert torch.equal(gt, gt2) and torch.equal(zy, zy2)
    assert torch.equal(ga, ga2) and torch.equal(za, za2) and torch.equal(gm, gm2) and torch.equal(zm, zm2)
    assert torch.equal(ga2.long(), gt2.argmax(1)) and torch.equal(gm2, gt2.max(1).values)
    # the same call with the four call arrays in PINNED HOST memory (slices at odd offsets): the kernel writes them over PCIe, no copy
    pin = [torch.full((n + 7,), 99, dtype=dt, pin_memory=True) for dt in (torch.uint8, torch.uint8, torch.float32, torch.float32)]
    gt3, zy3, *_ = c.pileup_forward_windows_calls(counts, centers, calls_out=tuple(t[3:3 + n] for t in pin))
    torch.cuda.synchronize()
    assert torch.equal(gt3, gt2) and torch.equal(zy3, zy2)
    for t, want in zip(pin, (ga2, za2, gm2, zm2)):
        assert torch.equal(t[3:3 + n], want.cpu()) and (t[:3] == 99).all() and (t[3 + n:] == 99).all()
    with pytest.raises(_lib.NanoSNPError):
        c.pileup_forward_windows_calls(counts, centers, calls_out=(torch.empty(n, dtype=torch.uint8),) * 2 + (torch.empty(n),) * 2)   # pageable host memory
    c.close()


def test_fused_call_with_no_sites_and_bad_arguments(pileup_weights):
    import ctypes
    import torch
    from nanosnp_amd import _lib
    c = _lib.Context(0)
    c.pileup_load_weights(pileup_weights)
    lib, P = _lib.load(), ctypes.c_void_p
    z = torch.zeros(8, device="cuda")
    assert lib.nsnp_pileup_forward_windows_calls(c.handle, None, None, 0, None, None, None, None, None, None, None) == 0
    assert lib.nsnp_pileup_forward_windows_calls(c.handle, P(z.data_ptr()), P(z.data_ptr()), 1, P(z.data_ptr()), P(z.data_ptr()), None, None,
                                                 None, None, None) == -1                # argmax / max outputs are required
    assert lib.nsnp_pileup_forward_windows_calls(None, None, None, 0, None, None, None, None, None, None, None) == -1
    c.close()


def test_bf16x3_mode_carries_the_full_fp32_operand_width(pileup_weights):
    """pileup_precision 2: every operand as three bf16 terms (8 + 8 + 8 significand bits, fp32 exponent range), six bf16 MFMAs per
    product, fp32 accumulation.  (a) its error against a FLOAT64 evaluation of the model (oracle.pileup_forward_f64) is the error of
    the exact-fp32 MFMA path itself - the two paths differ from each other by fp32 summation-order noise (2-3e-6 on 1 M windows:
    the fp32 kernels sit 1-2e-6 from float64 themselves), not by operand width; (b) every site-group shape of the two recurrence
    kernels gives the same bits; (c) counts of every split level - one bf16 (|x| <= 256), two (<= 65536), three - against the
    oracle at the 1e-4 contract, incl. magnitudes the f16x3 mode saturates at (> 131008)"""
    import torch
    from nanosnp_amd import _lib, host
    from oracle import oracle
    c2 = _lib.Context(0); c2.pileup_load_weights(pileup_weights); c2.set_option("pileup_precision", 2)
    c0 = _lib.Context(0); c0.pileup_load_weights(pileup_weights)
    n = 4096
    cols = host.synth_columns(20260002, n * 33, coverage=30, window=33)
    counts, _, _ = oracle.encode_columns(cols.bases, cols.col_off, cols.ref)
    rng = np.random.default_rng(5)
    xr = (rng.integers(0, 50, (1531, 33, 18)) - 10).astype(np.int32)
    for x_np in (counts.reshape(n, 33, 18), xr):
        x = torch.from_numpy(np.ascontiguousarray(x_np)).cuda()
        g2, z2 = c2.pileup_forward(x); g0, z0 = c0.pileup_forward(x)
        torch.cuda.synchronize()
        assert torch.isfinite(g2).all() and torch.isfinite(z2).all()
        # (a) against float64 on the first 1024 sites
        m = 1024
        g64, z64 = oracle.pileup_forward_f64(pileup_weights, x_np[:m])
        e2 = max(np.abs(g2[:m].cpu().numpy() - g64).max(), np.abs(z2[:m].cpu().numpy() - z64).max())
        e0 = max(np.abs(g0[:m].cpu().numpy() - g64).max(), np.abs(z0[:m].cpu().numpy() - z64).max())
        assert e2 <= max(1.5 * e0, 1e-6) and e2 < 3e-6, (e2, e0)
        assert max((g2 - g0).abs().max().item(), (z2 - z0).abs().max().item()) < 5e-6
        # (b) launch shapes
        for l0g, l1g in ((1, 1), (2, 2), (4, 4), (1, 4), (4, 1)):
            c2.set_option("l0_site_groups", l0g); c2.set_option("l1_site_groups", l1g)
            g, z = c2.pileup_forward(x)
            assert torch.equal(g, g2) and torch.equal(z, z2), (l0g, l1g)
        c2.set_option("l0_site_groups", 0); c2.set_option("l1_site_groups", 0)
    # (c) split levels of the input counts
    x = np.zeros((48, 33, 18), np.int32)
    x[:16] = rng.integers(-256, 257, (16, 33, 18))                      # one bf16 term
    x[16:32] = rng.integers(-300, 301, (16, 33, 18)); x[16:32, ::5, 3] = 65536; x[20, 7, 1] = -40000      # two terms
    x[32:] = rng.integers(-100, 101, (16, 33, 18)); x[33, 16, :4] = (1 << 20) + 3; x[40, 3, 9] = -16777215; x[41, 30, 2] = 200001
    xt = torch.from_numpy(x).cuda()
    g2, z2 = c2.pileup_forward(xt)
    torch.cuda.synchronize()
    og, oz = oracle.pileup_forward(pileup_weights, x, nthreads=8)
    assert np.abs(g2.cpu().numpy() - og).max() < PROB_ATOL and np.abs(z2.cpu().numpy() - oz).max() < PROB_ATOL
    # a 16-site group with a large count must not disturb its neighbours: sites are independent bit for bit
    sub, _ = c2.pileup_forward(xt[:16].contiguous())
    assert torch.equal(sub, g2[:16])
    c2.close(); c0.close()


@pytest.mark.parametrize("ckpt", ["ont_pileup", "hg001_e186"])
def test_bf16x3_split_levels_of_different_site_groups_in_one_workgroup(ckpt):
    """The bf16x3 layer-0 kernels run a step at the split level of the LARGEST count any site group of the WORKGROUP staged for it
    (1 term up to 256, 2 up to 65536, 3 beyond); a group below that level multiplies its own planes 1 and 2 as well, which must then be
    zeros.  (Round 4, found by tests/stress/b3_consistency.py: they held whatever an earlier step had left there - a workgroup with one
    large-count site gave wrong probabilities, by up to 0.08, for the sites of its OTHER group.  Coverage beyond 256x only.)
    8192 sites = the skewed two-group kernel, with large counts in the first group of some workgroups, the second group of others,
    both, at single steps and at all steps; every site must equal, bit for bit, its result in a 16-site batch of its own group (one
    group per workgroup), and the forced plain kernels with 2 and 4 groups per workgroup must agree; all against the oracle."""
    import torch
    from nanosnp_amd import _lib
    from oracle import oracle
    from tests.helpers import load_pileup_weights
    pileup_weights = load_pileup_weights(golden(CHECKPOINTS[ckpt][1]) if CHECKPOINTS[ckpt][1] else None)      # epoch 186 holds the largest weights shipped (max |W| 2.32)
    rng = np.random.default_rng(91)
    n = 8192
    x = (rng.integers(0, 60, (n, 33, 18)) - 12).astype(np.int32)
    x[5, 3, 2] = 70000                                   # workgroup 0: group A at one step (three terms) ...
    x[17] *= 300                                         # ... group B at every step (two terms)
    x[40, 10:20, 1] = 1000                               # workgroup 1: group A only, ten steps
    x[32 * 7 + 20, 32, 17] = -300                        # workgroup 7: group B only, last step
    x[32 * 100 + 3, 0, 0] = 1 << 22; x[32 * 100 + 19, 0, 0] = 257       # workgroup 100: both groups, different levels, first step
    x[4095, 16, 5] = 5000; x[4096, 16, 5] = -5000        # neighbours across workgroups
    xt = torch.from_numpy(x).cuda()
    c = _lib.Context(0); c.pileup_load_weights(pileup_weights); c.set_option("pileup_precision", 2)
    g, z = c.pileup_forward(xt); torch.cuda.synchronize()
    touched = sorted({s // 16 for s in (5, 17, 40, 32 * 7 + 20, 32 * 100 + 3, 32 * 100 + 19, 4095, 4096)} | {0, 1, 2, 3, 14, 15, 200, 201, 255, 256})
    for grp in touched:                                  # the group itself and the other group of its workgroup
        for gg in (grp, grp ^ 1):
            a = 16 * gg
            gs, zs = c.pileup_forward(xt[a:a + 16].contiguous())
            assert torch.equal(gs, g[a:a + 16]) and torch.equal(zs, z[a:a + 16]), gg
    for l0g in (1, 2, 4):
        c.set_option("l0_site_groups", l0g)
        g2, z2 = c.pileup_forward(xt)
        assert torch.equal(g2, g) and torch.equal(z2, z), l0g
    c.set_option("l0_site_groups", 0)
    m = 640                                              # workgroups 0 .. 19 against the oracle (incl. 0, 1, 7)
    og, oz = oracle.pileup_forward(pileup_weights, x[:m], nthreads=8)
    assert np.abs(g[:m].cpu().numpy() - og).max() < PROB_ATOL and np.abs(z[:m].cpu().numpy() - oz).max() < PROB_ATOL
    og, oz = oracle.pileup_forward(pileup_weights, x[3200:3232], nthreads=8)
    assert np.abs(g[3200:3232].cpu().numpy() - og).max() < PROB_ATOL and np.abs(z[3200:3232].cpu().numpy() - oz).max() < PROB_ATOL
    c.close()
