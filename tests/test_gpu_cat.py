"""Legacy CatModel forward (ResCRNN + percentage RNN) on the GPU vs the reference-module golden and the oracle."""
import numpy as np
import pytest

from tests.helpers import golden, seeded_cat_weights, synth_cat_groups

pytestmark = pytest.mark.gpu

PROB_ATOL = 1e-4        # BASELINE.json north_star tolerance on probabilities


@pytest.fixture(scope="module", params=[2, 1, 0], ids=["bf16x3", "f16x3", "fp32"])
def cat_model(request, gpu_ctx):
    """the forward tests run in all three modes: exact fp32 (library default), bf16x3 and the opt-in f16x3"""
    ws = seeded_cat_weights(21)
    gpu_ctx.cat_load_weights(ws)
    gpu_ctx.set_option("cat_precision", request.param)
    gpu_ctx.test_cat_precision = request.param
    yield gpu_ctx, ws
    gpu_ctx.set_option("cat_precision", 0)


def _fwd(ctx, g0, g1):
    import torch
    out = ctx.cat_forward(torch.from_numpy(np.ascontiguousarray(g0, dtype=np.float32)).cuda(),
                          torch.from_numpy(np.ascontiguousarray(g1, dtype=np.float32)).cuda())
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_cat_forward_golden(cat_model):
    ctx, _ = cat_model
    z = np.load(golden("cat_fwd.npz"))
    assert int(z["seed"]) == 21
    got = _fwd(ctx, z["g0"], z["g1"])
    assert np.abs(got - z["gt"]).max() < PROB_ATOL
    assert np.array_equal(got.argmax(1), z["gt"].argmax(1))


@pytest.mark.parametrize("prec", [0, 1, 2], ids=["fp32", "f16x3", "bf16x3"])
def test_cat_forward_large_golden_incl_edge_sites(prec):
    """cat_fwd_large.npz: 256 sites of the reference's CatModel.predict, incl. empty tags, one-read tags and saturated tensors"""
    from nanosnp_amd import _lib
    from tests.helpers import seeded_cat_weights
    z = np.load(golden("cat_fwd_large.npz"))
    c = _lib.Context(0)
    c.cat_load_weights(seeded_cat_weights(int(z["seed"])))
    c.set_option("cat_precision", prec)
    got = _fwd(c, z["g0"], z["g1"])
    assert np.isfinite(got).all() and np.abs(got - z["gt"]).max() < PROB_ATOL
    top2 = np.sort(z["gt"], 1)[:, -2:]
    assert np.all((got.argmax(1) == z["gt"].argmax(1)) | (top2[:, 1] - top2[:, 0] < 1e-3))
    c.close()


@pytest.mark.parametrize("N", [1, 127, 129, 300])
def test_cat_forward_vs_oracle(cat_model, N):
    from oracle import oracle
    ctx, ws = cat_model
    g0, g1 = synth_cat_groups(1000 + N, N)
    got = _fwd(ctx, g0, g1)
    want = oracle.cat_forward(ws, g0, g1, nthreads=8)
    assert got.shape == (N, 10)
    assert np.abs(got - want).max() < PROB_ATOL
    assert np.allclose(got.sum(1), 1.0, atol=1e-5)


@pytest.mark.parametrize("mode", [1, 2], ids=["f16x3", "bf16x3"])
def test_cat_forward_split_modes(cat_model, mode):
    """cat_precision = 1: every product as three fp16 MFMAs; 2: six bf16 MFMAs on three bf16 terms per operand (as hap_precision);
    fp32 accumulation, same goldens"""
    from oracle import oracle
    ctx, ws = cat_model
    z = np.load(golden("cat_fwd.npz"))
    g0, g1 = synth_cat_groups(555, 200)
    ctx.set_option("cat_precision", 0)
    ref32 = _fwd(ctx, g0, g1)
    ctx.set_option("cat_precision", mode)
    try:
        got = _fwd(ctx, z["g0"], z["g1"])
        assert np.abs(got - z["gt"]).max() < PROB_ATOL and np.array_equal(got.argmax(1), z["gt"].argmax(1))
        got = _fwd(ctx, g0, g1)
        d32 = np.abs(got - ref32).max()
        print("cat split mode", mode, "vs fp32 path", d32)
        assert d32 < 2e-5
        assert np.abs(got - oracle.cat_forward(ws, g0, g1, nthreads=8)).max() < PROB_ATOL
        assert np.array_equal(got, _fwd(ctx, g0, g1))            # run-to-run deterministic
    finally:
        ctx.set_option("cat_precision", ctx.test_cat_precision)


def test_cat_forward_multi_chunk_consistency(cat_model):
    """more sites than one internal pass (4096): every site's result is independent of its batch position"""
    ctx, _ = cat_model
    g0, g1 = synth_cat_groups(77, 64)
    reps = 4096 // 64 + 3
    big0 = np.tile(g0, (reps, 1, 1, 1)); big1 = np.tile(g1, (reps, 1, 1, 1))
    got = _fwd(ctx, big0, big1)
    small = _fwd(ctx, g0, g1)
    assert np.array_equal(got.reshape(reps, 64, 10), np.broadcast_to(small, (reps, 64, 10)))


def test_cat_forward_empty_and_errors(cat_model):
    import torch
    from nanosnp_amd._lib import NanoSNPError
    ctx, _ = cat_model
    out = ctx.cat_forward(torch.empty((0, 40, 11, 5), device="cuda"), torch.empty((0, 40, 11, 5), device="cuda"))
    assert out.shape == (0, 10)
    with pytest.raises(NanoSNPError):
        ctx.cat_forward(torch.zeros((2, 40, 5, 5), device="cuda"), torch.zeros((2, 40, 5, 5), device="cuda"))


def test_cat_forward_requires_weights():
    import torch
    from nanosnp_amd._lib import Context, NanoSNPError
    ctx = Context(0)
    with pytest.raises(NanoSNPError):
        ctx.cat_forward(torch.zeros((1, 40, 11, 5), device="cuda"), torch.zeros((1, 40, 11, 5), device="cuda"))
    ctx.close()


def test_cat_groups_vs_oracle(gpu_ctx):
    import torch
    from nanosnp_amd._lib import NanoSNPError
    from oracle import oracle
    rng = np.random.default_rng(9)
    N, L = 37, 11
    mk = lambda D: [rng.integers(-2, 5, (N, D, L)).astype(np.int32), rng.integers(0, 61, (N, D, L)).astype(np.int32),
                    rng.integers(0, 61, (N, D, L)).astype(np.int32)]
    t1, t2 = mk(24), mk(40)
    got = gpu_ctx.cat_groups([torch.from_numpy(a).cuda() for a in t1], [torch.from_numpy(a).cuda() for a in t2])
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), oracle.cat_groups(t1, t2))
    short = [torch.from_numpy(a[:, :10].copy()).cuda() for a in t1]
    with pytest.raises(NanoSNPError):
        gpu_ctx.cat_groups(short, [torch.from_numpy(a).cuda() for a in t2])


@pytest.mark.parametrize("prec", [0, 1, 2], ids=["fp32", "f16x3", "bf16x3"])
def test_cat_conv_kernels_agree(prec):
    """the two convolution kernels of the ResCRNN - k_cat_conv (block + halo staged in LDS once; option cat_conv_lds = 1, the default)
    and the gathering implicit GEMM (cat_conv_lds = 0) - sum the same products in a different K order: both inside the tolerance of
    the reference golden, within 1e-5 of each other on 300 random sites, each run-to-run deterministic"""
    from nanosnp_amd import _lib
    z = np.load(golden("cat_fwd_large.npz"))
    g0, g1 = synth_cat_groups(4242, 300)
    c = _lib.Context(0)
    c.cat_load_weights(seeded_cat_weights(int(z["seed"])))
    c.set_option("cat_precision", prec)
    out = {}
    for lds in (1, 0):
        c.set_option("cat_conv_lds", lds)
        got = _fwd(c, z["g0"], z["g1"])
        assert np.isfinite(got).all() and np.abs(got - z["gt"]).max() < PROB_ATOL
        out[lds] = _fwd(c, g0, g1)
        assert np.array_equal(out[lds], _fwd(c, g0, g1))
    d = np.abs(out[0] - out[1]).max()
    print("cat conv kernels, precision", prec, "max difference", d)
    assert d < 1e-5
    c.close()


def test_reference_style_cat_model_interface():
    """nanosnp_amd.cat_model.CatModel mirrors the legacy CatModel as HaplotypeModel/predict.py uses it: CatModel(nc0, nc1, nc2, nclass, nh),
    load_state_dict, eval, to, predict(g0, g1, g2, g3) -> [N, 10]; the golden written by the reference module"""
    import torch
    from nanosnp_amd import _lib
    from nanosnp_amd.cat_model import CatModel
    from nanosnp_amd.fixtures import cat_weight_names, seeded_cat_weights
    z = np.load(golden("cat_fwd.npz"))
    m = CatModel(nc0=5, nc1=5, nc2=2, nclass=10, nh=256).to("cuda")
    g0 = torch.from_numpy(z["g0"].astype(np.float32)).cuda(); g1 = torch.from_numpy(z["g1"].astype(np.float32)).cuda()
    with pytest.raises(_lib.NanoSNPError):
        m.predict(g0, g1, None, None)                                                   # weights not loaded
    sd = {k: torch.from_numpy(w) for k, w in zip(cat_weight_names(), seeded_cat_weights(int(z["seed"])))}
    sd["haplotype_base.cnn.0.bn1.num_batches_tracked"] = torch.tensor(0)              # (extra buffers of a real state dict are ignored)
    m.load_state_dict(sd)
    m.eval()
    got = m.predict(g0, g1, None, None).cpu().numpy()
    assert np.abs(got - z["gt"]).max() < PROB_ATOL and np.array_equal(got.argmax(1), z["gt"].argmax(1))
    with pytest.raises(KeyError):
        CatModel().load_state_dict({k: v for k, v in list(sd.items())[:10]})
    with pytest.raises(_lib.NanoSNPError):
        CatModel(nh=128)
    with pytest.raises(_lib.NanoSNPError):
        m.predict(g0.cpu(), g1.cpu(), None, None)
