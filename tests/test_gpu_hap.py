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
