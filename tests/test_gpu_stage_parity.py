"""Parity of the BENCH CONFIGURATION itself ("Next round" 1): the scheduling bench.py times - the column encode
on its own stream into a RING of count buffers, the forwards of 32 contexts on 32 streams behind event dependencies - against the
oracle, every site; and concurrent contexts against their sequential results, bit for bit (the soak of tests/manual/soak.py).
Reference behaviour held: PileupModel/predict.py:49-57 (forward, argmax / max per site)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check_every_site(stage, n_sites):
    from oracle import oracle
    torch = stage.torch
    stage.sync()
    m = n_sites * 33
    oc, od, of = oracle.encode_columns(stage.cols.bases[:int(stage.cols.col_off[m])], stage.cols.col_off[:m + 1], stage.cols.ref[:m])
    ogt, ozy = oracle.pileup_forward(stage.weights, oc.reshape(n_sites, 33, 18), nthreads=8)
    gt, zy = stage.gt_all[:n_sites].cpu().numpy(), stage.zy_all[:n_sites].cpu().numpy()
    assert np.isfinite(gt).all() and np.isfinite(zy).all()
    d = max(np.abs(gt - ogt).max(), np.abs(zy - ozy).max())
    assert d < 1e-4, d                                           # BASELINE north_star: logits within 1e-4 abs
    r = {k: v[:n_sites].cpu().numpy() for k, v in stage.res.items()}
    assert np.array_equal(r["ga"], gt.argmax(1)) and np.array_equal(r["za"], zy.argmax(1))     # predict.py:54-57
    assert np.array_equal(r["gm"], gt.max(1)) and np.array_equal(r["zm"], zy.max(1))
    return d, (oc, od, of)


@pytest.mark.parametrize("precision", [0, 1])
def test_32_streams_ring_of_two_two_sweeps_every_site_against_the_oracle(precision):
    """16 k windows in batches of 512 = 32 batches on 32 streams, 4 batches per encode launch into a ring of TWO count buffers (every
    slot is re-used four times per sweep, so a forward that read a slot after its re-encode, or an encode that did not wait for
    the slot's last readers, shows up as wrong probabilities), two sweeps back to back without a synchronisation between them"""
    from tools.pileup_stage import PileupStage
    st = PileupStage(0, 16384, batch=512, streams=32, coverage=30.0, seed=20261234, precision=precision, timing_streams=4, enc_group=4, ring=2)
    assert st.n_batches == 32 and st.R == 2 and st.G == 4
    st.run(0, 2 * st.n_batches)                                  # two sweeps, no sync in between
    d, (oc, od, of) = _check_every_site(st, st.n_windows)
    # the ring still holds the last two groups of the second sweep: bit for bit
    snap = st.snapshot(st.parity_ranges(st.n_windows, per_batch=64, n_ranges=32), ring_batches=4)
    assert len(snap["ring"]) == 2 and sorted(s["c0"] for s in snap["ring"]) == [24 * st.mcols, 28 * st.mcols]
    for s in snap["ring"]:
        c0, m = s["c0"], s["m"]
        assert np.array_equal(s["counts"], oc[c0:c0 + m]) and np.array_equal(s["depth"], od[c0:c0 + m]) and np.array_equal(s["flags"], of[c0:c0 + m])
    par = st.parity_check(snap)
    assert par["ok"] and par["encode_bit_exact"] and par["sites"] == 32 * 64 and par["max_abs_dp"] < 1e-4, par
    # a sweep through ONE stream writes the same bits (fp32: every launch shape is bit-identical)
    if precision == 0:
        g0, z0 = st.gt_all.clone(), st.zy_all.clone()
        st.torch.cuda.synchronize()
        st.gt_all.zero_(); st.zy_all.zero_()
        st.torch.cuda.synchronize()
        st.run(0, st.n_batches, single_stream=True); st.sync()
        assert st.torch.equal(st.gt_all, g0) and st.torch.equal(st.zy_all, z0)


def test_parity_check_sees_a_corrupted_run():
    """the checker of bench.py's parity_sample is not vacuous: a stale count buffer, a wrong probability and a wrong call all fail it"""
    from tools.pileup_stage import PileupStage
    st = PileupStage(0, 4096, batch=512, streams=8, coverage=30.0, seed=20261235, timing_streams=0, enc_group=2, ring=2)
    st.run(0, st.n_batches)
    snap = st.snapshot(st.parity_ranges(st.n_windows, per_batch=128, n_ranges=8))
    assert st.parity_check(snap)["ok"]
    bad = dict(snap, gt=snap["gt"].copy()); bad["gt"][5, 0] += 3e-4
    r = st.parity_check(bad)
    assert not r["ok"] and r["max_abs_dp"] > 1e-4
    bad = dict(snap, ring=[dict(s, counts=s["counts"].copy()) for s in snap["ring"]]); bad["ring"][0]["counts"][7, 3] += 1
    r = st.parity_check(bad)
    assert not r["ok"] and not r["encode_bit_exact"]
    bad = dict(snap, res=dict(snap["res"], ga=snap["res"]["ga"].copy())); bad["res"]["ga"][9] = (bad["res"]["ga"][9] + 1) % 21
    r = st.parity_check(bad)
    assert not r["ok"] and not r["calls_equal_own_argmax"]


@pytest.mark.parametrize("precision", [0, 2], ids=["fp32", "bf16x3"])
def test_concurrent_contexts_equal_their_sequential_results_bit_for_bit(pileup_weights, precision):
    """eight contexts on eight streams, ragged batch sizes, 20 rounds: the same bits as one after the other (tests/manual/soak.py)"""
    import torch
    from nanosnp_amd import _lib
    rng = np.random.default_rng(4321)
    ctxs = [_lib.Context(0) for _ in range(8)]
    streams = [torch.cuda.Stream() for _ in range(8)]
    for c in ctxs:
        c.pileup_load_weights(pileup_weights)
        c.set_option("pileup_precision", precision)
    xs = [torch.from_numpy((rng.integers(0, 50, (int(rng.integers(100, 6000)), 33, 18)) - 10).astype(np.int32)).cuda() for _ in range(8)]
    seq = [c.pileup_forward(x) for c, x in zip(ctxs, xs)]
    torch.cuda.synchronize()
    for rep in range(20):
        outs = [c.pileup_forward(x, stream=s) for c, x, s in zip(ctxs, xs, streams)]
        torch.cuda.synchronize()
        for (g, z), (g0, z0) in zip(outs, seq):
            assert torch.equal(g, g0) and torch.equal(z, z0), rep
    for c in ctxs:
        c.close()


def test_hap_stage_parity_sample():
    """the HaplotypeModel stage of the haplotype / two-stage / deep60 lines: what run_batch leaves behind against the oracle's chain"""
    from tools.hap_bench import HapStage
    hs = HapStage(0, 700, 256, 30.0, 90, 20261236, timing=False)
    for i in range(hs.n_batches):
        hs.run_batch(i)
    snap = hs.snapshot(range(hs.n_batches), per_batch=48)
    assert [c for _, c in snap["ranges"]] == [48, 48, 48]
    par = hs.parity_check(snap, nthreads=8)
    assert par["ok"] and par["sites"] == 144 and par["max_abs_dp"] < 1e-4, par
    bad = dict(snap, zy=snap["zy"].copy()); bad["zy"][3, 1] += 5e-4
    assert not hs.parity_check(bad, nthreads=8)["ok"]
