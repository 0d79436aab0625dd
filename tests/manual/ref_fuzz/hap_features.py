#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): dataset_dev.get_frequency_feature against oracle.hap_features on planes outside
the generator's range (codes the reference ignores, huge and negative qualities, all padding, one-element read sets), float64 bit for bit.
    python tests/manual/ref_fuzz/hap_features.py"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object}), ("tables", {"Filters": lambda **k: None})):
    m = types.ModuleType(name); [setattr(m, k, v) for k, v in attrs.items()]; sys.modules[name] = m
sys.path.insert(0, "/root/reference/HaplotypeModel")
import dataset_dev
from oracle import oracle
import inspect
print([n for n in dir(oracle) if "hap" in n.lower() or "feat" in n.lower()])
print(inspect.signature(oracle.hap_features))
import warnings
bad = 0
for seed in range(1, 9):
    rng = np.random.default_rng(seed)
    for trial in range(60):
        L = int(rng.choice([33, 11, 1, 5]))
        D = int(rng.choice([1, 2, 7, 30, 90, 180, 200]))
        mode = trial % 6
        if mode == 0:      # in-range random
            seq = rng.integers(-2, 5, (D, L)); hap = rng.integers(-2, 4, (D, L)); bq = rng.integers(-2, 61, (D, L)); mq = rng.integers(-2, 61, (D, L))
        elif mode == 1:    # out-of-range codes
            seq = rng.integers(-5, 9, (D, L)); hap = rng.integers(-4, 7, (D, L)); bq = rng.integers(-50, 300, (D, L)); mq = rng.integers(-50, 300, (D, L))
        elif mode == 2:    # huge qualities (sums beyond float32 integers / int32?)
            seq = rng.integers(1, 5, (D, L)); hap = rng.integers(0, 4, (D, L)); bq = rng.integers(0, 2**20, (D, L)); mq = rng.integers(0, 2**20, (D, L))
        elif mode == 3:    # all padding / all deletions / all one base
            v = int(rng.choice([-2, -1, 0, 3])); seq = np.full((D, L), v); hap = rng.integers(0, 4, (D, L)); bq = rng.integers(0, 61, (D, L)); mq = rng.integers(0, 61, (D, L))
        elif mode == 4:    # realistic: hap constant per row
            seq = rng.integers(-1, 5, (D, L)); h = rng.integers(0, 4, (D, 1)); hap = np.where(seq != 0, np.repeat(h, L, 1), 0); bq = rng.integers(0, 61, (D, L)); mq = rng.integers(0, 61, (D, L))
        else:              # a set present only through one element
            seq = rng.integers(1, 5, (D, L)); hap = np.zeros((D, L), int); hap[rng.integers(0, D), rng.integers(0, L)] = int(rng.integers(1, 4)); bq = rng.integers(0, 61, (D, L)); mq = rng.integers(0, 61, (D, L))
        seq, hap, bq, mq = (a.astype(np.int32) for a in (seq, hap, bq, mq))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = dataset_dev.get_frequency_feature(seq, bq, mq, hap)
        ref_row = rng.integers(0, 5, L).astype(np.int32)
        got = oracle.hap_features(seq, bq, mq, hap, ref_row)
        got = np.asarray(got)
        g104 = got[:104] if got.shape[0] == 105 else got
        same = want.shape == g104.shape and np.array_equal(want, g104.astype(np.float64) if g104.dtype != np.float64 else g104, equal_nan=True)
        if not same:
            bad += 1
            d = np.argwhere(~np.isclose(want, g104, rtol=0, atol=0, equal_nan=True))
            print("seed", seed, "trial", trial, "mode", mode, "D", D, "L", L, "dtype", g104.dtype, "n diff", len(d), "first", d[:3].tolist(), [ (want[tuple(i)], g104[tuple(i)]) for i in d[:3]])
print("bad", bad)
