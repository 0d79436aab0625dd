#!/usr/bin/env python3
"""Shader clock the hot kernels actually run at (diagnostic build only):

    tools/build_variant.sh clock -DNSNP_DEV_CLOCK
    NANOSNP_DEV_LIB_OVERRIDE=1 NANOSNP_HIP_LIB=build_tmp/libs/libnanosnp_hip_clock.so python tools/clock_probe.py

Runs the bench's pileup stage (32 streams, batches of 4096 windows), the same kernels alone, and the HaplotypeModel stage, each for
about two seconds, and prints clock = 0.1 GHz x shader cycles / 100 MHz ticks summed over the workgroups of each kernel, the cycles a workgroup lives
(mean, shortest, longest) and, for a single launch, the time from the first workgroup's start to the last one's end
(nanosnp_amd/csrc/nsnp_devclock.hpp; MI355X_MICROARCH.md "DVFS give-back" item 6).  The peak of bench.py's rooflines is priced at
2.4 GHz; this says how much of a fraction below 1 is clock the chip did not run."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from nanosnp_amd import _lib
from tools.pileup_stage import PileupStage
from tools.hap_bench import HapStage

lib = _lib.load()
if not hasattr(lib, "nsnp_devclk_read_pileup"):
    sys.exit("clock_probe.py: the loaded library is not a -DNSNP_DEV_CLOCK build (set NANOSNP_DEV_LIB_OVERRIDE=1 NANOSNP_HIP_LIB=...)")
NAMES = {("pileup", 0): "k_pileup_l0_rs32", ("pileup", 1): "k_pileup_l1_rs4", ("hap", 2): "k_hap_gemm"}


def read(tag, show=True):
    out = {}
    M = (1 << 64) - 1
    for tu in ("pileup", "hap"):
        buf = (ctypes.c_ulonglong * 56)()                                   # [8][3] sums, then [8][4] extremes
        assert getattr(lib, f"nsnp_devclk_read_{tu}")(buf) == 0
        for slot in range(8):
            cyc, ticks, n = buf[3 * slot], buf[3 * slot + 1], buf[3 * slot + 2]
            if n and (tu, slot) in NAMES:
                e = buf[24 + 4 * slot: 28 + 4 * slot]
                out[NAMES[(tu, slot)]] = (0.1 * cyc / max(ticks, 1), n, cyc / n, M - e[3], e[2], (e[1] - (M - e[0])) * 10)
    if show:
        for k, (ghz, n, cyc, cmin, cmax, span) in out.items():
            print(f"{tag:44s} {k:17s} {ghz:6.3f} GHz  {n:8d} workgroups: {cyc:8.0f} cycles each (min {cmin}, max {cmax})"
                  + (f"  first start -> last end {span / 1e3:.1f} us" if span < 5e6 else ""))
    return out


secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
ps = PileupStage(0, 64 * 4096, batch=4096, streams=32, coverage=60.0, enc_group=32)
ps.run(0, ps.n_batches); ps.sync(); read("", show=False)                  # warm-up, discarded
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    ps.run(0, ps.n_batches); n += ps.n_windows
ps.sync(); dt = time.time() - t0
read(f"pileup stage, 32 streams: {n / dt / 1e6:.2f} M sites/s")
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    ps.run(0, ps.n_batches, single_stream=True); n += ps.n_windows
ps.sync(); dt = time.time() - t0
read(f"pileup stage, one stream: {n / dt / 1e6:.2f} M sites/s")
for _ in range(3):
    ps.run(0, 1, single_stream=True); ps.sync(); r = read("", show=False)
ps.run(0, 1, single_stream=True); ps.sync()
read("one batch of 4096 windows alone")
del ps
hs = HapStage(0, 65536, 16384, 30.0, 90, 20260401, timing=False)
for i in range(hs.n_batches): hs.run_batch(i)
hs.sync(); read("", show=False)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for i in range(hs.n_batches): hs.run_batch(i)
    n += hs.n
hs.sync(); dt = time.time() - t0
read(f"haplotype stage: {n / dt / 1e3:.0f} k sites/s")
