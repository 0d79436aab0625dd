import sys, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch
from tools.hap_bench import HapStage
hs = HapStage(0, 16384, 16384, 30.0, 90, 20261236, timing=True)
b0, b1 = hs.batch_range(0)
for warm, timed in ((0, 8), (8, 8), (32, 16), (128, 32)):
    hs.sync(); torch.cuda.synchronize(); hs.ctx.read_timing()
    import time; time.sleep(0.2)
    for _ in range(warm): hs.features(b0, b1, which=(0,))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(hs.stream)
    for _ in range(timed): hs.features(b0, b1, which=(0,))
    e1.record(hs.stream); hs.sync(); torch.cuda.synchronize()
    ms, n = hs.ctx.read_timing()["hap_features"]
    print(f"warm {warm:4d} timed {timed:3d}: events around the timed launches {e0.elapsed_time(e1)/timed:.4f} ms per launch; per-kernel events over all {n} launches {ms/n:.4f} ms")
