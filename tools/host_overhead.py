#!/usr/bin/env python3
"""Host time per submitted batch (GPU box): engine call alone, + asynchronous status read, the module's
boxes_from_features (1 and 3 streams).  Submission only - the loop is timed before the device is waited for,
with a queue deep enough that the host never blocks on the GPU (sync every 16 steps excluded)."""
import sys
import time
import cProfile
import pstats

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
import imagematching_oetr_amd as pkg  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
model, weights, f1, f2, p1, p2, hf, hf2 = bench.synthetic_inputs(8, 640, 640, dev)
model = model.to(dev)
model.hip_freeze_weights = True
hw = (640, 640)
eng = pkg.HotPathEngine(weights, device=dev)


def timed(fn, n=64):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for blk in range(n // 16):
        t0 = time.perf_counter()
        for _ in range(16):
            fn()
        tot += time.perf_counter() - t0
        torch.cuda.synchronize()
    return tot / n * 1e6


print('eng.forward                      %7.1f us' % timed(lambda: eng.forward(f1, f2, p1, p2, hw, hw)))
print('eng.forward + read_flags_async   %7.1f us' % timed(lambda: (eng.forward(f1, f2, p1, p2, hw, hw), eng.read_flags_async())))
for k in (1, 3):
    model.hip_streams = k
    model.hip_flush()
    print('model.boxes_from_features k=%d    %7.1f us' % (k, timed(lambda: model.boxes_from_features(f1, f2, p1, p2, hw, hw))))
    model.hip_flush()
model.hip_streams = 3
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    model.boxes_from_features(f1, f2, p1, p2, hw, hw)
model.hip_flush()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
