"""Throughput mode: does queueing a second batch on every side stream (model.hip_queue_depth = 2: the host runs a whole
round of batches ahead instead of waiting for a stream's previous batch to publish its status word) shorten the step?
    python tools/queue_depth_probe.py"""
import sys, time, statistics
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval().to(dev)
model.hip_freeze_weights = True
n, hf = 8, 20
f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev); f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
pos = model.pos_encoding(f1).contiguous(); hw = (640, 640)
model.hip_streams = 3
def region(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.boxes_from_features(f1, f2, pos, pos, hw, hw)
    model.hip_flush()
    torch.cuda.synchronize()
    return time.perf_counter() - t0
region(60)
for rnd in range(3):
    for depth in (1, 2, 3):
        model.hip_flush(); model.hip_queue_depth = depth
        region(30)
        out = []
        for steps in (20, 60, 200):
            t = statistics.median(region(steps) for _ in range(9))
            out.append(f'{steps}: {n * steps / t / 1e3:.2f}k ({t / steps * 1e6:.0f} us/step)')
        print(f'round {rnd} hip_queue_depth={depth}: ' + '  '.join(out), flush=True)
