"""Overlapped-mode throughput (model.hip_streams = 3, product path) against the number of steps in a timed region:
what a short region pays for pipeline fill / drain and for 20 not being a multiple of 3.
    python tools/steps_sweep.py"""
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
for k in (3, 2, 4):
    model.hip_flush(); model.hip_streams = k
    def region(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.boxes_from_features(f1, f2, pos, pos, hw, hw)
        model.hip_flush()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    region(30)
    out = []
    for steps in (6, 12, 18, 20, 21, 24, 40, 100, 200):
        t = statistics.median(region(steps) for _ in range(9))
        out.append(f'{steps}: {n * steps / t / 1e3:.2f}k ({t / steps * 1e6:.0f} us/step, {t * 1e3:.2f} ms)')
    print(f'hip_streams={k}: ' + '  '.join(out), flush=True)
