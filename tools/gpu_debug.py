"""Print per-stage max errors of the HIP path vs the CPU oracle (no asserts)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from oracle import oetr_oracle as orc
from imagematching_oetr_amd import HotPathEngine, linear_attention, full_attention

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
print(torch.cuda.get_device_name(0))


def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return f'max|d|={float((a-b).abs().max()):.3e} (ref absmax {float(b.abs().max()):.3e})'


def run_case(wseed, sharp, fseed, n, g1, g2, im1, im2):
    w = orc.make_hot_weights(wseed, sharpen=sharp)
    f1 = orc.make_features(fseed, n, *g1)
    f2 = orc.make_features(fseed + 100, n, *g2)
    p1, p2 = orc.position_table(*g1), orc.position_table(*g2)
    ref = orc.hot_path(f1, f2, w, im1, im2, return_stages=True)
    eng = HotPathEngine(w, device=dev)
    for nl in (1, 2, 8):
        x1, x2 = orc.encoder_stack(orc.tokens(f1), orc.tokens(f2), orc.tokens(p1), orc.tokens(p2), w, n_layers=nl)
        if nl < 8:
            out = eng.forward(f1.to(dev), f2.to(dev), p1.to(dev), p2.to(dev), im1, im2, stages=True, enc_layers=nl)
            print(f'  enc_layers={nl} x1 {err(out["memory1"], x1)} x2 {err(out["memory2"], x2)}')
    out = eng.forward(f1.to(dev), f2.to(dev), p1.to(dev), p2.to(dev), im1, im2, stages=True)
    torch.cuda.synchronize()
    for k in ('memory1', 'memory2', 'hs1', 'hs2', 'logits1', 'logits2', 'cxy1', 'cxy2', 'tlbr1', 'tlbr2', 'box1', 'box2'):
        print(f'  {k:8s} {err(out[k], ref[k])}')
    for s in ('1', '2'):
        iou = orc.bbox_iou_aligned(out['box' + s].cpu(), ref['box' + s])
        print(f'  iou{s} min {float(iou.min()):.6f}')


for case in [(0, False, 10, 2, (20, 20), (20, 20), (640, 640), (640, 640)),
             (2, False, 12, 2, (20, 20), (40, 40), (640, 640), (1280, 1280)),
             (4, True, 14, 3, (15, 20), (25, 10), (480, 640), (800, 320))]:
    print('case', case)
    try:
        run_case(*case)
    except Exception as e:  # keep going: one call should tell us as much as possible
        import traceback; traceback.print_exc()

print('attention cores')
for (L, S) in [(1, 1), (77, 33), (400, 400), (400, 1600)]:
    g = torch.Generator().manual_seed(5)
    q = (torch.rand(2, L, 8, 32, generator=g) - 0.5) * 4
    k = (torch.rand(2, S, 8, 32, generator=g) - 0.5) * 4
    v = (torch.rand(2, S, 8, 32, generator=g) - 0.5) * 2
    try:
        print(f'  L{L} S{S} linear {err(linear_attention(q.to(dev), k.to(dev), v.to(dev)), orc.linear_attention(q, k, v))}')
        print(f'  L{L} S{S} full   {err(full_attention(q.to(dev), k.to(dev), v.to(dev)), orc.full_attention(q, k, v))}')
    except Exception as e:
        import traceback; traceback.print_exc()

# quick timing at the bench shape
w = orc.make_hot_weights(0)
eng = HotPathEngine(w, device=dev)
n = 8
f1 = orc.make_features(1, n, 20, 20).to(dev); f2 = orc.make_features(2, n, 20, 20).to(dev)
p1 = orc.position_table(20, 20).to(dev)
for _ in range(5):
    eng.forward(f1, f2, p1, p1, (640, 640), (640, 640))
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 50
for _ in range(K):
    eng.forward(f1, f2, p1, p1, (640, 640), (640, 640))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f'N=8 640^2 hot path: {dt*1e3:.3f} ms/batch -> {n/dt:.1f} pairs/s -> {n/dt*8.365e-3:.2f} TFLOP/s')
