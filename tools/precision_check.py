"""Both GEMM modes vs the fp64 oracle on the golden cases + timing."""
import sys, time, glob
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
from oracle import oetr_oracle as orc
from tests.test_oracle_golden import load_hot_case
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
for path in sorted(glob.glob(str(REPO / 'tests/golden/hot_*.npz'))):
    g, w, f1, f2 = load_hot_case(path)
    im1, im2 = tuple(int(v) for v in g['img1']), tuple(int(v) for v in g['img2'])
    p1, p2 = orc.position_table(*g['grid1']), orc.position_table(*g['grid2'])
    s64 = orc.hot_path(f1.double(), f2.double(), orc.cast_weights(w, torch.float64), im1, im2, return_stages=True)
    s32 = orc.hot_path(f1, f2, w, im1, im2, return_stages=True)
    for prec in ('f32', 'f32_split_f16'):
        eng = pkg.HotPathEngine(w, device=dev, precision=prec)
        out = eng.forward(f1.to(dev), f2.to(dev), p1.to(dev), p2.to(dev), im1, im2, stages=True)
        errs = {k: float((out[k].cpu().double().reshape(s64[k].shape) - s64[k]).abs().max()) for k in ('memory1', 'memory2', 'hs1', 'logits1', 'cxy1', 'cxy2', 'box1', 'box2')}
        iou = torch.cat([orc.bbox_iou_aligned(out['box1'].cpu().double(), s64['box1']), orc.bbox_iou_aligned(out['box2'].cpu().double(), s64['box2'])])
        print(f'{Path(path).stem[4:]:18s} {prec:14s} ' + ' '.join(f'{k}={v:.1e}' for k, v in errs.items()) + f' 1-iou={float((1-iou).max()):.1e}')
    errs = {k: float((s32[k].double() - s64[k]).abs().max()) for k in ('memory1', 'memory2', 'hs1', 'logits1', 'cxy1', 'cxy2', 'box1', 'box2')}
    print(f'{"":18s} {"torch-cpu-f32":14s} ' + ' '.join(f'{k}={v:.1e}' for k, v in errs.items()))
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
w = model.hot_path_state()
n = 8
f1 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev); f2 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev)
pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
for prec in ('f32', 'f32_split_f16'):
    eng = pkg.HotPathEngine(w, device=dev, precision=prec)
    for _ in range(10): eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    with pkg.KernelTrace(eng) as tr:
        for _ in range(20): eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
        torch.cuda.synchronize()
    print(f'{prec}: {dt*1e3:.3f} ms/step {n/dt:.0f} pairs/s  ' + ' '.join(f'{k.replace("k_","")}={v[1]/v[0]*1e3:.1f}' for k, v in tr.summary().items()))
