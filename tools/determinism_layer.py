"""Repeat a forward that stops after k encoder launches and compare the whole workspace with the
first run: which buffer / tile / lane differs first (tools/variants/<name> library)."""
import os, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import imagematching_oetr_amd as pkg
from imagematching_oetr_amd import hip_engine
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
w = model.hot_path_state()
prec, tile, k, variant = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
fill = sys.argv[5] if len(sys.argv) > 5 else ''
hip_engine._lib = hip_engine.load_library(str(REPO / 'tools' / 'variants' / variant / 'liboetr_hip.so'))
eng = pkg.HotPathEngine(w, device=dev, precision=prec, enc_tile=tile)
n, h1, w1, h2, w2 = 2, 20, 20, 20, 20
f1 = (torch.rand(n, 256, h1, w1) - 0.5).to(dev); f2 = (torch.rand(n, 256, h2, w2) - 0.5).to(dev)
p1 = model.pos_encoding(f1.cpu()).contiguous().to(dev); p2 = model.pos_encoding(f2.cpu()).contiguous().to(dev)
c = (f1, f2, p1, p2, (h1 * 32, w1 * 32), (h2 * 32, w2 * 32))
L1, L2 = h1 * w1, h2 * w2
rows = n * (L1 + L2); nt32 = n * ((L1 + 31) // 32 + (L2 + 31) // 32)
al = lambda fl: (fl * 4 + 255) // 256 * 64
layout = (('x', rows * 256, 256), ('qp', (rows + 2 * n * 64) * 256, 256), ('pos', (L1 + L2) * 256, 256),
          ('kvp0', nt32 * 8192, 8192), ('ksp0', nt32 * 256, 256), ('kvp1', nt32 * 8192, 8192), ('ksp1', nt32 * 256, 256),
          ('att0', nt32 * 256, 256), ('z0', nt32 * 8, 8), ('dkv1', nt32 * 8192, 8192), ('dks1', nt32 * 256, 256))
def run():
    eng.forward(*c, stages=True, enc_layers=8)      # a full forward first: every buffer holds this shape's data
    torch.cuda.synchronize()
    ws = eng._current_ws()
    if fill:
        body = ws.view(torch.float32)[64:]
        body.fill_(float('nan')) if fill == 'nan' else body.zero_()
    eng.forward(*c, stages=True, enc_layers=k)
    torch.cuda.synchronize()
    return eng._current_ws().clone().view(torch.float32)
ref = run()
print('flags word', int(eng._current_ws().view(torch.int32)[0]))
shown = 0
for it in range(400):
    b = run()
    off = 64; bad = False
    for nm, fl, unit in layout:
        a_, b_ = ref[off:off + fl], b[off:off + fl]
        ne = torch.nonzero((a_ != b_) & ~(torch.isnan(a_) & torch.isnan(b_))).flatten()
        if ne.numel() and shown < 4:
            bad = True
            units = sorted(set((ne // unit).tolist())); inu = sorted(set((ne % unit).tolist()))
            print(f'it {it} {nm}: {ne.numel()} differ; units {units[:12]} ({len(units)}); in-unit {inu[:10]}..{inu[-4:]} ({len(inu)}) max {float((a_-b_)[ne].abs().max()):.3e}')
            if nm.startswith('kvp'):
                u = units[0]; d = (a_ != b_)[u * 8192:(u + 1) * 8192].view(8, 4, 64, 4)   # [head][quad][lane][4]
                print('    unit', u, 'heads', sorted(set(torch.nonzero(d)[:, 0].tolist())), 'quads', sorted(set(torch.nonzero(d)[:, 1].tolist())),
                      'lanes', sorted(set(torch.nonzero(d)[:, 2].tolist())))
        off += al(fl)
    shown += bad
print('done')
