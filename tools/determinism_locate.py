"""Locate the first stage whose output depends on the workspace's previous contents."""
import os, sys
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
name, prec = sys.argv[1], sys.argv[2]
hip_engine._lib = hip_engine.load_library(str(REPO / 'tools' / 'variants' / name / 'liboetr_hip.so'))
eng = pkg.HotPathEngine(w, device=dev, precision=prec, enc_tile=64)
for n, hf in ((2, 20), (8, 20)):
    f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev); f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
    pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
    hw = (hf * 32, hf * 32)
    def run(k):
        out = {kk: v.clone() for kk, v in eng.forward(f1, f2, pos, pos, hw, hw, stages=True, enc_layers=k).items() if torch.is_tensor(v)}
        torch.cuda.synchronize()
        out['_ws'] = eng._current_ws().clone()
        return out
    for k in range(1, 9):
        a = run(k)
        run(1 + (k + 1) % 3)          # disturb the workspace
        b = run(k)
        diffs = {kk: (a[kk] - b[kk]).abs().max().item() for kk in a if a[kk].shape == b[kk].shape and not torch.equal(a[kk], b[kk])}
        diffs.pop('_ws', None)
        if diffs:
            # workspace layout (api.hip: carve): status 256 B | x | qp (tile-major, padded) | pos | kvp0 | ksp0 | kvp1 | ksp1
            L = hf * hf; rows = 2 * n * L; nt32 = 2 * n * ((L + 31) // 32)
            al = lambda fl: (fl * 4 + 255) // 256 * 64   # floats, 256-B aligned
            wa, wb = a['_ws'].view(torch.float32), b['_ws'].view(torch.float32)
            off = 64
            for nm, fl in (('x', rows * 256), ('qp', (rows + 2 * n * 64) * 256), ('pos', 2 * L * 256), ('kvp0', nt32 * 8192), ('ksp0', nt32 * 256), ('kvp1', nt32 * 8192), ('ksp1', nt32 * 256)):
                sa, sb = wa[off:off + fl], wb[off:off + fl]
                ne = torch.nonzero(sa != sb).flatten()
                print(f'   ws.{nm}: {ne.numel()} differing floats' + (f' first at {ne[0].item()} (unit {ne[0].item() // (8192 if nm.startswith("kvp") else 256)}) last {ne[-1].item()}' if ne.numel() else ''))
                if nm.startswith('kvp') and ne.numel():
                    u = ne[0].item() // 8192
                    d = (sa - sb)[u * 8192:(u + 1) * 8192].view(8, 4, 64, 4)     # head, q, lane, j
                    hd = torch.nonzero(d.abs().amax(dim=(1, 2, 3)) > 0).flatten().tolist()
                    print(f'      slot {u}: heads {hd}')
                    for h in hd:
                        m = d[h].abs() > 0                     # q, lane, j
                        print(f'      head {h}: per-register (r = 4q + j) count of differing lanes:', m.permute(0, 2, 1).reshape(16, 64).sum(1).tolist())
                        print(f'      head {h}: differing lanes:', torch.nonzero(m.any(dim=0).any(dim=1)).flatten().tolist())
                        print(f'      head {h}: max |diff| {d[h].abs().max().item():.3e}, max |value| {sa[u*8192+h*1024:u*8192+(h+1)*1024].abs().max().item():.3e}')
                off += al(fl)
            m = (a['memory1'] - b['memory1']).abs().amax(dim=(0, 2)) if 'memory1' in a else None
            rows = torch.nonzero(m > 0).flatten().tolist() if m is not None else []
            m2 = (a['memory2'] - b['memory2']).abs().amax(dim=(0, 2))
            rows2 = torch.nonzero(m2 > 0).flatten().tolist()
            print(f'n={n} hf={hf} enc_layers={k}: differs {diffs}; memory1 token rows {rows[:12]}..({len(rows)}) memory2 rows {rows2[:12]}..({len(rows2)})')
            pn = torch.nonzero((a['memory1'] - b['memory1']).abs().amax(dim=(1, 2)) > 0).flatten().tolist()
            print('   pairs with differing memory1:', pn)
            break
    else:
        print(f'n={n} hf={hf}: all prefixes identical')
