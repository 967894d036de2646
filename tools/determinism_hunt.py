"""Hunt for a forward whose result depends on timing / workspace history (rare): repeat staged
forwards of several shapes, compare every stage output and workspace buffer with the first run
of that shape, stop at the first difference and say where it is."""
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
prec = sys.argv[1] if len(sys.argv) > 1 else 'f32_split_f16'
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 64
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
if len(sys.argv) > 4 and sys.argv[4]:
    hip_engine._lib = hip_engine.load_library(str(REPO / 'tools' / 'variants' / sys.argv[4] / 'liboetr_hip.so'))
eng = pkg.HotPathEngine(w, device=dev, precision=prec, enc_tile=tile, attention=os.environ.get('HUNT_ATTENTION', 'linear'))
if os.environ.get('HUNT_PREREDUCE'): eng.set_state_prereduce(int(os.environ['HUNT_PREREDUCE']))
if os.environ.get('HUNT_DECSPLIT'): eng.set_decoder_split(int(os.environ['HUNT_DECSPLIT']))
if os.environ.get('HUNT_TAILMODE'): eng.set_tail_mode(int(os.environ['HUNT_TAILMODE']))
shapes = [(2, 20, 20, 20, 20), (8, 20, 20, 20, 20), (2, 10, 10, 6, 20), (3, 25, 25, 25, 25), (1, 32, 32, 32, 32)]
if os.environ.get('HUNT_SHAPES'):
    shapes = [shapes[int(i)] for i in os.environ['HUNT_SHAPES'].split(',')]
VERBOSE = int(os.environ.get('HUNT_VERBOSE', '2'))
cases = []
for n, h1, w1, h2, w2 in shapes:
    f1 = (torch.rand(n, 256, h1, w1) - 0.5).to(dev); f2 = (torch.rand(n, 256, h2, w2) - 0.5).to(dev)
    p1 = model.pos_encoding(f1.cpu()).contiguous().to(dev); p2 = model.pos_encoding(f2.cpu()).contiguous().to(dev)
    cases.append((f1, f2, p1, p2, (h1 * 32, w1 * 32), (h2 * 32, w2 * 32)))
# HUNT_MASKS=1: forward_dummy's masks (with holes) on every shape - the MASKED kernel instantiations
MASKS = [None] * len(cases)
if os.environ.get('HUNT_MASKS'):
    from oracle import oetr_oracle as orc
    MASKS = [dict(mask1=orc.make_masks(50 + i, n, h1, w1, 'holes').to(dev), mask2=orc.make_masks(60 + i, n, h2, w2, 'holes').to(dev))
             for i, (n, h1, w1, h2, w2) in enumerate(shapes)]
FILL = os.environ.get('HUNT_FILL', '')
THRASH = torch.zeros(int(os.environ.get('HUNT_THRASH_MB', '0')) * 262144 + 1, device=dev)
def run(c, k=8, fill='', masks=None):
    if fill:
        ws = eng._current_ws()
        if ws is not None:
            body = ws.view(torch.float32)[hip_engine.WORKSPACE_STATUS_BYTES // 4:]     # (behind the status block: the library's own)
            if fill == 'zero': body.zero_()
            elif fill == 'nan': body.fill_(float('nan'))
            elif fill == 'big': body.fill_(3.0e4)
            elif fill == 'rand': body.copy_(torch.randn(body.shape, device=dev) * 10)
    out = {kk: v.clone() for kk, v in eng.forward(*c, stages=True, enc_layers=k, **(masks or {})).items() if torch.is_tensor(v)}
    torch.cuda.synchronize()
    out['_ws'] = eng._current_ws().clone()
    return out
refs = [run(c, masks=m) for c, m in zip(cases, MASKS)]
t0 = time.time(); runs = 0; found = 0
while time.time() - t0 < budget:
    for ci, c in enumerate(cases):
        if runs % 3 == 0: run(c, 1 + runs % 5, masks=MASKS[ci])
        if THRASH.numel() > 1: THRASH.add_(1.0)      # cold L2 / MALL for the measured run
        b = run(c, 8, FILL, MASKS[ci]); runs += 1
        diffs = {kk: float((refs[ci][kk] - b[kk]).abs().max()) for kk in b if kk != '_ws' and not torch.equal(refs[ci][kk], b[kk])}
        if diffs:
            found += 1
            if found > VERBOSE: continue
            n_, h1_, w1_, h2_, w2_ = shapes[ci]
            L1, L2 = h1_ * w1_, h2_ * w2_
            rows = n_ * (L1 + L2); nt32 = n_ * ((L1 + 31) // 32 + (L2 + 31) // 32)
            al = lambda fl: (fl * 4 + 255) // 256 * 64
            wa, wb = refs[ci]['_ws'].view(torch.float32), b['_ws'].view(torch.float32)
            off = 64
            for nm, fl, unit in (('x', rows * 256, 256), ('qp', (rows + 2 * n_ * 64) * 256, 256), ('pos', (L1 + L2) * 256, 256),
                                 ('kvp0', nt32 * 8192, 8192), ('ksp0', nt32 * 256, 256), ('kvp1', nt32 * 8192, 8192), ('ksp1', nt32 * 256, 256),
                                 ('att0', nt32 * 256, 256), ('z0', nt32 * 8, 8), ('dkv1', nt32 * 8192, 8192), ('dks1', nt32 * 256, 256)):
                ne = torch.nonzero(wa[off:off + fl] != wb[off:off + fl]).flatten()
                if ne.numel():
                    units = sorted(set((ne // unit).tolist()))
                    inunit = sorted(set((ne % unit).tolist()))
                    print(f'   ws.{nm}: {ne.numel()} floats differ; units {units[:10]}; offsets in unit {inunit[:6]}..{inunit[-3:]} ({len(inunit)}) maxdiff {float((wa[off:off+fl]-wb[off:off+fl]).abs().max()):.3e}')
                off += al(fl)
            print(f'run {runs} shape {shapes[ci]}: {diffs}')
            for s in ('1', '2'):
                d = (refs[ci]['memory' + s] - b['memory' + s]).abs()
                if d.max() > 0:
                    pairs = torch.nonzero(d.amax(dim=(1, 2)) > 0).flatten().tolist()
                    rows = torch.nonzero(d.amax(dim=(0, 2)) > 0).flatten().tolist()
                    cols = torch.nonzero(d.amax(dim=(0, 1)) > 0).flatten().tolist()
                    print(f'   memory{s}: pairs {pairs} rows {rows[:8]}..{rows[-4:]} ({len(rows)}) channels {cols[:8]}..{cols[-4:]} ({len(cols)})')
print(f'{prec}@{tile}: {found} differing of {runs} runs in {time.time() - t0:.0f} s')
