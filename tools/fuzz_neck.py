"""Random-shape fuzz of the HIP neck vs the CPU oracle (run on the GPU box)."""
import sys, random
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
from oracle import oetr_oracle as orc
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
engines = {}
bad = 0
for case in range(ncase):
    ws = rng.randrange(3)
    if ws not in engines:
        engines[ws] = (orc.make_neck_weights(50 + ws), None)
        engines[ws] = (engines[ws][0], pkg.NeckEngine(engines[ws][0], device=dev))
    w, eng = engines[ws]
    n, hb, wb = rng.randrange(1, 5), rng.randrange(2, 48), rng.randrange(2, 48)
    if case % 3 == 0:      # wide maps: the row-window conv kernel (output map >= 16 wide)
        wb = rng.randrange(32, 130)
    kind = ('auto', 'row_window', 'row_window_1w', 'gather')[case % 4] if wb // 2 >= 16 else 'auto'
    eng.set_conv_kernel(kind)
    bb = orc.make_backbone_features(5000 + case, n, hb, wb)
    out = eng.forward(bb.to(dev)).cpu()
    ref = orc.neck(bb, w)
    e = (out - ref).abs().max().item()
    ok = e <= 5e-5 and out.shape == ref.shape
    bad += not ok
    print(f'{"OK " if ok else "BAD"} case {case}: n={n} {hb}x{wb} {kind} err={e:.2e}', flush=True)
print(f'{bad} bad of {ncase}')
