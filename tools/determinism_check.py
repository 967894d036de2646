"""Is a forward a pure function of its inputs?  For every tools/variants/*/liboetr_hip.so: the
boxes of repeated forwards, with encoder-prefix runs in between (they leave the workspace in a
different state), must be bit-identical.  Prints the number of differing runs."""
import os, sys, glob
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
names = sys.argv[1:] or sorted(Path(p).parent.name for p in glob.glob(str(REPO / 'tools/variants/*/liboetr_hip.so')))
for name in names:
    hip_engine._lib = hip_engine.load_library(str(REPO / 'tools' / 'variants' / name / 'liboetr_hip.so'))
    for prec, tile in (('f32_split_f16', 64), ('f32_split_qk16', 64), ('f16', 64), ('f32_split_f16', 32)):
        eng = pkg.HotPathEngine(w, device=dev, precision=prec, enc_tile=tile)
        bad = 0
        for n, hf in ((2, 20), (8, 20), (3, 25), (5, 20), (1, 32)):
            f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev); f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
            pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
            hw = (hf * 32, hf * 32)
            ref = eng.forward(f1, f2, pos, pos, hw, hw, stages=True)
            for it in range(12):
                if it % 2 == 0:
                    eng.forward(f1, f2, pos, pos, hw, hw, stages=True, enc_layers=1 + it % 3)
                b = eng.forward(f1, f2, pos, pos, hw, hw)
                if not (torch.equal(b[0], ref['box1']) and torch.equal(b[1], ref['box2'])):
                    bad += 1
                    d = (b[0] - ref['box1']).abs().max().item()
                    if os.environ.get('VERBOSE'): print(f'  {name} {prec}@{tile} n={n} hf={hf} it={it}: box1 differs by {d:.3e}')
        print(f'{name:20s} {prec}@{tile}: {bad} differing runs')
