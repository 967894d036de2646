"""Time every tools/variants/*/liboetr_hip.so in ONE process, interleaved
rounds (per-kernel HIP-event durations)."""
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
n = int(os.environ.get('PAIRS', 8))
hf = int(os.environ.get('HF', 20))
f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
names = sys.argv[1:] or sorted(Path(p).parent.name for p in glob.glob(str(REPO / 'tools/variants/*/liboetr_hip.so')))
engines = {}
for name in names:
    hip_engine._lib = hip_engine.load_library(str(REPO / 'tools' / 'variants' / name / 'liboetr_hip.so'))
    engines[name] = pkg.HotPathEngine(w, device=dev, precision=os.environ.get('PREC', 'f32_split_f16'),
                                      enc_tile=int(os.environ.get('TILE', 0)) or None,
                                      attention=os.environ.get('ATTN', 'linear'))
if os.environ.get('TAILMODE'):
    for e in engines.values(): e.set_tail_mode(int(os.environ['TAILMODE']))
if os.environ.get('DECSPLIT'):
    for e in engines.values(): e.set_decoder_split(int(os.environ['DECSPLIT']))
if os.environ.get('PREREDUCE'):
    for e in engines.values(): e.set_state_prereduce(int(os.environ['PREREDUCE']))
hw = (hf * 32, hf * 32)
ref = None
acc = {k: {} for k in names}
for rnd in range(4):
    for name in names:
        eng = engines[name]
        for _ in range(3):
            b = eng.forward(f1, f2, pos, pos, hw, hw)
        with pkg.KernelTrace(eng, max_launches=1024) as tr:
            for _ in range(20):
                eng.forward(f1, f2, pos, pos, hw, hw)
            torch.cuda.synchronize()
        for k, v in tr.summary().items():
            a = acc[name].setdefault(k, [0, 0.0]); a[0] += v[0]; a[1] += v[1]
        if ref is None:
            ref = b[0].clone()
        elif rnd == 0:
            print(f'  [{name}] box1 max diff vs first variant: {(b[0]-ref).abs().max().item():.2e}')
for name in names:
    tot = sum(v[1] for v in acc[name].values()) / 80 * 1e3
    print(f'{name:24s} total={tot:7.1f}us  ' + ' '.join(f'{k.replace("k_","")}={v[1]/v[0]*1e3:.1f}' for k, v in acc[name].items()))
