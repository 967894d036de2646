"""Time k_neck_* of several tools/variants/<name>/liboetr_hip.so builds (interleaved)."""
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
w = {k: v for k, v in model.state_dict().items() if k in pkg.neck_keys()}
n, hb = int(os.environ.get('IMGS', 16)), int(os.environ.get('HB', 40))
bb = torch.relu(torch.randn(n, 1024, hb, hb, device=dev))
names = sys.argv[1:]
engines = {}
for name in names:
    hip_engine._lib = hip_engine.load_library(str(REPO / 'tools' / 'variants' / name / 'liboetr_hip.so'))
    engines[name] = pkg.NeckEngine(w, device=dev)
    engines[name].set_conv_kernel(os.environ.get('KIND', 'auto'))   # gather | row_window | row_window_1w
acc = {k: {} for k in names}
for rnd in range(3):
    for name in names:
        eng = engines[name]
        for _ in range(2): eng.forward(bb)
        with pkg.KernelTrace(eng, max_launches=256) as tr:
            for _ in range(10): eng.forward(bb)
            torch.cuda.synchronize()
        for k, v in tr.summary().items():
            a = acc[name].setdefault(k, [0, 0.0]); a[0] += v[0]; a[1] += v[1]
for name in names:
    print(f'{name:10s} ' + ' '.join(f'{k.replace("k_neck_","")}={v[1]/v[0]*1e3:.1f}' for k, v in acc[name].items()))
