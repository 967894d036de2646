"""Per-WAVE arrival times at the phase boundaries of k_encoder64<B,A> (library built with
-DOETR_PHASE_TIMING=2: tools/variants/timing2): how far apart the eight waves of a workgroup
reach each barrier - time the early ones spend waiting with the MFMA pipe idle."""
import ctypes, os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
os.environ['OETR_HIP_LIB'] = str(REPO / 'tools/variants/timing2/liboetr_hip.so')
import numpy as np, torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
n = 8
f1 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev); f2 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev)
pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
prec = sys.argv[1] if len(sys.argv) > 1 else 'f32_split_f16'
eng = pkg.HotPathEngine(model.hot_path_state(), device=dev, precision=prec, enc_tile=64)
lib = eng.lib
for _ in range(3):
    eng.forward(f1, f2, pos, pos, (640, 640), (640, 640), stages=True, enc_layers=3)
eng.forward(f1, f2, pos, pos, (640, 640), (640, 640), stages=True, enc_layers=8)
torch.cuda.synchronize()
nb = 16 * 8
buf = (ctypes.c_longlong * (16 * nb))()
lib.oetr_debug_read_tbuf.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.oetr_debug_read_tbuf(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(16, 8, 16)[:, :, :9].astype(np.float64)
t -= t[:, :, :1].min(axis=1, keepdims=True)          # relative to the workgroup's first wave at stamp 0
names = ['start', 'prologue done', 'apply done', 'merge done', 'LN2 done', 'MLP1 done', 'MLP2a done', '-', 'x stored']
np.set_printoptions(linewidth=200, precision=0, suppress=True)
for wg in (0, 1, 7):
    print(f'workgroup {wg}: cycles since the first wave started, per wave (columns = waves 0..7)')
    for i, nm in enumerate(names):
        print(f'  {nm:14s}', t[wg, :, i], ' spread', t[wg, :, i].max() - t[wg, :, i].min())
print('mean spread per boundary over 16 workgroups:',
      [(names[i], float((t[:, :, i].max(axis=1) - t[:, :, i].min(axis=1)).mean())) for i in range(9)])
