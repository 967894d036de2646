import ctypes, os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
os.environ['OETR_HIP_LIB'] = str(REPO / 'tools/variants/timing/liboetr_hip.so')
import numpy as np, torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
n = 8
f1 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev); f2 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev)
pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
eng = pkg.HotPathEngine(model.hot_path_state(), device=dev)
for _ in range(3): eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
torch.cuda.synchronize()
nb = 16
buf = (ctypes.c_longlong * (16 * nb))()
eng.lib.oetr_debug_read_tbuf_decoder.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert eng.lib.oetr_debug_read_tbuf_decoder(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(nb, 16)[:, :10].astype(np.float64)
d = np.diff(t, axis=1)
names = ['state reduce + att0', 'S1 Wm_c0', 'S2 LN3+W1', 'S3 W2', 'S4 LN1+qkv+selfattn', 'S5 Wm_s1', 'S6 LN2+Wq+crossattn', 'S7 Wm_c1', 'S8/9 LN3+MLP']
for i, nm in enumerate(names):
    print(f'{nm:24s} {d[:, i].mean():9.0f} cycles  ({d[:, i].mean()/2300:.2f} us @2.3GHz)')
print('total', (t[:, 9] - t[:, 0]).mean())
