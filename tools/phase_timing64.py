"""Per-phase cycle breakdown of k_encoder64<B,A> (library built with -DOETR_PHASE_TIMING)."""
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
NAMES = ['loads+kvreduce', 'attn-apply (+Z)', 'merge GEMM', 'LN2 (registers)', 'MLP1a, MLP1b+GELU(a)', 'MLP2a+GELU(b)', '-', 'MLP2b+stage+x store',
         'LN-A', 'Q GEMM', 'K GEMM+phi(Q) store, V GEMM', 'KV state']
prec = sys.argv[1] if len(sys.argv) > 1 else 'f32_split_f16'
eng = pkg.HotPathEngine(model.hot_path_state(), device=dev, precision=prec, enc_tile=64)
lib = eng.lib
print('precision', prec)
for _ in range(3):
    eng.forward(f1, f2, pos, pos, (640, 640), (640, 640), stages=True, enc_layers=3)
# stop after the encoder with 3 layers: last launch is <B> only; the one before is <B,A>
eng.forward(f1, f2, pos, pos, (640, 640), (640, 640), stages=True, enc_layers=8)
torch.cuda.synchronize()
nb = 112
buf = (ctypes.c_longlong * (16 * nb))()
lib.oetr_debug_read_tbuf.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.oetr_debug_read_tbuf(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(nb, 16)[:, :13].astype(np.float64)
d = np.diff(t, axis=1)
print('cycles per phase (mean over workgroups; phases after MLP2b are stale from an earlier <B,A> launch):')
for i, nm in enumerate(NAMES):
    print(f'  {nm:20s} {d[:, i].mean():9.0f}  (min {d[:, i].min():9.0f} max {d[:, i].max():9.0f})')
print(f'  phase B total       {(t[:, 8] - t[:, 0]).mean():9.0f}')
