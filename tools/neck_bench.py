"""HIP neck vs the torch/MIOpen neck on the bench shape (16 images of 40x40 =
both sides of 8 pairs at 640x640), per-kernel durations from HIP events."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval().to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bb = torch.relu(torch.randn(n, 1024, hb, hb, device=dev))
eng = pkg.NeckEngine({k: v for k, v in model.state_dict().items() if k in pkg.neck_keys()}, device=dev)
if len(sys.argv) > 3:
    eng.set_conv_kernel(sys.argv[3])        # 'gather' | 'row_window' (default: auto)

def timed(fn, it=20):
    for _ in range(3): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3, out
t_torch, ref = timed(lambda: model.input_proj2(model.patchmerging(model.input_proj(bb))))
t_hip, out = timed(lambda: eng.forward(bb))
with pkg.KernelTrace(eng) as tr:
    for _ in range(20): eng.forward(bb)
    torch.cuda.synchronize()
ks = {k: round(v[1] / v[0] * 1e3, 1) for k, v in tr.summary().items()}
gflop = n * hb * hb * (2 * 1024 * 256 + (2 * 256 * (16 * 256 + 64 * 128 + 256 * 128) + 2 * 512 * 256) / 4) / 1e9
print(f'n={n} {hb}x{hb}: torch {t_torch:.3f} ms, hip {t_hip:.3f} ms ({t_torch / t_hip:.1f}x), '
      f'{gflop / t_hip:.1f} TFLOP/s algorithmic; kernels us {ks}; max |diff| vs torch {float((out - ref).abs().max()):.2e}')
