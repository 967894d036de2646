"""Where the end-to-end forward_dummy time goes (torch front end vs HIP hot path)."""
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
n, s = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 640
img = torch.rand(n, s, s, 3, device=dev)

def timed(fn, it=10):
    for _ in range(3): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3, out

for cl in (False, True):
    if cl:
        model = model.to(memory_format=torch.channels_last)
    t_bb, f = timed(lambda: model.backbone(img))
    t_ip, g = timed(lambda: model.input_proj(f))
    t_pm, h = timed(lambda: model.patchmerging(g))
    t_ip2, _ = timed(lambda: model.input_proj2(h))
    t_all, _ = timed(lambda: model.forward_dummy(img, img))
    print(f'channels_last={cl} per image-batch of {n}: backbone {t_bb:.2f} ms, input_proj {t_ip:.2f}, patchmerging {t_pm:.2f}, '
          f'input_proj2 {t_ip2:.2f}; forward_dummy (2 images) {t_all:.2f} ms = {n / t_all * 1e3:.0f} pairs/s; backbone out {tuple(f.shape)}')
